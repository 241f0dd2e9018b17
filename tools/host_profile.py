"""Host-side cost of a training step: counts and times every C-ABI call and torch.empty during `bench.py --config X`
(run on the GPU box:  python tools/host_profile.py C1|C3|DEC)."""
import os
import sys, time, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = ['bench.py', '--config', sys.argv[1], '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-kernel-timing']
import torch
from vqcpc_bach_amd import hip, ops
stats = collections.defaultdict(lambda: [0, 0.0])
raw_call = hip.call
def timed_call(name, *a):
    t0 = time.perf_counter(); raw_call(name, *a); dt = time.perf_counter() - t0
    s = stats[name]; s[0] += 1; s[1] += dt
hip.call = timed_call
raw_empty = torch.empty
def timed_empty(*a, **k):
    t0 = time.perf_counter(); r = raw_empty(*a, **k); dt = time.perf_counter() - t0
    s = stats['torch.empty']; s[0] += 1; s[1] += dt
    return r
torch.empty = timed_empty
def wrap_fn(cls, label):
    for meth in ('forward', 'backward'):
        raw = getattr(cls, meth)
        def timed(*a, _raw=raw, _k=f'{label}.{meth}', **k):
            t0 = time.perf_counter(); r = _raw(*a, **k); dt = time.perf_counter() - t0
            s = stats['py:' + _k]; s[0] += 1; s[1] += dt
            return r
        setattr(cls, meth, staticmethod(timed))
for name in ('EncoderLayerFn', 'LinearFn', 'AttnXFn', 'AddLayerNormFn', 'FFNFn', 'CrossProjFn', 'EmbeddingFn', 'GRULayerFn', 'VQFn',
             'NCEFn', 'SoftmaxCEFn', 'EmbedPosFn', 'UpscaleFn', 'BlockTableGatherFn'):
    if hasattr(ops, name):
        wrap_fn(getattr(ops, name), name)
# statistics restart when the warm-up steps are over (first calls pay lazy kernel loading / data-dependent codebook init)
WARMUP, STEPS = 3, 10
seen = [0]
def count_steps(cls):
    raw = cls.train_step
    def step(self, *a, **k):
        if seen[0] == WARMUP:
            stats.clear()
        seen[0] += 1
        return raw(self, *a, **k)
    cls.train_step = step
from vqcpc_bach_amd.vqcpc_encoder_trainer import VQCPCEncoderTrainer
from vqcpc_bach_amd.student_encoder_trainer import StudentEncoderTrainer
from vqcpc_bach_amd.decoders.decoder import Decoder
for c in (VQCPCEncoderTrainer, StudentEncoderTrainer, Decoder):
    count_steps(c)
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'), run_name='__main__')
calls = {k: v for k, v in stats.items() if k != 'torch.empty' and not k.startswith('py:')}
tot = sum(v[1] for v in calls.values()); n = sum(v[0] for v in calls.values())
print(f'hip.call: {n/STEPS:.0f} calls/step, {1e3*tot/STEPS:.2f} ms/step, {1e6*tot/n:.1f} us/call')
e = stats['torch.empty']; print(f'torch.empty: {e[0]/STEPS:.0f} calls/step, {1e3*e[1]/STEPS:.2f} ms/step')
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f'  {k:36s} {v[0]/STEPS:7.1f}/step {1e6*v[1]/v[0]:7.1f} us/call')
