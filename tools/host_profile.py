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
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'), run_name='__main__')
tot = sum(v[1] for k, v in stats.items() if k != 'torch.empty'); n = sum(v[0] for k, v in stats.items() if k != 'torch.empty')
print(f'hip.call: {n/13:.0f} calls/step, {1e3*tot/13:.2f} ms/step, {1e6*tot/n:.1f} us/call')
e = stats['torch.empty']; print(f'torch.empty: {e[0]/13:.0f} calls/step, {1e3*e[1]/13:.2f} ms/step')
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f'  {k:36s} {v[0]/13:7.1f}/step {1e6*v[1]/v[0]:7.1f} us/call')
