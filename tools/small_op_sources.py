"""Which Python lines launch the small torch kernels (fill / copy / add / cat ...) of a training step: wraps the tensor
methods involved, runs a few eager bench steps and counts the calls by their first stack frame inside the package (ops
issued by the autograd engine itself -- gradient accumulation, zero materialisation -- have no Python frame and are not seen).
    python tools/small_op_sources.py [C1|C3|DEC]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C1'
STEPS = 4
sys.argv = ['bench.py', '--config', cfg, '--steps', str(STEPS), '--warmup', '1', '--no-cpu-baseline', '--no-kernel-timing', '--no-graph']
import torch  # noqa: E402

count = collections.Counter()


def where():
    for f in reversed(traceback.extract_stack()[:-2]):
        if 'vqcpc_bach_amd' in f.filename or f.filename.endswith('bench.py'):
            return f'{os.path.basename(f.filename)}:{f.lineno} {f.line[:70]}'
    return '?'


def wrap(obj, name, label):
    raw = getattr(obj, name)

    def w(*a, **k):
        t = a[0] if a and torch.is_tensor(a[0]) else None
        if t is None or t.is_cuda or label in ('zeros', 'cat', 'stack', 'zeros_like', 'empty_like'):
            count[(label, where())] += 1
        return raw(*a, **k)
    setattr(obj, name, w)


for n in ('zero_', 'fill_', 'copy_', 'clone', 'contiguous', 'add_', '__add__', '__iadd__', '__mul__', 'sum', 'float', 'to'):
    wrap(torch.Tensor, n, n)
for n in ('zeros', 'cat', 'stack', 'zeros_like', 'where'):
    wrap(torch, n, n)
import bench  # noqa: E402

bench.main()
print('calls by source line (whole run: 1 warm-up epoch step + 12 sampled + 2 x %d steps):' % STEPS)
for (name, frame), n in count.most_common(70):
    print(f'{n:6d}  {name:12s} {frame}')
