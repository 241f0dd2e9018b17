"""Shapes of the small torch ops (fill / copy / add / cat ...) of one eager training step, including the ones the autograd
engine issues itself (gradient accumulation, zero materialisation): python tools/small_op_shapes.py [C1|C3|DEC]"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C1'
STEPS = 3
sys.argv = ['bench.py', '--config', cfg, '--steps', str(STEPS), '--warmup', '1', '--no-cpu-baseline', '--no-kernel-timing', '--no-graph']
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    bench.main()
count = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::add_', 'aten::add', 'aten::cat', 'aten::stack', 'aten::sum',
                   'aten::mul', 'aten::clone', 'aten::zeros', 'aten::zeros_like', 'aten::contiguous'):
        count[(ev.name, str(ev.input_shapes)[:90])] += 1
print('op, input shapes: count over the run (1 warm-up + 12 sampled + 2 x %d steps)' % STEPS)
for (name, shp), n in count.most_common(60):
    print(f'{n:6d}  {name:16s} {shp}')
