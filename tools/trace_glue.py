"""Which Python lines of the training step still launch torch kernels (adds, fills, cats, copies)?  One eager C1 step under
torch.profiler with stacks; prints every aten op that launched a device kernel with the innermost frames inside this repo.
    python tools/trace_glue.py [C1|C3|DEC]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import configs, getters, hip
from vqcpc_bach_amd.parallel import DataParallelContext
from torch.profiler import profile, ProfilerActivity

name = sys.argv[1] if len(sys.argv) > 1 else 'C1'
hip.load(); hip.set_gemm_mode(1)
dp = DataParallelContext()
torch.manual_seed(0)
config = configs.make_config(name, dropout=0.2 if name == 'DEC' else 0.1)
dlg = getters.get_dataloader_generator(config['dataset'], config['training_method'],
                                       dict(config['dataloader_generator_kwargs'], seed=1, rank=0, device=dp.device))
if config['training_method'].lower() == 'decoder':
    enc_cfg = config['config_encoder']
    enc_dlg = getters.get_dataloader_generator(enc_cfg['dataset'], enc_cfg['training_method'],
                                               dict(enc_cfg['dataloader_generator_kwargs'], seed=1, rank=0, device=dp.device))
    enc = getters.get_encoder('/tmp/vqcpc_trace', enc_dlg, enc_cfg)
    data_processor = getters.get_data_processor(dlg, config['data_processor_type'], config['data_processor_kwargs'])
    tr = getters.get_decoder('/tmp/vqcpc_trace', dlg, data_processor, enc, config['decoder_type'], config['decoder_kwargs'])
else:
    enc = getters.get_encoder('/tmp/vqcpc_trace', dlg, config)
    tr = getters.get_encoder_trainer('/tmp/vqcpc_trace', dlg, config['training_method'], enc, config['auxiliary_networks_kwargs'])
tr.to(dp.device); tr.init_optimizers(lr=config['lr'], schedule_lr=False, dp=dp); tr.train()
stream = dlg.dataloaders(batch_size=config['batch_size'])[0]
batches = [next(stream) for _ in range(4)]
for b in batches[:3]:
    tr.train_step(b, train=True)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

here = os.path.realpath(os.path.join(os.path.dirname(__file__), '..'))
agg = collections.Counter()
SKIP = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.transpose', 'aten.slice', 'aten.select', 'aten.as_strided',
        'aten.expand', 'aten.permute', 'aten.unsqueeze', 'aten.squeeze', 'aten.alias', 'aten.empty', 'aten.reshape', 'aten.unbind',
        'aten.split', 'aten._local_scalar_dense', 'aten.lift_fresh', 'aten.is_', 'aten.new_empty', 'aten.empty_like', 'aten.stride',
        'aten.sym_', 'aten.size', 'aten._reshape_alias', 'aten.chunk', 'aten.narrow', 'aten.unflatten', 'aten.flatten')


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            frames = [f for f in traceback.extract_stack() if 'vqcpc_bach_amd' in f.filename]
            where = ' <- '.join(f'{os.path.relpath(f.filename, here)}:{f.lineno}' for f in reversed(frames[-3:]))
            agg[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    tr.train_step(batches[3], train=True)
torch.cuda.synchronize()
for (op, where), n in sorted(agg.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f'{n:3d} x {op:34s} {where}')
print('total', sum(agg.values()))
