"""Host cost of one HIP-graph replay of the training step versus the eager enqueue (GPU only):
    python tools/graph_replay_cost.py [C1|C3|DEC] [batch]
Synchronises, then times (a) the host call that enqueues ONE step and (b) the time until the GPU has finished it."""
import contextlib
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import configs, getters, hip  # noqa: E402
from vqcpc_bach_amd.parallel import DataParallelContext  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'C1'
    hip.load()
    hip.set_gemm_mode(1)
    dp = DataParallelContext()
    config = configs.make_config(name, dropout=0.2 if name == 'DEC' else 0.1)
    B = int(sys.argv[2]) if len(sys.argv) > 2 else config['batch_size']
    dlg = getters.get_dataloader_generator(config['dataset'], config['training_method'],
                                           dict(config['dataloader_generator_kwargs'], seed=1234, rank=0, device=dp.device))
    if config['training_method'].lower() == 'decoder':
        ec = config['config_encoder']
        edlg = getters.get_dataloader_generator(ec['dataset'], ec['training_method'],
                                                dict(ec['dataloader_generator_kwargs'], seed=1234, rank=0, device=dp.device))
        enc = getters.get_encoder('/tmp/m', edlg, ec)
        proc = getters.get_data_processor(dlg, config['data_processor_type'], config['data_processor_kwargs'])
        tr = getters.get_decoder('/tmp/m', dlg, proc, enc, config['decoder_type'], config['decoder_kwargs'])
    else:
        enc = getters.get_encoder('/tmp/m', dlg, config)
        tr = getters.get_encoder_trainer('/tmp/m', dlg, config['training_method'], enc, config['auxiliary_networks_kwargs'])
    tr.to(dp.device)
    tr.init_optimizers(lr=config['lr'], schedule_lr=config.get('schedule_lr', False), dp=dp)
    tr.train()
    batch = next(dlg.dataloaders(batch_size=B)[0])
    kw = dict(masked_event_index=7) if name == 'C3' else {}
    for graph in (False, True):
        tr.enable_step_graph(graph)
        with contextlib.redirect_stdout(sys.stderr):
            for _ in range(5):
                tr.train_step(batch, train=True, **kw)
        host, total = [], []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.train_step(batch, train=True, **kw)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(1e3 * (t1 - t0))
            total.append(1e3 * (t2 - t0))
        print(f'{name} B={B} graph={graph}: host call {statistics.median(host):.2f} ms, step done after '
              f'{statistics.median(total):.2f} ms (median of 20 isolated steps)', flush=True)
    tr.enable_step_graph(False)


if __name__ == '__main__':
    main()
