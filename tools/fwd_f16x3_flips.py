"""Code-assignment agreement of the FULL-SIZE C1 training forward under the opt-in f16x3 forward arithmetic (GPU;
`python tools/fwd_f16x3_flips.py [C1|C4]`).  The same batch and parameters, dropout off: codes and losses of compute_losses() with
the forward GEMMs on (a) the bf16x6 split (default), (b) the three-product fp16 kernel (ops.FWD_ARITH = 'f16x3'), (c) the exact
fp32-MFMA kernels (GEMM mode 0) -- the codes that differ between any two of the three fp32-class arithmetics are near ties."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqcpc_bach_amd import configs, getters, hip, ops  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'C1'
    torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)          # parameters; the batch is seeded by the generator
    config = configs.make_config(name, dropout=0.0)
    dlg = getters.get_dataloader_generator('bach', 'vqcpc', dict(config['dataloader_generator_kwargs'], device='cuda', seed=7))
    enc = getters.get_encoder('/tmp/vqcpc_fwd_flips', dlg, config)
    tr = getters.get_encoder_trainer('/tmp/vqcpc_fwd_flips', dlg, 'vqcpc', enc, config['auxiliary_networks_kwargs'])
    tr.to('cuda')
    tr.init_optimizers(lr=1e-4, schedule_lr=False)
    batch = next(dlg.dataloaders(batch_size=config['batch_size'])[0])
    hip.set_gemm_mode(1)
    tr.eval()
    with torch.no_grad():
        tr.compute_losses(batch)                                   # data-dependent codebook initialisation
    tr.train()
    res, f16 = {}, 0
    for label, mode, fwd in (('bf16x6', 1, 'six'), ('f16x3', 1, 'f16x3'), ('fp32 MFMA', 0, 'six')):
        hip.set_gemm_mode(mode)
        ops.FWD_ARITH = fwd
        raw, n3 = hip.call, [0]

        def counting(fn, *args):
            n3[0] += fn in ('vqcpc_gemm_nt_f16x3',) or (fn == 'vqcpc_gemm_nt_grad')
            return raw(fn, *args)
        hip.call = counting
        try:
            for _ in range(2):                                     # second pass: the scales have followed
                with torch.enable_grad(), ops.forward_arithmetic(tr.flat):
                    loss, out = tr.compute_losses(batch)
        finally:
            hip.call = raw
        if fwd == 'f16x3':
            f16 = n3[0] // 2
        res[label] = (float(loss), torch.cat([out[k].reshape(-1) for k in ('idx_left', 'idx_right', 'idx_negative')]).cpu())
    hip.set_gemm_mode(1)
    ops.FWD_ARITH = 'six'
    names = list(res)
    n = res[names[0]][1].numel()
    print(f'{name}: {f16} forward launches per pass on the three-product kernel')
    for i in range(3):
        for j in range(i + 1, 3):
            a, b = res[names[i]], res[names[j]]
            print(f'{names[i]:10s} vs {names[j]:10s}: {int((a[1] != b[1]).sum())} of {n} codes differ, loss {a[0]:.7f} vs {b[0]:.7f}')


if __name__ == '__main__':
    main()
