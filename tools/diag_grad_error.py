"""Diagnostic (GPU box): per-parameter gradient error of the product against the fp32 oracle AND against the same oracle
evaluated in float64 (the truth), at a BASELINE configuration's model dimensions.
    python tools/diag_grad_error.py C1 8 [gemm_mode]"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import vqcpc_oracle as O  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def main():
    name, B = sys.argv[1], int(sys.argv[2])
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    from test_configs_gpu import _data_placed_codebooks
    from test_trainer_gpu import build_trainer
    from vqcpc_bach_amd import hip
    cfg = O.make_cfg(name, B=B)
    sd = O.init_state(cfg, seed=31)
    batch = O.synthetic_batch(cfg, seed=32)
    _data_placed_codebooks(cfg, sd, batch)
    names = list(sd.keys())
    # fp32 oracle
    P32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out32 = O.cpc_losses(batch, P32, cfg, training=True)
    g32 = torch.autograd.grad(out32['loss'], [P32[k] for k in names], allow_unused=True)
    # float64 oracle with the fp32 oracle's code assignment (the argmin itself is not under test here)
    P64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    real_assign = O.vq_assign
    idx_iter = iter([out32['idx_negative'], out32['idx_left'], out32['idx_right']])
    O.vq_assign = lambda z, cbs: next(idx_iter).reshape(-1, cfg['ncb'])
    try:
        out64 = O.cpc_losses(batch, P64, cfg, training=True)
    finally:
        O.vq_assign = real_assign
    g64 = torch.autograd.grad(out64['loss'], [P64[k] for k in names], allow_unused=True)
    hip.load()
    hip.set_gemm_mode(mode)
    if os.environ.get('VQCPC_TABLE_RATIO'):            # 0: always the first-layer block table; 1000000000: never
        from vqcpc_bach_amd.downscalers.relative_transformer_downscaler import RelativeTransformerDownscaler as D
        D.table_lookup_min_ratio = int(os.environ['VQCPC_TABLE_RATIO'])
    tr = build_trainer(cfg, sd, lr=1e-4)
    tr.train()
    loss, out = tr.compute_losses(batch)
    tr.flat.zero_grad()
    loss.backward()
    got = dict(tr.named_parameters())
    print(f'loss: product {float(loss):.7f}  oracle32 {float(out32["loss"]):.7f}  oracle64 {float(out64["loss"]):.9f}')
    print(f'{"parameter":70s} {"prod-vs-o32":>11s} {"prod-vs-o64":>11s} {"o32-vs-o64":>11s}')
    for k, a, b in zip(names, g32, g64):
        if a is None:
            continue
        p = got[k].grad.cpu()
        print(f'{k:70s} {rel(p, a):11.2e} {rel(p, b):11.2e} {rel(a, b):11.2e}')


if __name__ == '__main__':
    main()
