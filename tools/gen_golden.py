"""Generate golden input/output vectors by IMPORTING the reference (container-only tool).

The reference tree (/root/reference) is read-only, pure Python and has no tests of its own,
so the only way to pin the oracle (oracle/vqcpc_oracle.py) is to run the reference's own
classes here, on seeded synthetic inputs, and commit the resulting tensors as data fixtures
under tests/golden/*.npz.  Nothing in tests/, bench.py or the package reads /root/reference
at run time; this script is the only file that does, and it never copies source text.

Two sys.modules stubs are needed because `torch.utils.tensorboard` and `music21` are not
installed in this image (VQCPCB/encoder.py:6, VQCPCB/dataloaders/bach_cpc_dataloader.py:2);
getters/dataloaders are bypassed and the modules are constructed directly.

Run:  python tools/gen_golden.py            (writes tests/golden/*.npz, prints a summary)
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
sys.path.insert(0, REF)

_tb = types.ModuleType('torch.utils.tensorboard')


class _NoWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass


_tb.SummaryWriter = _NoWriter
sys.modules['torch.utils.tensorboard'] = _tb
sys.modules['music21'] = types.ModuleType('music21')

from VQCPCB.data_processor.bach_cpc_data_processor import BachCPCDataProcessor  # noqa: E402
from VQCPCB.downscalers.relative_transformer_downscaler import RelativeTransformerDownscaler  # noqa: E402
from VQCPCB.encoder import Encoder  # noqa: E402
from VQCPCB.quantizer.vector_quantizer import ProductVectorQuantizer  # noqa: E402
from VQCPCB.transformer.subsampled_relative_attention import SubsampledRelativeAttention  # noqa: E402
from VQCPCB.transformer.transformer_custom import TransformerEncoderLayerCustom  # noqa: E402
from VQCPCB.upscalers.mlp_upscaler import MlpUpscaler  # noqa: E402
from VQCPCB.vqcpc_encoder_trainer import VQCPCEncoderTrainer  # noqa: E402
from VQCPCB.vqcpc_helper import CModule, FksModule, nce_loss, quantization_loss  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)
META = dict(torch_version=np.array(torch.__version__), cpu_capability=np.array(torch.backends.cpu.get_cpu_capability()))


def npy(t):
    return t.detach().cpu().numpy().copy()   # copy: parameters are updated in place later


def save(name, **arrays):
    arrays.update(META)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name:28s} {os.path.getsize(path) / 1024:8.1f} KiB  ' + ' '.join(sorted(k for k in arrays if k not in META))[:150])


def sd_arrays(prefix, module):
    return {f'{prefix}/{k}': npy(v) for k, v in module.state_dict().items()}


# ----------------------------------------------------------------------------------------------
# 1. ProductVectorQuantizer: forward (idx, quantized_sg, loss) + gradients wrt inputs / codebooks
# ----------------------------------------------------------------------------------------------
def gen_quantizer(name, R, nb, K, D, ncb, squared, seed, ties=False, scale=1.0):
    torch.manual_seed(seed)
    q = ProductVectorQuantizer(codebook_size=K, codebook_dim=D, commitment_cost=0.25, num_codebooks=ncb,
                               use_batch_norm=False, initialize=False, squared_l2_norm=squared)
    q.eval()
    z = (torch.randn(R, nb, D) * scale).requires_grad_(True)
    if ties:
        # exact ties: duplicated codebook rows (first index must win) and inputs sitting exactly on codes
        with torch.no_grad():
            for e in q.embeddings:
                e[K // 2:] = e[:K - K // 2]
                e[1] = e[0]
            dsub = D // ncb
            flat = z.view(-1, D)
            for c, e in enumerate(q.embeddings):
                flat[:K, c * dsub:(c + 1) * dsub] = e
    zq, idx, loss = q(z, corrupt_labels=False)
    g_zq = torch.randn_like(zq)
    g_loss = torch.randn_like(loss)
    (zq * g_zq).sum().add((loss * g_loss).sum()).backward()
    # top-2 distance gap (classifies a potential argmin mismatch as near-tie or bug)
    gaps = []
    for xc, e in zip(z.detach().view(-1, D).chunk(ncb, dim=1), q.embeddings):
        d = ((xc.unsqueeze(1) - e.detach().unsqueeze(0)) ** 2).sum(2)
        top2 = torch.topk(d, 2, dim=1, largest=False)[0]
        gaps.append((top2[:, 1] - top2[:, 0]))
    save(name, z=npy(z), codebooks=np.stack([npy(e) for e in q.embeddings]), idx=npy(idx), zq=npy(zq), loss=npy(loss),
         g_zq=npy(g_zq), g_loss=npy(g_loss), dz=npy(z.grad), dE=np.stack([npy(e.grad) for e in q.embeddings]),
         top2_gap=npy(torch.stack(gaps, 1)), squared=np.array(squared), beta=np.array(0.25))


# ----------------------------------------------------------------------------------------------
# 1b. ProductVectorQuantizer at the C1 quantiser shape (2 x 512 codes of dim 16), including the reference's own
#     data-dependent initialisation (_initialize, vector_quantizer.py:57-70) under a fixed global seed.
#     Slim fixture: inputs, the codebooks the reference initialised, indices (int16), loss, top-2 gaps.
# ----------------------------------------------------------------------------------------------
def gen_quantizer_init(name, R, K, D, ncb, seed, init_seed):
    torch.manual_seed(seed)
    q = ProductVectorQuantizer(codebook_size=K, codebook_dim=D, commitment_cost=0.25, num_codebooks=ncb,
                               use_batch_norm=False, initialize=True, squared_l2_norm=True)
    q.eval()
    z = torch.randn(R, 1, D)
    torch.manual_seed(init_seed)          # the permutations of _initialize come from the global CPU generator
    zq, idx, loss = q(z, corrupt_labels=False)
    assert not q.initialize
    gaps = []
    for xc, e in zip(z.view(-1, D).chunk(ncb, dim=1), q.embeddings):
        d = ((xc.unsqueeze(1) - e.detach().unsqueeze(0)) ** 2).sum(2)
        top2 = torch.topk(d, 2, dim=1, largest=False)[0]
        gaps.append((top2[:, 1] - top2[:, 0]))
    save(name, z=npy(z), codebooks=np.stack([npy(e) for e in q.embeddings]), idx=npy(idx).astype(np.int16),
         loss=npy(loss), top2_gap=npy(torch.stack(gaps, 1)), init_seed=np.array(init_seed), squared=np.array(True),
         beta=np.array(0.25))


# ----------------------------------------------------------------------------------------------
# 2. SubsampledRelativeAttention bias and one TransformerEncoderLayerCustom (eval mode)
# ----------------------------------------------------------------------------------------------
def gen_relbias(name, n, H, L, hd, seed):
    torch.manual_seed(seed)
    m = SubsampledRelativeAttention(head_dim=hd, num_heads=H, seq_len_src=L, seq_len_tgt=L)
    # cuda_variable() is a no-op here (no GPU in this container)
    q = torch.randn(n * H, L, hd)
    save(name, q=npy(q), e1=npy(m.e1), e2=npy(m.e2), bias=npy(m(q)), H=np.array(H))


def gen_layer(name, n, H, L, d, ff, seed):
    torch.manual_seed(seed)
    layer = TransformerEncoderLayerCustom(d_model=d, nhead=H, attention_bias_type='relative_attention', num_channels=1,
                                          num_events=L, dim_feedforward=ff, dropout=0.0)
    with torch.no_grad():  # biases/LN are initialised to 0/1: perturb them so the fixture exercises them
        for k, p in layer.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    layer.eval()
    x = torch.randn(L, n, d, requires_grad=True)  # time-first (L, N, E) as the reference feeds it
    y, att = layer(x)
    g = torch.randn_like(y)
    (y * g).sum().backward()
    arrays = sd_arrays('sd', layer)
    arrays.update({f'grad/{k}': npy(p.grad) for k, p in layer.named_parameters()})
    save(name, x=npy(x), y=npy(y), attn=npy(att['a_self_encoder']), g=npy(g), dx=npy(x.grad), H=np.array(H), **arrays)


# ----------------------------------------------------------------------------------------------
# 3. CModule / FksModule / nce_loss / quantization_loss
# ----------------------------------------------------------------------------------------------
def gen_cpc_heads(name, B, Kl, Kr, N, zdim, cdim, hidden, seed):
    torch.manual_seed(seed)
    cm = CModule(input_dim=zdim, hidden_size=hidden, output_dim=cdim, num_layers=2, dropout=0.0)
    fk = FksModule(z_dim=zdim, c_dim=cdim, k_max=Kr)
    z_left = torch.randn(B, Kl, zdim, requires_grad=True)
    z_right = torch.randn(B, Kr, zdim, requires_grad=True)
    z_neg = torch.randn(B, N, Kr, zdim, requires_grad=True)  # (batch, negatives, k, z)
    c = cm(z_left, h=None)
    f_pos = fk(c, z_right)
    zn = z_neg.permute(1, 0, 2, 3).contiguous().view(N * B, Kr, zdim)
    f_neg = fk(c.repeat(N, 1), zn).view(N, B, Kr).contiguous().permute(1, 2, 0)
    loss = nce_loss(f_pos, f_neg)
    acc = (f_pos > f_neg.max(2)[0]).sum(0).float() / B
    loss.backward()
    ql, qr, qn = torch.rand(B, Kl), torch.rand(B, Kr), torch.rand(B, N, Kr, 1)
    arrays = sd_arrays('c_module', cm)
    arrays.update(sd_arrays('fks_module', fk))
    arrays.update({f'grad/c_module/{k}': npy(p.grad) for k, p in cm.named_parameters()})
    arrays.update({'grad/fks_module/W': npy(fk.W.grad)})
    save(name, z_left=npy(z_left), z_right=npy(z_right), z_neg=npy(z_neg), c=npy(c), f_pos=npy(f_pos), f_neg=npy(f_neg),
         loss=npy(loss), acc=npy(acc), dz_left=npy(z_left.grad), dz_right=npy(z_right.grad), dz_neg=npy(z_neg.grad),
         ql=npy(ql), qr=npy(qr), qn=npy(qn), qloss=npy(quantization_loss(ql, qn, qr)), **arrays)


# ----------------------------------------------------------------------------------------------
# 4. Encoder.forward and VQCPCEncoderTrainer.epoch (train with dropout=0, eval)
# ----------------------------------------------------------------------------------------------
class FakeDataloaderGenerator:
    """Only what VQCPCEncoderTrainer.__init__ reads (vqcpc_encoder_trainer.py:58)."""

    def __init__(self, num_blocks_left, num_blocks_right, num_negative_samples):
        self.num_blocks_left = num_blocks_left
        self.num_blocks_right = num_blocks_right
        self.num_negative_samples = num_negative_samples
        self.num_tokens_per_block = 16
        self.num_channels = 4


def build_encoder(cfg):
    dp = BachCPCDataProcessor(embedding_size=cfg['emb'], num_events=(cfg['Kl'] + cfg['Kr']) * 4, num_channels=4,
                              num_tokens_per_channel=cfg['vocab'], num_tokens_per_block=16)
    ds = RelativeTransformerDownscaler(input_dim=cfg['emb'], output_dim=cfg['D'], num_channels=4, downscale_factors=[4, 4],
                                       d_model=cfg['d'], n_head=cfg['H'], list_of_num_layers=cfg['layers'],
                                       dim_feedforward=cfg['ff'], dropout=0.0)
    q = ProductVectorQuantizer(codebook_size=cfg['K'], codebook_dim=cfg['D'], commitment_cost=0.25,
                               num_codebooks=cfg['ncb'], use_batch_norm=False, initialize=False, squared_l2_norm=True)
    up = MlpUpscaler(input_dim=cfg['D'], output_dim=cfg['zdim'], hidden_size=cfg['up_hidden'], dropout=0.0)
    return Encoder('/tmp/vqcpc_golden_model', dp, ds, q, up)


def synth_batch(cfg, gen):
    B, N, Kl, Kr = cfg['B'], cfg['N'], cfg['Kl'], cfg['Kr']
    V = cfg['vocab'][0]
    return {
        'x_left': torch.randint(0, V, (B, Kl * 4, 4), generator=gen),
        'x_right': torch.randint(0, V, (B, Kr * 4, 4), generator=gen),
        'negative_samples': torch.randint(0, V, (B, N, Kr, 4, 4), generator=gen),
        'negative_samples_back': torch.randint(0, V, (B, N, Kr, 4, 4), generator=gen),
    }


def perturb_1d(module, std=0.05):
    with torch.no_grad():
        for _, p in module.named_parameters():
            if p.dim() == 1:
                p.add_(std * torch.randn_like(p))


def gen_encoder_and_epoch(name, cfg, seed):
    torch.manual_seed(seed)
    enc = build_encoder(cfg)
    perturb_1d(enc)
    enc_gen = torch.Generator().manual_seed(seed + 2)
    with torch.no_grad():  # codebooks placed on (perturbed) downscaler outputs so that many codes are in use
        from VQCPCB.utils import flatten as _fl
        probe = torch.randint(0, cfg['vocab'][0], (4 * cfg['K'], 4, 4), generator=enc_gen)
        zp = enc.downscaler(_fl(enc.data_processor.embed(enc.data_processor.preprocess(probe)))).view(-1, cfg['D'])
        dsub = cfg['D'] // cfg['ncb']
        for c, e in enumerate(enc.quantizer.embeddings):
            e.copy_(zp[c:c + 4 * cfg['K']:4, c * dsub:(c + 1) * dsub] + 0.01 * torch.randn(cfg['K'], dsub, generator=enc_gen))
    tr = VQCPCEncoderTrainer('/tmp/vqcpc_golden_model', FakeDataloaderGenerator(cfg['Kl'], cfg['Kr'], cfg['N']), enc,
                             c_net_kwargs=dict(output_dim=cfg['cdim'], hidden_size=cfg['gru_hidden'], num_layers=2,
                                               dropout=0.0, bidirectional=cfg.get('bidirectional', False)),
                             quantization_weighting=cfg.get('qw', 0.5))
    gen = torch.Generator().manual_seed(seed + 1)
    batch = synth_batch(cfg, gen)
    arrays = {}
    for k in ('encoder', 'c_module', 'fks_module'):
        arrays.update(sd_arrays(f'sd0/{k}', getattr(tr, k)))
    if tr.c_module_back is not None:
        arrays.update(sd_arrays('sd0/c_module_back', tr.c_module_back))
        arrays.update(sd_arrays('sd0/fks_module_back', tr.fks_module_back))
    arrays.update({f'batch/{k}': npy(v) for k, v in batch.items()})

    # ---- Encoder.forward stages on x_left (eval mode, no dropout anyway)
    enc.eval()
    x_proc = enc.data_processor.preprocess(batch['x_left'])
    x_embed = enc.data_processor.embed(x_proc)
    from VQCPCB.utils import flatten
    z = enc.downscaler(flatten(x_embed))
    zq, idx, ql = enc.quantizer(z, corrupt_labels=False)
    z_up = enc.upscaler(zq)
    arrays.update(fwd_tokens=npy(x_proc), fwd_embed=npy(x_embed), fwd_z=npy(z), fwd_idx=npy(idx), fwd_zq=npy(zq),
                  fwd_qloss=npy(ql), fwd_zup=npy(z_up))
    full = enc(batch['x_left'])
    assert torch.equal(full[0], z_up) and torch.equal(full[1], idx)

    # ---- eval epoch (vqcpc_encoder_trainer.py:169-354, train=False)
    tr.init_optimizers(lr=1e-3, schedule_lr=False)
    ev = tr.epoch(iter([batch]), train=False, num_batches=1, corrupt_labels=False)
    for k, v in ev.items():
        arrays[f'eval/{k}'] = np.asarray(v, dtype=np.float64)

    # ---- train epoch, dropout=0: capture the gradients BEFORE clip_grad_norm_ rescales them in place
    pre_clip = {}
    orig_clip = torch.nn.utils.clip_grad_norm_

    def spy(parameters, max_norm, *a, **k):
        params = list(parameters)
        names = {id(p): n for n, p in tr.named_parameters()}
        for p in params:
            if p.grad is not None:
                pre_clip[names[id(p)]] = p.grad.detach().clone()
        return orig_clip(params, max_norm, *a, **k)

    torch.nn.utils.clip_grad_norm_ = spy
    import VQCPCB.vqcpc_encoder_trainer as vt
    vt.nn.utils.clip_grad_norm_ = spy
    try:
        trn = tr.epoch(iter([batch]), train=True, num_batches=1, corrupt_labels=False)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
        vt.nn.utils.clip_grad_norm_ = orig_clip
    for k, v in trn.items():
        arrays[f'train/{k}'] = np.asarray(v, dtype=np.float64)
    for k, g in pre_clip.items():
        arrays[f'grad/{k}'] = npy(g)
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in pre_clip.values()))
    arrays['grad_total_norm'] = np.asarray(float(total))
    for k in ('encoder', 'c_module', 'fks_module'):
        arrays.update(sd_arrays(f'sd1/{k}', getattr(tr, k)))
    if tr.c_module_back is not None:
        arrays.update(sd_arrays('sd1/c_module_back', tr.c_module_back))
        arrays.update(sd_arrays('sd1/fks_module_back', tr.fks_module_back))
    arrays['cfg_json'] = np.array(__import__('json').dumps(cfg))
    arrays['lr'] = np.array(1e-3)
    save(name, **arrays)
    print('   eval :', {k: (round(v, 5) if not isinstance(v, list) else [round(a, 3) for a in v]) for k, v in ev.items()})
    print('   train:', {k: (round(v, 5) if not isinstance(v, list) else [round(a, 3) for a in v]) for k, v in trn.items()})
    print('   grad total norm', float(total))


if __name__ == '__main__':
    gen_quantizer('vq_ncb1', R=64, nb=3, K=8, D=4, ncb=1, squared=True, seed=1)
    gen_quantizer('vq_ncb2', R=96, nb=2, K=16, D=8, ncb=2, squared=True, seed=2)
    gen_quantizer('vq_ncb2_d32', R=128, nb=1, K=64, D=32, ncb=2, squared=True, seed=3)
    gen_quantizer('vq_wide_d64', R=40, nb=1, K=32, D=64, ncb=1, squared=True, seed=4)     # dsub > 16: torch vector-sum order
    gen_quantizer('vq_ties', R=48, nb=2, K=16, D=8, ncb=2, squared=True, seed=5, ties=True)
    gen_quantizer('vq_l2norm', R=32, nb=2, K=8, D=4, ncb=1, squared=False, seed=6)
    gen_relbias('relbias_L16', n=3, H=2, L=16, hd=8, seed=10)
    gen_relbias('relbias_L4', n=5, H=4, L=4, hd=4, seed=11)
    gen_layer('layer_L16', n=6, H=2, L=16, d=32, ff=64, seed=20)
    gen_layer('layer_L4', n=6, H=2, L=4, d=32, ff=48, seed=21)   # hd = 16
    gen_cpc_heads('cpc_heads', B=5, Kl=3, Kr=4, N=6, zdim=8, cdim=6, hidden=12, seed=30)
    tiny = dict(emb=8, vocab=[11, 11, 11, 11], d=32, H=2, layers=[2, 1], ff=64, D=4, K=8, ncb=1, zdim=8, up_hidden=16,
                cdim=8, gru_hidden=16, B=6, N=3, Kl=2, Kr=2)
    gen_encoder_and_epoch('epoch_tiny', tiny, seed=40)
    bidir = dict(tiny, bidirectional=True, Kl=3, Kr=3, B=4)   # reference needs Kl == Kr for the backward direction
    gen_encoder_and_epoch('epoch_tiny_bidir', bidir, seed=41)
    gen_encoder_and_epoch('epoch_tiny_clip', dict(tiny, qw=60.0, B=5), seed=42)   # global grad norm > 5: clip active
    # round 2 additions (appended: every generator re-seeds, so the fixtures above are unchanged)
    gen_quantizer_init('vq_c1_init', R=4096, K=512, D=32, ncb=2, seed=7, init_seed=1234)
    acc = dict(emb=8, vocab=[11, 11, 11, 11], d=32, H=2, layers=[1, 1], ff=64, D=16, K=64, ncb=1, zdim=8, up_hidden=16,
               cdim=8, gru_hidden=16, B=16, N=2, Kl=2, Kr=2)
    gen_encoder_and_epoch('epoch_tiny_acc', acc, seed=43)     # accuracy is not all-zero: pins the hit counting
