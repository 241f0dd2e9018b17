"""SURVEY.md section 8(f) rows built on top of the hot path:
  N1  encoder inference path (indices only)                      decoders/decoder.py:327-336, encoder.py:97-110
  N2  'same_sequence' negatives constructed on the device        dataloaders/bach_cpc_dataloader.py:110-181
  N3  checkpoint round trip incl. the optimiser state (extension) encoder.py:47-74, vqcpc_encoder_trainer.py:117-151"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import vqcpc_oracle as O
from test_trainer_gpu import build_trainer, golden_cfg_sd

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize('name', ['negatives_same_seq', 'negatives_same_seq_uneven'])
def test_same_sequence_negatives_kernel_bit_exact(name):
    from vqcpc_bach_amd import hip, ops
    hip.load()
    g = load_golden(name)
    xl, xr = T(g['x_left']).cuda(), T(g['x_right']).cuda()
    assert torch.equal(ops.same_sequence_negatives(xl, xr).cpu(), T(g['negative_samples']))
    if 'negative_samples_back' in g:
        assert torch.equal(ops.same_sequence_negatives(xr, xl).cpu(), T(g['negative_samples_back']))


def test_same_sequence_negatives_kernel_vs_oracle_full_size():
    """C1 shape: B = 256, 8 + 8 blocks -> (256, 15, 8, 4, 4) int64; integer gather, bit-exact."""
    from vqcpc_bach_amd import hip, ops
    hip.load()
    gen = torch.Generator().manual_seed(3)
    xl = torch.randint(0, 56, (256, 32, 4), generator=gen)
    xr = torch.randint(0, 56, (256, 32, 4), generator=gen)
    got = ops.same_sequence_negatives(xl.cuda(), xr.cuda())
    assert got.shape == (256, 15, 8, 4, 4)
    assert torch.equal(got.cpu(), O.same_sequence_negatives(xl, xr))


def test_same_sequence_training_step_matches_oracle():
    """One bidirectional training step on a same_sequence batch built on the device == the oracle on the same batch."""
    from vqcpc_bach_amd.dataloaders.synthetic_cpc_dataloader import SyntheticCPCDataloaderGenerator
    cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=64, H=4, layers=[1, 1], ff=128, D=16, K=16, ncb=1, zdim=16, up_hidden=32,
                     cdim=16, gru_hidden=32, B=6, Kl=3, Kr=3, N=5, bidirectional=True)
    dlg = SyntheticCPCDataloaderGenerator(num_blocks_left=3, num_blocks_right=3, negative_sampling_method='same_sequence',
                                          vocab=cfg['vocab'], device='cuda', seed=11)
    assert dlg.num_negative_samples == 5
    batch = next(dlg.dataloaders(batch_size=cfg['B'])[0])
    assert batch['negative_samples'].shape == (6, 5, 3, 4, 4) and batch['negative_samples'].is_cuda
    host = {k: v.cpu() for k, v in batch.items()}
    assert torch.equal(host['negative_samples'], O.same_sequence_negatives(host['x_left'], host['x_right']))
    assert torch.equal(host['negative_samples_back'], O.same_sequence_negatives(host['x_right'], host['x_left']))
    sd = O.init_state(cfg, seed=2)
    st = {}
    O.encoder_forward(host['x_left'], sd, cfg, stages=st)
    sd['encoder.quantizer.embeddings.0'] = st['z'].reshape(-1, cfg['D'])[:cfg['K']].clone() + 0.01
    otr = O.OracleTrainer(cfg, sd, lr=1e-3)
    ref = otr.step(host, train=True)
    tr = build_trainer(cfg, sd, lr=1e-3)
    tr.train()
    loss, out = tr.compute_losses(batch)
    assert torch.equal(out['idx_negative'].cpu().reshape(ref['idx_negative'].shape), ref['idx_negative'])
    assert abs(float(out['loss']) - float(ref['loss'])) < 5e-5 * max(1.0, abs(float(ref['loss'])))


def test_encode_indices_inference_path():
    g = load_golden('epoch_tiny')
    cfg, sd = golden_cfg_sd(g)
    tr = build_trainer(cfg, sd)
    tr.eval()
    x = T(g['batch/x_left'])
    idx = tr.encoder.encode_indices(x)
    assert idx.dtype == torch.int64 and torch.equal(idx.cpu(), T(g['fwd_idx']))       # == the reference's indices
    with torch.no_grad():
        _, idx_full, _ = tr.encoder(x)
    assert torch.equal(idx, idx_full)
    merged = tr.encoder.encode_indices(x, merged=True)
    assert merged.shape == idx.shape[:-1] and torch.equal(merged, tr.encoder.merge_codes(idx))
    assert not idx.requires_grad


def test_encode_indices_two_codebooks_merge():
    cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=64, H=4, layers=[1, 1], ff=128, D=16, K=16, ncb=2, zdim=16, up_hidden=32,
                     cdim=16, gru_hidden=32, B=4, Kl=2, Kr=2, N=3)
    sd = O.init_state(cfg, seed=4)
    tr = build_trainer(cfg, sd)
    tr.eval()
    x = O.synthetic_batch(cfg, seed=5)['x_left']
    idx = tr.encoder.encode_indices(x)
    ref = O.encoder_forward(x, sd, cfg)[1]
    assert torch.equal(idx.cpu(), ref)
    assert torch.equal(tr.encoder.encode_indices(x, merged=True).cpu(), O.merge_codes(ref, cfg['K']))


def test_checkpoint_round_trip_with_optimizer_state(tmp_path):
    """save -> new trainer -> load -> init_optimizers resumes Adam moments, step count and the LR-schedule position;
    the next step of the resumed trainer equals the next step of the original one bit for bit (dropout off)."""
    cfg = O.make_cfg(emb=16, vocab=[30] * 4, d=64, H=4, layers=[1, 1], ff=128, D=16, K=16, ncb=1, zdim=16, up_hidden=32,
                     cdim=16, gru_hidden=32, B=4, Kl=2, Kr=2, N=3)
    sd = O.init_state(cfg, seed=7)
    batches = [O.synthetic_batch(cfg, seed=20 + i) for i in range(3)]
    a = build_trainer(cfg, sd, lr=1e-3)
    a.model_dir = a.encoder.model_dir = str(tmp_path / 'model')
    a.schedule_lr = True
    a.train()
    for b in batches[:2]:
        a.train_step(b, train=True)
    a.save(early_stopped=False)
    files = sorted(p.name for p in (tmp_path / 'model' / 'overfitted').iterdir())
    assert files == ['c_module', 'data_processor', 'downscaler', 'fks_module', 'optimizer', 'quantizer', 'upscaler']

    b_tr = build_trainer(cfg, O.init_state(cfg, seed=99), lr=1e-3)        # different weights: load must overwrite them
    b_tr.model_dir = b_tr.encoder.model_dir = str(tmp_path / 'model')
    b_tr.load(early_stopped=False, device='cuda')
    b_tr.init_optimizers(lr=1e-3, schedule_lr=True)
    assert b_tr.global_step == 2 and b_tr.optimizer.step_count == 2
    assert torch.equal(b_tr.optimizer.m, a.optimizer.m) and torch.equal(b_tr.optimizer.v, a.optimizer.v)
    assert torch.equal(b_tr.flat.flat, a.flat.flat)
    assert b_tr.current_lr() == a.current_lr()
    a.train_step(batches[2], train=True)
    b_tr.train()
    b_tr.train_step(batches[2], train=True)
    assert torch.equal(b_tr.flat.flat, a.flat.flat)


@pytest.mark.parametrize('method', ['vqcpc', 'student'])
def test_train_model_loop_writes_the_reference_checkpoint_layout(tmp_path, method):
    """EncoderTrainer.train_model (encoder.py:244-325) through the getters: epochs, validation, `overfitted` and
    `early_stopped` checkpoints with the reference's file names, reloadable into a freshly built trainer."""
    from vqcpc_bach_amd import configs, getters
    import os
    if method == 'vqcpc':
        config = configs.make_config('C0', dropout=0.1)
        config['dataloader_generator_kwargs']['device'] = 'cuda'
        files = {'data_processor', 'downscaler', 'quantizer', 'upscaler', 'c_module', 'fks_module', 'optimizer'}
    else:
        config = configs.make_config('C3', dropout=0.1)
        config['downscaler_kwargs'].update(d_model=64, n_head=2, list_of_num_layers=[1, 1], dim_feedforward=128)
        aux = config['auxiliary_networks_kwargs']
        aux['teacher_kwargs'].update(num_layers=2, d_model=64, n_head=2, dim_feedforward=128)
        aux['auxiliary_decoder_kwargs'].update(d_model=64, n_head=2, dim_feedforward=128, list_of_num_layers=[1, 1])
        config['dataloader_generator_kwargs'].update(sequences_size=8, device='cuda')
        files = {'data_processor', 'downscaler', 'quantizer', 'decoder', 'teacher', 'optimizer'}
    model_dir = str(tmp_path / 'model')

    def build():
        dlg = getters.get_dataloader_generator(config['dataset'], config['training_method'],
                                               dict(config['dataloader_generator_kwargs']))
        enc = getters.get_encoder(model_dir, dlg, config)
        tr = getters.get_encoder_trainer(model_dir, dlg, config['training_method'], enc,
                                         dict(config['auxiliary_networks_kwargs']))
        tr.to('cuda')
        return tr

    tr = build()
    hist = tr.train_model(batch_size=4, num_batches=2, num_epochs=2, lr=1e-4, corrupt_labels=False, schedule_lr=True,
                          plot=False, num_workers=0)
    assert len(hist) == 2 and all(np.isfinite(v) for v in (hist[-1][0]['loss_monitor'], hist[-1][1]['loss_monitor']))
    for sub in ('overfitted', 'early_stopped'):
        assert set(os.listdir(os.path.join(model_dir, sub))) == files, sub
    tr2 = build()
    tr2.load(early_stopped=False, device='cuda')
    tr2.init_optimizers(lr=1e-4, schedule_lr=True)
    assert tr2.global_step == tr.global_step == 4
    assert torch.equal(tr2.flat.flat, tr.flat.flat)
