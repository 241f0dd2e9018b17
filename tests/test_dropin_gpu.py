"""The Python face of the drop-in boundary: after `vqcpc_bach_amd.install_as_vqcpcb()` the call
sequence of the reference's main_encoder.py:56-97 runs unchanged through `from VQCPCB.getters import ...` -- with a config
dict that has the reference's exact key set (VQCPCB/configs/encoder_random_transfo_config.py:11-87; the values are
scaled down, and the two keys that file forgets although main_encoder.py:91 / vqcpc_encoder_trainer.py:53 read them --
`schedule_lr`, `c_net_kwargs.bidirectional` -- are present).  The config below is DATA written for this test."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REFERENCE_CONFIG_KEYS = {
    'training_method', 'dataset', 'dataloader_generator_kwargs', 'subdivision', 'data_processor_type',
    'data_processor_kwargs', 'downscaler_type', 'downscaler_kwargs', 'quantizer_type', 'quantizer_kwargs', 'upscaler_type',
    'upscaler_kwargs', 'auxiliary_networks_kwargs', 'lr', 'batch_size', 'num_batches', 'num_epochs',
    'quantizer_regularization', 'timestamp', 'savename'}


def make_reference_style_config():
    num_tokens_per_block = 1 * 4 * 4
    return {
        'training_method': 'vqcpc',
        'dataset': 'bach',
        'dataloader_generator_kwargs': dict(num_tokens_per_block=num_tokens_per_block, num_blocks_left=3, num_blocks_right=3,
                                            negative_sampling_method='random', num_negative_samples=15, sequences_size=1),
        'subdivision': 4,
        'data_processor_type': 'bach_cpc',
        'data_processor_kwargs': dict(embedding_size=32),
        'downscaler_type': 'relative_transformer_downscaler',
        'downscaler_kwargs': dict(downscale_factors=[4, 4], num_channels=4, d_model=128, n_head=8, list_of_num_layers=[2, 2],
                                  dim_feedforward=256, dropout=0.1),
        'quantizer_type': 'commitment',
        'quantizer_kwargs': dict(num_codebooks=1, codebook_size=32, codebook_dim=3, commitment_cost=0.25,
                                 use_batch_norm=False, squared_l2_norm=True),
        'upscaler_type': 'mlp_upscaler',
        'upscaler_kwargs': dict(output_dim=32, hidden_size=64, dropout=0.1),
        'auxiliary_networks_kwargs': {
            'quantization_weighting': 0.5,
            'c_net_kwargs': dict(output_dim=32, hidden_size=64, num_layers=2, dropout=0.1, bidirectional=False),
        },
        'lr': 1e-4,
        'schedule_lr': True,
        'batch_size': 16,
        'num_batches': 2,
        'num_epochs': 2,
        'quantizer_regularization': dict(corrupt_labels=False),
        'timestamp': 'test',
        'savename': 'encoder_dropin',
    }


def run_main_encoder(config, train, load, model_root, num_workers=0):
    """main_encoder.py:24-97 with the config module replaced by a dict and the cluster plots (music21) left out."""
    from VQCPCB.encoder import EncoderTrainer                                                  # main_encoder.py:12-13
    from VQCPCB.getters import get_dataloader_generator, get_encoder, get_encoder_trainer
    gpu_ids = [int(gpu) for gpu in range(torch.cuda.device_count())]
    device = 'cpu' if len(gpu_ids) == 0 else 'cuda'
    model_dir = f'{model_root}/{config["savename"]}_{config["timestamp"]}'
    config['quantizer_kwargs']['initialize'] = not load                                        # :51
    dataloader_generator = get_dataloader_generator(dataset=config['dataset'], training_method=config['training_method'],
                                                    dataloader_generator_kwargs=config['dataloader_generator_kwargs'])
    encoder = get_encoder(model_dir=model_dir, dataloader_generator=dataloader_generator, config=config)
    encoder_trainer = get_encoder_trainer(model_dir=model_dir, dataloader_generator=dataloader_generator,
                                          training_method=config['training_method'], encoder=encoder,
                                          auxiliary_networks_kwargs=config['auxiliary_networks_kwargs'])
    assert isinstance(encoder_trainer, EncoderTrainer)
    if load:
        encoder_trainer.load(early_stopped=False, device=device)
    encoder_trainer.to(device)
    history = None
    if train:
        os.makedirs(model_dir, exist_ok=True)
        history = encoder_trainer.train_model(batch_size=config['batch_size'], num_batches=config['num_batches'],
                                              num_epochs=config['num_epochs'], lr=config['lr'],
                                              schedule_lr=config['schedule_lr'],
                                              corrupt_labels=config['quantizer_regularization']['corrupt_labels'],
                                              plot=True, num_workers=num_workers)
    return encoder_trainer, model_dir, history


def test_main_encoder_sequence_through_the_vqcpcb_alias(tmp_path):
    import vqcpc_bach_amd
    for k in [k for k in sys.modules if k == 'VQCPCB' or k.startswith('VQCPCB.')]:
        del sys.modules[k]
    vqcpc_bach_amd.install_as_vqcpcb()
    import VQCPCB.getters
    import VQCPCB.vqcpc_encoder_trainer
    assert VQCPCB.getters is sys.modules['vqcpc_bach_amd.getters']
    assert VQCPCB.vqcpc_encoder_trainer.VQCPCEncoderTrainer is vqcpc_bach_amd.vqcpc_encoder_trainer.VQCPCEncoderTrainer

    config = make_reference_style_config()
    assert set(config) == REFERENCE_CONFIG_KEYS | {'schedule_lr'}
    torch.manual_seed(0)
    trainer, model_dir, history = run_main_encoder(config, train=True, load=False, model_root=str(tmp_path))
    assert len(history) == 2
    for train_m, val_m in history:
        for m in (train_m, val_m):                                                             # :343-354 contract
            assert set(m) == {'loss', 'accuracy', 'loss_quantize', 'loss_contrastive', 'num_codewords',
                              'num_codewords_negative', 'loss_monitor'}
            assert np.isfinite(m['loss']) and len(m['accuracy']) == 3 and 1 <= m['num_codewords'] <= 32
    assert trainer.global_step == 4 and not trainer.encoder.quantizer.initialize
    # checkpoint layout of encoder.py:47-56 / vqcpc_encoder_trainer.py:117-131
    for sub in ('overfitted', 'early_stopped'):
        files = set(os.listdir(f'{model_dir}/{sub}'))
        assert {'data_processor', 'downscaler', 'quantizer', 'upscaler', 'c_module', 'fks_module'} <= files, files
    sd = torch.load(f'{model_dir}/overfitted/downscaler', map_location='cpu')
    assert sd['transformers.0.layers.0.self_attn.in_proj_weight'].shape == (384, 128)
    assert sd['transformers.1.layers.1.self_attn.attn_bias.e1'].shape == (8 * 4, 16)
    assert torch.load(f'{model_dir}/overfitted/quantizer', map_location='cpu')['embeddings.0'].shape == (32, 3)

    # `-l -t`: resume.  initialize = not load, the loaded codebooks must survive the first batch
    saved = {n: p.detach().clone() for n, p in trainer.named_parameters()}
    config2 = make_reference_style_config()
    config2['num_epochs'] = 1
    trainer2, _, history2 = run_main_encoder(config2, train=False, load=True, model_root=str(tmp_path))
    for n, p in trainer2.named_parameters():
        assert torch.equal(p.detach(), saved[n]), n
    # even a quantizer built with initialize=True must not re-initialise after load()
    config3 = make_reference_style_config()
    config3['num_epochs'] = 1
    from VQCPCB.getters import get_dataloader_generator, get_encoder, get_encoder_trainer
    config3['quantizer_kwargs']['initialize'] = True
    dlg = get_dataloader_generator(config3['dataset'], config3['training_method'], config3['dataloader_generator_kwargs'])
    enc = get_encoder(model_dir, dlg, config3)
    tr3 = get_encoder_trainer(model_dir, dlg, 'vqcpc', enc, config3['auxiliary_networks_kwargs'])
    tr3.load(early_stopped=False, device='cuda')
    tr3.to('cuda')
    assert not enc.quantizer.initialize
    tr3.init_optimizers(lr=1e-4, schedule_lr=True)
    assert tr3.global_step == 4, 'optimiser / schedule state resumes (extension over the reference)'
    cb = enc.quantizer.embeddings[0].detach().clone()
    gen_train, _, _ = dlg.dataloaders(batch_size=16)
    tr3.epoch(gen_train, train=False, num_batches=1, corrupt_labels=False)
    assert torch.equal(cb, enc.quantizer.embeddings[0].detach())
    with pytest.raises(KeyError):                       # the reference requires the key too (getters.py:146)
        bad = make_reference_style_config()
        get_encoder(model_dir, dlg, bad)


def test_student_config_through_the_alias(tmp_path):
    """main_encoder.py with training_method 'Student' (configs/encoder_student_config.py schema) at reduced sizes."""
    import vqcpc_bach_amd
    from vqcpc_bach_amd import configs
    vqcpc_bach_amd.install_as_vqcpcb()
    config = configs.make_student_config(dropout=0.1)
    config['downscaler_kwargs'].update(d_model=128, dim_feedforward=256, list_of_num_layers=[1, 1])
    config['auxiliary_networks_kwargs']['teacher_kwargs'].update(num_layers=2, d_model=128, dim_feedforward=256)
    config['auxiliary_networks_kwargs']['auxiliary_decoder_kwargs'].update(d_model=128, dim_feedforward=256,
                                                                            list_of_num_layers=[1, 1])
    config.update(batch_size=4, num_batches=2, num_epochs=1, timestamp='t', schedule_lr=False)
    trainer, model_dir, history = run_main_encoder(config, train=True, load=False, model_root=str(tmp_path))
    assert len(history) == 1 and np.isfinite(history[0][0]['loss_encdec'])
    assert os.path.exists(f'{model_dir}/overfitted/quantizer')


def test_train_model_defaults_are_the_benchmarked_path():
    """A caller who follows main_encoder.py and chooses nothing gets what bench.py measures: bf16x6 GEMM arithmetic and
    HIP-graph replay of the training step ("the benched path is not the drop-in default").  Fresh process, so
    that no earlier test has made a process-wide choice; explicit choices still win."""
    import subprocess
    from conftest import ROOT
    code = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, 'tests'))
sys.path.insert(0, %r)
import vqcpc_bach_amd
from vqcpc_bach_amd import hip, ops
from test_dropin_gpu import make_reference_style_config, run_main_encoder
vqcpc_bach_amd.install_as_vqcpcb()
hip.load()
assert hip.get_gemm_mode() == 0                       # bare library default: exact fp32 MFMA
assert ops.GRAD_ARITH == 'six' and ops.FWD_ARITH == 'six'   # ... and the six-product split in backward and in every forward too
seen, seen_fwd = [], []
raw_enter = ops.direct_weight_gradients.__enter__
ops.direct_weight_gradients.__enter__ = lambda self: (seen.append(ops.GRAD_ARITH), raw_enter(self))[1]
raw_fenter = ops.forward_arithmetic.__enter__
ops.forward_arithmetic.__enter__ = lambda self: (seen_fwd.append(ops.FWD_ARITH), raw_fenter(self))[1]
cfg = make_reference_style_config()
cfg.update(num_batches=6, num_epochs=1)
tr, _, hist = run_main_encoder(cfg, train=True, load=False, model_root=sys.argv[1])
assert tr.trained_gemm_mode == 1, tr.trained_gemm_mode    # train_model chose bf16x6 for its epochs ...
assert hip.get_gemm_mode() == 0                       # ... and put the process-wide setting back
assert seen and set(seen) == {'f16x3'}, seen          # backward passes ran under the f16x3 gradient arithmetic (round 5) ...
assert seen_fwd and set(seen_fwd) == {'f16x3'}, seen_fwd      # ... and the training forwards under the f16x3 forward arithmetic
assert ops.GRAD_ARITH == 'six' and ops.FWD_ARITH == 'six'     # ... which train_model() put back as well
assert tr._graph is not None and tr._graph.replays >= 3, 'training steps are graph replays by default'
print('REPLAYS', tr._graph.replays)
# explicit choices win
hip.set_gemm_mode(0)
cfg2 = make_reference_style_config(); cfg2.update(num_batches=4, num_epochs=1, timestamp='t2')
from VQCPCB.getters import get_dataloader_generator, get_encoder, get_encoder_trainer
dlg = get_dataloader_generator(cfg2['dataset'], cfg2['training_method'], cfg2['dataloader_generator_kwargs'])
cfg2['quantizer_kwargs']['initialize'] = True
enc = get_encoder(sys.argv[1] + '/m2', dlg, cfg2)
tr2 = get_encoder_trainer(sys.argv[1] + '/m2', dlg, 'vqcpc', enc, cfg2['auxiliary_networks_kwargs'])
tr2.to('cuda'); tr2.enable_step_graph(False)
tr2.train_model(batch_size=16, num_batches=4, num_epochs=1, lr=1e-4, schedule_lr=False, corrupt_labels=False)
assert hip.get_gemm_mode() == 0 and tr2._graph is None
print('OK')
''' % (ROOT, ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = {k: v for k, v in os.environ.items() if k not in ('VQCPC_GEMM_MODE', 'VQCPC_STEP_GRAPH', 'VQCPC_GRAD_ARITH', 'VQCPC_FWD_ARITH')}
        r = subprocess.run([sys.executable, '-c', code, d], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'OK' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
