"""BASELINE.json configurations in the reference's config-dict schema (VQCPCB/configs/encoder_random_transfo_config.py)."""
import copy

_SIZES = {
    # name: d_model, heads, layers, ff, codebook_dim, codebook_size, num_codebooks, B, Kl, Kr
    'C0': (128, 4, [1, 1], 512, 16, 64, 1, 8, 2, 2),
    'C1': (256, 8, [2, 2], 1024, 32, 512, 2, 256, 8, 8),
    'C4': (512, 8, [4, 4], 2048, 64, 1024, 4, 256, 16, 16),
}


def make_student_config(dropout=0.1, **over):
    """BASELINE.json configs[3] / SURVEY.md C3: VQCPCB/configs/encoder_student_config.py:5-99."""
    cfg = {
        'training_method': 'Student', 'dataset': 'bach',
        'dataloader_generator_kwargs': dict(sequences_size=24),
        'subdivision': 4,
        'data_processor_type': 'bach', 'data_processor_kwargs': dict(embedding_size=32),
        'downscaler_type': 'relative_transformer_downscaler_linear',
        'downscaler_kwargs': dict(downscale_factors=[4, 4], d_model=512, n_head=8, list_of_num_layers=[4, 4],
                                  dim_feedforward=2048, attention_bias_type='relative_attention', dropout=dropout),
        'quantizer_type': 'commitment',
        'quantizer_kwargs': dict(num_codebooks=1, codebook_size=32, codebook_dim=3, commitment_cost=0.25,
                                 use_batch_norm=False, squared_l2_norm=True, initialize=True),
        'upscaler_type': None,
        'auxiliary_networks_kwargs': {
            'quantization_weighting': 0.1, 'num_events_masked': 4, 'teacher_type': 'relative',
            'teacher_kwargs': dict(data_processor_config=dict(data_processor_type='bach',
                                                              data_processor_kwargs=dict(embedding_size=32)),
                                   num_layers=8, positional_embedding_size=8, d_model=512, dim_feedforward=2048, n_head=8,
                                   dropout=dropout),
            'auxiliary_decoder_type': 'relative',
            'auxiliary_decoder_kwargs': dict(positional_embedding_size=8, d_model=512, dim_feedforward=2048, n_head=8,
                                             dropout=dropout, list_of_num_layers=[4, 4]),
        },
        'lr': 1e-5, 'schedule_lr': False, 'batch_size': 8, 'num_batches': 512, 'num_epochs': 1,
        'quantizer_regularization': dict(corrupt_labels=False), 'timestamp': None, 'savename': 'encoder_student_C3',
    }
    cfg = copy.deepcopy(cfg)
    for k, v in over.items():
        cfg[k] = v
    return cfg


def make_decoder_config(dropout=0.2, **over):
    """VQCPCB/configs/decoder_relative_AC_AC_C_random.py:4-46 (24 beats = 96 ticks = 384
    target tokens, d_model 512, 8 heads, 3 + 3 layers, ff 1024, dropout 0.2, batch 32, scheduled lr) on the frozen
    transformer encoder of VQCPCB/configs/encoder_random_transfo_config.py:26-63 (d_model 512, 8 heads, [2, 2] layers,
    ff 2048, 1 x 32 codes of dim 3) -- the shipped decoder configs point at LSTM-downscaler encoders, which are out of
    scope."""
    enc = make_config('C1', dropout=0.1)
    enc['downscaler_kwargs'].update(d_model=512, n_head=8, list_of_num_layers=[2, 2], dim_feedforward=2048)
    enc['quantizer_kwargs'].update(num_codebooks=1, codebook_size=32, codebook_dim=3, initialize=False)
    cfg = {
        'config_encoder': enc, 'training_method': 'decoder', 'dataset': 'bach',
        'dataloader_generator_kwargs': dict(sequences_size=24),
        'data_processor_type': 'bach', 'data_processor_kwargs': dict(embedding_size=32),
        'decoder_type': 'transformer_relative',
        'decoder_kwargs': dict(d_model=512, n_head=8, num_encoder_layers=3, num_decoder_layers=3, dim_feedforward=1024,
                               positional_embedding_size=8, dropout=dropout),
        'lr': 1e-4, 'schedule_lr': True, 'batch_size': 32, 'num_batches': 2048, 'num_epochs': 1, 'timestamp': None,
        'savename': 'decoder_relative_AC_AC_C',
    }
    cfg = copy.deepcopy(cfg)
    for k, v in over.items():
        cfg[k] = v
    return cfg


def make_config(name='C1', dropout=0.1, **over):
    if name == 'C3':
        return make_student_config(dropout=dropout, **over)
    if name == 'DEC':
        return make_decoder_config(dropout=dropout, **over)
    d, H, layers, ff, D, K, ncb, B, Kl, Kr = _SIZES[name]
    cfg = {
        'training_method': 'vqcpc', 'dataset': 'bach',
        'dataloader_generator_kwargs': dict(num_tokens_per_block=16, num_blocks_left=Kl, num_blocks_right=Kr,
                                            negative_sampling_method='random', num_negative_samples=15,
                                            sequences_size=1),
        'subdivision': 4,
        'data_processor_type': 'bach_cpc', 'data_processor_kwargs': dict(embedding_size=32),
        'downscaler_type': 'relative_transformer_downscaler',
        'downscaler_kwargs': dict(downscale_factors=[4, 4], num_channels=4, d_model=d, n_head=H,
                                  list_of_num_layers=layers, dim_feedforward=ff, dropout=dropout),
        'quantizer_type': 'commitment',
        'quantizer_kwargs': dict(num_codebooks=ncb, codebook_size=K, codebook_dim=D, commitment_cost=0.25,
                                 use_batch_norm=False, squared_l2_norm=True, initialize=True),
        'upscaler_type': 'mlp_upscaler', 'upscaler_kwargs': dict(output_dim=32, hidden_size=512, dropout=dropout),
        'auxiliary_networks_kwargs': {'quantization_weighting': 0.5,
                                      'c_net_kwargs': dict(output_dim=32, hidden_size=512, num_layers=2, dropout=dropout,
                                                           bidirectional=False)},
        'lr': 1e-4, 'schedule_lr': False, 'batch_size': B, 'num_batches': 256, 'num_epochs': 1,
        'quantizer_regularization': dict(corrupt_labels=False), 'timestamp': None, 'savename': f'encoder_cpc_{name}',
    }
    cfg = copy.deepcopy(cfg)
    for k, v in over.items():
        cfg[k] = v
    return cfg
