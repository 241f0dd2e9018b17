"""Factories with the reference's names and config schema (reference: VQCPCB/getters.py:24-45,48-175,431-443,487-514),
restricted to the encoder / vqcpc branches that `main_encoder.py` reaches."""
from .data_processor.bach_cpc_data_processor import BachCPCDataProcessor
from .dataloaders.synthetic_cpc_dataloader import SyntheticCPCDataloaderGenerator
from .downscalers.relative_transformer_downscaler import RelativeTransformerDownscaler
from .encoder import Encoder
from .quantizer.vector_quantizer import NoQuantization, ProductVectorQuantizer
from .upscalers.mlp_upscaler import MlpUpscaler
from .vqcpc_encoder_trainer import VQCPCEncoderTrainer


def get_dataloader_generator(dataset, training_method, dataloader_generator_kwargs):
    if dataset.lower() in ('bach', 'synthetic') and training_method.lower() == 'vqcpc':
        # the music21 Bach corpus is replaced by a synthetic generator with the same tensor contract
        return SyntheticCPCDataloaderGenerator(**dataloader_generator_kwargs)
    raise NotImplementedError('only the vqcpc training method is on the hot path (student/decoder/prior: out of scope)')


def get_downscaler(downscaler_type, downscaler_kwargs):
    if downscaler_type == 'relative_transformer_downscaler':
        k = downscaler_kwargs
        return RelativeTransformerDownscaler(input_dim=k['input_dim'], output_dim=k['output_dim'],
                                             downscale_factors=k['downscale_factors'], num_channels=k['num_channels'],
                                             d_model=k['d_model'], n_head=k['n_head'],
                                             list_of_num_layers=k['list_of_num_layers'],
                                             dim_feedforward=k['dim_feedforward'], dropout=k['dropout'])
    raise NotImplementedError(f'{downscaler_type}: only the relative transformer downscaler is on the hot path')


def get_upscaler(upscaler_type, upscaler_kwargs):
    if upscaler_type == 'mlp_upscaler':
        return MlpUpscaler(input_dim=upscaler_kwargs['input_dim'], output_dim=upscaler_kwargs['output_dim'],
                           hidden_size=upscaler_kwargs['hidden_size'], dropout=upscaler_kwargs['dropout'])
    if upscaler_type is None:
        return None
    raise NotImplementedError


def get_data_processor(dataloader_generator, data_processor_type, data_processor_kwargs):
    if data_processor_type != 'bach_cpc':
        raise NotImplementedError
    dataset = dataloader_generator.dataset_positive
    num_events = dataset.sequences_size * dataset.subdivision
    num_tokens_per_channel = [len(d) for d in dataset.index2note_dicts]
    dp = BachCPCDataProcessor(embedding_size=data_processor_kwargs['embedding_size'], num_events=num_events,
                              num_channels=dataloader_generator.num_channels,
                              num_tokens_per_channel=num_tokens_per_channel,
                              num_tokens_per_block=dataloader_generator.num_tokens_per_block)
    assert dataloader_generator.num_channels == dp.num_channels
    return dp


def get_encoder(model_dir, dataloader_generator, config):
    if config['training_method'].lower() != 'vqcpc':
        raise NotImplementedError
    quantizer_kwargs, downscaler_kwargs = config['quantizer_kwargs'], config['downscaler_kwargs']
    data_processor = get_data_processor(dataloader_generator, config['data_processor_type'], config['data_processor_kwargs'])
    downscaler_kwargs['input_dim'] = data_processor.embedding_size
    downscaler_kwargs['output_dim'] = quantizer_kwargs['codebook_dim']
    downscaler_kwargs['num_tokens'] = data_processor.num_events * data_processor.num_channels
    downscaler_kwargs['num_channels'] = data_processor.num_channels
    downscaler = get_downscaler(config['downscaler_type'], downscaler_kwargs)
    if config['quantizer_type'] == 'commitment':
        quantizer = ProductVectorQuantizer(codebook_size=quantizer_kwargs['codebook_size'],
                                           num_codebooks=quantizer_kwargs['num_codebooks'],
                                           codebook_dim=quantizer_kwargs['codebook_dim'],
                                           initialize=quantizer_kwargs.get('initialize', True),
                                           squared_l2_norm=quantizer_kwargs['squared_l2_norm'],
                                           use_batch_norm=quantizer_kwargs['use_batch_norm'],
                                           commitment_cost=quantizer_kwargs['commitment_cost'])
    elif config['quantizer_type'] is None:
        quantizer = NoQuantization(codebook_dim=quantizer_kwargs['codebook_dim'])
    else:
        raise NotImplementedError
    upscaler = None
    if config['upscaler_type'] is not None:
        upscaler_kwargs = config['upscaler_kwargs']
        upscaler_kwargs['input_dim'] = quantizer_kwargs['codebook_dim']
        upscaler = get_upscaler(config['upscaler_type'], upscaler_kwargs)
    return Encoder(model_dir=model_dir, data_processor=data_processor, downscaler=downscaler, quantizer=quantizer,
                   upscaler=upscaler)


def get_encoder_trainer(model_dir, dataloader_generator, training_method, encoder, auxiliary_networks_kwargs):
    if training_method.lower() == 'vqcpc':
        return VQCPCEncoderTrainer(model_dir=model_dir, dataloader_generator=dataloader_generator, encoder=encoder,
                                   c_net_kwargs=auxiliary_networks_kwargs['c_net_kwargs'],
                                   quantization_weighting=auxiliary_networks_kwargs['quantization_weighting'])
    raise NotImplementedError('student trainer: SURVEY.md section 8 config C3, not built yet')
