"""Factories with the reference's names and config schema (reference: VQCPCB/getters.py:24-45,48-175,221-270,431-514),
restricted to the encoder branches (vqcpc, student) that `main_encoder.py` reaches, plus `get_decoder` (:274-392) for
the relative decoder training step."""
import numpy as np

from .auxiliary_decoders.auxiliary_decoder_relative import AuxiliaryDecoderRelative
from .data_processor.bach_cpc_data_processor import BachCPCDataProcessor
from .data_processor.bach_data_processor import BachDataProcessor
from .dataloaders.synthetic_cpc_dataloader import SyntheticCPCDataloaderGenerator
from .dataloaders.synthetic_student_dataloader import SyntheticStudentDataloaderGenerator
from .decoders.decoder import Decoder
from .downscalers.relative_transformer_downscaler import RelativeTransformerDownscaler
from .downscalers.relative_transformer_downscaler_linear import RelativeTransformerDownscalerLinear
from .encoder import Encoder
from .student_encoder_trainer import StudentEncoderTrainer
from .teachers.teacher_relative import TeacherRelative
from .quantizer.vector_quantizer import NoQuantization, ProductVectorQuantizer
from .upscalers.mlp_upscaler import MlpUpscaler
from .vqcpc_encoder_trainer import VQCPCEncoderTrainer


def get_dataloader_generator(dataset, training_method, dataloader_generator_kwargs):
    if dataset.lower() in ('bach', 'synthetic') and training_method.lower() == 'vqcpc':
        # the music21 Bach corpus is replaced by a synthetic generator with the same tensor contract
        return SyntheticCPCDataloaderGenerator(**dataloader_generator_kwargs)
    if dataset.lower() in ('bach', 'synthetic') and training_method.lower() in ('student', 'decoder'):
        return SyntheticStudentDataloaderGenerator(**dataloader_generator_kwargs)     # {'x': (B, events, voices)}
    raise NotImplementedError('only the vqcpc, student and decoder training methods are on the path (prior: out of scope)')


def get_downscaler(downscaler_type, downscaler_kwargs):
    if downscaler_type == 'relative_transformer_downscaler':
        k = downscaler_kwargs
        return RelativeTransformerDownscaler(input_dim=k['input_dim'], output_dim=k['output_dim'],
                                             downscale_factors=k['downscale_factors'], num_channels=k['num_channels'],
                                             d_model=k['d_model'], n_head=k['n_head'],
                                             list_of_num_layers=k['list_of_num_layers'],
                                             dim_feedforward=k['dim_feedforward'], dropout=k['dropout'])
    if downscaler_type == 'relative_transformer_downscaler_linear':
        k = downscaler_kwargs
        return RelativeTransformerDownscalerLinear(input_dim=k['input_dim'], output_dim=k['output_dim'],
                                                   downscale_factors=k['downscale_factors'],
                                                   num_channels=k['num_channels'], d_model=k['d_model'],
                                                   n_head=k['n_head'], list_of_num_layers=k['list_of_num_layers'],
                                                   dim_feedforward=k['dim_feedforward'], dropout=k['dropout'])
    raise NotImplementedError(f'{downscaler_type}: only the relative transformer downscalers are on the path')


def get_upscaler(upscaler_type, upscaler_kwargs):
    if upscaler_type == 'mlp_upscaler':
        return MlpUpscaler(input_dim=upscaler_kwargs['input_dim'], output_dim=upscaler_kwargs['output_dim'],
                           hidden_size=upscaler_kwargs['hidden_size'], dropout=upscaler_kwargs['dropout'])
    if upscaler_type is None:
        return None
    raise NotImplementedError


def get_data_processor(dataloader_generator, data_processor_type, data_processor_kwargs):
    if data_processor_type == 'bach':
        dataset = dataloader_generator.dataset
        return BachDataProcessor(embedding_size=data_processor_kwargs['embedding_size'],
                                 num_events=dataset.sequences_size * dataset.subdivision,
                                 num_tokens_per_channel=[len(d) for d in dataset.index2note_dicts])
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


def get_teacher(teacher_type, teacher_kwargs, dataloader_generator):
    if teacher_type != 'relative':
        raise NotImplementedError(f'teacher_type {teacher_type}: the student configuration uses the relative teacher')
    dpc = teacher_kwargs['data_processor_config']
    data_processor = get_data_processor(dataloader_generator, dpc['data_processor_type'], dpc['data_processor_kwargs'])
    k = teacher_kwargs
    return TeacherRelative(num_layers=k['num_layers'], num_tokens_per_channel=k['num_tokens_per_channel'],
                           d_model=k['d_model'], positional_embedding_size=k['positional_embedding_size'],
                           dim_feedforward=k['dim_feedforward'], n_head=k['n_head'], dropout=k['dropout'],
                           num_tokens=k['num_tokens'], data_processor=data_processor)


def get_auxiliary_decoder(auxiliary_decoder_type, auxiliary_decoder_kwargs):
    if auxiliary_decoder_type != 'relative':
        raise NotImplementedError(f'auxiliary_decoder_type {auxiliary_decoder_type}: only the relative decoder is built')
    k = auxiliary_decoder_kwargs
    return AuxiliaryDecoderRelative(num_tokens_per_channel=k['num_tokens_per_channel'], codebook_dim=k['codebook_dim'],
                                    upscale_factors=k['upscale_factors'], n_head=k['n_head'],
                                    dim_feedforward=k['dim_feedforward'], list_of_num_layers=k['list_of_num_layers'],
                                    d_model=k['d_model'], num_tokens_bottleneck=k['num_tokens_bottleneck'],
                                    dropout=k['dropout'])


def get_encoder(model_dir, dataloader_generator, config):
    if config['training_method'].lower() not in ('vqcpc', 'student'):
        raise NotImplementedError
    quantizer_kwargs, downscaler_kwargs = config['quantizer_kwargs'], config['downscaler_kwargs']
    data_processor = get_data_processor(dataloader_generator, config['data_processor_type'], config['data_processor_kwargs'])
    downscaler_kwargs['input_dim'] = data_processor.embedding_size
    downscaler_kwargs['output_dim'] = quantizer_kwargs['codebook_dim']
    downscaler_kwargs['num_tokens'] = data_processor.num_tokens
    downscaler_kwargs['num_channels'] = data_processor.num_channels
    downscaler = get_downscaler(config['downscaler_type'], downscaler_kwargs)
    if config['quantizer_type'] == 'commitment':
        quantizer = ProductVectorQuantizer(codebook_size=quantizer_kwargs['codebook_size'],
                                           num_codebooks=quantizer_kwargs['num_codebooks'],
                                           codebook_dim=quantizer_kwargs['codebook_dim'],
                                           initialize=quantizer_kwargs['initialize'],
                                           squared_l2_norm=quantizer_kwargs['squared_l2_norm'],
                                           use_batch_norm=quantizer_kwargs['use_batch_norm'],
                                           commitment_cost=quantizer_kwargs['commitment_cost'])
    elif config['quantizer_type'] is None:
        quantizer = NoQuantization(codebook_dim=quantizer_kwargs['codebook_dim'])
    else:
        raise NotImplementedError
    upscaler = None
    if config.get('upscaler_type') is not None:
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
    if training_method.lower() == 'student':                                           # getters.py:444-482
        teacher_kwargs = auxiliary_networks_kwargs['teacher_kwargs']
        teacher_kwargs['num_tokens_per_channel'] = encoder.data_processor.num_tokens_per_channel
        teacher_kwargs['num_tokens'] = encoder.data_processor.num_tokens
        teacher = get_teacher(auxiliary_networks_kwargs['teacher_type'], teacher_kwargs, dataloader_generator)
        dec_kwargs = auxiliary_networks_kwargs['auxiliary_decoder_kwargs']
        dec_kwargs['num_tokens_per_channel'] = encoder.data_processor.num_tokens_per_channel
        dec_kwargs['codebook_dim'] = encoder.quantizer.codebook_dim
        dec_kwargs['upscale_factors'] = list(reversed(encoder.downscaler.downscale_factors))
        dec_kwargs['num_tokens_bottleneck'] = (encoder.data_processor.num_tokens
                                               // int(np.prod(encoder.downscaler.downscale_factors)))
        decoder = get_auxiliary_decoder(auxiliary_networks_kwargs['auxiliary_decoder_type'], dec_kwargs)
        return StudentEncoderTrainer(model_dir=model_dir, dataloader_generator=dataloader_generator, encoder=encoder,
                                     teacher=teacher, auxiliary_decoder=decoder,
                                     quantization_weighting=auxiliary_networks_kwargs['quantization_weighting'],
                                     num_events_masked=auxiliary_networks_kwargs['num_events_masked'])
    raise NotImplementedError(training_method)


def get_decoder(model_dir, dataloader_generator, data_processor, encoder, decoder_type, decoder_kwargs):
    """getters.py:274-392.  'transformer_relative' (anticausal source / anticausal cross / causal target) and
    'transformer_relative_fullCross'; the absolute and diagonal variants are out of scope."""
    kinds = {'transformer_relative': ('anticausal', 'anticausal'), 'transformer_relative_fullCross': ('anticausal', 'full')}
    if decoder_type not in kinds:
        raise NotImplementedError(f'decoder_type {decoder_type}: only the relative decoders {sorted(kinds)} are built')
    num_channels_decoder = data_processor.num_channels
    num_events_decoder = data_processor.num_events
    num_channels_encoder = 1
    num_events_encoder = int((num_events_decoder * num_channels_decoder)
                             // (np.prod(encoder.downscaler.downscale_factors) * num_channels_encoder))
    k = decoder_kwargs
    return Decoder(model_dir=model_dir, dataloader_generator=dataloader_generator, data_processor=data_processor,
                   encoder=encoder, transformer_type='relative', encoder_attention_type=kinds[decoder_type][0],
                   cross_attention_type=kinds[decoder_type][1], d_model=k['d_model'],
                   num_encoder_layers=k['num_encoder_layers'], num_decoder_layers=k['num_decoder_layers'],
                   n_head=k['n_head'], dim_feedforward=k['dim_feedforward'], dropout=k['dropout'],
                   positional_embedding_size=k['positional_embedding_size'], num_channels_encoder=num_channels_encoder,
                   num_events_encoder=num_events_encoder, num_channels_decoder=num_channels_decoder,
                   num_events_decoder=num_events_decoder)
