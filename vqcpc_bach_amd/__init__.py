"""vqcpc_bach_amd -- MI355X-native VQ-CPC encoder training step behind the reference's Python surface.

Module names mirror SonyCSLParis/vqcpc-bach's `VQCPCB` package for the hot path only
(`encoder`, `vqcpc_encoder_trainer`, `student_encoder_trainer`, `vqcpc_helper`, `quantizer.vector_quantizer`,
`downscalers.relative_transformer_downscaler[_linear]`, `transformer.*`, `data_processor.*`, `upscalers.mlp_upscaler`,
`teachers.teacher_relative`, `auxiliary_decoders.auxiliary_decoder_relative`, `decoders.decoder` (training step only), `getters`), so `main_encoder.py` runs unchanged after `vqcpc_bach_amd.install_as_vqcpcb()`.
All numerical work goes through libvqcpc_hip.so (include/vqcpc.h); there is no CPU fallback.
"""
import sys

__version__ = '0.1.0'


def install_as_vqcpcb():
    """Alias this package as `VQCPCB` so that `from VQCPCB.getters import get_encoder, ...` resolves here."""
    import importlib
    pkg = sys.modules[__name__]
    sys.modules.setdefault('VQCPCB', pkg)
    for sub in ('utils', 'encoder', 'getters', 'vqcpc_helper', 'vqcpc_encoder_trainer', 'quantizer',
                'quantizer.vector_quantizer', 'downscalers', 'downscalers.relative_transformer_downscaler',
                'transformer', 'transformer.transformer_custom', 'transformer.multihead_attention_custom',
                'transformer.subsampled_relative_attention', 'data_processor', 'data_processor.data_processor',
                'data_processor.bach_cpc_data_processor', 'upscalers', 'upscalers.mlp_upscaler', 'dataloaders',
                'dataloaders.synthetic_cpc_dataloader', 'dataloaders.synthetic_student_dataloader',
                'student_encoder_trainer', 'teachers', 'teachers.teacher_relative', 'auxiliary_decoders',
                'auxiliary_decoders.auxiliary_decoder_relative', 'data_processor.bach_data_processor',
                'downscalers.relative_transformer_downscaler_linear', 'decoders', 'decoders.decoder'):
        sys.modules.setdefault('VQCPCB.' + sub, importlib.import_module(__name__ + '.' + sub))
    return pkg
