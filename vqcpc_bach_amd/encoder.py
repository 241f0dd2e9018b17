"""Encoder composition and the trainer base class (reference: VQCPCB/encoder.py:12-110, 216-325)."""
import os

import torch
from torch import nn

from . import ops
from .utils import dict_pretty_print, flatten


class Encoder(nn.Module):
    """data_processor -> downscaler -> quantizer -> upscaler  (encoder.py:31-45, 76-95)."""

    def __init__(self, model_dir, data_processor, downscaler, quantizer, upscaler):
        super().__init__()
        self.data_processor = data_processor
        self.downscaler = downscaler
        self.quantizer = quantizer
        self.upscaler = upscaler
        self.model_dir = model_dir

    # ---- checkpoints: one torch.save(state_dict) per sub-module, reference file names (encoder.py:47-74) --------
    def _dir(self, early_stopped):
        return f'{self.model_dir}/early_stopped' if early_stopped else f'{self.model_dir}/overfitted'

    def save(self, early_stopped):
        model_dir = self._dir(early_stopped)
        os.makedirs(model_dir, exist_ok=True)
        torch.save(self.data_processor.state_dict(), f'{model_dir}/data_processor')
        torch.save(self.downscaler.state_dict(), f'{model_dir}/downscaler')
        torch.save(self.quantizer.state_dict(), f'{model_dir}/quantizer')
        if self.upscaler is not None:
            torch.save(self.upscaler.state_dict(), f'{model_dir}/upscaler')

    def load(self, early_stopped, device):
        print(f'Loading models {self.__repr__()}')
        model_dir = self._dir(early_stopped)
        if not os.path.exists(model_dir):      # older layout
            model_dir = self.model_dir
        ml = torch.device(device)
        self.data_processor.load_state_dict(torch.load(f'{model_dir}/data_processor', map_location=ml))
        self.downscaler.load_state_dict(torch.load(f'{model_dir}/downscaler', map_location=ml))
        self.quantizer.load_state_dict(torch.load(f'{model_dir}/quantizer', map_location=ml))
        if self.upscaler:
            self.upscaler.load_state_dict(torch.load(f'{model_dir}/upscaler', map_location=ml))
        # loaded codebooks must never be overwritten by the data-dependent initialisation of the first batch
        # (main_encoder.py:51 builds the quantizer with initialize = not load for the same reason)
        if hasattr(self.quantizer, 'initialize'):
            self.quantizer.initialize = False

    # ---- forward ---------------------------------------------------------------------------------------------
    def encode_many(self, xs, corrupt_flags=None, with_loss_rows=False):
        """Encode several token tensors (batch_i, ticks_i, voices) in ONE pass: all 16-token blocks are independent
        (relative_transformer_downscaler.py:98-115), so the calls of vqcpc_encoder_trainer.py:201-231 merge into one
        launch sequence.  The FIRST tensor plays the role of the reference's first call for the data-dependent
        codebook initialisation.  Returns a list of (z_quantized, encoding_indices, quantization_loss)."""
        corrupt_flags = corrupt_flags or [False] * len(xs)
        tpb = self.downscaler.sequence_length
        toks = [self.data_processor.preprocess(x) for x in xs]               # (batch_i, nb_i, 16)
        # BachDataProcessor (student configuration) keeps (batch, events, voices): cut it into blocks of `tpb` tokens,
        # token p of a block = (event p // voices, voice p % voices), i.e. utils.flatten + the view of
        # relative_transformer_downscaler_linear.py:104-107
        toks = [t if t.shape[-1] == tpb else t.reshape(t.shape[0], -1, tpb) for t in toks]
        sizes = [t.shape[0] * t.shape[1] for t in toks]
        tokens = torch.cat([t.reshape(-1, tpb) for t in toks], dim=0) if len(toks) > 1 else toks[0].reshape(-1, tpb)
        tokens = self.data_processor.checked(tokens)
        z = self.downscaler.forward_tokens(tokens.unsqueeze(0), self.data_processor)[0]        # (R, D)
        starts = [sum(sizes[:i]) for i in range(len(sizes))]
        corrupt_rows = None
        if any(corrupt_flags):
            corrupt_rows = torch.cat([torch.arange(s, s + n, device=z.device)
                                      for s, n, f in zip(starts, sizes, corrupt_flags) if f])
        zq, idx, ql = self.quantizer(z, corrupt_labels=any(corrupt_flags), init_rows=slice(0, sizes[0]),
                                     corrupt_rows=corrupt_rows)
        if self.upscaler is not None:
            zq = self.upscaler(zq)
        out = []
        zq_parts = ops.SplitRowsFn.apply(zq, *sizes) if (len(sizes) > 1 and zq.requires_grad) else None
        for i, (t, s, n) in enumerate(zip(toks, starts, sizes)):
            b, nb = t.shape[0], t.shape[1]
            idx_i = idx[s:s + n].view(b, nb, -1) if idx is not None else None
            zq_i = zq_parts[i] if zq_parts is not None else zq[s:s + n]
            out.append((zq_i.view(b, nb, -1), idx_i, ql[s:s + n].view(b, nb)))
        # with_loss_rows: also the quantisation loss of ALL rows of the pass (the trainer's q-loss sums every segment)
        return (out, ql) if with_loss_rows else out

    def forward(self, x, corrupt_labels=False):
        """x (batch, num_ticks, num_voices) ints from the dataloader ->
        z_quantized (batch, nb, z_dim), encoding_indices (batch, nb, num_codebooks), quantization_loss (batch, nb)."""
        return self.encode_many([x], [corrupt_labels])[0]

    def forward_embedded(self, x_embed, corrupt_labels=False):
        """API-compatible tail of the reference forward for callers that already hold embeddings (encoder.py:84-95)."""
        z = self.downscaler.forward(flatten(x_embed))
        zq, idx, ql = self.quantizer(z, corrupt_labels=corrupt_labels)
        if self.upscaler is not None:
            zq = self.upscaler(zq)
        return zq, idx, ql

    @torch.no_grad()
    def encode_indices(self, x, merged=False):
        """Inference path (what decoders/decoder.py:327-336 and the cluster tools encoder.py:137-159 consume): token
        tensor (batch, ticks, voices) -> encoding_indices (batch, nb, num_codebooks) int64 [or merged codes
        (batch, nb)], without materialising the straight-through output, the loss or the upscaler.  Uses the modules'
        current train/eval mode for dropout like `forward` does; call `.eval()` first for deterministic codes."""
        tpb = self.downscaler.sequence_length
        t = self.data_processor.preprocess(x)
        t = t if t.shape[-1] == tpb else t.reshape(t.shape[0], -1, tpb)
        t = self.data_processor.checked(t)
        z = self.downscaler.forward_tokens(t.unsqueeze(0), self.data_processor)[0]             # (batch, nb, D)
        q = self.quantizer
        assert not q.initialize, 'the codebooks are still waiting for their data-dependent initialisation'
        from . import ops
        idx = ops.vq_assign(z.reshape(-1, z.shape[-1]), torch.stack(list(q.embeddings), dim=0))
        idx = idx.view(z.shape[0], z.shape[1], -1)
        return self.merge_codes(idx) if merged else idx

    def quantizer_needs_init(self):
        """True while the codebooks still wait for their data-dependent initialisation (first training batch)."""
        return bool(getattr(self.quantizer, 'initialize', False))

    def merge_codes(self, codes):
        """sum_c codes[..., c] * codebook_size**c  (encoder.py:97-110, without its aliasing `+=`)."""
        ret = codes[..., 0].clone()
        for c in range(1, codes.shape[-1]):
            ret = ret + codes[..., c] * (self.quantizer.codebook_size ** c)
        return ret

    def plot_clusters(self, *a, **k):
        raise NotImplementedError('cluster visualisation needs the music21 corpus: out of scope (SURVEY.md section 2, #4)')

    show_nn_clusters = scatterplot_clusters_3d = plot_clusters


class EncoderTrainer(nn.Module):
    """Epoch loop, checkpoints, logging (encoder.py:216-325)."""

    def __init__(self, dataloader_generator):
        super().__init__()
        self.dataloader_generator = dataloader_generator
        self.writer = None

    def train_model(self, batch_size, num_batches, num_epochs, lr, corrupt_labels, schedule_lr, plot=False, num_workers=0,
                    **kwargs):
        if plot:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(f'{self.model_dir}')
            except Exception:                  # tensorboard is optional (not installed in the ROCm image)
                self.writer = None
        best_val = 1e8
        from . import hip
        from . import ops
        mode_before, arith_before = hip.gemm_mode_state(), ops.gradient_arithmetic_state()
        if hasattr(self, 'use_training_defaults'):
            self.use_training_defaults()           # bf16x6 GEMMs + step-graph replay unless the caller chose otherwise
        self.trained_gemm_mode = hip.get_gemm_mode()        # what the epochs below run in (0 fp32 MFMA / 1 bf16x6 / 2 bf16)
        try:
            return self._train_epochs(batch_size, num_batches, num_epochs, lr, corrupt_labels, schedule_lr, plot, num_workers,
                                      best_val)
        finally:
            # the GEMM arithmetic is a process-wide setting: a caller who chose nothing gets back what was there before
            # (evaluation / generation code that runs after training sees the mode it would have seen without it)
            hip.restore_gemm_mode_state(mode_before)
            ops.restore_gradient_arithmetic_state(arith_before)

    def _train_epochs(self, batch_size, num_batches, num_epochs, lr, corrupt_labels, schedule_lr, plot, num_workers, best_val):
        self.init_optimizers(lr=lr, schedule_lr=schedule_lr)
        history = []
        for epoch_id in range(num_epochs):
            gen_train, gen_val, _ = self.dataloader_generator.dataloaders(batch_size=batch_size, num_workers=num_workers)
            train = self.epoch(data_loader=gen_train, train=True, num_batches=num_batches, corrupt_labels=corrupt_labels)
            del gen_train
            val = self.epoch(data_loader=gen_val, train=False,
                             num_batches=num_batches // 2 if num_batches is not None else None,
                             corrupt_labels=corrupt_labels)
            del gen_val
            if getattr(self, 'is_main', True):
                print(f'======= Epoch {epoch_id} =======')
                print('---Train---')
                dict_pretty_print(train, endstr=' ' * 5)
                print()
                print('---Val---')
                dict_pretty_print(val, endstr=' ' * 5)
                print('\n')
                self.save(early_stopped=False)
                if val['loss_monitor'] < best_val:
                    self.save(early_stopped=True)
                    best_val = val['loss_monitor']
                if plot and self.writer is not None:
                    self.plot(epoch_id, train, val)
            history.append((train, val))
        return history

    def plot(self, epoch_id, monitored_quantities_train, monitored_quantities_val, index_encoder=None):
        suffix = f'_{index_encoder})' if index_encoder is not None else ''
        for tag, quantities in (('train', monitored_quantities_train), ('val', monitored_quantities_val)):
            if quantities is None:
                continue
            for k, v in quantities.items():
                if isinstance(v, list):
                    for ind, elem in enumerate(v):
                        self.writer.add_scalar(f'{k}_{ind}{suffix}/{tag}', elem, epoch_id)
                else:
                    self.writer.add_scalar(f'{k}{suffix}/{tag}', v, epoch_id)
