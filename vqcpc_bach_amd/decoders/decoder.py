"""Decoder training step (reference: VQCPCB/decoders/decoder.py:25-543): a seq2seq
relative transformer that reconstructs the token sequence from the frozen encoder's codes.

Covered: `transformer_type='relative'` with causal target self-attention, anticausal or full source self-attention and
anticausal or full cross-attention -- getters.py decoder_type 'transformer_relative' / 'transformer_relative_fullCross'.
`__init__`, `forward`, `epoch`, `train_model`, `init_optimizers`, `save` / `load` keep the reference's names, argument
meaning, state_dict keys and return contracts.  Generation / re-harmonisation (:552-1062), the absolute-position and
'diagonal' variants are out of scope and raise.

Hot path (`compute_loss`), all numerics in libvqcpc_hip.so:
  * source: `source_embeddings` lookup of the merged codes (gather + deterministic segment-sum gradient);
  * target: `linear_target(cat[embed(x), channel emb, event-in-code emb])` depends only on (token, position in a code
    block), so it is evaluated on the vmax * U table rows (U = total_upscaling) by ONE small GEMM and looked up per
    token -- with the start-of-sentence row appended, the reference's shift-by-one (:474-480) is the same lookup with
    shifted indices;
  * transformer: masked / rectangular relative attention kernels (csrc/relattn_x.hip), fused add + LayerNorm, GEMMs;
  * per-voice output projections on the strided rows of each voice, softmax cross-entropy kernels.

Reference defect fixed: `epoch` (:327-344) passes the quantizer's (B, S, num_codebooks) indices to `forward`, which
raises; the codes are merged with `Encoder.merge_codes` first, as `generate` (:600) does.
"""
import os
from itertools import islice

import numpy as np
import torch
from torch import nn

from .. import ops
from ..graphs import GraphedTraining
from ..parallel import DataParallelContext, FlatParameters
from ..transformer.transformer_custom import (TransformerCustom, TransformerDecoderCustom, TransformerDecoderLayerCustom,
                                              TransformerEncoderCustom, TransformerEncoderLayerCustom, mask_code)
from ..utils import dict_pretty_print, flatten, SEEDS, STEP_LOCK


class HeadsFn(torch.autograd.Function):
    """Per-voice output projections (decoder.py:523-526): logits_c = out[c::nc] W_c^T + b_c on the row-strided view of
    voice c; the input gradient of every voice is written straight into its rows of ONE buffer."""

    @staticmethod
    def forward(ctx, out, nc, *params):
        out = out.contiguous()
        ws, bs = params[0::2], params[1::2]
        logits, padded = [], []
        for c in range(nc):
            N = ws[c].shape[0]
            pn = -N % 4
            wp = torch.nn.functional.pad(ws[c], (0, 0, 0, pn)) if pn else ws[c]
            bp = torch.nn.functional.pad(bs[c], (0, pn)) if pn else bs[c]
            padded.append(wp)
            logits.append(ops.gemm_nt(out[c::nc], wp, bias=bp)[:, :N])
        ctx.save_for_backward(out, *padded)
        ctx.nc = nc
        ctx.sizes = [w.shape[0] for w in ws]
        return tuple(logits)

    @staticmethod
    def backward(ctx, *grads):
        out, *padded = ctx.saved_tensors
        nc = ctx.nc
        d_out = torch.empty_like(out)
        d_params = []
        for c in range(nc):
            N, wp = ctx.sizes[c], padded[c]
            g = grads[c]
            g = torch.nn.functional.pad(g, (0, wp.shape[0] - N)) if wp.shape[0] != N else g.contiguous()
            ops.gemm_nt(g, ops.transpose(wp), out=d_out[c::nc])
            dw, db = ops.gemm_tn(g, out[c::nc])
            d_params += [dw[:N], db[:N]]
        return (d_out, None, *d_params)


class Decoder(GraphedTraining, nn.Module):
    def __init__(self, model_dir, dataloader_generator, data_processor, encoder, transformer_type, encoder_attention_type,
                 cross_attention_type, d_model, num_encoder_layers, num_decoder_layers, n_head, dim_feedforward,
                 positional_embedding_size, num_channels_encoder, num_events_encoder, num_channels_decoder,
                 num_events_decoder, dropout):
        super().__init__()
        if transformer_type != 'relative':
            raise NotImplementedError("transformer_type 'absolute' (learned absolute positions, nn attention without "
                                      'relative bias) is not on the path: SURVEY.md section 8(f) N4 covers the relative decoder')
        assert encoder_attention_type in ['anticausal', 'causal', 'full']
        assert cross_attention_type in ['anticausal', 'causal', 'diagonal', 'full']
        if cross_attention_type == 'diagonal':
            raise NotImplementedError("cross_attention_type 'diagonal' (TransformerAlignedDecoderLayerCustom) is out of scope")
        if cross_attention_type == 'causal':
            raise NotImplementedError                      # as the reference (decoder.py:487-488)
        self.transformer_type = transformer_type
        self.encoder_attention_type = encoder_attention_type
        self.cross_attention_type = cross_attention_type
        self.model_dir = model_dir
        self.encoder = encoder
        self.encoder.eval()                                # frozen (:72-75)
        for p in self.encoder.parameters():
            p.requires_grad = False
        self.dataloader_generator = dataloader_generator
        self.data_processor = data_processor
        self.num_tokens_per_channel = self.data_processor.num_tokens_per_channel
        self.num_channels = len(self.num_tokens_per_channel)
        self.d_model = d_model
        self.num_tokens_target = self.data_processor.num_tokens
        self.total_upscaling = int(np.prod(self.encoder.downscaler.downscale_factors))
        assert self.num_tokens_target % self.total_upscaling == 0
        assert self.num_tokens_target == num_channels_decoder * num_events_decoder
        self.target_channel_embeddings = nn.Parameter(torch.randn((1, self.num_channels, positional_embedding_size)))
        self.num_events_per_code = self.total_upscaling // self.num_channels
        self.target_events_positioning_embeddings = nn.Parameter(
            torch.randn((1, self.num_events_per_code, positional_embedding_size)))
        encoder_layer = TransformerEncoderLayerCustom(d_model=d_model, nhead=n_head,
                                                      attention_bias_type='relative_attention',
                                                      num_channels=num_channels_encoder, num_events=num_events_encoder,
                                                      dim_feedforward=dim_feedforward, dropout=dropout)
        decoder_layer = TransformerDecoderLayerCustom(d_model=d_model, nhead=n_head,
                                                      attention_bias_type_self='relative_attention',
                                                      attention_bias_type_cross='relative_attention_target_source',
                                                      num_channels_encoder=num_channels_encoder,
                                                      num_events_encoder=num_events_encoder,
                                                      num_channels_decoder=num_channels_decoder,
                                                      num_events_decoder=num_events_decoder,
                                                      dim_feedforward=dim_feedforward, dropout=dropout)
        self.transformer = TransformerCustom(
            d_model=self.d_model, nhead=n_head,
            custom_encoder=TransformerEncoderCustom(encoder_layer=encoder_layer, num_layers=num_encoder_layers),
            custom_decoder=TransformerDecoderCustom(decoder_layer=decoder_layer, num_layers=num_decoder_layers))
        self.linear_target = nn.Linear(self.data_processor.embedding_size + positional_embedding_size * 2, self.d_model)
        self.sos = nn.Parameter(torch.randn((1, 1, self.d_model)))
        if type(self.encoder.quantizer).__name__ == 'NoQuantization':
            raise NotImplementedError('continuous (NoQuantization) sources are out of scope')
        codebook_size = self.encoder.quantizer.codebook_size ** self.encoder.quantizer.num_codebooks
        self.source_embeddings = nn.Embedding(codebook_size, self.d_model)
        self.pre_softmaxes = nn.ModuleList([nn.Linear(self.d_model, n) for n in self.num_tokens_per_channel])
        self.num_tokens_source = num_channels_encoder * num_events_encoder
        self.optimizer = None
        self.scheduler = None
        self.dp = None
        self.is_main = True
        self.global_step = 0

    def __repr__(self):
        names = dict(anticausal='AC', causal='C', full='F', diagonal='D')
        return f'Decoder-{self.transformer_type}-{names[self.encoder_attention_type]}-{names[self.cross_attention_type]}'

    # ---- masks: API-compatible matrices (:292-308); the kernels evaluate the same rules from indices ---------------
    def _generate_square_subsequent_mask(self, sz):
        mask = (torch.triu(torch.ones(sz, sz)) == 1).transpose(0, 1)
        return mask.float().masked_fill(mask == 0, float('-inf')).masked_fill(mask == 1, float(0.0))

    def _generate_anticausal_mask(self, sz, sz_tgt=None):
        mask = self._generate_square_subsequent_mask(sz).t()
        if sz_tgt is not None:
            assert sz_tgt % sz == 0
            mask = torch.repeat_interleave(mask, sz_tgt // sz, dim=0)
        return mask.to(self.sos.device)

    def _generate_causal_mask(self, sz):
        return self._generate_square_subsequent_mask(sz).to(self.sos.device)

    # ---- the trainable parameters (the frozen encoder is excluded) -------------------------------------------------
    def _trainable(self):
        return [self.data_processor, self.target_channel_embeddings, self.target_events_positioning_embeddings,
                self.transformer, self.linear_target, self.sos, self.source_embeddings, self.pre_softmaxes]

    @staticmethod
    def lr_lambda(step):
        """LambdaLR factor of init_optimizers (:237-249)."""
        warmup, lo, hi = 10000, 0.1, 1.0
        s1 = (hi - lo) / warmup
        return max(min(lo + s1 * step, hi + (step - warmup) * (-s1 * 0.1)), lo)

    def init_optimizers(self, lr, schedule_lr, dp=None):
        dev = self.sos.device
        assert dev.type == 'cuda', 'call .to(device) first: the training step has no CPU path'
        self.dp = dp if dp is not None else (self.dp or DataParallelContext(device=dev))
        self.is_main = self.dp.rank == 0
        SEEDS.set_rank(self.dp.rank)            # per-rank dropout masks, whatever the launcher seeded
        self.flat = FlatParameters(self._trainable())
        self.dp.broadcast_(self.flat.flat, src=0)
        self.lr, self.schedule_lr = lr, schedule_lr
        self.optimizer = ops.FlatAdam(self.flat.flat, self.flat.flat_grad, lr=lr, max_norm=5.0)
        self.scheduler = self.lr_lambda if schedule_lr else None
        self.global_step = 0
        st = getattr(self, '_resume_state', None)
        if st is not None and st['m'].numel() == self.optimizer.m.numel():
            self.optimizer.m.copy_(st['m'])
            self.optimizer.v.copy_(st['v'])
            self.optimizer.step_count = int(st['step'])
            self.global_step = int(st['global_step'])
            self.restore_dropout_stream(st.get('dropout_stream'))
        self._resume_state = None

    def current_lr(self):
        return self.lr * (self.lr_lambda(self.global_step) if self.schedule_lr else 1.0)

    # ---- checkpoints (:254-274): one file `decoder` holding the whole state_dict, encoder included -------------------
    def _dir(self, early_stopped):
        return f'{self.model_dir}/early_stopped' if early_stopped else f'{self.model_dir}/overfitted'

    def save(self, early_stopped):
        model_dir = self._dir(early_stopped)
        os.makedirs(model_dir, exist_ok=True)
        torch.save(self.state_dict(), f'{model_dir}/decoder')
        if self.optimizer is not None:       # extension: the reference restarts Adam on every resume
            torch.save(dict(m=self.optimizer.m, v=self.optimizer.v, step=self.optimizer.step_count,
                            global_step=self.global_step, dropout_stream=self.dropout_stream_state()), f'{model_dir}/decoder_optimizer')

    def load(self, early_stopped, device):
        print(f'Loading models {self.__repr__()}')
        model_dir = self._dir(early_stopped)
        ml = torch.device(device)
        self.load_state_dict(torch.load(f'{model_dir}/decoder', map_location=ml))
        opt = f'{model_dir}/decoder_optimizer'
        self._resume_state = torch.load(opt, map_location=ml) if os.path.exists(opt) else None

    def train(self, mode=True):
        super().train(mode)
        self.encoder.eval()                  # :319-320: the encoder stays in eval mode
        return self

    # ---- forward ---------------------------------------------------------------------------------------------------
    def _target_rows(self, x):
        """(B, events, channels) int64 on the device -> (B * T, d_model) shifted target rows (see module docstring)."""
        B = x.shape[0]
        nc, U, d = self.num_channels, self.total_upscaling, self.d_model
        tables = self.data_processor.stacked_tables()                                       # (nc, vmax, emb)
        vmax = tables.shape[1]
        dev = x.device
        syn = torch.arange(vmax, device=dev).repeat_interleave(U)                            # table row v * U + u holds token v
        x_table = ops.EmbedPosFn.apply(syn, tables, self.target_channel_embeddings.view(nc, -1),
                                       self.target_events_positioning_embeddings.view(self.num_events_per_code, -1), U)
        tgt_table = ops.linear(x_table, self.linear_target.weight, self.linear_target.bias)  # (vmax * U, d)
        table = torch.cat([tgt_table, self.sos.view(1, d)], dim=0)                           # + start-of-sentence row
        tok = flatten(x)                                                                     # (B, T), t = event * nc + voice
        T = tok.shape[1]
        idx = tok * U + (torch.arange(T, device=dev) % U)
        shifted = torch.cat([torch.full((B, 1), vmax * U, dtype=idx.dtype, device=dev), idx[:, :-1]], dim=1)
        return ops.EmbeddingFn.apply(table, shifted.reshape(-1))

    def compute_loss(self, source, x):
        """source (B, S) merged codes, x (B, events, channels) device int64 -> (loss, logits per voice, attentions)."""
        B, S = source.shape
        nc = self.num_channels
        assert S == self.num_tokens_source and x.shape[1] * nc == self.num_tokens_target
        src = ops.EmbeddingFn.apply(self.source_embeddings.weight, source.reshape(-1))      # (B * S, d)
        tgt = self._target_rows(x)
        out, att_dec, att_enc = self.transformer.forward_rows(
            src, tgt, B, mask_code(self.encoder_attention_type), ops.MASK_CAUSAL, mask_code(self.cross_attention_type))
        params = [t for m in self.pre_softmaxes for t in (m.weight, m.bias)]
        logits = HeadsFn.apply(out, nc, *params)                                            # nc x (B * events, V_c)
        ce = sum(ops.SoftmaxCEFn.apply(lg, x[:, :, c].reshape(-1), None) for c, lg in enumerate(logits))
        loss = ce.mean()                                                                    # :529-535
        E = x.shape[1]
        return loss, [lg.reshape(B, E, -1) for lg in logits], att_dec, att_enc

    def forward(self, source, target):
        """API-compatible `Decoder.forward` (:431-543): source (B, S) merged codes, target (B, events, channels)."""
        target = self.data_processor.checked(self.data_processor.preprocess(target))
        loss, logits, att_dec, att_enc = self.compute_loss(source.to(target.device), target)
        return {'loss': loss, 'attentions_decoder': att_dec, 'attentions_encoder': att_enc,
                'weights_per_category': logits, 'monitored_quantities': {'loss': loss.item()}}

    def encode(self, x):
        """:327-336 + the merge the reference forgot: frozen encoder, inference only -> merged codes (B, S)."""
        return self.encoder.encode_indices(x, merged=True)

    def _step_compute(self, tensor_dict):
        x = self.data_processor.checked(self.data_processor.preprocess(tensor_dict['x']))
        codes = self.encode(tensor_dict['x'])
        # (round 6: the decoder's forward products inside ops.forward_arithmetic -- f16x3 scale table + the weights' fp16 planes of this
        # step; the frozen encoder above is inference and stays outside, on six products)
        with torch.enable_grad(), ops.forward_arithmetic(self.flat):      # whatever the caller's ambient grad mode: this IS the training step
            loss, _, _, _ = self.compute_loss(codes, x)
        self.flat.zero_grad()
        with ops.direct_weight_gradients(self.flat):
            loss.backward()
        return loss.detach()

    def _step_apply(self, loss):
        self.optimizer.step(lr=self.current_lr(), grad_scale=1.0 / self.dp.world_size)       # clip 5 + Adam (:345-346)
        return loss

    def _train_step_body(self, tensor_dict):
        loss = self._step_compute(tensor_dict)
        self._all_reduce_gradients()
        return self._step_apply(loss)

    def _graph_optimizers(self):
        return [self.optimizer]

    def train_step(self, tensor_dict, train=True):
        if not train:
            with STEP_LOCK:
                x = self.data_processor.checked(self.data_processor.preprocess(tensor_dict['x']))
                codes = self.encode(tensor_dict['x'])
                with torch.no_grad():
                    return self.compute_loss(codes, x)[0].detach()
        with SEEDS.stream_of(self):            # this trainer's own dropout-seed stream (utils.DropoutSeeds.stream_of)
            out = self._graphed_step(tensor_dict, self._train_step_body, parts=(self._step_compute, self._step_apply))
            if out is None:
                out = self._train_step_body(tensor_dict)
        self.global_step += 1
        return out

    def epoch(self, data_loader, train=True, num_batches=None):
        assert self.optimizer is not None, 'call init_optimizers(lr, schedule_lr) first'
        self.train() if train else self.eval()
        total = torch.zeros((), dtype=torch.float32, device=self.sos.device)
        n = 0
        for tensor_dict in islice(data_loader, num_batches):
            total += self.train_step(tensor_dict, train=train)
            n += 1
        total /= max(n, 1)
        if self.dp.distributed:
            self.dp.all_reduce_sum_(total)
            total /= self.dp.world_size
        means = {'loss': float(total.item())}                    # the host sync of the epoch
        self.data_processor.raise_if_bad_tokens(dp=self.dp)
        self.encoder.data_processor.raise_if_bad_tokens(dp=self.dp)
        if train:
            self._report_scale_saturation(means)
        return means

    def train_model(self, batch_size, num_batches, num_epochs, lr, schedule_lr, plot=False, num_workers=0, **kwargs):
        from .. import hip, ops
        mode_before, arith_before = hip.gemm_mode_state(), ops.gradient_arithmetic_state()
        self.use_training_defaults()               # bf16x6 GEMMs + step-graph replay unless the caller chose otherwise
        self.trained_gemm_mode = hip.get_gemm_mode()
        try:
            return self._train_epochs(batch_size, num_batches, num_epochs, lr, schedule_lr, num_workers)
        finally:
            hip.restore_gemm_mode_state(mode_before)      # process-wide settings: put back what the caller had (encoder.py)
            ops.restore_gradient_arithmetic_state(arith_before)

    def _train_epochs(self, batch_size, num_batches, num_epochs, lr, schedule_lr, num_workers):
        best_val = 1e8
        self.init_optimizers(lr=lr, schedule_lr=schedule_lr)
        history = []
        for epoch_id in range(num_epochs):
            gen_train, gen_val, _ = self.dataloader_generator.dataloaders(batch_size=batch_size, num_workers=num_workers)
            train = self.epoch(data_loader=gen_train, train=True, num_batches=num_batches)
            del gen_train
            val = self.epoch(data_loader=gen_val, train=False,
                             num_batches=num_batches // 2 if num_batches is not None else None)
            del gen_val
            if self.is_main:
                print(f'======= Epoch {epoch_id} =======')
                print('---Train---')
                dict_pretty_print(train, endstr=' ' * 5)
                print()
                print('---Val---')
                dict_pretty_print(val, endstr=' ' * 5)
                print('\n')
                self.save(early_stopped=False)
                if val['loss'] < best_val:
                    self.save(early_stopped=True)
                    best_val = val['loss']
            history.append((train, val))
        return history

    def generate(self, *a, **k):
        raise NotImplementedError('generation / re-harmonisation (decoder.py:552-1062) is out of scope: SURVEY.md section 2')

    generate_from_code_long = generate_reharmonisation = generate_alla_mano = check_duplicate = compute_start_end_times = \
        init_generation = plot = generate
