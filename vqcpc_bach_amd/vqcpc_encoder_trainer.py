"""VQCPCEncoderTrainer -- the hot path (reference: VQCPCB/vqcpc_encoder_trainer.py:14-354).

One iteration of `epoch()`:
  1. every block of the batch (negatives, [negatives_back,] left, right: R = B (N Kr [x2] + Kl + Kr) blocks) goes through
     the encoder in ONE pass (Encoder.encode_many);
  2. GRU context on z_left, fused bilinear scores + InfoNCE + hits (vqcpc_nce_*), quantisation loss;
  3. backward (hand-scheduled per layer), ONE RCCL all-reduce of the flat gradient, global-norm clip + Adam as two flat
     kernels (no host synchronisation anywhere in the step);
  4. metrics accumulate on the device and are read once at the end of the epoch (the reference syncs 6+ times a step).
"""
import os
from itertools import islice

import numpy as np
import torch

from . import hip, ops, vqcpc_helper
from .encoder import EncoderTrainer
from .graphs import GraphedTraining
from .utils import SEEDS, STEP_LOCK
from .parallel import DataParallelContext, FlatParameters
from .vqcpc_helper import cpc_scores_and_loss


class VQCPCEncoderTrainer(GraphedTraining, EncoderTrainer):
    def __init__(self, model_dir, dataloader_generator, encoder, c_net_kwargs, quantization_weighting):
        super().__init__(dataloader_generator=dataloader_generator)
        self.model_dir = model_dir
        self.dataloader_generator = dataloader_generator
        self.encoder = encoder
        self.data_processor = encoder.data_processor
        self.codebook_dim = self.encoder.quantizer.codebook_dim
        self.upscale_factors = list(reversed(self.encoder.downscaler.downscale_factors))
        self.num_tokens_per_channel = self.encoder.data_processor.num_tokens_per_channel
        self.num_channels = len(self.num_tokens_per_channel)
        assert self.data_processor.num_tokens % np.prod(self.upscale_factors) == 0
        z_dim = encoder.upscaler.output_dim if encoder.upscaler is not None else self.codebook_dim
        c_dim = c_net_kwargs['output_dim']
        k_max = self.dataloader_generator.num_blocks_right

        def c_net():
            return vqcpc_helper.CModule(input_dim=z_dim, hidden_size=c_net_kwargs['hidden_size'], output_dim=c_dim,
                                        num_layers=c_net_kwargs['num_layers'], dropout=c_net_kwargs['dropout'])

        self.c_module = c_net()
        self.fks_module = vqcpc_helper.FksModule(z_dim=z_dim, c_dim=c_dim, k_max=k_max)
        # the shipped transformer configs omit 'bidirectional' (defect 2): default to False
        if c_net_kwargs.get('bidirectional', False):
            self.c_module_back = c_net()
            self.fks_module_back = vqcpc_helper.FksModule(z_dim=z_dim, c_dim=c_dim, k_max=k_max)
        else:
            self.c_module_back = None
        self.quantization_weighting = quantization_weighting
        self.optimizer = None
        self.scheduler = None
        self.schedule_lr = False
        self.flat = None
        self.dp = None
        self.lr = None
        self.global_step = 0

    # ---- optimiser ---------------------------------------------------------------------------------------------
    def _modules_with_params(self):
        mods = [self.c_module, self.fks_module, self.encoder]
        if self.c_module_back is not None:
            mods += [self.fks_module_back, self.c_module_back]
        return mods

    @staticmethod
    def lr_lambda(step):
        """init_optimizers' LambdaLR factor (:96-107): linear warm-up 0.1 -> 1 over 10 000 steps, then a 10x slower
        linear decay, floored at 0.1."""
        warmup, lo, hi = 10000, 0.1, 1.0
        s1 = (hi - lo) / warmup
        return max(min(lo + s1 * step, hi + (step - warmup) * (-s1 * 0.1)), lo)

    def init_optimizers(self, lr, schedule_lr, dp=None):
        dev = next(self.encoder.parameters()).device
        assert dev.type == 'cuda', 'call .to(device) first: the training step has no CPU path'
        self.dp = dp if dp is not None else (self.dp or DataParallelContext(device=dev))
        self.is_main = self.dp.rank == 0
        SEEDS.set_rank(self.dp.rank)            # per-rank dropout masks, whatever the launcher seeded
        self.flat = FlatParameters(self._modules_with_params())
        self.dp.broadcast_(self.flat.flat, src=0)                       # identical replicas
        if self.dp.distributed:
            self.encoder.quantizer.init_broadcast = lambda tensors: [self.dp.broadcast_(t.data, 0) for t in tensors]
        self.lr, self.schedule_lr = lr, schedule_lr
        self.optimizer = ops.FlatAdam(self.flat.flat, self.flat.flat_grad, lr=lr, max_norm=5.0)
        self.scheduler = self.lr_lambda if schedule_lr else None
        self.global_step = 0
        self._apply_resume_state()

    def _apply_resume_state(self):
        """Extension over the reference (which restarts Adam and the LR schedule on every resume):
        `save` writes `optimizer`; `load` stashes it; it is applied here once the flat buffers exist."""
        st = getattr(self, '_resume_state', None)
        if st is None:
            return
        if st['m'].numel() != self.optimizer.m.numel():
            print('optimizer state ignored: parameter count differs from the checkpoint')
            return
        self.optimizer.m.copy_(st['m'])
        self.optimizer.v.copy_(st['v'])
        self.optimizer.step_count = int(st['step'])
        self.global_step = int(st['global_step'])
        self.restore_dropout_stream(st.get('dropout_stream'))
        self._resume_state = None

    def current_lr(self):
        return self.lr * (self.lr_lambda(self.global_step) if self.schedule_lr else 1.0)

    def to(self, device):
        for m in self._modules_with_params():
            m.to(device)
        return self

    # ---- checkpoints (:117-151) ----------------------------------------------------------------------------------
    def _dir(self, early_stopped):
        return f'{self.model_dir}/early_stopped' if early_stopped else f'{self.model_dir}/overfitted'

    def save(self, early_stopped):
        model_dir = self._dir(early_stopped)
        os.makedirs(model_dir, exist_ok=True)
        self.encoder.save(early_stopped=early_stopped)
        torch.save(self.c_module.state_dict(), f'{model_dir}/c_module')
        torch.save(self.fks_module.state_dict(), f'{model_dir}/fks_module')
        if self.c_module_back is not None:
            torch.save(self.c_module_back.state_dict(), f'{model_dir}/c_module_back')
            torch.save(self.fks_module_back.state_dict(), f'{model_dir}/fks_module_back')
        if self.optimizer is not None:       # extension: the reference drops optimiser state on resume
            torch.save(dict(m=self.optimizer.m, v=self.optimizer.v, step=self.optimizer.step_count,
                            global_step=self.global_step, dropout_stream=self.dropout_stream_state()),
                       f'{model_dir}/optimizer')

    def load(self, early_stopped, device):
        print(f'Loading models {self.__repr__()}')
        model_dir = self._dir(early_stopped)
        if not os.path.exists(model_dir):
            model_dir = self.model_dir
        ml = torch.device(device)
        self.encoder.load(early_stopped=early_stopped, device=device)
        self.c_module.load_state_dict(torch.load(f'{model_dir}/c_module', map_location=ml))
        self.fks_module.load_state_dict(torch.load(f'{model_dir}/fks_module', map_location=ml))
        if self.c_module_back is not None:
            self.c_module_back.load_state_dict(torch.load(f'{model_dir}/c_module_back', map_location=ml))
            self.fks_module_back.load_state_dict(torch.load(f'{model_dir}/fks_module_back', map_location=ml))
        opt = f'{model_dir}/optimizer'
        self._resume_state = torch.load(opt, map_location=ml) if os.path.exists(opt) else None

    def train(self, mode=True):
        for m in self._modules_with_params():
            m.train(mode)
        return self

    def eval(self):
        return self.train(False)

    # ---- one step ------------------------------------------------------------------------------------------------
    def compute_losses(self, tensor_dict, corrupt_labels=False):
        """Forward half of one iteration (:195-307).  Returns (loss, dict of device tensors)."""
        neg = tensor_dict['negative_samples']
        B, N, Kr, num_events, num_channels = neg.shape
        xs = [neg.reshape(B * N * Kr, num_events, num_channels)]
        corrupt = [corrupt_labels]
        bidir = self.c_module_back is not None
        if bidir:
            xs.append(tensor_dict['negative_samples_back'].reshape(B * N * Kr, num_events, num_channels))
            corrupt.append(corrupt_labels)
        xs += [tensor_dict['x_left'], tensor_dict['x_right']]
        corrupt += [False, False]
        enc, ql_rows = self.encoder.encode_many(xs, corrupt, with_loss_rows=True)
        (z_neg, idx_neg, ql_neg) = enc[0]
        (z_left, idx_left, ql_left), (z_right, idx_right, ql_right) = enc[-2], enc[-1]
        zdim = z_neg.shape[-1]
        z_neg = z_neg.reshape(B, N, Kr, -1, zdim)[:, :, :, 0, :]                  # (B, N, K, z), first block (:245)

        c = self.c_module(z_left, h=None)
        contrastive, hits = cpc_scores_and_loss(self.fks_module, c, z_right, z_neg)
        # quantization_loss (vqcpc_helper.py:32-51) sums the per-block losses of EVERY encoder call of the step: one sum over
        # the rows of the merged pass instead of one per segment (and their slice / add backward nodes)
        ql_total = ql_rows.sum()
        n_terms = 3 * B
        hits_back = None
        if bidir:                                                                 # :277-296
            z_nb = enc[1][0]
            # reference quirk kept for parity: the backward negatives are NOT permuted to negative-major before
            # FksModule's raw .view (:286-292), so window b is scored against rows n*B + b of the (B*N, K, z) tensor
            z_nb = z_nb.reshape(B, N, Kr, -1, zdim)[:, :, :, 0, :].reshape(B * N, Kr, zdim)
            z_nb = z_nb.reshape(N, B, Kr, zdim).permute(1, 0, 2, 3)
            c_back = self.c_module_back(z_right.flip(dims=[1]), h=None)
            contrastive_back, hits_back = cpc_scores_and_loss(self.fks_module_back, c_back, z_left, z_nb)
            contrastive = contrastive + contrastive_back
            n_terms = 4 * B                                                       # ql_total already holds the fourth segment
        q_loss = ql_total / n_terms                                               # quantization_loss, helper :32-51
        loss = contrastive + self.quantization_weighting * q_loss
        accuracy = hits.mean(0)
        if hits_back is not None:
            accuracy = (accuracy + hits_back.mean(0)) / 2
        return loss, dict(loss=loss.detach(), loss_contrastive=contrastive.detach(), loss_quantize=q_loss.detach(),
                          accuracy=accuracy, idx_left=idx_left, idx_right=idx_right, idx_negative=idx_neg)

    def _count_codewords(self, *idx_tensors):
        """len(torch.unique(.)) without a host sync: sort + count boundaries.  With a product quantiser
        (num_codebooks > 1, which the reference's epoch() cannot run: SURVEY.md section 0 defect 1) the count is over
        merged code tuples sum_c idx_c K^c -- the number of distinct product codes in use.  Without a quantiser
        (NoQuantization returns encoding_indices None) the counters stay 0, as in the reference (:325)."""
        if any(i is None for i in idx_tensors):
            return torch.zeros((), dtype=torch.float32, device=self.flat.flat.device)
        flat = [i.reshape(-1, i.shape[-1]).contiguous() for i in idx_tensors]
        ncb, K = flat[0].shape[1], int(self.encoder.quantizer.codebook_size)
        if flat[0].is_cuda and len(flat) <= 2 and hip.query('vqcpc_count_distinct_codes_supported', ncb, K):
            # one launch (a bit per possible product code in LDS) instead of cat + sort + compare + sum
            out = torch.empty(1, dtype=torch.float32, device=flat[0].device)
            b = flat[1] if len(flat) == 2 else None
            hip.call('vqcpc_count_distinct_codes', flat[0], flat[0].shape[0], b, 0 if b is None else b.shape[0], ncb, K, out)
            return out[0]
        merged = torch.cat([self.encoder.merge_codes(i) for i in flat])
        s = torch.sort(merged)[0]
        return (s[1:] != s[:-1]).sum().float() + 1.0

    def _step_metrics(self, out):
        """The per-step summands of epoch()'s means as ONE device vector: loss, quantize, contrastive, num_codewords,
        num_codewords_negative, accuracy[k] (vqcpc_encoder_trainer.py:320-340).  Computed inside the step so that a
        captured step (graphs.py) replays it with the rest."""
        return torch.cat([torch.stack([out['loss'], out['loss_quantize'], out['loss_contrastive'],
                                       self._count_codewords(out['idx_left'], out['idx_right']),
                                       self._count_codewords(out['idx_negative'])]), out['accuracy']])

    def _step_compute(self, tensor_dict, corrupt_labels=False):
        """zero_grad / forward / backward (:310-312) + the step's metric vector: everything before the gradient all-reduce."""
        with torch.enable_grad(), ops.forward_arithmetic(self.flat):     # whatever the caller's ambient grad mode: this IS the training step
            loss, out = self.compute_losses(tensor_dict, corrupt_labels)
        self.flat.zero_grad()
        with ops.direct_weight_gradients(self.flat):
            loss.backward()
        out['metrics'] = self._step_metrics(out)
        return out

    def _step_apply(self, out):
        """clip + Adam (:313-316) on the (all-reduced) flat gradient; the rank sum becomes a mean inside the kernels."""
        self.optimizer.step(lr=self.current_lr(), grad_scale=1.0 / self.dp.world_size)
        return out

    def _train_step_body(self, tensor_dict, corrupt_labels=False):
        """Everything a step enqueues on the device: compute, ONE all-reduce of the gradient bucket, apply."""
        out = self._step_compute(tensor_dict, corrupt_labels)
        self._all_reduce_gradients()
        return self._step_apply(out)

    def _graph_optimizers(self):
        return [self.optimizer]

    def train_step(self, tensor_dict, train=True, corrupt_labels=False):
        """One iteration.  Returns device-side metrics.  With `enable_step_graph()` a training step is a HIP-graph replay
        (graphs.py) once the first eager steps have done the lazy initialisations."""
        if not train:
            with STEP_LOCK, torch.no_grad():
                out = self.compute_losses(tensor_dict, corrupt_labels)[1]
                out['metrics'] = self._step_metrics(out)
                return out
        out = None
        with SEEDS.stream_of(self):            # this trainer's own dropout-seed stream (utils.DropoutSeeds.stream_of)
            if not corrupt_labels and not self.encoder.quantizer_needs_init():
                out = self._graphed_step(tensor_dict, self._train_step_body, parts=(self._step_compute, self._step_apply))
            if out is None:
                out = self._train_step_body(tensor_dict, corrupt_labels)
        self.global_step += 1
        return out

    def epoch(self, data_loader, train, num_batches, corrupt_labels):
        assert self.optimizer is not None, 'call init_optimizers(lr, schedule_lr) first'
        dev = self.flat.flat.device
        if self.is_main:
            print(f'lr: {self.current_lr()}')
        self.train() if train else self.eval()
        k_r = self.dataloader_generator.num_blocks_right
        sums = torch.zeros(5 + k_r, dtype=torch.float32, device=dev)   # loss, quantize, contrastive, ncw, ncw_neg, acc[k]
        dproc = self.encoder.data_processor
        n = 0
        for tensor_dict in islice(data_loader, num_batches):
            out = self.train_step(tensor_dict, train=train, corrupt_labels=corrupt_labels)
            sums += out['metrics']
            n += 1
        sums /= max(n, 1) * self.dp.world_size
        flag = dproc.bad_token_flag()
        if flag is not None:
            sums = torch.cat([sums, flag.float()])
        # metrics are means over ranks; the out-of-range-token flag rides in the same all-reduce, so EVERY rank raises
        # together (a rank raising alone would leave the others blocked in their next collective)
        self.dp.all_reduce_sum_(sums)
        host = sums.cpu().tolist()                                    # the only host sync of the epoch
        if flag is not None:
            dproc.raise_if_bad_tokens(host.pop())                     # nn.Embedding's IndexError, one epoch late at most
        means = dict(loss=host[0], accuracy=host[5:], loss_quantize=host[1], loss_contrastive=host[2],
                     num_codewords=host[3], num_codewords_negative=host[4])
        means['loss_monitor'] = -sum(means['accuracy']) / len(means['accuracy'])
        if train:
            self._report_scale_saturation(means)      # graphs.GraphedTraining: means['f16x3_scale_saturations'] + marked steps
        return means
