"""StudentEncoderTrainer -- distilled VQ-VAE training (reference: VQCPCB/student_encoder_trainer.py:13-293;
SURVEY.md section 8 row A23, BASELINE configs[3]).

One iteration of `epoch()`:
  1. one masked event index per batch from the host generator (:159-160, no device sync), events [m-k, m+k] replaced by
     the mask token of their voice;
  2. teacher forward on the masked tokens, hard-target cross-entropy on the masked event (:120-142);
  3. encoder (linear-aggregation downscaler + VQ) -> auxiliary decoder, soft-target cross-entropy against the detached
     teacher logits of the masked event, + weighted quantisation loss (:186-218);
  4. ONE backward over both (disjoint) graphs, ONE RCCL all-reduce of the flat gradient, then three global-norm clips
     (teacher | auxiliary decoder | encoder, :252, :268-269) + Adam as flat kernels.  The reference steps the teacher
     before the encoder/decoder forward; the latter only reads teacher logits computed before that step, so deferring
     the teacher update to the end of the iteration gives the same numbers.  Adam is element-wise: the reference's
     single Adam over decoder + encoder parameters (:51-55) equals one Adam per clip group with a shared step count.
  5. metrics accumulate on the device; the reference's five `.item()` syncs per step (:137, :211-215) and the
     384 `.item()` calls of its loss loop (utils.py:155) disappear.
"""
import os
from itertools import islice

import numpy as np
import torch

from . import ops
from .encoder import EncoderTrainer
from .graphs import GraphedTraining
from .utils import SEEDS, STEP_LOCK
from .parallel import DataParallelContext, FlatParameters
from .vqcpc_encoder_trainer import VQCPCEncoderTrainer


class StudentEncoderTrainer(GraphedTraining, EncoderTrainer):
    def __init__(self, model_dir, dataloader_generator, encoder, num_events_masked, teacher, auxiliary_decoder,
                 quantization_weighting, num_gpus=1):
        super().__init__(dataloader_generator=dataloader_generator)
        self.model_dir = model_dir
        self.dataloader_generator = dataloader_generator
        self.encoder = encoder
        self.teacher = teacher
        self.auxiliary_decoder = auxiliary_decoder
        self.codebook_dim = self.encoder.quantizer.codebook_dim
        self.upscale_factors = self.auxiliary_decoder.upscale_factors
        self.num_tokens_per_channel = self.encoder.data_processor.num_tokens_per_channel
        self.num_channels = len(self.num_tokens_per_channel)
        self.num_events_masked = num_events_masked
        assert self.encoder.data_processor.num_tokens % np.prod(self.upscale_factors) == 0
        self.quantization_weighting = quantization_weighting
        self.optimizer_enc_dec = None      # (decoder, encoder) clip groups
        self.optimizer_teacher = None
        self.scheduler_enc_dec = None
        self.scheduler_teacher = None
        self.schedule_lr = False
        self.flat = None
        self.dp = None
        self.lr = None
        self.global_step = 0

    # ---- optimiser (:49-76) --------------------------------------------------------------------------------------
    def _modules_with_params(self):
        return [self.teacher, self.auxiliary_decoder, self.encoder]

    lr_lambda = staticmethod(VQCPCEncoderTrainer.lr_lambda)

    def init_optimizers(self, lr, schedule_lr, dp=None):
        dev = next(self.encoder.parameters()).device
        assert dev.type == 'cuda', 'call .to(device) first: the training step has no CPU path'
        self.dp = dp if dp is not None else (self.dp or DataParallelContext(device=dev))
        self.is_main = self.dp.rank == 0
        SEEDS.set_rank(self.dp.rank)            # per-rank dropout masks, whatever the launcher seeded
        self.flat = FlatParameters(self._modules_with_params())
        self.dp.broadcast_(self.flat.flat, src=0)
        if self.dp.distributed:
            self.encoder.quantizer.init_broadcast = lambda tensors: [self.dp.broadcast_(t.data, 0) for t in tensors]
        self.lr, self.schedule_lr = lr, schedule_lr

        def adam(module):
            a, b = self.flat.range_of(module)
            return ops.FlatAdam(self.flat.flat[a:b], self.flat.flat_grad[a:b], lr=lr, max_norm=5.0)

        self.optimizer_teacher = adam(self.teacher)
        self.optimizer_enc_dec = (adam(self.auxiliary_decoder), adam(self.encoder))
        self.scheduler_enc_dec = self.scheduler_teacher = self.lr_lambda if schedule_lr else None
        self.global_step = 0
        st = getattr(self, '_resume_state', None)      # extension: Adam moments + schedule position survive a resume
        if st is not None and st['m'].numel() == self.flat.numel:
            for opt in (self.optimizer_teacher,) + self.optimizer_enc_dec:
                opt.step_count = int(st['step'])
            m, v = st['m'].to(dev), st['v'].to(dev)
            for opt, mod in zip((self.optimizer_teacher,) + self.optimizer_enc_dec, self._modules_with_params()):
                a, b = self.flat.range_of(mod)
                opt.m.copy_(m[a:b])
                opt.v.copy_(v[a:b])
            self.global_step = int(st['global_step'])
            self.restore_dropout_stream(st.get('dropout_stream'))
            self._resume_state = None

    def current_lr(self):
        return self.lr * (self.lr_lambda(self.global_step) if self.schedule_lr else 1.0)

    def to(self, device):
        for m in self._modules_with_params():
            m.to(device)
        return self

    # ---- checkpoints (:85-107): encoder files + `decoder` + `teacher` -------------------------------------------------
    def _dir(self, early_stopped):
        return f'{self.model_dir}/early_stopped' if early_stopped else f'{self.model_dir}/overfitted'

    def save(self, early_stopped):
        model_dir = self._dir(early_stopped)
        os.makedirs(model_dir, exist_ok=True)
        self.encoder.save(early_stopped=early_stopped)
        torch.save(self.auxiliary_decoder.state_dict(), f'{model_dir}/decoder')
        torch.save(self.teacher.state_dict(), f'{model_dir}/teacher')
        if self.optimizer_teacher is not None:      # extension: the reference drops optimiser state on resume
            opts = (self.optimizer_teacher,) + self.optimizer_enc_dec
            torch.save(dict(m=torch.cat([o.m for o in opts]), v=torch.cat([o.v for o in opts]),
                            step=self.optimizer_teacher.step_count, global_step=self.global_step, dropout_stream=self.dropout_stream_state()),
                       f'{model_dir}/optimizer')

    def load(self, early_stopped, device):
        print(f'Loading models {self.__repr__()}')
        print(f'Loading from {self.model_dir}')
        model_dir = self._dir(early_stopped)
        ml = torch.device(device)
        self.encoder.load(early_stopped=early_stopped, device=device)
        self.auxiliary_decoder.load_state_dict(torch.load(f'{model_dir}/decoder', map_location=ml))
        self.teacher.load_state_dict(torch.load(f'{model_dir}/teacher', map_location=ml))
        opt = f'{model_dir}/optimizer'
        self._resume_state = torch.load(opt, map_location=ml) if os.path.exists(opt) else None

    def train(self, mode=True):
        for m in self._modules_with_params():
            m.train(mode)
        return self

    def eval(self):
        return self.train(False)

    # ---- masking (:144-184) --------------------------------------------------------------------------------------
    def draw_masked_event(self, num_events):
        """One index per batch from the global CPU generator, exactly the reference's draw (:159-160)."""
        return int(torch.randint(high=num_events, size=()).item())

    def mask_teacher(self, x, num_events_masked, masked_event_index=None):
        """x (batch, num_events, num_channels) -> (masked_x, notes_to_be_predicted), same shapes: the tokens of events
        [m - k, m + k] become the mask token V_c of their voice, notes_to_be_predicted marks event m."""
        B, E, C = x.shape
        assert C == self.num_channels
        m = self.draw_masked_event(E) if masked_event_index is None else int(masked_event_index)
        lo, hi = max(m - num_events_masked, 0), min(m + num_events_masked + 1, E)
        masked_x = x.clone()
        mt = getattr(self, '_mask_tokens', None)            # cached: a host -> device copy is not capturable (graphs.py)
        if mt is None or mt.device != x.device or mt.dtype != x.dtype:
            mt = self._mask_tokens = torch.tensor(self.num_tokens_per_channel, dtype=x.dtype, device=x.device).view(1, 1, C)
        masked_x[:, lo:hi] = mt
        notes = torch.zeros_like(x)
        notes[:, m] = 1
        self._last_masked_event = m
        return masked_x, notes

    # ---- the two forward halves ------------------------------------------------------------------------------------
    def forward_teacher(self, x, masked_event_index=None):
        """:120-142.  `weights_per_category` holds the logits of the masked event only, (batch, V_c) per voice."""
        masked_x, notes = self.mask_teacher(x, self.num_events_masked, masked_event_index)
        m = self._last_masked_event
        logits = self.teacher.forward_events(masked_x, m)
        loss = sum(ops.SoftmaxCEFn.apply(lg, x[:, m, c], None) for c, lg in enumerate(logits)).mean()
        return {'loss': loss, 'notes_to_be_predicted': notes, 'weights_per_category': logits, 'masked_event_index': m,
                'monitored_quantities': {'loss_teacher': loss.detach()}}

    def _encode_decode(self, x, m):
        """Encoder + auxiliary decoder up to the student logits of event m (independent of the teacher)."""
        z_quantized, encoding_indices, quantization_loss = self.encoder(x)
        logits = self.auxiliary_decoder.forward_events(z_quantized, m)
        return encoding_indices, quantization_loss, logits

    def _encdec_losses(self, encoding_indices, quantization_loss, logits, weights_per_category_teacher):
        rec = sum(ops.SoftmaxCEFn.apply(lg, None, t.detach()) for lg, t in zip(logits, weights_per_category_teacher))
        q_mean, rec_mean = quantization_loss.mean(), rec.mean()
        loss = self.quantization_weighting * q_mean + rec_mean
        return {'loss': loss, 'encoding_indices': encoding_indices, 'weights_per_category': logits,
                'monitored_quantities': {'loss_quantization': q_mean.detach(), 'loss_reconstruction': rec_mean.detach(),
                                         'loss_encdec': loss.detach(), 'loss_monitor': rec_mean.detach()}}

    def forward_encdec(self, x, weights_per_category_teacher, notes_to_be_predicted, masked_event_index=None):
        """:186-218 with the teacher / student logits restricted to the masked event."""
        m = self._last_masked_event if masked_event_index is None else masked_event_index
        return self._encdec_losses(*self._encode_decode(x, m), weights_per_category_teacher)

    overlap_streams = True      # teacher and encoder/decoder are independent graphs: run them on two HIP streams

    def compute_losses(self, tensor_dict, masked_event_index=None):
        x = self.teacher.data_processor.checked(self.teacher.data_processor.preprocess(tensor_dict['x']))
        m = self.draw_masked_event(x.shape[1]) if masked_event_index is None else int(masked_event_index)
        if self.overlap_streams and x.is_cuda:
            # at the reference's batch of 8 every GEMM has only 3072 rows and fills a fraction of the 256 CUs: the two
            # independent halves of the step (teacher | encoder + decoder) run concurrently, forward and -- because
            # autograd replays every node on the stream of its forward -- backward
            main = torch.cuda.current_stream()
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = torch.cuda.Stream()
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                enc = self._encode_decode(x, m)
            t = self.forward_teacher(x, m)
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():     # a graph's private pool is not recycled across streams
                for tensor in (enc[0], enc[1], *enc[2]):
                    if tensor is not None:
                        tensor.record_stream(main)
            e = self._encdec_losses(*enc, t['weights_per_category'])
        else:
            t = self.forward_teacher(x, m)
            e = self.forward_encdec(x, t['weights_per_category'], t['notes_to_be_predicted'], m)
        out = dict(t['monitored_quantities'], **e['monitored_quantities'])
        out.update(masked_event_index=t['masked_event_index'], encoding_indices=e['encoding_indices'],
                   teacher_logits=[lg.detach() for lg in t['weights_per_category']],
                   student_logits=[lg.detach() for lg in e['weights_per_category']])     # detached: `out` must not keep the step's autograd graph alive
        return t['loss'], e['loss'], out

    def _step_compute(self, tensor_dict, masked_event_index=None):
        m = self._graph_m if masked_event_index is None else masked_event_index
        # (round 6: the forward products of the step inside ops.forward_arithmetic, as in the CPC trainer -- their f16x3 scale table
        # and the weights' fp16 planes for this step; the masked event index selects rows, not shapes: one table serves every index)
        with torch.enable_grad(), ops.forward_arithmetic(self.flat):      # whatever the caller's ambient grad mode: this IS the training step
            loss_teacher, loss_encdec, out = self.compute_losses(tensor_dict, m)
        self.flat.zero_grad()
        with ops.direct_weight_gradients(self.flat):
            (loss_teacher + loss_encdec).backward()          # disjoint graphs: the teacher logits are detached
        return out

    def _step_apply(self, out):
        lr, scale = self.current_lr(), 1.0 / self.dp.world_size
        self.optimizer_teacher.step(lr=lr, grad_scale=scale)
        for opt in self.optimizer_enc_dec:
            opt.step(lr=lr, grad_scale=scale)
        return out

    def _train_step_body(self, tensor_dict, masked_event_index=None):
        if self._dp_bucketed():
            out = self._step_compute_teacher(tensor_dict, masked_event_index)
            self._all_reduce_teacher_async()
            try:
                out = self._step_compute_encdec(out)
            except BaseException:
                self._join_teacher_work()     # never leave the asynchronous collective dangling behind an exception
                raise
            self._all_reduce_encdec_and_join()
            return self._step_apply(out)
        out = self._step_compute(tensor_dict, masked_event_index)
        self._all_reduce_gradients()
        return self._step_apply(out)

    # ---- bucketed data-parallel step (bucketed + overlapped all-reduce) ---------------------------
    # The flat gradient of this step is 327 MB at C3 (82 M parameters): one blocking all-reduce of it behind the backward
    # pass is ~7 ms of xGMI time against an 11.5 ms step.  The teacher's graph and the encoder + decoder's are disjoint (the
    # teacher logits are detached), and their parameters are two contiguous ranges of the flat buffers: with several ranks
    # the step is  [teacher forward + backward]  ->  all-reduce(teacher range), asynchronously on RCCL's own stream  ->
    # [encoder / decoder forward + backward, under that collective]  ->  all-reduce(rest), join  ->  clip + Adam.
    # Price: the two halves no longer fill the GPU side by side on two streams (compute_losses' overlap_streams).
    # OPT-IN (VQCPC_DP_BUCKETS=2); the default is the single all-reduce.  UNMEASURED: no multi-GPU node was available to the
    # builder, so the form that has never met RCCL on two physical GPUs is not what a distributed run gets by default; on the
    # two-ranks-on-one-GPU harness (gloo) the replicas stay bit-identical and equal the single-call form's gradients.
    def _dp_bucketed(self):
        return self.dp is not None and self.dp.distributed and os.environ.get('VQCPC_DP_BUCKETS', '1') == '2'

    _teacher_range_cache = None

    def _teacher_range(self):
        if self._teacher_range_cache is None or self._teacher_range_cache[0] is not self.flat:
            self._teacher_range_cache = (self.flat, self.flat.range_of(self.teacher))   # once per flat buffer, not per step
        return self._teacher_range_cache[1]

    def _step_compute_teacher(self, tensor_dict, masked_event_index=None):
        m = self._graph_m if masked_event_index is None else masked_event_index
        x = self.teacher.data_processor.checked(self.teacher.data_processor.preprocess(tensor_dict['x']))
        m = self.draw_masked_event(x.shape[1]) if m is None else int(m)
        self.flat.zero_grad()
        with torch.enable_grad(), ops.forward_arithmetic(self.flat, tag='teacher'):
            t = self.forward_teacher(x, m)
        with ops.direct_weight_gradients(self.flat, tag='teacher'):       # own f16x3 scale table: two backward passes per step
            t['loss'].backward()
        return dict(x=x, m=m, teacher_logits=[lg.detach() for lg in t['weights_per_category']],
                    monitored=dict(t['monitored_quantities']))

    def _step_compute_encdec(self, st):
        with torch.enable_grad(), ops.forward_arithmetic(self.flat, tag='encdec'):
            e = self._encdec_losses(*self._encode_decode(st['x'], st['m']), st['teacher_logits'])
        with ops.direct_weight_gradients(self.flat, tag='encdec'):
            e['loss'].backward()
        out = dict(st['monitored'], **e['monitored_quantities'])
        out.update(masked_event_index=st['m'], encoding_indices=e['encoding_indices'], teacher_logits=st['teacher_logits'],
                   student_logits=[lg.detach() for lg in e['weights_per_category']])
        return out

    def _all_reduce_teacher_async(self):
        import torch.distributed as dist
        a, b = self._teacher_range()
        self._teacher_work = dist.all_reduce(self.flat.flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True)

    def _all_reduce_encdec_and_join(self):
        import torch.distributed as dist
        a, b = self._teacher_range()
        g = self.flat.flat_grad
        try:
            if a > 0:
                dist.all_reduce(g[:a], op=dist.ReduceOp.SUM)
            if b < g.numel():
                dist.all_reduce(g[b:], op=dist.ReduceOp.SUM)
        finally:
            self._join_teacher_work()

    def _join_teacher_work(self):
        work, self._teacher_work = getattr(self, '_teacher_work', None), None
        if work is not None:
            work.wait()                       # the current stream waits for the collective (no host block with RCCL)

    def _dp_stages(self, parts):
        if self._dp_bucketed():
            return ([self._step_compute_teacher, self._step_compute_encdec, self._step_apply],
                    [self._all_reduce_teacher_async, self._all_reduce_encdec_and_join])
        return super()._dp_stages(parts)

    _graph_m = None

    def _graph_optimizers(self):
        return [self.optimizer_teacher] + list(self.optimizer_enc_dec)

    def _graph_key(self, batch):
        return self._graph_m           # one captured step per masked event index (it selects rows by host-side slicing)

    def precapture_step_graphs(self, tensor_dict):
        """Captures the step for EVERY masked event index on `tensor_dict`'s shapes (one graph each, one shared memory
        pool), so that no capture happens later inside a training loop.  Call after the warm-up steps."""
        assert self._graph_on and not self.encoder.quantizer_needs_init()
        if self._graph is None:
            self._graph = self._new_step_graph(self._train_step_body, (self._step_compute, self._step_apply))
        self._graph_eager_steps = self.graph_warmup_steps
        for m in range(tensor_dict['x'].shape[1]):
            self._graph_m = m
            if self._graph._signature(tensor_dict) not in self._graph.graphs:
                self._graph.capture(tensor_dict)
        return len(self._graph.graphs)

    def train_step(self, tensor_dict, train=True, masked_event_index=None):
        if not train:
            with STEP_LOCK, torch.no_grad():
                return self.compute_losses(tensor_dict, masked_event_index)[2]
        # the masked event is drawn on the host exactly as the reference does (student_encoder_trainer.py:159-160)
        x = tensor_dict['x']
        self._graph_m = self.draw_masked_event(x.shape[1]) if masked_event_index is None else int(masked_event_index)
        out = None
        with SEEDS.stream_of(self):            # this trainer's own dropout-seed stream (utils.DropoutSeeds.stream_of)
            if not self.encoder.quantizer_needs_init():
                out = self._graphed_step(tensor_dict, self._train_step_body, parts=(self._step_compute, self._step_apply))
            if out is None:
                out = self._train_step_body(tensor_dict, self._graph_m)
        self.global_step += 1
        return out

    KEYS = ('loss_teacher', 'loss_quantization', 'loss_reconstruction', 'loss_encdec', 'loss_monitor')

    def epoch(self, data_loader, train=True, num_batches=None, corrupt_labels=False):
        assert self.optimizer_teacher is not None, 'call init_optimizers(lr, schedule_lr) first'
        self.train() if train else self.eval()
        sums = torch.zeros(len(self.KEYS), dtype=torch.float32, device=self.flat.flat.device)
        n = 0
        for tensor_dict in islice(data_loader, num_batches):
            out = self.train_step(tensor_dict, train=train)
            sums += torch.stack([out[k] for k in self.KEYS])
            n += 1
        sums /= max(n, 1)
        if self.dp.distributed:
            self.dp.all_reduce_sum_(sums)
            sums /= self.dp.world_size
        means = dict(zip(self.KEYS, sums.cpu().tolist()))     # the host sync of the epoch
        self.teacher.data_processor.raise_if_bad_tokens(dp=self.dp)
        self.encoder.data_processor.raise_if_bad_tokens(dp=self.dp)
        if train:
            self._report_scale_saturation(means)
        return means
