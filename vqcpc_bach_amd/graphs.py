"""Whole-step HIP-graph replay of a training step (no counterpart in the reference: its step is a Python loop over eager
PyTorch ops).

A training step of this library is ~700 kernel launches issued from Python (ctypes + autograd): 18 ms of host time per step
at C1, more than the GPU needs at the student step's batch of 8.  `StepGraph` captures ONE step -- zero_grad, forward,
backward, clip, Adam -- into a HIP graph (torch.cuda.CUDAGraph: every launch of the library goes to torch's current
stream, so stream capture records them) and replays it per batch:

  * inputs: the batch is copied into static device buffers (the only per-step device work issued from the host, besides
    a 4-byte learning-rate copy);
  * dropout: seeds are kernel ARGUMENTS and therefore frozen in the graph; the graph's first node
    (`vqcpc_rng_salt_advance`) increments a device-side step counter and derives a fresh salt that every RNG call XORs
    into its seed (csrc/common.h), so masks still change every step;
  * Adam: `vqcpc_adam_step_dev` reads the learning rate and the step count (bias corrections) from device memory;
  * outputs (losses, accuracy, code indices) are the graph's static output tensors.

Multi-rank runs replay TWO graphs per step around ONE eagerly issued RCCL all-reduce of the flat gradient bucket:
graph 1 = zero_grad, forward, backward (and the step's metric vector), then `dist.all_reduce` on the same stream, then
graph 2 = clip + Adam.  The collective itself is not captured (`VQCPC_DP_GRAPH=capture` records it inside a single graph
instead -- RCCL supports stream capture -- and `VQCPC_DP_GRAPH=off` keeps multi-rank steps eager): what a graph buys is the
~18 ms of host launch work per step, and that is all in the two halves; a stream-ordered collective between two replays
costs the host one call.  So the data-parallel step is the same captured step that the single-GPU benchmark replays.

Steps whose batch shapes differ from the captured ones, evaluation steps and label-corruption steps run eagerly, exactly
as before.

Pitfall met on ROCm 7.2 / torch 2.10: ending a capture while the PREVIOUS eager step's autograd graph is still alive
(e.g. a step output that was not detached and is still referenced by the caller) segfaults inside capture_end; every
output of the trainers' steps is detached for that reason.
"""
import os

import torch
import torch.distributed

from . import hip


def dp_graph_mode():
    """How a multi-rank step is replayed: 'split' (two graphs around an eager all-reduce, default), 'capture' (one graph,
    the RCCL all-reduce recorded in it) or 'off' (eager multi-rank steps)."""
    mode = os.environ.get('VQCPC_DP_GRAPH', 'split')
    assert mode in ('split', 'capture', 'off'), mode
    return mode


class StepGraph:
    def __init__(self, step_fn, optimizers, lr_fn, device, key_fn=None, seed_base=0x5EED5A17, between_fn=None,
                 finish_fn=None, stages=None, betweens=None):
        """step_fn(batch_dict) -> outputs (tensor / dict / tuple of tensors): one full training step on device tensors.
        optimizers: the ops.FlatAdam objects the step uses; lr_fn() -> current learning rate (host float);
        key_fn(batch) -> extra hashable that selects the graph (the student step's masked event index).
        Two-graph form (multi-rank): step_fn records everything up to the gradients, `between_fn()` runs EAGERLY between the
        two replays (the all-reduce of the gradient bucket), finish_fn(outputs of step_fn) -> outputs records the rest."""
        self.step_fn, self.optimizers, self.lr_fn, self.key_fn = step_fn, list(optimizers), lr_fn, key_fn
        self.between_fn, self.finish_fn = between_fn, finish_fn
        assert (between_fn is None) == (finish_fn is None)
        # General form: `stages` = [f0, f1, .., fk] (f0(batch) -> outputs, fi(outputs) -> outputs), one graph each, and
        # `betweens` = [b0, .., b(k-1)] run EAGERLY between consecutive replays (bucketed all-reduces).  The two-graph form
        # above is stages = [step_fn, finish_fn], betweens = [between_fn].
        if stages is not None:
            assert step_fn is None and between_fn is None and len(betweens) == len(stages) - 1 >= 1
            self.step_fn, self.finish_fn, self.between_fn = stages[0], stages[-1], betweens[0]
            self.stages, self.betweens = list(stages), list(betweens)
        elif finish_fn is not None:
            self.stages, self.betweens = [step_fn, finish_fn], [between_fn]
        else:
            self.stages, self.betweens = [step_fn], []
        self.device = torch.device(device)
        self.seed_base = int(seed_base)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)       # device step counter (uint64 bits)
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._lr_set = None         # the value lr_dev holds (as far as the stream order is concerned)
        self.graphs = {}            # key -> (CUDAGraph, static inputs, static outputs)
        self.pool = None
        self.replays = 0
        self._counter_host = None   # host mirror of the device step counter
        for opt in self.optimizers:
            opt.use_device_scalars(self.lr_dev, self.counter)

    def _signature(self, batch):
        # tensors by shape / dtype; everything else (flags, python scalars) BY VALUE: it is baked into the captured launches
        sig = tuple((k, tuple(v.shape), v.dtype) if torch.is_tensor(v) else (k, repr(v)) for k, v in sorted(batch.items()))
        return (sig, self.key_fn(batch) if self.key_fn is not None else None)

    def _set_lr(self):
        """The learning rate travels as a kernel ARGUMENT of a fill launch (stream-ordered, no host buffer the next step
        could overwrite while this step's copy is still queued), and only when it changed."""
        lr = float(self.lr_fn())
        if lr != self._lr_set:
            self.lr_dev.fill_(lr)
            self._lr_set = lr

    def _stage(self, i, arg):
        """Stage i of a step.  The salt is process-wide device state: every captured stage that may draw random numbers sets
        it in its first node (stage 0 advances the step counter, later ones re-derive the same salt from it) and puts it
        back to 0 in its last, so that whatever runs after a replay -- an eager step of ANOTHER trainer, an evaluation pass,
        the eager collective between two stages -- sees the seeds it was given."""
        last = len(self.stages) - 1
        if i == 0:
            hip.call('vqcpc_rng_salt_advance', self.counter, self.seed_base)
        elif i < last:
            hip.call('vqcpc_rng_salt_from_counter', self.counter, self.seed_base)
        out = self.stages[i](arg)
        if i < last or last == 0:
            hip.call('vqcpc_rng_salt_set', 0)
        return out

    def capture(self, batch):
        """Records one step on `batch`'s shapes.  Nothing is executed: the caller replays afterwards."""
        static = {k: (torch.empty(v.shape, dtype=v.dtype, device=self.device) if torch.is_tensor(v) else v)
                  for k, v in batch.items()}
        graph = torch.cuda.CUDAGraph()
        counts = [opt.step_count for opt in self.optimizers]
        torch.cuda.synchronize()
        # only calls of THIS thread may invalidate the capture: with a process group alive its watchdog thread polls events while we
        # capture, and a second trainer stepped from another Python thread (utils.STEP_LOCK keeps its STEPS out of this capture)
        # still synchronises its own stream and copies its batches between its steps -- under the 'global' mode such a call, landing
        # inside this thread's capture window, invalidated the capture (round 6: one run in three of the two-threads test)
        mode = 'thread_local'
        graphs = [graph]
        try:
            with torch.cuda.graph(graph, pool=self.pool, capture_error_mode=mode):
                out = self._stage(0, static)
            if self.pool is None:
                self.pool = graph.pool()
            for i in range(1, len(self.stages)):         # later stages: same memory pool, replayed right after the previous one
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode=mode):
                    out = self._stage(i, out)
                graphs.append(g)
        finally:
            for opt, c in zip(self.optimizers, counts):  # capture ran the Python side of optimizer.step(): undo its count
                opt.step_count = c
        graph2 = graphs[1] if len(graphs) > 1 else None
        entry = (tuple(graphs), static, out)
        self.graphs[self._signature(batch)] = entry
        return entry

    def __call__(self, batch):
        entry = self.graphs.get(self._signature(batch))
        if entry is None:
            entry = self.capture(batch)
        graphs, static, out = entry
        for k, v in batch.items():
            if torch.is_tensor(v):
                static[k].copy_(v, non_blocking=True)
        # the device step counter (Adam's t, the salt sequence) continues where the eager steps left off
        want = self.optimizers[0].step_count
        if want != self._counter_host:
            self.counter.fill_(int(want))
        self._set_lr()
        graphs[0].replay()
        for g, between in zip(graphs[1:], self.betweens):
            between()
            g.replay()
        for opt in self.optimizers:
            opt.step_count += 1
        self._counter_host = want + 1
        self.replays += 1
        return out

    def release(self):
        for opt in self.optimizers:
            opt.use_device_scalars(None, None)
        self.graphs.clear()
        hip.call('vqcpc_rng_salt_set', 0)


class GraphedTraining:
    """Mixin of the trainers: `enable_step_graph()` makes `train_step(train=True)` replay a captured step after a few
    eager steps (lazy initialisations -- codebook data init, kernel attributes, allocator -- happen there)."""

    graph_warmup_steps = 2
    _graph_on = False
    _graph = None
    _graph_eager_steps = 0

    _graph_explicit = False

    def use_training_defaults(self):
        """Called by `train_model()` (the reference's entry point, encoder.py:244): a caller who chose nothing gets the
        configuration bench.py measures -- bf16x6 GEMM arithmetic, f16x3 gradient GEMMs inside backward, step-graph replay
        (`VQCPC_STEP_GRAPH=0`, `VQCPC_GEMM_MODE=...`, `VQCPC_GRAD_ARITH=six`, `enable_step_graph(False)`, `hip.set_gemm_mode()` or
        `ops.set_gradient_arithmetic()` beforehand override it)."""
        hip.use_training_default_gemm_mode()
        from . import ops
        ops.use_training_default_gradient_arithmetic()      # f16x3 gradient GEMMs (active in the bf16x6 mode only)
        if not self._graph_explicit and os.environ.get('VQCPC_STEP_GRAPH', '1') != '0':
            self.enable_step_graph(True)
            self._graph_explicit = False

    def _report_scale_saturation(self, means):
        """End of a TRAINING epoch, every trainer (the host has just synchronised for the metric means): the f16x3 scale tables of
        this trainer's flat parameters are asked whether a tensor outgrew the 16-32 x head-room of its previous-step scale (its
        largest elements were clamped to 65504 / scale for that ONE step -- in a forward product that can move a loss or a code
        assignment of that step; the next step already runs under the followed scale).  `means['f16x3_scale_saturations']` = the
        number of (call site, operand) pairs it happened to during THIS epoch (0.0 in every run of this repository); when non-zero
        the marked step indices are logged through `warnings` and kept in `self.scale_saturation_log`."""
        from . import ops
        flat = getattr(self, 'flat', None)
        if flat is None or not getattr(flat, '_grad_scales', None):
            return means
        total = ops.scale_saturations(flat)
        seen = getattr(self, '_scale_saturations_seen', 0)
        means['f16x3_scale_saturations'] = float(total - seen)
        if total > seen:
            import warnings
            rep = ops.scale_saturation_report(flat)
            self.scale_saturation_log = rep
            warnings.warn(f'f16x3 GEMM arithmetic: {total - seen} operand tensors outgrew the fp16 range under their previous-step scale '
                          f'during this epoch (clamped for one step each; marked steps by scale table: '
                          f'{ {k: v["step_indices"] for k, v in rep.items()} }); '
                          'ops.set_gradient_arithmetic("six") / ops.set_forward_arithmetic("six") select the scale-free arithmetic')
            self._scale_saturations_seen = total
        return means

    def seed_dropout(self, base):
        """Re-seeds THIS trainer's dropout-seed stream (utils.DropoutSeeds.stream_of): the per-trainer counterpart of
        SEEDS.manual_seed(), which only reaches trainers that have not taken a step yet."""
        self._dropout_stream = (int(base) & 0xFFFFFFFF, 0)      # two fields: immune to SEEDS.manual_seed() generations

    def dropout_stream_state(self):
        """(base, counter) of this trainer's dropout-seed stream, for checkpoints (None before the first step)."""
        st = getattr(self, '_dropout_stream', None)
        return None if st is None else (int(st[0]), int(st[1]))

    def restore_dropout_stream(self, state):
        if state is not None:
            self._dropout_stream = (int(state[0]) & 0xFFFFFFFF, int(state[1]))

    def enable_step_graph(self, enabled=True):
        self._graph_on = bool(enabled)
        self._graph_explicit = True
        if not enabled and self._graph is not None:
            self._graph.release()
            self._graph = None
        return self

    def _graph_optimizers(self):
        raise NotImplementedError

    def _graph_key(self, batch):
        return None

    def _all_reduce_gradients(self):
        self.dp.all_reduce_sum_(self.flat.flat_grad)

    def _dp_stages(self, parts):
        """(stages, betweens) of the multi-rank step: by default [compute, apply] around ONE all-reduce of the flat gradient
        bucket.  A trainer whose step has independent halves overrides this (student: bucketed all-reduces)."""
        return [parts[0], parts[1]], [self._all_reduce_gradients]

    def _new_step_graph(self, body, parts):
        """parts = (compute, apply): the halves of `body` before / after the gradient all-reduce."""
        dev = self.flat.flat.device
        if self.dp.distributed and not (dp_graph_mode() == 'capture' and self.dp.backend == 'nccl'):
            stages, betweens = self._dp_stages(parts)
            return StepGraph(None, self._graph_optimizers(), self.current_lr, dev, key_fn=self._graph_key, stages=stages,
                             betweens=betweens)
        return StepGraph(body, self._graph_optimizers(), self.current_lr, dev, key_fn=self._graph_key)

    def _graphed_step(self, batch, body, parts=None):
        """Returns body's outputs from a graph replay, or None when this step has to run eagerly."""
        if not self._graph_on:
            return None
        if self.dp.distributed and (parts is None or dp_graph_mode() == 'off'):
            return None
        if self._graph_eager_steps < self.graph_warmup_steps:
            self._graph_eager_steps += 1
            return None
        if self._graph is None:
            self._graph = self._new_step_graph(body, parts)
        try:
            return self._graph(batch)
        except RuntimeError as e:
            # a failed CAPTURE (e.g. a runtime that cannot record one of the step's calls) must not take the training run
            # down: fall back to eager steps for good and say so once.  Errors of a replayed step are real errors.
            if self._graph.replays > 0:
                raise
            import warnings
            warnings.warn(f'step-graph capture failed ({str(e)[:200]}); continuing with eager steps')
            torch.cuda.synchronize()
            hip.clear_runtime_error()       # the failed capture's HIP error must not be reported by the next kernel launch check
            self._graph_on = False
            self._graph.release()
            self._graph = None
            return None
