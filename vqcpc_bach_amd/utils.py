"""Small helpers with the reference's names (VQCPCB/utils.py:5-21,52-81)."""
import torch


class _StepLock:
    """ONE training / evaluation step of ONE trainer at a time, process-wide (SURVEY.md section 8(b) "re-entrant").  The host layer
    keeps per-step state in module globals (ops: the open gradient scope, its deferred weight gradients, transposed weights and
    f16x3 scale table; this module: the dropout-seed source) and torch.autograd executes every backward node of a device on one
    engine thread, so two trainers stepping from two Python threads would interleave there whatever the callers do; the device
    side has one RNG salt per translation unit that a replayed step graph sets and clears.  The lock serialises the steps on the
    host, and the event it hands from one holder to the next orders them on the device as well (the next holder's stream waits
    for the previous holder's last kernel), so threads with their own HIP streams are safe too.  Re-entrant; a few microseconds
    per step."""

    def __init__(self):
        import threading
        self._lock = threading.RLock()
        self._depth = 0
        self._event = None

    def __enter__(self):
        self._lock.acquire()
        self._depth += 1
        try:
            if self._depth == 1 and self._event is not None:
                torch.cuda.current_stream().wait_event(self._event)
        except BaseException:                    # __exit__ is not called when __enter__ raises: give the lock back here
            self._depth -= 1
            self._lock.release()
            raise
        return self

    def __exit__(self, *exc):
        try:
            if self._depth == 1 and torch.cuda.is_available() and torch.cuda.is_initialized():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._event = ev
        finally:
            self._depth -= 1
            self._lock.release()
        return False


STEP_LOCK = _StepLock()


class DropoutSeeds:
    """Counter-based seed source for the in-kernel dropout RNG: every dropout site draws a fresh 64-bit seed, so masks
    are reproducible from (base seed, call order) and never stored."""

    def __init__(self, base=0x5EED):
        self.base = int(base) & 0xFFFFFFFF
        self.counter = 0
        self.generation = 0         # bumped by manual_seed(): a trainer whose stream was forked earlier warns once (see stream_of)
        self.rank_salt = 0          # set once per process by the trainers' init_optimizers (data-parallel rank)

    def set_rank(self, rank):
        """Data-parallel replicas must draw DIFFERENT dropout masks (they see different windows of one global batch); the
        salt survives manual_seed(), so seeding every rank alike -- as a launcher script naturally does -- is safe."""
        self.rank_salt = (int(rank) * 0x9E3779B1) & 0xFFFFFFFF

    def manual_seed(self, base):
        """Re-seeds the process-wide source, i.e. every trainer that has NOT taken a step yet.  A trainer that has stepped owns
        its stream (two trainers seeded one after the other in one process must not reach into each other: tests/test_graphs_gpu.py
        ::test_two_trainers_interleaved_...) and keeps it: `trainer.seed_dropout(base)` re-seeds that trainer; its next step
        warns once that this call did not."""
        self.base = int(base) & 0xFFFFFFFF
        self.counter = 0
        self.generation += 1

    def next(self):
        self.counter += 1
        return (((self.base + self.rank_salt) & 0xFFFFFFFF) << 32) + self.counter * 0x10000

    def stream_of(self, owner):
        """Context manager around one training step of `owner` (a trainer): inside it the seeds come from the owner's OWN
        stream, forked from this object's (base, counter) at the owner's FIRST step (seed, build, step -- as with one trainer
        -- gives the sequence it always gave; `trainer.seed_dropout(base)` re-seeds a trainer that has already stepped).
        Two trainers stepping alternately in one process therefore draw exactly the seeds each would draw alone."""
        return _OwnedStream(self, owner)


class _OwnedStream:
    def __init__(self, seeds, owner):
        self.seeds, self.owner = seeds, owner

    def __enter__(self):
        STEP_LOCK.__enter__()                    # one training step at a time, process-wide (see _StepLock)
        try:
            s, st = self.seeds, getattr(self.owner, '_dropout_stream', None)
            outer = (s.base, s.counter)
            if st is not None and len(st) > 2 and st[2] != s.generation and not getattr(self.owner, '_dropout_reseed_warned', False):
                import warnings
                self.owner._dropout_reseed_warned = True
                warnings.warn('SEEDS.manual_seed() was called after this trainer took its first step: the trainer keeps its own '
                              'dropout-seed stream (call trainer.seed_dropout(base) to re-seed it)', stacklevel=3)
            self._gen = st[2] if (st is not None and len(st) > 2) else s.generation     # the generation this stream belongs to
            if st is None:
                st = (s.base, s.counter)
        except BaseException:                    # (warnings as errors, a broken owner): no __exit__ will follow -- release the lock
            STEP_LOCK.__exit__(None, None, None)
            raise
        self.outer = outer
        s.base, s.counter = st[0], st[1]
        return s

    def __exit__(self, *exc):
        s = self.seeds
        self.owner._dropout_stream = (s.base, s.counter, self._gen)
        s.base, s.counter = self.outer
        STEP_LOCK.__exit__(*exc)
        return False


SEEDS = DropoutSeeds()


def current_device():
    """Device of the calling rank.  The reference's cuda_variable (utils.py:5-9) moves to the DEFAULT 'cuda' device;
    with one process per GPU the launcher calls torch.cuda.set_device(local_rank) first, so this is the same thing."""
    if not torch.cuda.is_available():
        raise RuntimeError('vqcpc_bach_amd needs an MI355X: there is no CPU path (the CPU oracle lives in oracle/)')
    return torch.device('cuda', torch.cuda.current_device())


def cuda_variable(tensor):
    return tensor.to(current_device(), non_blocking=True)


def to_numpy(tensor):
    return tensor.detach().to('cpu').numpy()


def dict_pretty_print(d, endstr='\n'):
    for key, value in d.items():
        if isinstance(value, list):
            print(f'{key.capitalize()}: [%s]' % ', '.join(map(str, value)))
        else:
            print(f'{key.capitalize()}: {value:.6}', end=endstr)


def flatten(x):
    """(batch, num_events, num_channels, ...) -> (batch, num_events * num_channels, ...)"""
    size = x.size()
    assert len(size) >= 3
    return x.reshape(size[0], size[1] * size[2], *size[3:])


def unflatten(sequence, num_channels):
    size = sequence.size()
    assert len(size) >= 2 and size[1] % num_channels == 0
    return sequence.reshape(size[0], size[1] // num_channels, num_channels, *size[2:])


def categorical_crossentropy(value, target, mask=None):
    """utils.categorical_crossentropy (reference utils.py:24-49): value = list of (batch, events, V_c) logits,
    target / mask (batch, events, channels); CE on the masked positions, summed over channels.
    API-compatible helper (boolean selection = one host sync); the training step uses the event-only fast path."""
    from . import ops
    total = 0
    for c, logits in enumerate(value):
        sel = mask[..., c].bool()
        total = total + ops.SoftmaxCEFn.apply(logits[sel], target[..., c][sel], None)
    return total


def distilled_categorical_crossentropy(value, target, mask=None):
    """utils.distilled_categorical_crossentropy (reference utils.py:131-159): soft-target CE against teacher logits on
    every (channel, event) whose mask is on for more than half of the batch; -> (batch,)."""
    from . import ops
    total = 0
    for c, (student, teacher) in enumerate(zip(value, target)):
        events = torch.nonzero(mask[:, :, c].float().mean(0) > 0.5).flatten().tolist()
        for e in events:
            total = total + ops.SoftmaxCEFn.apply(student[:, e], None, teacher[:, e].detach())
    return total
