"""``build_lr_scheduler`` with the reference's signature (engine/optimizer/scheduler.py:84-143).

The reference composes torch's CosineAnnealingLR / LambdaLR with a warm-up
wrapper; what reaches the optimizer is a pure function of the step index, which
is what this module computes (host side, float64):

    k <  warmup_iter : warmup_lr              if k == 0 or constant warm-up
                       base * k / warmup_iter  (linear warm-up)
    k >= warmup_iter : t = k - warmup_iter
                       cosine: base * (1 + cos(pi t / max_iter)) / 2
                       linear: base * (1 - t / max_iter)

(the successor starts counting only when the warm-up ends, scheduler.py:28-33).
``lr_table(n)`` exposes the same values as an array for the device-resident
multi-step path.
"""
from __future__ import annotations

import math

AVAI_SCHEDS = ["cosine", "linear"]
AVAI_WARMUP_SCHEDS = ["constant", "linear"]


class StepLR:
    def __init__(self, optimizer, kind, max_iter, warmup_iter=0, warmup_type=None, warmup_lr=None):
        self.optimizer = optimizer
        self.kind, self.max_iter = kind, float(max_iter)
        self.warmup_iter, self.warmup_type, self.warmup_lr = int(warmup_iter), warmup_type, warmup_lr
        self.base_lrs = [g.get("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def lr_at(self, k, base):
        if k < self.warmup_iter:
            if self.warmup_type == "constant" or k == 0:
                return float(self.warmup_lr)
            return base * k / self.warmup_iter
        t = k - self.warmup_iter
        if self.kind == "cosine":
            # cosine with period 2*max_iter, as CosineAnnealingLR continues past T_max
            return base * (1.0 + math.cos(math.pi * t / self.max_iter)) / 2.0
        return base * (1.0 - t / self.max_iter)

    def _apply(self):
        self._last_lr = [self.lr_at(self.last_epoch, b) for b in self.base_lrs]
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g["lr"] = lr

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else int(epoch)
        self._apply()

    def get_last_lr(self):
        return list(self._last_lr)

    def lr_table(self, n, start=None):
        """lr_at(k0 .. k0+n-1) as a list of Python floats (doubles: ``Hyper.lr`` / ``GroupItem.lr`` take doubles), bit-identical
        to ``lr_at`` entry by entry (tests/test_host_api_cpu.py pins it over a 12 800-step schedule): the base-independent
        factor of the tail -- ``1 + math.cos(pi t / max_iter)`` resp. ``1 - t / max_iter``, formed per element with the SAME
        scalar expression as ``lr_at``, never a vectorised ``numpy.cos`` whose last bit can differ -- is cached per (kind,
        max_iter) and shared by every head of a sweep; what remains per call, ``base * factor / 2`` resp. ``base * factor`` and
        the warm-up's ``base * k / warmup_iter``, are single IEEE double operations that numpy performs exactly as Python does."""
        import numpy as np
        k0 = self.last_epoch if start is None else start
        base = self.base_lrs[0]
        out = np.empty(n, dtype=np.float64)
        nw = min(max(self.warmup_iter - k0, 0), n)                 # leading entries still inside the warm-up
        if nw:
            if self.warmup_type == "constant":
                out[:nw] = float(self.warmup_lr)
            else:
                out[:nw] = base * np.arange(k0, k0 + nw, dtype=np.float64) / self.warmup_iter
                if k0 == 0:
                    out[0] = float(self.warmup_lr)
        if nw < n:
            t0 = k0 + nw - self.warmup_iter
            f = _tail_factors(self.kind, self.max_iter, t0 + (n - nw))[t0:t0 + (n - nw)]
            out[nw:] = base * f / 2.0 if self.kind == "cosine" else base * f
        return out.tolist()


_FACTORS = {}


def _tail_factors(kind, max_iter, upto):
    """factor[t] of the post-warm-up schedule for t < upto (grown on demand, shared between schedulers)."""
    import numpy as np
    key = (kind, max_iter)
    f = _FACTORS.get(key)
    if f is None or len(f) < upto:
        have = 0 if f is None else len(f)
        size = max(upto, 2 * have, 1024)
        if kind == "cosine":
            ext = [1.0 + math.cos(math.pi * t / max_iter) for t in range(have, size)]
        else:
            ext = [1.0 - t / max_iter for t in range(have, size)]
        f = np.concatenate([f, np.asarray(ext, dtype=np.float64)]) if have else np.asarray(ext, dtype=np.float64)
        _FACTORS[key] = f
    return f


def build_lr_scheduler(optimizer, lr_scheduler, warmup_iter, max_iter, warmup_type=None, warmup_lr=None,
                       verbose=False):
    """Same arguments and ValueErrors as the reference builder (scheduler.py:84-143)."""
    if verbose:
        print(f"Building scheduler: {lr_scheduler} with warmup: {warmup_type}")
    if lr_scheduler not in AVAI_SCHEDS:
        raise ValueError(f"scheduler must be one of {AVAI_SCHEDS}, but got {lr_scheduler}")
    if warmup_iter > 0 and warmup_type not in AVAI_WARMUP_SCHEDS:
        raise ValueError(f"warmup_type must be one of {AVAI_WARMUP_SCHEDS}, but got {warmup_type}")
    return StepLR(optimizer, lr_scheduler, max_iter, warmup_iter, warmup_type, warmup_lr)
