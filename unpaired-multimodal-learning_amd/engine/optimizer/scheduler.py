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
        """lr_at(k0 .. k0+n-1) as a list of Python floats (doubles: ``Hyper.lr`` / ``GroupItem.lr`` take doubles).  Every entry
        is formed by the SAME expression as ``lr_at`` -- the cosine tail with ``math.cos`` per element, not a vectorised
        ``numpy.cos``, whose last bit can differ -- so a blockwise run (lr_table) and a stepwise run (lr_at) see bit-identical
        learning rates (tests/test_host_api_cpu.py pins it over a 12 800-step schedule)."""
        k0 = self.last_epoch if start is None else start
        base = self.base_lrs[0]
        return [self.lr_at(k, base) for k in range(k0, k0 + n)]


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
