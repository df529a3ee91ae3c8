"""Plain data parallelism for the UML head step: one process per GPU, replicas of
the head, each rank feeds its own shard of the rows, ONE sum all-reduce of the flat
gradient buffer per step (RCCL over xGMI when the backend is "nccl").

The reference is single-process / single-device; this layer is
new functionality.  Semantics: the result equals the single-GPU step on the
concatenation of all ranks' rows -- every rank divides its partial gradient and its
partial loss sums by the GLOBAL row counts before the all-reduce
(``RowBatch.global_rows``), so SUM over ranks is the global mean.

Message: [g_head | g_proj | g_scales(2) | scalars(8)] fp32 -- 2.05 MB for the
ImageNet CLIP-B/16 head (C=1000, d=512), one bucket, one collective.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class DataParallelStepper:
    """Wraps an engine exposing ``train_step`` / ``grad_step`` / ``apply_update``
    (``umlh.HeadEngine``).  ``equal_shards=True`` (default) assumes every rank passes the
    same number of rows per modality each step, which holds for equally sized shards
    cycled with the same batch size; otherwise pass ``global_rows`` in the RowBatch."""

    def __init__(self, engine, group: Optional[dist.ProcessGroup] = None):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, tensors, src: int = 0) -> None:
        if self.world > 1:
            for t in tensors:
                if t is not None:
                    dist.broadcast(t, src=src, group=self.group)

    def _with_global(self, b):
        if b is not None and b.global_rows is None:
            b.global_rows = b.n_rows() * self.world
        return b

    def step(self, img, txt, lr: float, step: int, alpha: float = 1.0, img_alpha: float = 1.0, scalars_out=None):
        if self.world == 1:
            return self.engine.train_step(img, txt, lr=lr, step=step, alpha=alpha, img_alpha=img_alpha,
                                          scalars_out=scalars_out)
        flat = self.engine.grad_step(self._with_global(img), self._with_global(txt), alpha=alpha, img_alpha=img_alpha)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return self.engine.apply_update(lr=lr, step=step, scalars_out=scalars_out)
