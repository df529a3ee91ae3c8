"""Plain data parallelism for the UML head step: one process per GPU, replicas of
the head, each rank feeds its own shard of the rows, ONE sum all-reduce of the flat
gradient buffer per step (RCCL over xGMI when the backend is "nccl").

The reference is single-process / single-device; this layer is
new functionality.  Semantics: the result equals the single-GPU step on the
concatenation of all ranks' rows -- every rank divides its partial gradient and its
partial loss sums by the GLOBAL row counts before the all-reduce
(``RowBatch.global_rows``), so SUM over ranks is the global mean.

Message: [g_head | g_proj | g_scales(2) | scalars(12)] fp32 -- 2.05 MB for the
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
        self._own_weights = False      # True once this stepper performed the last write of the weights
        self._flat = None

    def attach_rccl(self) -> bool:
        """Hand the engine an RCCL communicator of its own so that ``umlh_train_steps`` runs whole data-parallel
        steps from C (grad -> ncclAllReduce -> update on the step's stream, no Python per step).  Needs an initialised
        process group to exchange the unique id; returns False when the library was built without RCCL."""
        if self.world <= 1 or not hasattr(self.engine, "init_rccl"):
            return False
        return bool(self.engine.init_rccl(self.group))

    def attach_p2p(self) -> bool:
        """Hand the engine the direct peer-to-peer all-reduce (``HeadEngine.init_p2p``: reduce-scatter + all-gather over
        hipIpc-mapped exchange regions) instead of RCCL; ``umlh_train_steps`` then runs the data-parallel steps from C as with
        ``attach_rccl``.  Linear heads only; opt-in (unmeasured on a multi-GPU node)."""
        if self.world <= 1 or not hasattr(self.engine, "init_p2p") or self.engine.has_proj:
            return False
        self.engine.init_p2p(self.group)
        return True

    def invalidate(self) -> None:
        """Call after writing the parameters from outside (load_state_dict, re-init)."""
        self._own_weights = False

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
        try:
            flat = self.engine.grad_step(self._with_global(img), self._with_global(txt), alpha=alpha, img_alpha=img_alpha,
                                         weights_unchanged=self._own_weights)
        except TypeError:              # engines without the hint (test doubles)
            flat = self.engine.grad_step(self._with_global(img), self._with_global(txt), alpha=alpha, img_alpha=img_alpha)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        try:
            out = self.engine.apply_update(lr=lr, step=step, scalars_out=scalars_out, alpha=alpha, img_alpha=img_alpha)
        except TypeError:              # engines without the loss weights in apply_update (test doubles)
            out = self.engine.apply_update(lr=lr, step=step, scalars_out=scalars_out)
        self._own_weights = True
        return out

    def step_indexed(self, idx_img, idx_txt, lr: float, step: int, alpha: float = 1.0, img_alpha: float = 1.0,
                     scalars_out=None, global_img=None, global_txt=None):
        """Lean variant for device-resident tables registered with ``engine.bind_tables``: per step only
        two index vectors change.  Equal shards assumed unless the global row counts are given."""
        gi = global_img if global_img is not None else (idx_img.numel() * self.world if idx_img is not None else 0)
        gt = global_txt if global_txt is not None else (idx_txt.numel() * self.world if idx_txt is not None else 0)
        self._flat = self.engine.grad_buffer()          # its length follows the diagnostics switch
        self.engine.grad_step_indexed(idx_img, idx_txt, gi, gt, alpha, img_alpha, weights_unchanged=self._own_weights)
        if self.world > 1:
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        out = self.engine.apply_update(lr=lr, step=step, scalars_out=scalars_out, alpha=alpha, img_alpha=img_alpha)
        self._own_weights = True
        return out
