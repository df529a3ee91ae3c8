"""HeadEngine: torch-owned buffers + one umlh handle = the fused UML head step.

Host-side plumbing only (device memory, streams); all arithmetic happens in the
HIP kernels behind the C ABI.  Mirrors what ``finetune.train`` does per step
with ``model``/``optimizer`` (vision_language/finetune.py:180-195).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import Batch, Buffers, Config, Hyper, OPT_IDS, PREC_IDS, N_SCALARS, Stream, UmlhError, check


# Set by finetune.sweep_farm for its duration: worker threads take it around their multi-step enqueue calls.  Many
# threads launching kernels at once contend inside the HIP runtime (measured: 12 free-running workers are 1.5x SLOWER
# than one); serialised enqueues keep the single-thread launch rate while the streams still overlap on the GPU.
ENQUEUE_LOCK = None


@dataclass
class RowBatch:
    """Rows of one modality for one step: a device-resident (feats, labels) table and
    optional int64 row ids into it (None = rows 0..rows-1)."""
    feats: torch.Tensor                 # [N, dim] fp32, device, contiguous
    labels: torch.Tensor                # [N] int64, device
    index: Optional[torch.Tensor] = None  # [rows] int64, device
    rows: Optional[int] = None
    global_rows: Optional[int] = None   # CE-mean denominator across ranks (default = rows)
    feats_bf16: Optional[torch.Tensor] = None   # [N, dim] bf16 shadow of feats (bf16 engines)

    def n_rows(self) -> int:
        if self.rows is not None:
            return int(self.rows)
        return int(self.index.numel() if self.index is not None else self.feats.shape[0])


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HeadEngine:
    def __init__(self, d_img: int, d_shared: int, num_classes: int, *, has_proj: bool = False,
                 learnable_temp: bool = False, optimizer: str = "adamw", weight_decay: float = 0.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, momentum: float = 0.9,
                 max_rows_img: int = 4096, max_rows_txt: int = 4096, precision: str = "fp32",
                 device="cuda:0", bias_from=None):
        """``bias_from = d``: a linear head WITH bias (engine/models/head.py:65,68 ``bias=True``) over d-wide features, run
        as a bias-free head over ``d_shared``-wide rows [x | 1 | 0...]: column d of every feature row is 1 (appended here, cached
        per table), column d of ``w_head`` is the bias, columns beyond are zero padding that stays zero (zero gradient; AdamW's
        decay of 0 is 0).  Same logits ``x W^T + b``, same gradients and optimizer recurrences for weight and bias.
        ``bias_from = (d_img, d_sh)`` with ``has_proj``: the 2-layer head with both biases -- image rows widen to ``d_img``
        columns with their 1 at column d_img, text rows to ``d_shared`` with their 1 at column d_sh; column d_img of ``w_proj``
        rows < d_sh is img_proj's bias, and row d_sh of ``w_proj`` is the constant row that copies the ones column into the
        projected rows (so that column d_sh of ``w_head`` acts as the head's bias on both modalities): the caller writes it
        (1 at column d_img) and the optimizer leaves it alone (``umlh_freeze_proj_row``)."""
        if optimizer not in OPT_IDS:   # engine/optimizer/optim.py:22
            raise AssertionError(f"Optimizer {optimizer} not found; available optimizers = {list(OPT_IDS)}")
        self.lib = _lib.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise UmlhError("HeadEngine needs a GPU device (gfx950); there is no CPU path")
        self.cfg = Config(d_img, d_shared, num_classes, int(has_proj), int(learnable_temp), OPT_IDS[optimizer],
                          PREC_IDS[precision], max_rows_img, max_rows_txt, betas[0], betas[1], eps, momentum,
                          weight_decay)
        nbytes = int(self.lib.umlh_workspace_bytes(C.byref(self.cfg)))
        if nbytes == 0:
            raise UmlhError(f"unsupported head config: d_img={d_img} d_shared={d_shared} C={num_classes} "
                            f"has_proj={has_proj} precision={precision}")
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):           # the handle remembers its device; its entry points switch to it
            check(self.lib.umlh_create(C.byref(self.cfg), C.byref(self.handle)), "umlh_create")
        f32 = dict(dtype=torch.float32, device=self.device)
        self.workspace = torch.empty(nbytes // 4, **f32)
        self.has_proj, self.learnable_temp, self.optimizer = bool(has_proj), bool(learnable_temp), optimizer
        self.precision = precision
        self.d_img, self.d_shared, self.num_classes = d_img, d_shared, num_classes
        self.bias_from = None
        self._from_img = self._from_txt = None       # user-visible widths of image / text rows of a head with bias
        if bias_from is not None:
            if has_proj:
                self._from_img, self._from_txt = (int(v) for v in bias_from)
                if not (0 < self._from_img < d_img and 0 < self._from_txt < d_shared):
                    raise UmlhError("bias_from: (d_img, d_sh) must leave room for the ones column in both widths")
                check(self.lib.umlh_freeze_proj_row(self.handle, self._from_txt), "umlh_freeze_proj_row")
            else:
                self._from_img = self._from_txt = int(bias_from)
                if d_img != d_shared or not 0 < self._from_txt < d_shared:
                    raise UmlhError("bias_from: a linear head needs d_img == d_shared > bias_from")
            self.bias_from = bias_from
        self._aug_cache = {}                 # (data_ptr, rows, version, width) -> (fp32 augmented rows, bf16 shadow or None)
        self.w_head = torch.zeros(num_classes, d_shared, **f32)
        self.m_head = torch.zeros_like(self.w_head)
        self.v_head = torch.zeros_like(self.w_head)
        self.w_proj = torch.zeros(d_shared, d_img, **f32) if has_proj else None
        self.m_proj = torch.zeros_like(self.w_proj) if has_proj else None
        self.v_proj = torch.zeros_like(self.w_proj) if has_proj else None
        self.scales = torch.ones(2, **f32)
        self.m_scales = torch.zeros(2, **f32)
        self.v_scales = torch.zeros(2, **f32)
        self._scalars = torch.zeros(N_SCALARS, **f32)
        self.rebind()

    # -- buffers -----------------------------------------------------------------
    def rebind(self, **tensors) -> None:
        """(Re)attach parameter/state tensors, e.g. ``rebind(w_head=model.head.weight.data)``
        so the kernels update the module's own storage in place."""
        for k, v in tensors.items():
            if not hasattr(self, k):
                raise KeyError(k)
            if v is not None:
                cur = getattr(self, k)
                if v.dtype != torch.float32 or not v.is_contiguous() or v.device != self.device:
                    raise UmlhError(f"rebind({k}): need a contiguous fp32 tensor on {self.device}")
                if cur is not None and tuple(cur.shape) != tuple(v.shape):
                    raise UmlhError(f"rebind({k}): shape {tuple(v.shape)} != {tuple(cur.shape)}")
            setattr(self, k, v)
        b = Buffers(_ptr(self.w_head), _ptr(self.m_head), _ptr(self.v_head), _ptr(self.w_proj), _ptr(self.m_proj),
                    _ptr(self.v_proj), _ptr(self.scales), _ptr(self.m_scales), _ptr(self.v_scales),
                    _ptr(self.workspace), self.workspace.numel() * 4)
        check(self.lib.umlh_bind(self.handle, C.byref(b)), "umlh_bind")

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.umlh_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _from_width(self, f, dim):
        """User-visible width of rows headed for a ``dim``-wide operand of a head with bias, or None (no bias / rows that
        already have the engine's width)."""
        if self.bias_from is None or f.dim() != 2:
            return None
        for width, frm in ((self.d_img, self._from_img), (self.d_shared, self._from_txt)):
            if dim == width and f.shape[1] == frm:
                return frm
        return None

    def _augment(self, f: torch.Tensor, dim: int, want16: bool):
        """[N, d] feature rows -> ([N, dim] rows [x | 1 | 0...], bf16 shadow or None) for a head with bias.  Tables are
        converted once (cached by address, row count and in-place version)."""
        frm = self._from_width(f, dim)
        if frm is None:
            return f, None
        key = (f.data_ptr(), int(f.shape[0]), int(f._version), int(dim))
        hit = self._aug_cache.pop(key, None)
        if hit is None:
            a = torch.zeros(f.shape[0], dim, dtype=torch.float32, device=f.device)
            a[:, :frm] = f
            a[:, frm] = 1.0
            hit = (a, None)
        if want16 and hit[1] is None:
            hit = (hit[0], to_bf16(hit[0]))
        self._aug_cache[key] = hit           # most recently used last
        while len(self._aug_cache) > 32:
            self._aug_cache.pop(next(iter(self._aug_cache)))
        return hit

    def _batch(self, b: Optional[RowBatch], dim: int) -> Optional[Batch]:
        if b is None:
            return None
        rows = b.n_rows()
        if rows == 0:
            # data-parallel split: an empty local shard still tells the update that the modality has rows elsewhere
            return Batch(None, None, None, 0, int(b.global_rows), None) if b.global_rows else None
        f, y = b.feats, b.labels
        f16_aug = None
        if self._from_width(f, dim) is not None:
            if f.dtype != torch.float32 or not f.is_contiguous():
                raise UmlhError(f"features must be contiguous fp32, got {f.dtype} {tuple(f.shape)}")
            f, f16_aug = self._augment(f, dim, self.precision == "bf16")
        if f.dtype != torch.float32 or not f.is_contiguous() or f.dim() != 2 or f.shape[1] != dim:
            raise UmlhError(f"features must be contiguous fp32 [N,{dim}], got {f.dtype} {tuple(f.shape)}")
        if y.dtype != torch.int64 or not y.is_contiguous():
            raise UmlhError("labels must be contiguous int64")
        if f.device != self.device or y.device != self.device:
            raise UmlhError(f"features/labels must live on {self.device}")
        if b.index is not None:
            if b.index.dtype != torch.int64 or not b.index.is_contiguous() or b.index.device != self.device:
                raise UmlhError("index must be contiguous int64 on the engine's device")
            if b.index.numel() < rows:
                raise UmlhError("index shorter than rows")
        elif f.shape[0] < rows:
            raise UmlhError("feature table shorter than rows")
        f16 = b.feats_bf16 if f16_aug is None else f16_aug
        if self.precision == "bf16":
            if f16 is None:                       # dense batch without a cached shadow: convert now (HIP kernel)
                f16 = b.feats_bf16 = to_bf16(f)
            if f16.dtype != torch.bfloat16 or f16.shape != f.shape or not f16.is_contiguous() or f16.device != self.device:
                raise UmlhError("feats_bf16 must be a contiguous bf16 tensor shaped like feats on the engine's device")
        return Batch(_ptr(f), _ptr(y), _ptr(b.index), rows, int(b.global_rows or rows), _ptr(f16))

    @staticmethod
    def _ref(x):
        return None if x is None else C.byref(x)

    # -- C ABI calls -----------------------------------------------------------------
    def zero_shot_init(self, text_feats: torch.Tensor, text_labels: torch.Tensor) -> None:
        """head.weight.data = get_zero_shot_weights(...)  (engine/models/head.py:22-37,96-98)."""
        if self.bias_from is not None:
            raise UmlhError("zero_shot_init: initialise the d-wide weight with a bias-free engine (get_zero_shot_weights)")
        tf = text_feats.to(self.device, torch.float32).contiguous()
        tl = text_labels.to(self.device, torch.int64).contiguous()
        if tf.shape[1] != self.d_shared:
            raise UmlhError(f"text features have dim {tf.shape[1]}, head expects {self.d_shared}")
        check(self.lib.umlh_zero_shot_init(self.handle, _ptr(tf), _ptr(tl), tf.shape[0], self._stream()),
              "umlh_zero_shot_init")
        torch.cuda.current_stream(self.device).synchronize()   # tf/tl are temporaries

    def logits(self, batch: RowBatch, modality: int) -> torch.Tensor:
        dim = self.d_img if modality == 0 else self.d_shared
        b = self._batch(batch, dim)
        out = torch.empty(batch.n_rows(), self.num_classes, dtype=torch.float32, device=self.device)
        if b is not None:
            check(self.lib.umlh_logits(self.handle, C.byref(b), modality, _ptr(out), self._stream()), "umlh_logits")
        return out

    def project(self, batch: RowBatch) -> torch.Tensor:
        """img_proj(feats) -> [rows, d_shared]  (engine/models/head.py:87-90)."""
        b = self._batch(batch, self.d_img)
        out = torch.empty(batch.n_rows(), self.d_shared, dtype=torch.float32, device=self.device)
        if b is not None:
            check(self.lib.umlh_project(self.handle, C.byref(b), _ptr(out), self._stream()), "umlh_project")
        return out

    def enable_diagnostics(self, on: bool = True) -> None:
        """Per-step gradient diagnostics (S_GRAD_* scalars; see ``grad_diagnostics``) on/off.  For a head with bias they cover
        the weight columns only, as the reference's do (finetune.py:190-191: ``model.head.weight``)."""
        if on and self.bias_from is not None:
            check(self.lib.umlh_set_diagnostic_columns(self.handle, int(self._from_txt)), "umlh_set_diagnostic_columns")
        check(self.lib.umlh_enable_diagnostics(self.handle, 1 if on else 0), "umlh_enable_diagnostics")

    def train_step(self, img: Optional[RowBatch], txt: Optional[RowBatch], lr: float, step: int,
                   alpha: float = 1.0, img_alpha: float = 1.0, scalars_out: Optional[torch.Tensor] = None):
        bi, bt = self._batch(img, self.d_img), self._batch(txt, self.d_shared)
        hy = Hyper(float(lr), int(step), float(alpha), float(img_alpha), 0, 0)
        so = scalars_out if scalars_out is not None else self._scalars
        check(self.lib.umlh_train_step(self.handle, self._ref(bi), self._ref(bt), C.byref(hy), _ptr(so),
                                       self._stream()), "umlh_train_step")
        return so

    def _make_stream(self, table, batches, dim, cap, n):
        """(ctypes Stream, keep-alive tuple) of one modality for ``n`` steps; see ``train_steps``."""
        if table is None:
            return None, None
        f, y = table[0], table[1]
        f16 = table[2] if len(table) > 2 else None
        if self._from_width(f, dim) is not None and f.dtype == torch.float32 and f.is_contiguous():
            f, f16 = self._augment(f, dim, self.precision == "bf16")
        if f.dtype != torch.float32 or not f.is_contiguous() or f.shape[1] != dim or y.dtype != torch.int64:
            raise UmlhError("train_steps: table must be contiguous fp32 [N,dim] + int64 labels")
        if self.precision == "bf16" and (f16 is None or f16.dtype != torch.bfloat16 or f16.shape != f.shape):
            raise UmlhError("train_steps: bf16 engine needs the table's bf16 shadow")
        # an entry is one step's index vector, or (index slice, [sizes]) covering several consecutive steps
        import numpy as np
        parts, sizes = [], []
        for b in batches:
            if isinstance(b, tuple):
                parts.append(b[0])
                sizes.extend(b[1])
            else:
                parts.append(b)
                sizes.append(b.numel())
        if len(sizes) != n:
            raise UmlhError("train_steps: one index vector per step required")
        offs_np = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.asarray(sizes, dtype=np.int32), out=offs_np[1:])
        largest = int(np.diff(offs_np).max()) if n else 0
        if largest > cap:
            raise UmlhError(f"train_steps: batch of {largest} rows exceeds capacity {cap}")
        idx = torch.cat(parts) if len(parts) > 1 else parts[0].contiguous()
        if idx.numel() != int(offs_np[n]):
            raise UmlhError("train_steps: index slices do not match their batch sizes")
        offs = offs_np.ctypes.data_as(C.POINTER(C.c_int32))
        keep = (f, y, f16, idx, offs_np)
        return Stream(_ptr(f), _ptr(f16), _ptr(y), _ptr(idx), offs), keep

    def train_steps(self, img_table, img_index_batches, txt_table, txt_index_batches, lrs, first_step: int,
                    alpha: float = 1.0, img_alpha: float = 1.0, scalars_out: Optional[torch.Tensor] = None) -> None:
        """``len(lrs)`` consecutive fused steps with no Python in between (umlh_train_steps).
        ``*_table`` = (feats, labels[, feats_bf16]) device tensors or None; ``*_index_batches``
        = one int64 device index vector per step, or (index slice, [batch sizes]) entries that each cover
        several consecutive steps."""
        self.train_steps_prepared(self.prepare_steps(img_table, img_index_batches, txt_table, txt_index_batches, lrs),
                                  first_step, alpha=alpha, img_alpha=img_alpha, scalars_out=scalars_out)

    def prepare_steps(self, img_table, img_index_batches, txt_table, txt_index_batches, lrs):
        """The host-side half of ``train_steps`` (argument checks, offsets, the concatenated index vector, ctypes arrays) done
        ahead of time -- e.g. for the NEXT block while the GPU still runs the current one; pass the result to
        ``train_steps_prepared``.  The learning rates are the ones of the steps the block will run."""
        n = len(lrs)
        si, keep_i = self._make_stream(img_table, img_index_batches, self.d_img, self.cfg.max_rows_img, n)
        st, keep_t = self._make_stream(txt_table, txt_index_batches, self.d_shared, self.cfg.max_rows_txt, n)
        lr_arr = (C.c_double * n)(*[float(x) for x in lrs])
        return (n, si, st, lr_arr, keep_i, keep_t)

    def train_steps_prepared(self, prep, first_step: int, alpha: float = 1.0, img_alpha: float = 1.0,
                             scalars_out: Optional[torch.Tensor] = None) -> None:
        n, si, st, lr_arr, keep_i, keep_t = prep
        lock = ENQUEUE_LOCK
        if lock is not None:
            lock.acquire()
        try:
            rc = self.lib.umlh_train_steps(self.handle, self._ref(si), self._ref(st), n, lr_arr, int(first_step),
                                           float(alpha), float(img_alpha), _ptr(scalars_out), self._stream())
        finally:
            if lock is not None:
                lock.release()
        check(rc, "umlh_train_steps")
        # index tensors must outlive the enqueued kernels: keep them until the next call
        self._keepalive = (keep_i, keep_t)

    def micro_launches(self) -> int:
        """Persistent micro-step launches this engine has taken part in."""
        v = C.c_int64(0)
        check(self.lib.umlh_micro_launches(self.handle, C.byref(v)), "umlh_micro_launches")
        return int(v.value)

    def micro_status(self) -> int:
        """0, or 1 + the step at which a bounded in-launch wait of the micro-step kernel gave up (host sync)."""
        v = C.c_int32(0)
        check(self.lib.umlh_micro_status(self.handle, C.byref(v)), "umlh_micro_status")
        return int(v.value)

    def step_launches(self) -> int:
        """One-launch steps (forward, dW and update as claimed tasks of ONE launch) this engine has taken."""
        v = C.c_int64(0)
        check(self.lib.umlh_step_launches(self.handle, C.byref(v)), "umlh_step_launches")
        return int(v.value)

    def step_status(self):
        """(code, task, launch tag, phase) of the first bounded in-launch wait of the one-launch step that gave up; code 0 =
        none did (host sync)."""
        v = (C.c_int32 * 4)()
        check(self.lib.umlh_step_status(self.handle, v), "umlh_step_status")
        return tuple(int(x) for x in v)

    def check_status(self) -> None:
        """Raise UmlhError if an in-launch wait of this engine's kernels has given up (the device was starved by another
        tenant, or a fault): steps from that point on applied no update.  Host sync; called wherever step scalars are read."""
        st = self.micro_status()
        if st != 0:
            raise UmlhError(f"micro-step kernel gave up waiting at step {st - 1} of a call (another process starving the "
                            "device of CUs?); the state of this head is undefined")
        code, task, tag, phase = self.step_status()
        if code != 0:
            raise UmlhError(f"one-launch step: a wait on task {task} (phase {phase}, launch tag {tag}) gave up after 50 ms "
                            "(another process starving the device of CUs?); no update was applied from that step on")

    def grad_step(self, img: Optional[RowBatch], txt: Optional[RowBatch], alpha: float = 1.0,
                  img_alpha: float = 1.0, weights_unchanged: bool = False) -> torch.Tensor:
        """``weights_unchanged=True``: the caller guarantees nobody but this engine wrote w_head since its
        last apply_update/train_step (lets a bf16 engine reuse the weight shadow its update kernel wrote)."""
        bi, bt = self._batch(img, self.d_img), self._batch(txt, self.d_shared)
        hy = Hyper(0.0, 1, float(alpha), float(img_alpha), 1 if weights_unchanged else 0, 0)
        check(self.lib.umlh_grad_step(self.handle, self._ref(bi), self._ref(bt), C.byref(hy), self._stream()),
              "umlh_grad_step")
        return self.grad_buffer()

    # -- lean per-step path for the data-parallel loop (no per-step validation / object churn) ----
    def bind_tables(self, img_table, txt_table) -> None:
        """Validate the device-resident tables once and cache their ctypes descriptors;
        afterwards ``grad_step_indexed`` only patches the index pointer and row counts."""
        def mk(tab, dim):
            if tab is None:
                return None
            rb = RowBatch(tab[0], tab[1], None, rows=1, feats_bf16=tab[2] if len(tab) > 2 else None)
            b = self._batch(rb, dim)
            b.keep = tab                      # keep the tensors alive
            return b
        self._tab = (mk(img_table, self.d_img), mk(txt_table, self.d_shared))
        self._hy = Hyper(0.0, 1, 1.0, 1.0, 0, 0)

    def grad_step_indexed(self, idx_img, idx_txt, global_img: int, global_txt: int, alpha: float = 1.0,
                          img_alpha: float = 1.0, weights_unchanged: bool = False) -> None:
        bi, bt = self._tab
        if bi is not None:
            bi.index, bi.rows, bi.global_rows = idx_img.data_ptr(), idx_img.numel(), global_img
        if bt is not None:
            bt.index, bt.rows, bt.global_rows = idx_txt.data_ptr(), idx_txt.numel(), global_txt
        hy = self._hy
        hy.alpha, hy.img_alpha, hy.flags = alpha, img_alpha, 1 if weights_unchanged else 0
        check(self.lib.umlh_grad_step(self.handle, self._ref(bi), self._ref(bt), C.byref(hy), self._stream()),
              "umlh_grad_step")

    def grad_buffer(self) -> torch.Tensor:
        """Flat fp32 view of the gradient message inside the workspace: [g_head | g_proj | g_scales(2) | scalars(12)], with the
        head part [g_img | g_txt] while the gradient diagnostics are enabled."""
        p, n = C.c_void_p(), C.c_uint64()
        check(self.lib.umlh_grad_buffer(self.handle, C.byref(p), C.byref(n)), "umlh_grad_buffer")
        off = (p.value - self.workspace.data_ptr()) // 4
        return self.workspace[off:off + n.value]

    # -- data-parallel transport ---------------------------------------------------------------------
    def init_rccl(self, group=None) -> bool:
        """Give this engine an RCCL communicator of its own over the ranks of ``group`` (``umlh_comm_init_rank``): rank 0
        draws the unique id, ``torch.distributed`` (whatever backend the group runs on) carries its 128 bytes to the
        others.  Afterwards ``train_steps`` runs whole data-parallel steps from C: gradients -> ncclAllReduce -> update on
        the step's stream.  Returns False if librccl could not be loaded."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        idb = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        ok = 1
        if rank == 0:
            ok = 1 if self.lib.umlh_comm_unique_id(idb) == 0 else 0
        dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.tensor([ok] + list(idb), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        vals = t.cpu().tolist()
        if not vals[0]:
            return False
        idb = (C.c_ubyte * _lib.COMM_ID_BYTES)(*vals[1:])
        check(self.lib.umlh_comm_init_rank(self.handle, idb, world, rank), "umlh_comm_init_rank")
        return True

    def init_p2p(self, group=None) -> None:
        """Attach the direct peer-to-peer all-reduce (csrc/umlh_p2p.hip) over the ranks of ``group``: every rank allocates its
        exchange region, ``torch.distributed`` (any backend) carries the 64-byte IPC handles, every rank maps its peers' regions
        and hands the table to ``umlh_p2p_attach``.  Linear heads only.  Unmeasured on a multi-GPU node (RCCL is the default)."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n_floats = C.c_uint64(0)
        p = C.c_void_p()
        check(self.lib.umlh_grad_buffer(self.handle, C.byref(p), C.byref(n_floats)), "umlh_grad_buffer")
        n_max = 2 * self.num_classes * self.d_shared + (self.d_shared * self.d_img if self.has_proj else 0) + 2 + _lib.N_SCALARS
        nbytes = int(self.lib.umlh_p2p_region_bytes(n_max, world))
        if nbytes == 0:
            raise UmlhError("init_p2p: unsupported world size")
        own = C.c_void_p()
        rc = self.lib.umlh_p2p_alloc(nbytes, C.byref(own))
        if rc:
            raise UmlhError(f"umlh_p2p_alloc failed (hip error {rc})")
        hb = (C.c_ubyte * 64)()
        rc = self.lib.umlh_p2p_export(own, hb)
        if rc:
            raise UmlhError(f"umlh_p2p_export failed (hip error {rc})")
        handles = [None] * world
        dist.all_gather_object(handles, bytes(hb), group=group)
        regions = (C.c_void_p * world)()
        opened = []
        for q in range(world):
            if q == rank:
                regions[q] = own
            else:
                peer = C.c_void_p()
                hq = (C.c_ubyte * 64)(*handles[q])
                rc = self.lib.umlh_p2p_open(hq, C.byref(peer))
                if rc:
                    raise UmlhError(f"umlh_p2p_open of rank {q}'s region failed (hip error {rc})")
                regions[q] = peer
                opened.append(peer)
        dist.barrier(group=group)                     # every rank has mapped every region before anybody's first step writes into one
        check(self.lib.umlh_p2p_attach(self.handle, regions, world, rank), "umlh_p2p_attach")
        self._p2p = (own, opened, regions)            # kept alive with the engine

    def detach_comm(self) -> None:
        """Drop the communicator (``umlh_set_comm(h, NULL, 1)``): ``train_steps`` is single-GPU again and the data-parallel
        step goes through ``grad_step`` / ``apply_update`` with the caller's all-reduce."""
        check(self.lib.umlh_set_comm(self.handle, None, 1), "umlh_set_comm")

    def set_allreduce(self, fn, n_ranks: int) -> None:
        """Custom transport for the C-level data-parallel loop (``umlh_set_allreduce``): ``fn(tensor)`` must SUM-all-reduce
        the given fp32 view of the gradient message in place, ordered with the current stream (tests: gloo).  ``fn=None``
        detaches."""
        if fn is None:
            self._ar_cb = None
            check(self.lib.umlh_set_allreduce(self.handle, _lib.ALLREDUCE_FN(), None, 1), "umlh_set_allreduce")
            return

        def cb(ctx, buf, n, stream):
            try:
                off = (int(buf) - self.workspace.data_ptr()) // 4
                fn(self.workspace[off:off + int(n)])
                return 0
            except Exception as exc:            # noqa: BLE001 -- the C side turns the code into an UmlhError
                print(f"umlh all-reduce callback failed: {exc!r}")
                return 1
        self._ar_cb = _lib.ALLREDUCE_FN(cb)     # keep the thunk alive as long as the engine
        check(self.lib.umlh_set_allreduce(self.handle, self._ar_cb, None, int(n_ranks)), "umlh_set_allreduce")

    def apply_update(self, lr: float, step: int, scalars_out: Optional[torch.Tensor] = None, alpha: float = 1.0,
                     img_alpha: float = 1.0):
        hy = Hyper(float(lr), int(step), float(alpha), float(img_alpha), 0, 0)
        so = scalars_out if scalars_out is not None else self._scalars
        check(self.lib.umlh_apply_update(self.handle, C.byref(hy), _ptr(so), self._stream()), "umlh_apply_update")
        return so

    PHASES = ("proj_fwd", "fwd_ce", "dw_head", "proj_bwd", "reduce_update")

    def profile(self, enable: bool = True) -> None:
        check(self.lib.umlh_profile_enable(self.handle, int(enable)), "umlh_profile_enable")

    def profile_read(self):
        """Per-phase milliseconds of the latest step (HIP events on the step's stream)."""
        ms = (C.c_float * len(self.PHASES))()
        check(self.lib.umlh_profile_read(self.handle, ms), "umlh_profile_read")
        return dict(zip(self.PHASES, [float(x) for x in ms]))

    def eval_rows(self, batch: RowBatch, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Per-row {CE, top-1 correct} of up to ``max_rows_img`` rows in one forward launch (``umlh_eval_rows``)."""
        b = self._batch(batch, self.d_img)
        n = batch.n_rows()
        if out is None:
            out = torch.empty(n, 2, dtype=torch.float32, device=self.device)
        check(self.lib.umlh_eval_rows(self.handle, self._ref(b), _ptr(out), self._stream()), "umlh_eval_rows")
        return out

    def eval_batch(self, batch: RowBatch, scalars_out: Optional[torch.Tensor] = None):
        b = self._batch(batch, self.d_img)
        so = scalars_out if scalars_out is not None else self._scalars
        check(self.lib.umlh_eval_batch(self.handle, self._ref(b), _ptr(so), self._stream()), "umlh_eval_batch")
        return so


def train_steps_grouped(jobs, n_steps: int) -> None:
    """``n_steps`` fused steps of MANY heads in grouped persistent launches (``umlh_train_steps_grouped``): the
    reference's sweep over a HYPER_DICT grid (finetune.py:406-448) as one [G, C, d] job.  ``jobs`` is a list of dicts
    with the keyword arguments of ``HeadEngine.train_steps`` plus ``engine``; all engines live on one device and the
    launches go to that device's current stream.  Bit-identical to calling ``train_steps`` on each engine."""
    if not jobs:
        return
    from ._lib import GroupItem
    items = (GroupItem * len(jobs))()
    keep = []
    for j, job in enumerate(jobs):
        e = job["engine"]
        lrs = job["lrs"]
        if len(lrs) != n_steps:
            raise UmlhError("train_steps_grouped: every head needs n_steps learning rates")
        si, ki = e._make_stream(job.get("img_table"), job.get("img_index_batches"), e.d_img, e.cfg.max_rows_img, n_steps)
        st, kt = e._make_stream(job.get("txt_table"), job.get("txt_index_batches"), e.d_shared, e.cfg.max_rows_txt, n_steps)
        lr_arr = (C.c_double * n_steps)(*[float(x) for x in lrs])
        so = job.get("scalars_out")
        items[j] = GroupItem(e.handle, C.pointer(si) if si is not None else None, C.pointer(st) if st is not None else None,
                             lr_arr, int(job["first_step"]), float(job.get("alpha", 1.0)), float(job.get("img_alpha", 1.0)),
                             _ptr(so))
        keep.append((si, st, ki, kt, lr_arr, so))
        e._keepalive = (ki, kt)
    e0 = jobs[0]["engine"]
    check(e0.lib.umlh_train_steps_grouped(items, len(jobs), int(n_steps), e0._stream()), "umlh_train_steps_grouped")


def grad_diagnostics(scalars, n_elements: int, rows_img: int = 1, rows_txt: int = 1) -> dict:
    """The reference's per-step gradient diagnostics (finetune.py:203-206,238) from one scalar row
    written by ``train_step`` / ``train_steps``: the kernels accumulate the raw sums
    (S_GRAD_DOT, S_GRAD_N2_IMG, S_GRAD_N2_TXT, S_GRAD_AGREE); the ratios are formed here, at logging
    time.  ``n_elements`` = C * d of the head weight.  With a modality absent the reference logs
    similarity 0 and agreement 0 (finetune.py:205-206)."""
    import math
    row = [float(x) for x in scalars[:_lib.N_SCALARS].tolist()]
    n2i, n2t = row[_lib.S_GRAD_N2_IMG], row[_lib.S_GRAD_N2_TXT]
    both = rows_img > 0 and rows_txt > 0
    den = math.sqrt(n2i) * math.sqrt(n2t)
    return {
        "grad_direction_sim": (row[_lib.S_GRAD_DOT] / den if den > 0 else float("nan")) if both else 0.0,
        "img_grad_norm": math.sqrt(n2i),
        "txt_grad_norm": math.sqrt(n2t),
        "grad_agreement_rate": row[_lib.S_GRAD_AGREE] / float(n_elements) if both else 0.0,
    }


def gather_rows(x: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """out[j] = x[index[j]] for a contiguous fp32 [N, d] device table (``umlh_gather_rows``)."""
    lib = _lib.load_library()
    out = torch.empty(index.numel(), x.shape[1], dtype=torch.float32, device=x.device)
    st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    check(lib.umlh_gather_rows(_ptr(x), _ptr(index), index.numel(), x.shape[1], _ptr(out), 0, st), "umlh_gather_rows")
    return out


def column_sums(x: torch.Tensor) -> torch.Tensor:
    """sum over the rows of a contiguous fp32 [M, N] device matrix (``umlh_colsum``)."""
    lib = _lib.load_library()
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    check(lib.umlh_colsum(_ptr(x), x.shape[0], x.shape[1], _ptr(out), st), "umlh_colsum")
    return out


def optimizer_step(name: str, param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: Optional[torch.Tensor], *,
                   lr: float, step: int, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                   momentum: float = 0.9) -> None:
    """Standalone ``optimizer.step()`` on one fp32 tensor through ``umlh_optimizer_step``."""
    lib = _lib.load_library()
    for t in (param, grad, m) + ((v,) if v is not None else ()):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != "cuda":
            raise UmlhError("optimizer_step: contiguous fp32 GPU tensors required")
    st = C.c_void_p(torch.cuda.current_stream(param.device).cuda_stream)
    check(lib.umlh_optimizer_step(OPT_IDS[name], _ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(),
                                  float(lr), int(step), float(betas[0]), float(betas[1]), float(eps), float(momentum),
                                  float(weight_decay), st), "umlh_optimizer_step")


def optimizer_step_multi(name: str, params, grads, ms, vs, *, lr: float, step: int, weight_decay: float = 0.0,
                         betas=(0.9, 0.999), eps: float = 1e-8, momentum: float = 0.9) -> None:
    """``optimizer.step()`` over a list of fp32 GPU tensors in one launch per 48 tensors (``umlh_optimizer_step_multi``)."""
    lib = _lib.load_library()
    n = len(params)
    if n == 0:
        return
    sgd = name == "sgd"
    fixed = []
    for i in range(n):
        g = grads[i]
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        fixed.append(g)
        for t in (params[i], g, ms[i]) + (() if sgd else (vs[i],)):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != "cuda":
                raise UmlhError("optimizer_step_multi: contiguous fp32 GPU tensors required")
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    cnt = (C.c_int64 * n)(*[p.numel() for p in params])
    st = C.c_void_p(torch.cuda.current_stream(params[0].device).cuda_stream)
    check(lib.umlh_optimizer_step_multi(OPT_IDS[name], n, arr(params), arr(fixed), arr(ms), None if sgd else arr(vs), cnt,
                                        float(lr), int(step), float(betas[0]), float(betas[1]), float(eps), float(momentum),
                                        float(weight_decay), st), "umlh_optimizer_step_multi")


def to_bf16(t: torch.Tensor) -> torch.Tensor:
    """bf16 shadow (round-to-nearest-even) of an fp32 GPU tensor through ``umlh_to_bf16``."""
    lib = _lib.load_library()
    if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != "cuda":
        raise UmlhError("to_bf16: contiguous fp32 GPU tensor required")
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    st = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    check(lib.umlh_to_bf16(_ptr(t), _ptr(out), t.numel(), st), "umlh_to_bf16")
    return out


def random_permutation(n: int, seed: int, device) -> torch.Tensor:
    """int64 permutation of 0..n-1 drawn on the device by ``umlh_random_permutation`` (sort-free)."""
    lib = _lib.load_library()
    out = torch.empty(n, dtype=torch.int64, device=device)
    st = C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
    check(lib.umlh_random_permutation(int(n), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(out), st), "umlh_random_permutation")
    return out
