"""CPU ORACLE for the UML head fine-tune hot path  --  TEST INFRASTRUCTURE ONLY.

This module is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``unpaired-multimodal-learning_amd/``) never routes
through it and fails loudly when the HIP extension is missing.

It is a plain-numpy restatement (float32 by default, float64 on request) of the
arithmetic the reference executes through third-party PyTorch on this path.
Every function cites the reference file:line (relative to ``/root/reference``)
it follows.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``oracle/make_golden.py`` (imports
``/root/reference/vision_language`` with stubs for absent third-party
packages) and committed as data under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks every function here against them.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- #
# forward: head (+ optional img_proj), scale, cross entropy
# --------------------------------------------------------------------------- #
def project(x: np.ndarray, w_proj: Optional[np.ndarray]) -> np.ndarray:
    """``img_proj`` Linear(num_features -> text_indim, bias=False), no activation.
    vision_language/engine/models/head.py:64-66,79."""
    if w_proj is None:
        return x
    return x @ w_proj.T


def head_logits(feats: np.ndarray, w_head: np.ndarray, scale) -> np.ndarray:
    """``head(feats) * scale``; head = Linear(shared_dim -> C, bias=False).
    vision_language/engine/models/head.py:68,80-82 (UML) and :124,133-135
    (UMLClip, scale = exp(logit_scale))."""
    dt = feats.dtype
    return (feats @ w_head.T) * np.asarray(scale, dtype=dt)


def log_softmax(z: np.ndarray) -> np.ndarray:
    m = z.max(axis=1, keepdims=True)
    e = np.exp(z - m)
    return (z - m) - np.log(e.sum(axis=1, keepdims=True))


def cross_entropy_mean(z: np.ndarray, y: np.ndarray) -> float:
    """``F.cross_entropy(z, y)`` with default args (mean over rows, no smoothing).
    Call sites: vision_language/finetune.py:186-187,304."""
    ls = log_softmax(z)
    return float(-ls[np.arange(z.shape[0]), y].mean(dtype=np.float64))


def top1_correct(z: np.ndarray, y: np.ndarray) -> np.ndarray:
    """``argmax(z, dim=1) == y`` (first maximal index wins, as torch.argmax).
    vision_language/finetune.py:197-198,303."""
    return (z.argmax(axis=1) == y)


# --------------------------------------------------------------------------- #
# model state
# --------------------------------------------------------------------------- #
@dataclass
class HeadState:
    """Parameters of the shared classifier head in reference ``parameters()``
    order.  UML (head.py:39-70): [vision_model params (none: identity over
    pre-extracted feature rows)], img_proj.weight, head.weight, img_scale,
    txt_scale  (the two scales are Parameters only if learnable_temp).
    UMLClip (head.py:101-125): head.weight only, fixed scale exp(logit_scale).
    """
    w_head: np.ndarray                       # [C, shared_dim]
    w_proj: Optional[np.ndarray] = None      # [shared_dim, d_img] or None
    img_scale: float = 1.0
    txt_scale: float = 1.0
    learnable_temp: bool = False

    def copy(self) -> "HeadState":
        return HeadState(self.w_head.copy(),
                         None if self.w_proj is None else self.w_proj.copy(),
                         self.img_scale, self.txt_scale, self.learnable_temp)

    def param_names(self) -> List[str]:
        names = []
        if self.w_proj is not None:
            names.append("w_proj")
        names.append("w_head")
        if self.learnable_temp:
            names += ["img_scale", "txt_scale"]
        return names

    def get(self, name):
        v = getattr(self, name)
        return np.asarray(v, dtype=self.w_head.dtype)

    def set(self, name, value):
        if name in ("img_scale", "txt_scale"):
            setattr(self, name, float(value))
        else:
            setattr(self, name, value)


def forward(state: HeadState, x_img: Optional[np.ndarray],
            x_txt: Optional[np.ndarray]) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
    """``model(images, text_features)`` -> (img_logits, txt_logits|None) with the
    backbone = identity over pre-extracted feature rows.
    vision_language/engine/models/head.py:77-84 / :131-137."""
    dt = state.w_head.dtype
    zi = zt = None
    if x_img is not None:
        zi = head_logits(project(x_img, state.w_proj), state.w_head, dt.type(state.img_scale))
    if x_txt is not None:
        zt = head_logits(x_txt, state.w_head, dt.type(state.txt_scale))
    return zi, zt


# --------------------------------------------------------------------------- #
# loss + closed-form gradient of one step
# --------------------------------------------------------------------------- #
@dataclass
class StepOut:
    loss_img: float
    loss_txt: float
    acc_img: float
    acc_txt: float
    grads: Dict[str, np.ndarray]
    zi: Optional[np.ndarray] = None
    zt: Optional[np.ndarray] = None
    # per-modality head gradients (the diagnostics of finetune.py:190-191)
    g_head_img: Optional[np.ndarray] = None
    g_head_txt: Optional[np.ndarray] = None


def _dlogits(z: np.ndarray, y: np.ndarray) -> np.ndarray:
    """d(mean CE)/dz = (softmax(z) - onehot(y)) / B."""
    p = np.exp(log_softmax(z))
    p[np.arange(z.shape[0]), y] -= 1
    return p / z.dtype.type(z.shape[0])


def step_grads(state: HeadState, x_img, y_img, x_txt, y_txt, alpha: float,
               img_alpha: float = 1.0) -> StepOut:
    """Loss and gradient of ``loss = img_alpha*CE(img) + alpha*CE(txt)``
    (vision_language/finetune.py:160,186-188,193); the gradient is what autograd
    produces for the graph of head.py:77-84:

        dW_head = s_i * dZi^T F_i + s_t * dZt^T X_t       (dZ incl. 1/B, alpha)
        dF_i    = s_i * dZi W_head ;  dW_proj = dF_i^T X_i
        ds_i    = sum(dZi * raw_i) ;  ds_t = sum(dZt * raw_t)   (raw = logits/scale)
    """
    dt = state.w_head.dtype
    grads: Dict[str, np.ndarray] = {}
    g_head = np.zeros_like(state.w_head)
    li = lt = 0.0
    ai = at = 0.0
    zi = zt = None
    g_img = g_txt = None
    if x_img is not None:
        f_i = project(x_img, state.w_proj)
        raw_i = f_i @ state.w_head.T
        zi = raw_i * dt.type(state.img_scale)
        li = cross_entropy_mean(zi, y_img)
        ai = float(top1_correct(zi, y_img).mean())
        dzi = _dlogits(zi, y_img)                       # d CE_img / d zi
        g_img = dt.type(state.img_scale) * (dzi.T @ f_i)  # finetune.py:190
        g_head += dt.type(img_alpha) * g_img
        if state.w_proj is not None:
            df = dt.type(img_alpha * state.img_scale) * (dzi @ state.w_head)
            grads["w_proj"] = df.T @ x_img
        if state.learnable_temp:
            grads["img_scale"] = np.asarray(img_alpha * float((dzi * raw_i).sum(dtype=np.float64)), dtype=dt)
    if x_txt is not None:
        raw_t = x_txt @ state.w_head.T
        zt = raw_t * dt.type(state.txt_scale)
        lt = cross_entropy_mean(zt, y_txt)
        at = float(top1_correct(zt, y_txt).mean())
        dzt = _dlogits(zt, y_txt)
        g_txt = dt.type(state.txt_scale) * (dzt.T @ x_txt)  # finetune.py:191
        g_head += dt.type(alpha) * g_txt
        if state.learnable_temp:
            grads["txt_scale"] = np.asarray(alpha * float((dzt * raw_t).sum(dtype=np.float64)), dtype=dt)
    elif state.learnable_temp:
        # txt_scale takes no part in the graph -> .grad stays None -> torch skips it
        pass
    grads["w_head"] = g_head
    return StepOut(li, lt, ai, at, grads, zi, zt, g_img, g_txt)


def grad_diagnostics(g_img: Optional[np.ndarray], g_txt: Optional[np.ndarray]) -> Dict[str, float]:
    """Per-step gradient diagnostics the reference logs (finetune.py:203-206,238) from the
    UNWEIGHTED per-modality head gradients (finetune.py:190-191; an absent modality is
    ``zeros_like`` there): cosine similarity and sign-agreement rate (both 0 unless both
    modalities are present, :205-206) and the two Frobenius norms (:238)."""
    both = g_img is not None and g_txt is not None
    gi = None if g_img is None else g_img.ravel().astype(np.float64)
    gt = None if g_txt is None else g_txt.ravel().astype(np.float64)
    ni = 0.0 if gi is None else float(np.sqrt((gi * gi).sum()))
    nt = 0.0 if gt is None else float(np.sqrt((gt * gt).sum()))
    sim = float((gi * gt).sum() / (ni * nt)) if both else 0.0
    agree = float((np.sign(g_img.ravel()) == np.sign(g_txt.ravel())).mean()) if both else 0.0
    return {"grad_direction_sim": sim, "grad_agreement_rate": agree, "img_grad_norm": ni, "txt_grad_norm": nt}


# --------------------------------------------------------------------------- #
# optimizers (torch.optim semantics; built by engine/optimizer/optim.py:15-71)
# --------------------------------------------------------------------------- #
ADAM_BETAS = (0.9, 0.999)     # engine/optimizer/optim.py:9
SGD_MOMENTUM = 0.9            # engine/optimizer/optim.py:12
ADAM_EPS = 1e-8               # torch default (optim.py:57-71 pass none)


@dataclass
class OptState:
    name: str                  # 'sgd' | 'adam' | 'adamw'
    weight_decay: float
    step: int = 0
    m: Dict[str, np.ndarray] = field(default_factory=dict)   # exp_avg / momentum_buffer
    v: Dict[str, np.ndarray] = field(default_factory=dict)   # exp_avg_sq


def optimizer_step(state: HeadState, grads: Dict[str, np.ndarray], opt: OptState, lr: float) -> None:
    """One ``optimizer.step()`` over every parameter that received a gradient.

    sgd   : torch.optim.SGD(lr, momentum=0.9, weight_decay, nesterov=False)
            engine/optimizer/optim.py:34-48.  g += wd*p; buf = g (first step) or
            0.9*buf + g; p -= lr*buf.
    adam  : torch.optim.Adam(lr, weight_decay, betas=(0.9,0.999))  optim.py:50-59.
            L2 form: g += wd*p, then the Adam recurrence.
    adamw : torch.optim.AdamW(...)  optim.py:61-71.  Decoupled: p *= 1-lr*wd.
    Adam recurrence (torch single-tensor path): m = b1*m + (1-b1)*g;
    v = b2*v + (1-b2)*g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
    """
    assert opt.name in ("sgd", "adam", "adamw"), opt.name  # optim.py:22
    opt.step += 1
    t = opt.step
    b1, b2 = ADAM_BETAS
    for name in state.param_names():
        if name not in grads:
            continue
        p = state.get(name)
        dt = p.dtype.type
        g = np.asarray(grads[name], dtype=p.dtype)
        wd = opt.weight_decay
        if opt.name == "sgd":
            if wd != 0:
                g = g + dt(wd) * p
            if name not in opt.m:
                opt.m[name] = g.copy()
            else:
                opt.m[name] = dt(SGD_MOMENTUM) * opt.m[name] + g
            p = p - dt(lr) * opt.m[name]
        else:
            if opt.name == "adamw":
                p = p * dt(1.0 - lr * wd)
            elif wd != 0:
                g = g + dt(wd) * p
            if name not in opt.m:
                opt.m[name] = np.zeros_like(p)
                opt.v[name] = np.zeros_like(p)
            opt.m[name] = opt.m[name] + (g - opt.m[name]) * dt(1 - b1)      # lerp_
            opt.v[name] = dt(b2) * opt.v[name] + dt(1 - b2) * g * g
            bc1 = 1.0 - b1 ** t
            bc2 = 1.0 - b2 ** t
            step_size = lr / bc1
            denom = np.sqrt(opt.v[name]) / dt(math.sqrt(bc2)) + dt(ADAM_EPS)
            p = p - dt(step_size) * (opt.m[name] / denom)
        state.set(name, p)


# --------------------------------------------------------------------------- #
# learning-rate schedule (engine/optimizer/scheduler.py:11-143)
# --------------------------------------------------------------------------- #
class LRSchedule:
    """lr used by optimizer step k (k = 0, 1, ...), i.e. the value in
    ``param_groups[0]['lr']`` when ``optimizer.step()`` runs for the k-th time,
    = ``scheduler.get_last_lr()[0]`` after k ``scheduler.step()`` calls.

    Successor: CosineAnnealingLR(T_max=max_iter, eta_min=0) or
    LambdaLR(1 - x/max_iter) (scheduler.py:110-119); wrapped, if warmup_iter>0,
    by Constant/LinearWarmupScheduler (scheduler.py:36-81,121-141).

    Quirks restated exactly:
      * the successor is constructed first, so its constructor's initial step
        leaves lr = base; then the wrapper's constructor step sets lr(0) =
        warmup_lr (linear: ``last_epoch == 0`` -> min_lr, scheduler.py:76-77;
        constant: cons_lr, :55).
      * wrapper.step() while last_epoch < warmup_iter: last_epoch += 1 and
        lr = base*last_epoch/warmup_iter  (linear, :78-80)  /  cons_lr; once
        last_epoch == warmup_iter the value is successor.get_last_lr() = base.
      * from then on (scheduler.py:28-31) each step advances only the successor,
        whose own last_epoch started at 0: torch's CosineAnnealingLR uses the
        *recursive* form  lr_t = (1+cos(pi t/T))/(1+cos(pi (t-1)/T)) * lr_{t-1}
        (eta_min = 0), which we follow in float64 like torch does in Python.
    """

    def __init__(self, base_lr: float, lr_scheduler: str, warmup_iter: int, max_iter: int,
                 warmup_type: Optional[str] = None, warmup_lr: Optional[float] = None):
        if lr_scheduler not in ("cosine", "linear"):               # scheduler.py:105-108
            raise ValueError(lr_scheduler)
        if warmup_iter > 0 and warmup_type not in ("constant", "linear"):   # :121-126
            raise ValueError(warmup_type)
        self.base, self.kind = float(base_lr), lr_scheduler
        self.warm, self.T = int(warmup_iter), float(max_iter)
        self.wtype, self.wlr = warmup_type, warmup_lr
        self._k = 0                       # number of scheduler.step() calls so far
        self._succ_t = 0                  # successor's last_epoch
        self._succ_lr = self.base
        self._lr = self._warm_value(0) if self.warm > 0 else self.base

    def _warm_value(self, last_epoch: int) -> float:
        if self.wtype == "constant":
            return float(self.wlr)
        if last_epoch == 0:
            return float(self.wlr)
        return self.base * last_epoch / self.warm

    def _succ_step(self) -> None:
        self._succ_t += 1
        t = self._succ_t
        if self.kind == "cosine":
            T = self.T
            if t - 1 - T == 0 or (t - 1 - T) % (2 * T) == 0:     # torch's restart branch
                self._succ_lr = self._succ_lr + self.base * (1 - math.cos(math.pi / T)) / 2
            else:
                self._succ_lr = ((1 + math.cos(math.pi * t / T)) /
                                 (1 + math.cos(math.pi * (t - 1) / T))) * self._succ_lr
        else:
            self._succ_lr = self.base * (1 - t / self.T)

    def get_last_lr(self) -> float:
        return self._lr

    def step(self) -> None:
        if self.warm > 0 and self._k < self.warm:
            self._k += 1
            self._lr = self._warm_value(self._k) if self._k < self.warm else self._succ_lr
        else:
            self._k += 1
            self._succ_step()
            self._lr = self._succ_lr

    def table(self, n: int) -> np.ndarray:
        """lr for steps 0..n-1 (fresh schedule)."""
        s = LRSchedule(self.base, self.kind, self.warm, int(self.T), self.wtype, self.wlr)
        out = np.empty(n, dtype=np.float64)
        for k in range(n):
            out[k] = s.get_last_lr()
            s.step()
        return out


# --------------------------------------------------------------------------- #
# zero-shot init + text dataset reductions
# --------------------------------------------------------------------------- #
def zero_shot_weights(text_feats: np.ndarray, text_labels: np.ndarray, num_classes: int) -> np.ndarray:
    """W[c] = mean of the text rows of class c (0 if none), then L2-normalise rows
    with F.normalize's eps=1e-12 clamp.  vision_language/engine/models/head.py:22-37."""
    d = text_feats.shape[1]
    w = np.zeros((num_classes, d), dtype=text_feats.dtype)
    for c in np.unique(text_labels):
        w[int(c)] = text_feats[text_labels == c].mean(axis=0)
    n = np.sqrt((w * w).sum(axis=1, keepdims=True))
    return w / np.maximum(n, text_feats.dtype.type(1e-12))


def text_average(feats: np.ndarray, labels: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """TextTensorDataset(n_shots='average'): one mean row per unique label, sorted.
    vision_language/engine/datasets/utils.py:88-98."""
    uniq = np.unique(labels)
    out = np.stack([feats[labels == c].mean(axis=0) for c in uniq])
    return out, uniq


def text_select_n_shots(labels, n_shots: int):
    """TextTensorDataset(n_shots=int): per unique label (sorted), a
    ``torch.randperm(count)[:n]`` subsample drawn from torch's *global* CPU
    generator -- restated with torch so the draws are seed-identical.
    vision_language/engine/datasets/utils.py:76-86."""
    import torch
    lab = torch.as_tensor(np.asarray(labels))
    chunks = []
    for c in torch.unique(lab):
        inds = (lab == c).nonzero(as_tuple=True)[0]
        n = min(n_shots, inds.size(0))
        chunks.append(inds[torch.randperm(inds.size(0))[:n]])
    return torch.cat(chunks).numpy()


# --------------------------------------------------------------------------- #
# loader index streams (torch DataLoader(shuffle=True, drop_last=False))
# --------------------------------------------------------------------------- #
class ShuffledBatches:
    """Index stream of ``DataLoader(ds, batch_size=B, shuffle=True,
    drop_last=False, num_workers=0)`` cycled forever by ``fetch_next``
    (vision_language/finetune.py:33-39,157-158,370-371).

    Per ``iter(loader)`` torch draws from the global CPU generator, in order:
    the iterator's base seed (``torch.empty((), int64).random_()``), then --
    lazily, at the first ``next`` -- RandomSampler's seed (same call), which
    seeds a private generator for ``torch.randperm(n)``.  We make the same
    calls so the permutation is identical under the same global seed.
    """

    def __init__(self, n: int, batch_size: int):
        self.n, self.bs = int(n), int(batch_size)
        self._perm = None
        self._pos = 0
        self._started = False

    def start(self):                      # == iter(loader)
        import torch
        torch.empty((), dtype=torch.int64).random_()          # _base_seed
        self._perm = None
        self._pos = 0
        self._started = True

    def _draw_perm(self):
        import torch
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator()
        g.manual_seed(seed)
        self._perm = torch.randperm(self.n, generator=g).numpy()
        self._pos = 0

    def next(self) -> np.ndarray:         # == fetch_next(loader, it)[0] indices
        if not self._started:
            self.start()
        if self._perm is None:
            self._draw_perm()
        if self._pos >= self.n:           # StopIteration -> iter(loader) again
            self.start()
            self._draw_perm()
        idx = self._perm[self._pos:self._pos + self.bs]
        self._pos += self.bs
        return idx


# --------------------------------------------------------------------------- #
# validate + train loop  (vision_language/finetune.py:120-315)
# --------------------------------------------------------------------------- #
def loader_iter_draw() -> None:
    """Every ``iter(DataLoader)`` -- shuffled or not -- draws one int64 from torch's
    global CPU generator for its ``_base_seed``; ``for batch in val_loader``
    (finetune.py:295) therefore advances the stream the training samplers seed
    themselves from.  Restated so the batch order stays seed-identical."""
    import torch
    torch.empty((), dtype=torch.int64).random_()


def validate(state: HeadState, feats: np.ndarray, labels: np.ndarray, batch_size: int,
             rng_draw: bool = True) -> Tuple[float, float]:
    """finetune.py:291-315: sequential batches; val_acc = global mean of correct;
    val_loss = mean of the per-batch mean losses (not sample weighted, :311-312,
    accumulated in float32 like torch.stack(loss).mean())."""
    if rng_draw:
        loader_iter_draw()
    losses, correct = [], 0
    for s in range(0, feats.shape[0], batch_size):
        z, _ = forward(state, feats[s:s + batch_size], None)
        y = labels[s:s + batch_size]
        losses.append(F32(cross_entropy_mean(z, y)))
        correct += int(top1_correct(z, y).sum())
    val_acc = float(F32(correct) / F32(feats.shape[0]))
    val_loss = float(np.mean(np.asarray(losses, dtype=F32), dtype=F32))
    return val_loss, val_acc


def train_loop(state: HeadState, opt: OptState, sched: LRSchedule,
               img: Optional[Tuple[np.ndarray, np.ndarray]], txt: Optional[Tuple[np.ndarray, np.ndarray]],
               val: Tuple[np.ndarray, np.ndarray], test: Optional[Tuple[np.ndarray, np.ndarray]],
               batch_size: int, max_iters: int, alpha: float, eval_freq: int = 100, patience: int = 5,
               record: Optional[dict] = None) -> dict:
    """Restatement of ``finetune.train`` (finetune.py:120-288) without the
    logging-only diagnostics: per step fetch img batch then txt batch (:164-174),
    loss = 1.0*L_img + alpha*L_txt (:160,:188), backward, optimizer.step,
    scheduler.step (:193-195); every ``eval_freq`` iters incl. i=0 (:247)
    snapshot -> validate -> strict '>' best tracking (:257) -> patience (:269);
    finally restore the best snapshot (:274).  Returns the same dict keys (:121).
    """
    assert img is not None or txt is not None                         # :123
    out = {"iter": None, "val_acc": None, "model": None, "val_classwise": None,
           "val_loss": None, "model_records": []}
    img_batches = ShuffledBatches(img[0].shape[0], batch_size) if img is not None else None
    txt_batches = ShuffledBatches(txt[0].shape[0], batch_size) if txt is not None else None
    if img_batches is not None:
        img_batches.start()                                           # iter(image_loader) :157
    if txt_batches is not None:
        txt_batches.start()                                           # iter(text_loader)  :158
    no_improve = 0
    for i in range(max_iters):
        xi = yi = xt = yt = None
        if img_batches is not None:
            ii = img_batches.next()
            xi, yi = img[0][ii], img[1][ii]
        if txt_batches is not None:
            ti = txt_batches.next()
            xt, yt = txt[0][ti], txt[1][ti]
        so = step_grads(state, xi, yi, xt, yt, alpha)
        lr = sched.get_last_lr()
        optimizer_step(state, so.grads, opt, lr)
        sched.step()
        if record is not None:
            record.setdefault("loss_img", []).append(so.loss_img)
            record.setdefault("loss_txt", []).append(so.loss_txt)
            record.setdefault("lr", []).append(lr)
            record.setdefault("grad_diag", []).append(grad_diagnostics(so.g_head_img, so.g_head_txt))
            if img_batches is not None:
                record.setdefault("idx_img", []).append(ii.copy())
            if txt_batches is not None:
                record.setdefault("idx_txt", []).append(ti.copy())
        if i % eval_freq == 0:
            snap = state.copy()
            val_loss, val_acc = validate(state, val[0], val[1], batch_size)
            if test is not None:
                validate(state, test[0], test[1], batch_size)
            if record is not None:
                record.setdefault("val_iter", []).append(i)
                record.setdefault("val_loss", []).append(val_loss)
                record.setdefault("val_acc", []).append(val_acc)
            if out["val_acc"] is None or val_acc > out["val_acc"]:
                out.update(iter=i, val_acc=val_acc, val_loss=val_loss, model=snap)
                no_improve = 0
            else:
                no_improve += 1
            if no_improve >= patience:
                break
    best = out["model"]
    state.w_head = best.w_head.copy()
    state.w_proj = None if best.w_proj is None else best.w_proj.copy()
    state.img_scale, state.txt_scale = best.img_scale, best.txt_scale
    return out
