#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference; the GPU box never sees
it).  Imports ``/root/reference/vision_language`` as Python with ``sys.modules``
stubs for third-party packages that are absent offline (torchvision, wandb,
ftfy, torchaudio, timm -- ordinary ModuleNotFoundErrors, nothing was denied)
and a fake ``timm.models.create_model`` returning an identity module with
``.num_features = d``: that turns the reference's ``UML`` into exactly
"pre-extracted feature row -> (img_proj) -> head", the regime this build
accelerates.

Only DATA is written: inputs and the reference's outputs, as .npz files.  No
reference source is copied.  Re-run with:  python oracle/make_golden.py
"""
from __future__ import annotations

import importlib.machinery
import io
import contextlib
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch
import transformers  # noqa: F401  (must be imported before the stubs go in)

REF = "/root/reference/vision_language"
REF_MB = "/root/reference/MultiBench"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub(name):
    m = MagicMock(name=name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


for _n in ["torchvision", "torchvision.datasets", "torchvision.datasets.folder",
           "torchvision.transforms", "torchvision.transforms.functional", "wandb", "ftfy",
           "torchaudio", "torchaudio.functional", "timm", "timm.models", "timm.data"]:
    try:
        __import__(_n)
    except Exception:
        _stub(_n)


class _IdentityBackbone(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.num_features = d
        self.num_classes = 0

    def forward(self, x):
        return x


_FEAT_D = {"d": 0}
sys.modules["timm.models"].create_model = lambda name, pretrained=True, **kw: _IdentityBackbone(_FEAT_D["d"])
sys.path.insert(0, REF)

with contextlib.redirect_stdout(io.StringIO()):
    import finetune as ref_finetune                                     # noqa: E402
    from engine.models.head import UML as RefUML, UMLClip as RefUMLClip, get_zero_shot_weights  # noqa: E402
    from engine.datasets.utils import TextTensorDataset as RefTextDS   # noqa: E402
    from engine.optimizer.optim import build_optimizer as ref_build_optimizer      # noqa: E402
    from engine.optimizer.scheduler import build_lr_scheduler as ref_build_sched  # noqa: E402
    from engine.tools.utils import set_random_seed                      # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_uml(d_img, text_indim, C, learnable_temp):
    _FEAT_D["d"] = d_img
    return quiet(RefUML, "identity", text_indim, C, bias=False, learnable_temp=learnable_temp, freeze_backbone=False)


def make_umlclip(d, C, logit):
    """UMLClip without its network-fetching __init__ (clip.load): build the object
    bare, give it the attributes forward() reads, and run the reference forward."""
    m = RefUMLClip.__new__(RefUMLClip)
    torch.nn.Module.__init__(m)
    m.num_classes = C
    m.img_proj = None

    class _Enc(torch.nn.Module):
        embed_dim = d

        def encode_image(self, x):
            return x
    m.vision_model = _Enc()
    m.shared_dim = d
    m.head = torch.nn.Linear(d, C, bias=False)
    m.logit_scale = torch.tensor(logit)
    return m


def synth(n, d, C, gen, proto=None, noise=1.0, normalize=True):
    y = torch.randint(0, C, (n,), generator=gen)
    x = torch.randn(n, d, generator=gen) * noise
    if proto is not None:
        x = x + proto[y]
    if normalize:
        x = torch.nn.functional.normalize(x, dim=1)
    return x.float(), y.long()


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------- #
# (i) single step: logits, losses, grads from the reference model + autograd
# --------------------------------------------------------------------------- #
def single_step(tag, kind, d_img, d_shared, C, Bi, Bt, alpha, scale=None, learnable=False, seed=0):
    gen = torch.Generator().manual_seed(seed)
    xi, yi = synth(Bi, d_img, C, gen)
    xt, yt = synth(Bt, d_shared, C, gen)
    torch.manual_seed(seed + 1)
    if kind == "clip":
        model = make_umlclip(d_shared, C, float(np.log(scale)))
        with torch.no_grad():
            model.head.weight.copy_(torch.nn.functional.normalize(torch.randn(C, d_shared, generator=gen), dim=1))
    else:
        model = make_uml(d_img, d_shared if kind == "mlp" else 0, C, learnable)
        if learnable:
            with torch.no_grad():
                model.img_scale.fill_(1.7)
                model.txt_scale.fill_(0.6)
    model.train()
    img_logits, txt_logits = model(xi, xt)
    # finetune.py:186-193
    image_loss = torch.nn.functional.cross_entropy(img_logits, yi)
    text_loss = torch.nn.functional.cross_entropy(txt_logits, yt)
    loss = 1.0 * image_loss + alpha * text_loss
    (g_img,) = torch.autograd.grad(image_loss, model.head.weight, retain_graph=True)
    (g_txt,) = torch.autograd.grad(text_loss, model.head.weight, retain_graph=True)
    loss.backward()
    out = dict(x_img=xi, y_img=yi, x_txt=xt, y_txt=yt, alpha=alpha,
               w_head=model.head.weight, img_logits=img_logits, txt_logits=txt_logits,
               loss_img=image_loss, loss_txt=text_loss, g_head=model.head.weight.grad,
               g_head_img=g_img, g_head_txt=g_txt,
               acc_img=(img_logits.argmax(1) == yi).float().mean(),
               acc_txt=(txt_logits.argmax(1) == yt).float().mean())
    if kind == "clip":
        out["scale_img"] = out["scale_txt"] = float(model.logit_scale.exp())
    else:
        out["scale_img"] = float(model.img_scale)
        out["scale_txt"] = float(model.txt_scale)
        if model.img_proj is not None:
            out["w_proj"] = model.img_proj.weight
            out["g_proj"] = model.img_proj.weight.grad
        if learnable:
            out["g_img_scale"] = model.img_scale.grad
            out["g_txt_scale"] = model.txt_scale.grad
    npz("step_" + tag, **out)


# --------------------------------------------------------------------------- #
# (ii)+(iii) optimizer/scheduler trajectories through the reference builders
# --------------------------------------------------------------------------- #
def optim_traj(tag, optim, sched, warmup_type, d_img, d_shared, C, B, steps, lr, wd, warm, max_iter,
               warm_lr, learnable, alpha, seed=0):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    model = make_uml(d_img, d_shared, C, learnable)
    proto_i = torch.randn(C, d_img, generator=gen)
    proto_t = torch.randn(C, d_shared if d_shared > 0 else d_img, generator=gen)
    optimizer = ref_build_optimizer(model.parameters(), optim, lr, wd)
    scheduler = ref_build_sched(optimizer, sched, warm, max_iter, warmup_type=warmup_type, warmup_lr=warm_lr)
    rec = dict(w_head0=model.head.weight.detach().clone())
    if model.img_proj is not None:
        rec["w_proj0"] = model.img_proj.weight.detach().clone()
    xs_i, ys_i, xs_t, ys_t, lrs, li, lt = [], [], [], [], [], [], []
    snaps = {}
    for k in range(steps):
        xi, yi = synth(B, d_img, C, gen, proto_i)
        xt, yt = synth(B, proto_t.shape[1], C, gen, proto_t)
        optimizer.zero_grad()
        a, b = model(xi, xt)
        il = torch.nn.functional.cross_entropy(a, yi)
        tl = torch.nn.functional.cross_entropy(b, yt)
        (1.0 * il + alpha * tl).backward()
        lrs.append(optimizer.param_groups[0]["lr"])
        optimizer.step()
        scheduler.step()
        xs_i.append(xi); ys_i.append(yi); xs_t.append(xt); ys_t.append(yt)
        li.append(float(il)); lt.append(float(tl))
        if k in (0, 1, 9, warm - 1, warm, warm + 1, steps - 1):
            snaps[f"w_head_after_{k}"] = model.head.weight.detach().clone()
            if model.img_proj is not None:
                snaps[f"w_proj_after_{k}"] = model.img_proj.weight.detach().clone()
    st = optimizer.state[model.head.weight]
    if optim == "sgd":
        rec["m_head_final"] = st["momentum_buffer"]
    else:
        rec["m_head_final"] = st["exp_avg"]
        rec["v_head_final"] = st["exp_avg_sq"]
    rec.update(snaps)
    rec.update(x_img=torch.stack(xs_i), y_img=torch.stack(ys_i), x_txt=torch.stack(xs_t), y_txt=torch.stack(ys_t),
               lr=np.asarray(lrs, dtype=np.float64), loss_img=np.asarray(li), loss_txt=np.asarray(lt),
               hyper=np.asarray([lr, wd, warm, max_iter, warm_lr, alpha], dtype=np.float64),
               img_scale_final=float(model.img_scale), txt_scale_final=float(model.txt_scale))
    npz("traj_" + tag, **rec)


def lr_traces():
    out = {}
    for tag, (sched, wtype, lr, warm, max_iter, wlr) in {
        "cos_lin": ("cosine", "linear", 1e-3, 50, 12800, 1e-5),       # HYPER_DICT['clip_linear']
        "cos_const": ("cosine", "constant", 1e-4, 20, 300, 1e-6),
        "lin_lin": ("linear", "linear", 5e-5, 10, 200, 1e-5),
        "cos_nowarm": ("cosine", None, 1e-3, 0, 100, None),
    }.items():
        p = torch.nn.Parameter(torch.zeros(1))
        opt = ref_build_optimizer([p], "adamw", lr, 0.0)
        s = ref_build_sched(opt, sched, warm, max_iter, warmup_type=wtype, warmup_lr=wlr)
        n = max_iter + 5 if max_iter <= 300 else max_iter
        vals, last = [], []
        for _ in range(n):
            vals.append(opt.param_groups[0]["lr"])
            last.append(s.get_last_lr()[0])
            p.grad = torch.zeros(1)
            opt.step()
            s.step()
        out[tag] = np.asarray(vals, dtype=np.float64)
        out[tag + "_last"] = np.asarray(last, dtype=np.float64)
        out[tag + "_cfg"] = np.asarray([lr, warm, max_iter, -1 if wlr is None else wlr], dtype=np.float64)
    npz("lr_traces", **out)


# --------------------------------------------------------------------------- #
# (iv)+(v) zero-shot weights, TextTensorDataset reductions
# --------------------------------------------------------------------------- #
def text_side():
    gen = torch.Generator().manual_seed(7)
    C, d = 12, 40
    counts = [5, 0, 3, 7, 1, 4, 4, 0, 6, 2, 5, 3]       # classes 1 and 7 have no text
    labels = torch.cat([torch.full((n,), c, dtype=torch.long) for c, n in enumerate(counts)])
    labels = labels[torch.randperm(labels.numel(), generator=gen)]
    feats = torch.randn(labels.numel(), d, generator=gen)
    eot = torch.randint(0, 77, (labels.numel(),), generator=gen)
    ds = quiet(RefTextDS, feats, labels, eot, n_shots=None)
    w = quiet(get_zero_shot_weights, ds, C, d, device="cpu")
    avg = quiet(RefTextDS, feats, labels, eot, n_shots="average")
    torch.manual_seed(123)
    shot = quiet(RefTextDS, feats, labels, eot, n_shots=3)
    npz("text_side", feats=feats, labels=labels, eot=eot, num_classes=C, zero_shot_w=w,
        avg_feats=avg.input_tensor, avg_labels=avg.label_tensor, avg_eot=avg.eot_indices,
        shot_seed=123, shot_feats=shot.input_tensor, shot_labels=shot.label_tensor, shot_eot=shot.eot_indices)


# --------------------------------------------------------------------------- #
# (vi) full finetune.train() runs
# --------------------------------------------------------------------------- #
class RecordingImageDS(torch.utils.data.Dataset):
    """Yields the dict samples DatasetWrapper yields (engine/datasets/utils.py:153-174)
    from pre-extracted rows, and records the index order the DataLoader asked for."""

    def __init__(self, x, y, log=None):
        self.x, self.y, self.log = x, y, log

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        if self.log is not None:
            self.log.append(int(i))
        return {"img": self.x[i], "label": int(self.y[i]), "classname": "c", "impath": "p"}


class RecordingTextDS(RefTextDS):
    def __init__(self, *a, log=None, **k):
        super().__init__(*a, **k)
        self.log = log

    def __getitem__(self, i):
        if self.log is not None:
            self.log.append(int(i))
        return super().__getitem__(i)


def train_run(tag, d_img, text_indim, C, n_img_per_class, n_txt, n_val, B, max_iters, eval_freq, patience,
              optim, lr, wd, alpha, learnable, zeroshot, seed, noise=2.0, modality="crossmodal"):
    from torch.utils.data import DataLoader
    gen = torch.Generator().manual_seed(1000 + seed)
    d_shared = text_indim if text_indim > 0 else d_img
    proto_i = torch.randn(C, d_img, generator=gen)
    proto_t = torch.randn(C, d_shared, generator=gen)
    xi = torch.cat([proto_i[c] + noise * torch.randn(n_img_per_class, d_img, generator=gen) for c in range(C)])
    yi = torch.arange(C).repeat_interleave(n_img_per_class)
    xi = torch.nn.functional.normalize(xi, dim=1)
    xt, yt = synth(n_txt, d_shared, C, gen, proto_t, noise=noise)
    xv, yv = synth(n_val, d_img, C, gen, proto_i, noise=noise)
    xe, ye = synth(n_val + 7, d_img, C, gen, proto_i, noise=noise)
    eot = torch.zeros(n_txt, dtype=torch.long)

    set_random_seed(seed)                                       # finetune.py:452-454
    img_log, txt_log = [], []
    text_ds = quiet(RecordingTextDS, xt, yt, eot, n_shots=None, log=txt_log)
    model = make_uml(d_img, text_indim, C, learnable)
    w_head_init = model.head.weight.detach().clone()
    w_proj_init = model.img_proj.weight.detach().clone() if model.img_proj is not None else None
    if zeroshot:                                                # finetune.py:362-363 (device quirk a5)
        model.head.weight.data = quiet(get_zero_shot_weights, RefTextDS(xt, yt, eot), C, d_shared, device="cpu")
    optimizer = ref_build_optimizer(model.parameters(), optim, lr, wd)
    scheduler = ref_build_sched(optimizer, "cosine", 50, max_iters, warmup_type="linear", warmup_lr=1e-5)
    image_loader = DataLoader(RecordingImageDS(xi, yi, img_log), batch_size=B, shuffle=True, num_workers=0, drop_last=False)
    text_loader = DataLoader(text_ds, batch_size=B, shuffle=True, num_workers=0, drop_last=False)
    val_loader = DataLoader(RecordingImageDS(xv, yv), batch_size=B, shuffle=False)
    test_loader = DataLoader(RecordingImageDS(xe, ye), batch_size=B, shuffle=False)
    if modality == "image":
        text_loader = None                                      # finetune.py:373-376

    # instrumentation: record every cross_entropy value (train: grad enabled, eval: not)
    F = torch.nn.functional
    orig_ce = F.cross_entropy
    train_ce, eval_ce = [], []

    def rec_ce(inp, tgt, *a, **k):
        v = orig_ce(inp, tgt, *a, **k)
        (train_ce if torch.is_grad_enabled() else eval_ce).append(float(v))
        return v
    F.cross_entropy = rec_ce
    orig_validate = ref_finetune.validate
    val_calls = []

    def rec_validate(m, loader, device="cpu"):
        r = orig_validate(m, loader, device=device)
        val_calls.append((len(loader.dataset), r[0], r[1]))
        return r
    ref_finetune.validate = rec_validate
    # the reference's per-modality head gradients (finetune.py:190-191): every torch.autograd.grad result
    orig_grad = torch.autograd.grad
    grad_calls = []

    def rec_grad(outputs, inputs, *a, **k):
        r = orig_grad(outputs, inputs, *a, **k)
        grad_calls.append(r[0].detach().clone())
        return r
    torch.autograd.grad = rec_grad
    try:
        out = quiet(ref_finetune.train, model, image_loader, text_loader, val_loader, test_loader, optimizer,
                    scheduler, device="cpu", max_iters=max_iters, alpha=alpha, eval_freq=eval_freq,
                    patience=patience, capture_features_during_training=False, args=None, logger=None)
    finally:
        F.cross_entropy = orig_ce
        ref_finetune.validate = orig_validate
        torch.autograd.grad = orig_grad
    test_loss, test_acc = ref_finetune.validate(model, test_loader, device="cpu")   # finetune.py:390
    n_steps = len(img_log) and (len(train_ce) // (2 if modality == "crossmodal" else 1))
    vals = [(l, a) for (n, l, a) in val_calls if n == n_val]
    # gradient diagnostics the reference logs per step, formed from ITS gradients with its own
    # expressions (finetune.py:203-206,238); image-only runs log similarity 0 / agreement 0
    diag = np.zeros((n_steps, 4), dtype=np.float64)      # sim, agreement, |g_img|, |g_txt|
    per = 2 if modality == "crossmodal" else 1
    for k in range(n_steps):
        gi = torch.flatten(grad_calls[per * k])
        gt = torch.flatten(grad_calls[per * k + 1]) if per == 2 else torch.zeros_like(gi)
        if per == 2:
            diag[k, 0] = float(torch.dot(gi, gt) / (torch.norm(gi) * torch.norm(gt)))
            diag[k, 1] = float(torch.mean((torch.sign(gi) == torch.sign(gt)).float()))
        diag[k, 2] = float(torch.norm(gi))
        diag[k, 3] = float(torch.norm(gt))
    rec = dict(x_img=xi, y_img=yi, x_txt=xt, y_txt=yt, x_val=xv, y_val=yv, x_test=xe, y_test=ye,
               w_head_init=w_head_init, idx_img=np.asarray(img_log, dtype=np.int64),
               idx_txt=np.asarray(txt_log, dtype=np.int64),
               train_ce=np.asarray(train_ce, dtype=np.float64), n_steps=n_steps, grad_diag=diag,
               val_loss=np.asarray([v[0] for v in vals]), val_acc=np.asarray([v[1] for v in vals]),
               best_iter=out["iter"], best_val_acc=out["val_acc"], best_val_loss=out["val_loss"],
               w_head_best=out["model"]["head.weight"], w_head_final=model.head.weight,
               test_loss=test_loss, test_acc=test_acc,
               cfg=np.asarray([d_img, text_indim, C, B, max_iters, eval_freq, patience, lr, wd, alpha,
                               int(learnable), int(zeroshot), seed], dtype=np.float64),
               optim=optim, modality=modality)
    if w_proj_init is not None:
        rec["w_proj_init"] = w_proj_init
        rec["w_proj_best"] = out["model"]["img_proj.weight"]
    if learnable:
        rec["img_scale_best"] = out["model"]["img_scale"]
        rec["txt_scale_best"] = out["model"]["txt_scale"]
    npz("train_" + tag, **rec)
    print(f"   {tag}: steps={n_steps} best_iter={out['iter']} best_val_acc={out['val_acc']:.4f} test_acc={test_acc:.4f}")


def main():
    torch.set_num_threads(4)
    # (i)
    single_step("clip_d64_c10", "clip", 64, 64, 10, 8, 6, alpha=0.5, scale=100.0, seed=1)
    single_step("clip_d512_c100", "clip", 512, 512, 100, 32, 32, alpha=1.0, scale=100.0, seed=2)
    single_step("clip_d128_c1000", "clip", 128, 128, 1000, 16, 12, alpha=1.0, scale=100.0, seed=3)
    single_step("lin_d96_c37", "lin", 96, 96, 37, 20, 9, alpha=2.0, seed=4)
    single_step("mlp_d48_t64_c10", "mlp", 48, 64, 10, 12, 10, alpha=0.7, learnable=True, seed=5)
    single_step("mlp_d96_t160_c20", "mlp", 96, 160, 20, 16, 16, alpha=1.0, learnable=False, seed=6)
    # (ii)
    optim_traj("adamw_lin", "adamw", "cosine", "linear", 64, 0, 10, 16, 70, 1e-3, 0.01, 50, 12800, 1e-5, False, 1.0)
    optim_traj("sgd_lin", "sgd", "cosine", "linear", 64, 0, 10, 16, 40, 1e-2, 0.001, 20, 200, 1e-5, False, 0.5)
    optim_traj("adam_lin", "adam", "linear", "constant", 64, 0, 10, 16, 40, 1e-3, 0.01, 10, 100, 1e-4, False, 1.0)
    optim_traj("adamw_mlp", "adamw", "cosine", "linear", 40, 56, 12, 16, 60, 1e-3, 0.01, 50, 12800, 1e-5, True, 1.0)
    # (iii)
    lr_traces()
    # (iv) (v)
    text_side()
    # (vi)
    train_run("lin_zs", 32, 0, 10, 16, 90, 40, 32, 400, 20, 3, "adamw", 1e-3, 0.01, 1.0, False, True, seed=1)
    train_run("mlp_lt", 24, 32, 8, 12, 70, 48, 16, 300, 25, 4, "adamw", 1e-3, 0.001, 0.5, True, True, seed=2)
    train_run("lin_imgonly", 32, 0, 10, 16, 90, 40, 32, 200, 20, 3, "sgd", 1e-2, 0.0, 1.0, False, False, seed=3,
              modality="image")


if __name__ == "__main__":
    main()
