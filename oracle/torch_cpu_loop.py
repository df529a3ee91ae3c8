"""CPU BASELINE for bench.py's ``cpu_baseline`` leg  --  TEST/BENCH INFRASTRUCTURE ONLY.

A PyTorch-CPU restatement ("port") of the reference's training loop as shipped
(vision_language/finetune.py:157-206), used only to time the CPU path beside the
GPU path on the same box.  The reference source cannot travel to the GPU box, so
this restates, with the same third-party PyTorch ops the reference calls, what
one iteration costs there:

  A. ``reference_shaped_steps``: per-sample ``Dataset.__getitem__`` + default collate
     through ``torch.utils.data.DataLoader`` (finetune.py:370-371), ``model(img, txt)``
     (head.py:77-84) incl. the second no-grad feature pass (:183), two
     ``F.cross_entropy`` (:186-187), the two diagnostic ``autograd.grad`` calls
     (:190-191), ``loss.backward(retain_graph=True)`` (:193), ``optimizer.step()``,
     ``scheduler.step()`` (:194-195) and the per-step accuracy / gradient-cosine
     scalars (:197-206).
  B. ``bare_math_steps``: only Linear + CE x2 + backward + AdamW on pre-batched tensors
     (an upper bound for any CPU implementation).

Calibration against the real reference loop (oracle/calibrate_cpu_port.py, build container, 8 cores,
torch 2.10 CPU, cfg2-shaped synthetic tensors): the reference's own ``finetune.train()`` runs at
5.6-6.0e4 samples/s (step times between successive optimizer.step() calls), this port at 5.8-6.5e4 =
1.02-1.09x of it, the bare math at 2.3-3.0e5.  The port is therefore a slightly OPTIMISTIC stand-in for the
reference (GPU/CPU ratios quoted against it are conservative by that factor).
"""
from __future__ import annotations

import math
import time

import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset

CALIBRATION = ("port / real reference finetune.train() = 1.02-1.09 (5.8-6.5e4 vs 5.6-6.0e4 samples/s, 8 cores, build "
               "container, oracle/calibrate_cpu_port.py)")


class _RowDictDS(Dataset):
    """What DatasetWrapper yields per sample (engine/datasets/utils.py:153-174), from rows."""

    def __init__(self, x, y):
        self.x, self.y = x, y

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return {"img": self.x[i], "label": int(self.y[i]), "classname": "c", "impath": "p"}


class _RowTupleDS(Dataset):
    """TextTensorDataset.__getitem__ (engine/datasets/utils.py:100-101)."""

    def __init__(self, x, y):
        self.x, self.y, self.e = x, y, torch.zeros_like(y)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i], self.e[i]


class _Head(torch.nn.Module):
    def __init__(self, d, C, scale):
        super().__init__()
        self.head = torch.nn.Linear(d, C, bias=False)
        self.scale = scale

    def forward(self, img, txt=None):
        a = self.head(img) * self.scale
        return a, (self.head(txt) * self.scale if txt is not None else None)

    def extract_features(self, img):
        return img


def _fetch(loader, it):
    try:
        b = next(it)
    except StopIteration:
        it = iter(loader)
        b = next(it)
    return b, it


def _cos_lr(opt, k, base, warm, T, wlr):
    lr = (wlr if k == 0 else base * k / warm) if k < warm else base * (1 + math.cos(math.pi * (k - warm) / T)) / 2
    for g in opt.param_groups:
        g["lr"] = lr


def reference_shaped_steps(x_img, y_img, x_txt, y_txt, C, batch, steps, warmup, scale=100.0, alpha=1.0, lr=1e-3, wd=0.01,
                           threads=None, w0=None):
    """Returns (samples_per_s, seconds, samples) over ``steps`` timed iterations."""
    if threads:
        torch.set_num_threads(threads)
    d = x_img.shape[1]
    model = _Head(d, C, scale)
    if w0 is not None:
        with torch.no_grad():
            model.head.weight.copy_(w0)
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd, betas=(0.9, 0.999))
    il = DataLoader(_RowDictDS(x_img, y_img), batch_size=batch, shuffle=True, num_workers=0, drop_last=False)
    tl = DataLoader(_RowTupleDS(x_txt, y_txt), batch_size=batch, shuffle=True, num_workers=0, drop_last=False)
    ii, ti = iter(il), iter(tl)
    samples, t0 = 0, None
    for k in range(warmup + steps):
        if k == warmup:
            t0 = time.perf_counter()
            samples = 0
        b, ii = _fetch(il, ii)
        images, image_labels = b["img"], b["label"]
        (text_features, text_labels, _), ti = _fetch(tl, ti)
        labels = torch.cat([image_labels, text_labels])
        flags = torch.cat([torch.ones_like(image_labels), torch.zeros_like(text_labels)])
        opt.zero_grad()
        image_logits, text_logits = model(images, text_features)
        with torch.no_grad():
            image_feature = model.extract_features(images).detach()  # noqa: F841
        im, tm = flags == 1, flags == 0
        image_loss = F.cross_entropy(image_logits, labels[im])
        text_loss = F.cross_entropy(text_logits, labels[tm])
        loss = 1.0 * image_loss + alpha * text_loss
        (g_i,) = torch.autograd.grad(image_loss, model.head.weight, retain_graph=True)
        (g_t,) = torch.autograd.grad(text_loss, model.head.weight, retain_graph=True)
        loss.backward(retain_graph=True)
        _cos_lr(opt, k, lr, 50, 12800, 1e-5)
        opt.step()
        _ = (image_logits.argmax(1) == labels[im]).float().mean().item()
        _ = (text_logits.argmax(1) == labels[tm]).float().mean().item()
        _ = torch.abs(g_i.mean(0)), torch.abs(g_t.mean(0))
        gi, gt = g_i.flatten(), g_t.flatten()
        _ = torch.dot(gi, gt) / (gi.norm() * gt.norm())
        _ = (torch.sign(gi) == torch.sign(gt)).float().mean()
        samples += images.shape[0] + text_features.shape[0]
    dt = time.perf_counter() - t0
    return samples / dt, dt, samples


def bare_math_steps(x_img, y_img, x_txt, y_txt, C, batch, steps, warmup, scale=100.0, alpha=1.0, lr=1e-3, wd=0.01,
                    threads=None):
    if threads:
        torch.set_num_threads(threads)
    d = x_img.shape[1]
    model = _Head(d, C, scale)
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=wd)
    g = torch.Generator().manual_seed(0)
    samples, t0 = 0, None
    for k in range(warmup + steps):
        if k == warmup:
            t0 = time.perf_counter()
            samples = 0
        ii = torch.randint(0, x_img.shape[0], (batch,), generator=g)
        ti = torch.randint(0, x_txt.shape[0], (min(batch, x_txt.shape[0]),), generator=g)
        xi, yi, xt, yt = x_img[ii], y_img[ii], x_txt[ti], y_txt[ti]
        opt.zero_grad()
        a, b = model(xi, xt)
        (F.cross_entropy(a, yi) + alpha * F.cross_entropy(b, yt)).backward()
        opt.step()
        samples += xi.shape[0] + xt.shape[0]
    dt = time.perf_counter() - t0
    return samples / dt, dt, samples
