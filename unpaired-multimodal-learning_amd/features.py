"""On-disk feature-tensor format of the reference's offline extractor and its path scheme
(reference: vision_language/features.py:32-44 paths, :96-103,:143-149 text payload, :180-184 and
:239-246 image payload).  Only the READ side is built here: extraction itself runs frozen
backbones that need downloaded weights (out of scope).

    image train file : {'train': {'features','labels','paths'}, 'val': {...}, 'lab2cname'}
    image test file  : {'features','labels','paths', 'lab2cname'}
    text file        : {'features','labels','eot_indices','prompts','lab2cname'[, 'cname2lab']}
"""
from __future__ import annotations

import os

import torch


def get_few_shot_setup_name(train_shot, seed):
    return f"shot_{train_shot}-seed_{seed}"


def img_outdir(outdir, encoder, ds, augmentation, tr_shot, seed, mode="train", return_tokens=False):
    sub = "patch-token" if return_tokens else ""
    enc = encoder.replace("/", "-")
    if mode == "train":
        return os.path.join(outdir, sub, "image", enc, ds, augmentation, f"{get_few_shot_setup_name(tr_shot, seed)}.pth")
    return os.path.join(outdir, sub, "image", enc, ds, "test.pth")


def text_outdir(outdir, encoder, ds, text_augmentation, return_tokens=False):
    sub = "patch-token" if return_tokens else ""
    return os.path.join(outdir, sub, "text", encoder.replace("/", "-"), ds, f"{text_augmentation}.pth")


def descriptor_outdir(outdir, encoder, ds, descriptor_type, return_tokens=False):
    return text_outdir(outdir, encoder, ds, descriptor_type, return_tokens)


def _load(path):
    # tensors + plain containers only: nothing from the file is executed
    return torch.load(path, map_location="cpu", weights_only=True)


def _pair(d, what):
    if "features" not in d or "labels" not in d:
        raise KeyError(f"{what}: expected keys 'features' and 'labels', found {sorted(d)}")
    f, y = d["features"], d["labels"]
    if f.dim() != 2 or y.dim() != 1 or f.shape[0] != y.shape[0]:
        raise ValueError(f"{what}: features {tuple(f.shape)} / labels {tuple(y.shape)} are not [N,d] / [N]")
    return f.float().contiguous(), y.long().contiguous()


def load_image_train_features(path):
    """-> {'train': (X, y), 'val': (X, y), 'lab2cname': ...}"""
    d = _load(path)
    return {"train": _pair(d["train"], f"{path}[train]"), "val": _pair(d["val"], f"{path}[val]"),
            "lab2cname": d.get("lab2cname")}


def load_image_test_features(path):
    d = _load(path)
    return {"test": _pair(d, path), "lab2cname": d.get("lab2cname")}


def load_text_features(path):
    """-> dict with 'features' [N,d] fp32, 'labels' [N] int64, 'eot_indices' [N], + passthrough keys."""
    d = _load(path)
    f, y = _pair(d, path)
    eot = d.get("eot_indices")
    if eot is None:
        eot = torch.zeros_like(y)
    out = dict(d)
    out.update(features=f, labels=y, eot_indices=eot)
    return out
