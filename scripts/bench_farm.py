#!/usr/bin/env python3
"""Sweep farm vs sequential sweep on one MI355X.  Measured: the sequential sweep runs
at ~2.75e4 steps/s, GPU-bound on three dependent small kernels per step; the farm reproduces every point bit for bit but
does not exceed ~2.0e4 steps/s -- the dispatch rate of dependent small kernels, not host threads, is the limit.

Workload: BASELINE configs[0]-shaped synthetic features (Caltech101 16-shot, CLIP ViT-B/16: d=512, C=100, 1600 train
rows, 3000 CUPL text rows, batch 32) x the `clip_linear` grid of engine/optimizer/default.py (6 AdamW points), replicated
over `--alphas` (the reference sweeps alpha and seeds on top of HYPER_DICT: configs/finetune.yaml:14-18), fixed
`--iters` steps per point (patience disabled so every point does the same work)."""
import argparse
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 6, 12])
    ap.add_argument("--replicas", type=int, default=3, help="copies of the 6-point grid (stand-ins for alpha / seed axes)")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--grouped", action="store_true", help="also time the sweep as one grouped job (sweep_mode='grouped': every "
                                                          "grid point a head of the same persistent launches)")
    a = ap.parse_args()
    import finetune as ft
    from engine.datasets.utils import TextTensorDataset
    from engine.optimizer.default import HYPER_DICT
    C, d = 100, 512
    g = torch.Generator().manual_seed(0)
    proto = torch.randn(C, d, generator=g)

    def draw(n):
        y = torch.randint(0, C, (n,), generator=g)
        return torch.nn.functional.normalize(proto[y] + 3.0 * torch.randn(n, d, generator=g), dim=1), y
    tr = (torch.nn.functional.normalize(torch.cat([proto[c] + 3.0 * torch.randn(16, d, generator=g) for c in range(C)]), dim=1),
          torch.arange(C).repeat_interleave(16))
    va, te, (xt, yt) = draw(400), draw(2465), draw(3000)
    text_ds = TextTensorDataset(xt, yt, torch.zeros(len(yt), dtype=torch.long))
    datasets = {"img_tr": tr, "img_val": va, "img_te": te, "text_ds": text_ds}
    grid = dict(HYPER_DICT["clip_linear"])
    grid["max_iter"] = [a.iters]
    grid["patience"] = [10 ** 6]
    grid["dropout"] = [None] + [0.1 * (k + 1) for k in range(a.replicas - 1)]   # unused by the head: replicates the grid, distinct result dirs
    n_points = len(ft._grid(grid))
    torch.manual_seed(0)        # (first call: torch.cuda.manual_seed_all -> device_count -> amdsmi_init, a one-time 0.1 s of the process)
    out = []
    for w in [1] + a.workers + (["grouped"] if a.grouped else []):   # the first pass is an untimed warm-up (module load, allocator)
        with tempfile.TemporaryDirectory() as tmp:
            args = types.SimpleNamespace(savepath=tmp, device="cuda:0", modality="crossmodal", alpha=1.0,
                                         classifier_init="zeroshot", use_clip=True, logit=4.60517, nclasses=C, seed=1,
                                         precision=a.precision, sweep_workers=w if w != "grouped" else 1,
                                         sweep_mode="grouped" if w == "grouped" else "", eval_test=False, order_rng="torch-cpu")
            devnull = open(os.devnull, "w")
            so = sys.stdout
            sys.stdout = devnull
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res, bv, bt = ft.sweep(datasets, grid, args)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                sys.stdout = so
            steps = n_points * a.iters
            if not out and w == 1 and len(out) == 0 and not getattr(main, "_warm", False):
                main._warm = True
                continue
            out.append((w, dt, steps / dt, bv))
            print(f"workers={w!s:>7}  points={n_points}  {dt:7.2f} s  {steps / dt:10.0f} steps/s  {64 * steps / dt:12.0f} samples/s  best_val={bv:.4f}", flush=True)
    base = out[0][2]
    for w, dt, r, _ in out[1:]:
        print(f"{'grouped launch' if w == 'grouped' else 'farm x%d' % w}: {r / base:.2f}x the sequential sweep")


if __name__ == "__main__":
    main()
