#!/usr/bin/env python3
"""cProfile of one small-batch grid point (where does a cfg1-shaped step spend host time?)."""
import cProfile
import os
import pstats
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch
import finetune as ft
from engine.datasets.utils import TextTensorDataset

C, d = 100, 512
g = torch.Generator().manual_seed(0)
proto = torch.randn(C, d, generator=g)


def draw(n):
    y = torch.randint(0, C, (n,), generator=g)
    return torch.nn.functional.normalize(proto[y] + 3.0 * torch.randn(n, d, generator=g), dim=1), y


tr, va, te, (xt, yt) = draw(1600), draw(400), draw(2465), draw(3000)
text_ds = TextTensorDataset(xt, yt, torch.zeros(len(yt), dtype=torch.long))
hp = {"optim": "adamw", "lr": 1e-3, "weight_decay": 0.01, "lr_scheduler": "cosine", "batch_size": 32, "max_iter": int(sys.argv[1]) if len(sys.argv) > 1 else 2000,
      "warmup_iter": 50, "warmup_type": "linear", "warmup_min_lr": 1e-5, "patience": 10 ** 6, "dropout": None, "learnable_temp": False}
so = sys.stdout
for rep in range(2):
    sys.stdout = open(os.devnull, "w")
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    ft.setup_feature_run(tr, va, te, text_ds, hp, num_classes=C, use_clip=True, device="cuda:0", eval_test=False)
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    sys.stdout = so
    print(f"rep {rep}: {dt:.3f} s  ({1e6 * dt / hp['max_iter']:.1f} us/step)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
