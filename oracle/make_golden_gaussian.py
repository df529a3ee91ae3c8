#!/usr/bin/env python3
"""Golden vectors for the Gaussian toy, produced by the reference itself (Gaussian_experiment/{model,data,dataset}.py
imported from /root/reference; test infrastructure only): generated data, initial weights, per-step losses of
`train_model_steps`'s body (main.py:47-61) for 6 Adam steps in 'xy' mode, gradients of the first step, final weights."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/Gaussian_experiment"
sys.path.insert(0, REF)
from data import generate_data                      # noqa: E402
from dataset import UnpairedDataset                 # noqa: E402
from model import SharedAutoencoder                 # noqa: E402
from utils import make_reproducible                 # noqa: E402
from torch.utils.data import DataLoader             # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    cfg = {"seed": 42, "num_samples": 600, "dim_c": 10, "dim_x": 5, "dim_y": 5, "dim_obs": 50, "noise_std": 0.09,
           "attenuate_x": True, "attenuation": 0.05, "shared_latent_distribution_type": "gaussian"}
    data = generate_data(cfg)
    val = generate_data(dict(cfg, seed=43, num_samples=64, attenuate_x=False))
    lap = generate_data(dict(cfg, seed=44, num_samples=8, shared_latent_distribution_type="laplace"))
    n = cfg["num_samples"]
    ds = UnpairedDataset(data["x"][:n // 2], data["y"][:n - n // 2])
    g = torch.Generator()
    g.manual_seed(42)
    loader = DataLoader(ds, batch_size=128, shuffle=True, drop_last=True, generator=g)
    make_reproducible(0)
    model = SharedAutoencoder(dim_obs=50, dim_common=128, dim_latent=10)
    init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    model.train()
    it = iter(loader)
    lx, ly, grads1 = [], [], None
    for step in range(6):                                            # main.py:40-61
        try:
            batch = next(it)
        except StopIteration:
            it = iter(loader)
            batch = next(it)
        opt.zero_grad()
        loss_x, loss_y, _, _ = model(batch["x"], batch["y"])
        loss = 1.0 * loss_x + 0.5 * loss_y
        loss.backward()
        if step == 0:
            grads1 = {k: p.grad.detach().clone().numpy() for k, p in model.named_parameters()}
        opt.step()
        lx.append(float(loss_x)); ly.append(float(loss_y))
    model.eval()
    with torch.no_grad():
        _, _, rvx, rvy = model(x=val["x"], y=val["y"])
        vlx, vly = float(model.loss_fn(rvx, val["x"])), float(model.loss_fn(rvy, val["y"]))
        ex, ey = model.get_embeddings(x=val["x"], y=val["y"])
    rec = {"data_x": data["x"].numpy(), "data_y": data["y"].numpy(), "val_x": val["x"].numpy(), "val_y": val["y"].numpy(),
           "laplace_y": lap["y"].numpy(), "loss_x": np.asarray(lx), "loss_y": np.asarray(ly), "val_loss_x": vlx, "val_loss_y": vly,
           "emb_x": ex.numpy(), "emb_y": ey.numpy()}
    for k, v in init.items():
        rec["init/" + k] = v
    for k, v in grads1.items():
        rec["grad1/" + k] = v
    for k, v in model.state_dict().items():
        rec["final/" + k] = v.detach().numpy()
    path = os.path.join(OUT, "gaussian_toy.npz")
    np.savez_compressed(path, **rec)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)  loss_x {lx[0]:.4f}->{lx[-1]:.4f}  loss_y {ly[0]:.4f}->{ly[-1]:.4f}")


if __name__ == "__main__":
    main()
