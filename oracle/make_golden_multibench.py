#!/usr/bin/env python3
"""Golden vectors for the MultiBench step by RUNNING THE REFERENCE's models.py (importable
as-is on CPU).  Eval mode (dropout off) so results are deterministic.
Writes tests/golden/mb_*.npz (data only).  Build container only."""
import contextlib
import io
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.path.insert(0, "/root/reference/MultiBench")
with contextlib.redirect_stdout(io.StringIO()):
    import models as R            # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def build(z, dx, dy, pos_embd, pos_learnable, seed):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = R.UML(R.Linear(dx, z), R.Linear(dy, z),
                  R.Transformer(z, z, nhead=5, num_layers=5, conv1d=True, out_last=False, pos_embd=pos_embd,
                                pos_learnable=pos_learnable, max_len=128),
                  [R.Linear(z, dx), R.Linear(z, dy)], modality="xy")     # MultiBench/main.py:117-121
    return m.eval()


def case(tag, z, dx, dy, B, T, pos_embd, pos_learnable, seed, ax, ay):
    m = build(z, dx, dy, pos_embd, pos_learnable, seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, dx, generator=g)
    y = torch.randn(B, T, dy, generator=g)
    lx = torch.randint(2, T + 1, (B,), generator=g)
    ly = torch.randint(2, T + 1, (B,), generator=g)
    lx[0] = T
    ly[0] = T
    out = m(x, y, lx, ly)
    loss = ax * out["loss_x"] + ay * out["loss_y"]           # train.py:394
    m.zero_grad()
    loss.backward()
    rec = {"x": x, "y": y, "lx": lx, "ly": ly, "alpha": np.asarray([ax, ay]),
           "loss_x": out["loss_x"].detach(), "loss_y": out["loss_y"].detach(), "loss_private": out["loss_private"].detach(),
           "zx": out["zx"].detach(), "zy": out["zy"].detach(), "x_recon": out["x_recon"].detach(), "y_recon": out["y_recon"].detach(),
           "diff_next_x": out["diff_next_x"].detach(), "diff_next_y": out["diff_next_y"].detach(),
           "cfg": np.asarray([z, dx, dy, B, T, int(pos_embd), int(pos_learnable)])}
    names = []
    for k, v in m.state_dict().items():
        rec["sd::" + k] = v.detach().clone()
    keep = ("xproj_in", "yproj_in", "decoders", "encoder.conv", "encoder.pos_embedding", "layers.0.self_attn", "layers.4.norm2",
            "layers.2.linear2.bias")
    for k, p in m.named_parameters():
        names.append(k)
        gnorm = float(p.grad.norm()) if p.grad is not None else -1.0
        rec["gn::" + k] = gnorm
        if p.grad is not None and any(s in k for s in keep):
            rec["g::" + k] = p.grad.detach().clone()
    # 4 Adam steps of the alternation loop (train.py:354-398): 2 "epochs" x 2 batches, step_k = 0
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)            # main.py:122
    losses = []
    for epoch in range(2):
        a0 = 0.0 if epoch <= 0 else ax                         # train.py:356-358 with step_k = 0, mode 'xy'
        for bidx in range(2):
            gb = torch.Generator().manual_seed(1000 * seed + 10 * epoch + bidx)
            xb = torch.randn(B, T, dx, generator=gb)
            yb = torch.randn(B, T, dy, generator=gb)
            o = m(xb, yb, lx, ly)
            l = a0 * o["loss_x"] + ay * o["loss_y"]
            opt.zero_grad()
            l.backward()
            opt.step()
            losses.append([float(o["loss_x"]), float(o["loss_y"]), float(l)])
    rec["traj_losses"] = np.asarray(losses)
    rec["traj_dec0_w"] = m.decoders[0].fc.weight.detach()
    rec["traj_xproj_w"] = m.xproj_in.fc.weight.detach()
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"mb_{tag}.npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in rec.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) loss_x={float(out['loss_x']):.5f} loss_y={float(out['loss_y']):.5f}")


if __name__ == "__main__":
    torch.set_num_threads(4)
    case("z20_sin", 20, 35, 300, 6, 12, True, False, seed=3, ax=1.0, ay=1.0)       # MOSEI dims, sinusoidal positions
    case("z40_learn", 40, 35, 74, 5, 9, True, True, seed=4, ax=0.5, ay=2.0)        # learnable positions
    case("z20_nopos", 20, 16, 24, 4, 7, False, False, seed=5, ax=1.0, ay=1.0)
