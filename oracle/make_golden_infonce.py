#!/usr/bin/env python3
"""Golden vectors for SequenceInfoNCELoss and for the UML step with infoNCE_loss=True, by RUNNING THE REFERENCE's
MultiBench/models.py on CPU (eval mode: dropout off).  Writes tests/golden/infonce.npz (data only).  Build container only."""
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


def loss_case(rec, tag, B, T, D, seed, masked, temperature=0.07):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(B, T, D, generator=g, requires_grad=True)
    tgt = torch.randn(B, T, D, generator=g)
    mask = None
    if masked:
        lengths = torch.randint(1, T + 1, (B,), generator=g)
        lengths[0] = T
        mask = torch.arange(T).unsqueeze(0) < lengths.unsqueeze(1)
    crit = R.SequenceInfoNCELoss(temperature)
    loss = crit(pred, tgt, mask=mask)
    loss.backward()
    rec[f"{tag}::pred"], rec[f"{tag}::tgt"] = pred.detach(), tgt
    rec[f"{tag}::mask"] = mask if mask is not None else torch.ones(B, T, dtype=torch.bool)
    rec[f"{tag}::masked"] = np.asarray(int(masked))
    rec[f"{tag}::temperature"] = np.asarray(temperature)
    rec[f"{tag}::loss"], rec[f"{tag}::dpred"] = loss.detach(), pred.grad.detach()
    print(tag, float(loss))


def model_case(rec, seed=11, z=20, dx=12, dy=18, B=5, T=9):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = R.UML(R.Linear(dx, z), R.Linear(dy, z),
                  R.Transformer(z, z, nhead=5, num_layers=2, conv1d=True, out_last=False, pos_embd=True, pos_learnable=False, max_len=128),
                  [R.Linear(z, dx), R.Linear(z, dy)], modality="xy", infoNCE_loss=True)
    m.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x, y = torch.randn(B, T, dx, generator=g), torch.randn(B, T, dy, generator=g)
    lx, ly = torch.randint(2, T + 1, (B,), generator=g), torch.randint(2, T + 1, (B,), generator=g)
    lx[0] = ly[0] = T
    out = m(x, y, lx, ly)
    (out["loss_x"] + out["loss_y"]).backward()
    rec["model::cfg"] = np.asarray([z, dx, dy, B, T])
    rec["model::x"], rec["model::y"], rec["model::lx"], rec["model::ly"] = x, y, lx, ly
    rec["model::loss_x"], rec["model::loss_y"] = out["loss_x"].detach(), out["loss_y"].detach()
    for k, v in m.state_dict().items():
        rec["model::sd::" + k] = v.detach().clone()
    for k, p in m.named_parameters():
        if p.grad is not None and any(s in k for s in ("decoders.1", "yproj_in", "encoder.conv", "layers.1.norm2", "layers.0.self_attn.in_proj_bias")):
            rec["model::g::" + k] = p.grad.detach().clone()
    print("model", float(out["loss_x"]), float(out["loss_y"]))


if __name__ == "__main__":
    torch.set_num_threads(4)
    rec = {}
    loss_case(rec, "small", 3, 5, 8, seed=1, masked=False)
    loss_case(rec, "masked", 6, 11, 35, seed=2, masked=True)
    loss_case(rec, "wide", 4, 17, 300, seed=3, masked=True, temperature=0.2)
    loss_case(rec, "one_row_seqs", 5, 2, 16, seed=4, masked=True)
    model_case(rec)
    path = os.path.join(OUT, "infonce.npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in rec.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
