#!/usr/bin/env python3
"""Golden vectors for heads WITH bias (engine/models/head.py:65,68,122 ``bias=True``) by RUNNING THE REFERENCE's head classes
and its own build_optimizer / build_lr_scheduler on CPU: logits, losses, autograd gradients of weight and bias, and a short
AdamW / SGD trajectory.  Writes tests/golden/bias_heads.npz (data only).  Build container only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G   # noqa: E402  (stubs the absent third-party packages, imports the reference)


def case(rec, tag, kind, d, C, Bi, Bt, optim, wd, steps, alpha, seed, text_indim=0):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    if kind == "clip":
        m = G.make_umlclip(d, C, 4.60517)
        m.head = torch.nn.Linear(d, C, bias=True)                       # head.py:122 with bias=True
    else:
        G._FEAT_D["d"] = d
        m = G.quiet(G.RefUML, "identity", text_indim, C, bias=True, learnable_temp=False, freeze_backbone=False)   # head.py:65,68
    with torch.no_grad():
        m.head.weight.mul_(3.0)
        m.head.bias.uniform_(-0.5, 0.5, generator=gen)
        if text_indim:
            m.img_proj.bias.uniform_(-0.3, 0.3, generator=gen)
    n = 200
    xi, yi = G.synth(n, d, C, gen)
    xt, yt = G.synth(n, text_indim or d, C, gen)
    rec[f"{tag}::w0"], rec[f"{tag}::b0"] = m.head.weight.detach().clone(), m.head.bias.detach().clone()
    if text_indim:
        rec[f"{tag}::pw0"], rec[f"{tag}::pb0"] = m.img_proj.weight.detach().clone(), m.img_proj.bias.detach().clone()
    rec[f"{tag}::xi"], rec[f"{tag}::yi"], rec[f"{tag}::xt"], rec[f"{tag}::yt"] = xi, yi, xt, yt
    opt = G.ref_build_optimizer(m.parameters(), optim, 1e-3, wd)
    sch = G.ref_build_sched(opt, "cosine", 2, 100, warmup_type="linear", warmup_lr=1e-5)
    idx_i, idx_t, losses, lrs = [], [], [], []
    for k in range(steps):
        ii = torch.randperm(n, generator=gen)[:Bi]
        ti = torch.randperm(n, generator=gen)[:Bt]
        li, lt = m(xi[ii], xt[ti])
        loss_i = torch.nn.functional.cross_entropy(li, yi[ii])
        loss_t = torch.nn.functional.cross_entropy(lt, yt[ti])
        loss = loss_i + alpha * loss_t                                 # finetune.py:186-188
        opt.zero_grad()
        loss.backward()
        if k == 0:
            rec[f"{tag}::logits_img0"], rec[f"{tag}::logits_txt0"] = li.detach().clone(), lt.detach().clone()
            rec[f"{tag}::gw0"], rec[f"{tag}::gb0"] = m.head.weight.grad.clone(), m.head.bias.grad.clone()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        idx_i.append(ii); idx_t.append(ti)
        losses.append([float(loss_i), float(loss_t)])
    rec[f"{tag}::idx_i"], rec[f"{tag}::idx_t"] = torch.stack(idx_i), torch.stack(idx_t)
    rec[f"{tag}::losses"], rec[f"{tag}::lrs"] = np.asarray(losses), np.asarray(lrs)
    rec[f"{tag}::w1"], rec[f"{tag}::b1"] = m.head.weight.detach().clone(), m.head.bias.detach().clone()
    if text_indim:
        rec[f"{tag}::pw1"], rec[f"{tag}::pb1"] = m.img_proj.weight.detach().clone(), m.img_proj.bias.detach().clone()
    rec[f"{tag}::cfg"] = np.asarray([d, C, Bi, Bt, steps, alpha, wd, {"adamw": 2, "adam": 1, "sgd": 0}[optim], text_indim], dtype=np.float64)
    print(tag, losses[0], losses[-1])


if __name__ == "__main__":
    torch.set_num_threads(4)
    rec = {}
    case(rec, "clip_d64_c10_adamw", "clip", 64, 10, 32, 32, "adamw", 0.01, 8, 1.0, seed=1)
    case(rec, "uml_d96_c37_sgd", "uml", 96, 37, 20, 33, "sgd", 1e-3, 8, 0.5, seed=2)
    case(rec, "clip_d512_c100_adam", "clip", 512, 100, 32, 32, "adam", 0.0, 6, 1.0, seed=3)
    case(rec, "mlp_d48_t64_c10_adamw", "uml", 48, 10, 24, 40, "adamw", 0.01, 8, 0.7, seed=4, text_indim=64)   # img_proj + both biases
    G.npz("bias_heads", **rec)
