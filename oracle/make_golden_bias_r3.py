#!/usr/bin/env python3
"""Round-3 golden vectors for heads WITH bias: (i) bias=True together with learnable_temp=True (head.py:65-70) and (ii) the
per-step gradient diagnostics of a bias head, formed as finetune.py:190-191,203-206 forms them (torch.autograd.grad of each
modality's loss with respect to model.head.weight ONLY).  RUNS THE REFERENCE's head class, build_optimizer and
build_lr_scheduler on CPU; writes tests/golden/bias_heads_r3.npz (data only).  Build container only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G   # noqa: E402  (stubs the absent third-party packages, imports the reference)


def case(rec, tag, d, C, Bi, Bt, optim, wd, steps, alpha, seed, text_indim=0, learnable=True):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    G._FEAT_D["d"] = d
    m = G.quiet(G.RefUML, "identity", text_indim, C, bias=True, learnable_temp=learnable, freeze_backbone=False)
    with torch.no_grad():
        m.head.weight.mul_(3.0)
        m.head.bias.uniform_(-0.5, 0.5, generator=gen)
        if text_indim:
            m.img_proj.bias.uniform_(-0.3, 0.3, generator=gen)
        if learnable:
            m.img_scale.fill_(2.0)
            m.txt_scale.fill_(1.5)
    n = 200
    xi, yi = G.synth(n, d, C, gen)
    xt, yt = G.synth(n, text_indim or d, C, gen)
    rec[f"{tag}::w0"], rec[f"{tag}::b0"] = m.head.weight.detach().clone(), m.head.bias.detach().clone()
    if text_indim:
        rec[f"{tag}::pw0"], rec[f"{tag}::pb0"] = m.img_proj.weight.detach().clone(), m.img_proj.bias.detach().clone()
    rec[f"{tag}::xi"], rec[f"{tag}::yi"], rec[f"{tag}::xt"], rec[f"{tag}::yt"] = xi, yi, xt, yt
    opt = G.ref_build_optimizer(m.parameters(), optim, 1e-3, wd)
    sch = G.ref_build_sched(opt, "cosine", 2, 100, warmup_type="linear", warmup_lr=1e-5)
    idx_i, idx_t, losses, lrs, diag, scales = [], [], [], [], [], []
    for k in range(steps):
        ii = torch.randperm(n, generator=gen)[:Bi]
        ti = torch.randperm(n, generator=gen)[:Bt]
        li, lt = m(xi[ii], xt[ti])
        loss_i = torch.nn.functional.cross_entropy(li, yi[ii])
        loss_t = torch.nn.functional.cross_entropy(lt, yt[ti])
        loss = loss_i + alpha * loss_t                                 # finetune.py:186-188 (img_alpha = 1)
        (gi,) = torch.autograd.grad(loss_i, m.head.weight, retain_graph=True)      # finetune.py:190-191
        (gt,) = torch.autograd.grad(loss_t, m.head.weight, retain_graph=True)
        opt.zero_grad()
        loss.backward()
        gi, gt = gi.flatten(), gt.flatten()                                            # finetune.py:203-206
        diag.append([float(torch.dot(gi, gt) / (torch.norm(gi) * torch.norm(gt))), float(torch.norm(gi)), float(torch.norm(gt)),
                     float(torch.mean((torch.sign(gi) == torch.sign(gt)).float()))])
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        idx_i.append(ii); idx_t.append(ti)
        losses.append([float(loss_i), float(loss_t)])
        scales.append([float(m.img_scale), float(m.txt_scale)])
    rec[f"{tag}::idx_i"], rec[f"{tag}::idx_t"] = torch.stack(idx_i), torch.stack(idx_t)
    rec[f"{tag}::losses"], rec[f"{tag}::lrs"] = np.asarray(losses), np.asarray(lrs)
    rec[f"{tag}::diag"], rec[f"{tag}::scales"] = np.asarray(diag), np.asarray(scales)
    rec[f"{tag}::w1"], rec[f"{tag}::b1"] = m.head.weight.detach().clone(), m.head.bias.detach().clone()
    if text_indim:
        rec[f"{tag}::pw1"], rec[f"{tag}::pb1"] = m.img_proj.weight.detach().clone(), m.img_proj.bias.detach().clone()
    rec[f"{tag}::cfg"] = np.asarray([d, C, Bi, Bt, steps, alpha, wd, {"adamw": 2, "adam": 1, "sgd": 0}[optim], text_indim, int(learnable)],
                                    dtype=np.float64)
    print(tag, losses[0], losses[-1], diag[0], scales[-1])


if __name__ == "__main__":
    torch.set_num_threads(4)
    rec = {}
    case(rec, "uml_d96_c37_adamw_learn", 96, 37, 20, 33, "adamw", 0.01, 8, 0.5, seed=11)
    case(rec, "mlp_d48_t64_c10_adamw_learn", 48, 10, 24, 40, "adamw", 0.01, 8, 0.7, seed=12, text_indim=64)
    case(rec, "uml_d128_c20_sgd_fixed", 128, 20, 32, 32, "sgd", 1e-3, 6, 1.0, seed=13, learnable=False)   # d + 1 > 128: packed width 256
    G.npz("bias_heads_r3", **rec)
