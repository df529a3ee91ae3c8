#!/usr/bin/env python3
"""Probe: one micro step, weight error vs the oracle by class row and column."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import numpy as np, torch, umlh
from oracle import uml_oracle as O
DEV = "cuda:0"
d, C, ri, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(1)
n_img = 300
T = lambda a, t=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(DEV, t).contiguous()
xi = rng.standard_normal((n_img, d)).astype(np.float32); xi /= np.linalg.norm(xi, axis=1, keepdims=True)
yi = rng.integers(0, C, n_img)
w0 = rng.standard_normal((C, d)).astype(np.float32); w0 /= np.linalg.norm(w0, axis=1, keepdims=True)
bi = [rng.permutation(n_img)[:ri] for k in range(steps)]
lrs = [0.05] * steps
st = O.HeadState(w0.copy(), None, 20.0, 20.0, False); opt = O.OptState("sgd", 0.0)
for k in range(steps):
    so = O.step_grads(st, xi[bi[k]], yi[bi[k]], None, None, 1.0); O.optimizer_step(st, so.grads, opt, lrs[k])
e = umlh.HeadEngine(d, d, C, optimizer="sgd", weight_decay=0.0, max_rows_img=64, max_rows_txt=64, device=DEV)
e.w_head.copy_(T(w0)); e.scales.fill_(20.0)
sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
e.train_steps((T(xi), T(yi, torch.int64)), [T(b, torch.int64) for b in bi], None, None, lrs, first_step=1, scalars_out=sc)
torch.cuda.synchronize()
err = np.abs(e.w_head.cpu().numpy() - st.w_head)
moved = np.abs(st.w_head - w0).max()
print(f"d={d} C={C} rows={ri} steps={steps}: launches {e.micro_launches()} moved {moved:.2e} max err {err.max():.2e}")
print(" per class :", " ".join(f"{x:.0e}" for x in err.max(1)))
print(" per column:", " ".join(f"{x:.0e}" for x in err.max(0)))
