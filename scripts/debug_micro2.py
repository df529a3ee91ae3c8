#!/usr/bin/env python3
"""Probe: the (d=96, C=10, 64 image rows, SGD) case of tests/test_micro_gpu.py, per-step loss error vs the oracle, repeated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import numpy as np, torch, umlh
from oracle import uml_oracle as O
DEV = "cuda:0"
d, C, ri, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 12
OPT = sys.argv[4]
rng = np.random.default_rng(d + 7 * C + ri)
n_img = 300
T = lambda a, t=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(DEV, t).contiguous()
xi = rng.standard_normal((n_img, d)).astype(np.float32); xi /= np.linalg.norm(xi, axis=1, keepdims=True)
xt = rng.standard_normal((260, d)).astype(np.float32)
yi = rng.integers(0, C, n_img); _ = rng.integers(0, C, 260)
w0 = rng.standard_normal((C, d)).astype(np.float32); w0 /= np.linalg.norm(w0, axis=1, keepdims=True)
bi = [rng.permutation(n_img)[:ri if k != 5 else ri - 5] for k in range(steps)]
lrs = [1e-3 * (k + 1) / steps for k in range(steps)]
st = O.HeadState(w0.copy(), None, 20.0, 20.0, False); opt = O.OptState(OPT, 1e-3)
ref = []
for k in range(steps):
    so = O.step_grads(st, xi[bi[k]], yi[bi[k]], None, None, 0.7); ref.append(so.loss_img); O.optimizer_step(st, so.grads, opt, lrs[k])
for rep in range(2):
    for micro in ("1", "0"):
        os.environ["UMLH_MICRO"] = micro
        e = umlh.HeadEngine(d, d, C, optimizer=OPT, weight_decay=1e-3, max_rows_img=64, max_rows_txt=64, device=DEV)
        e.w_head.copy_(T(w0)); e.scales.fill_(20.0)
        sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
        e.train_steps((T(xi), T(yi, torch.int64)), [T(b, torch.int64) for b in bi], None, None, lrs, first_step=1, alpha=0.7, scalars_out=sc)
        torch.cuda.synchronize()
        err = np.abs(sc.cpu().numpy()[:, 0] - np.asarray(ref))
        print(f"rep {rep} micro={micro} launches {e.micro_launches()}: max err {err.max():.2e} per step", " ".join(f"{x:.1e}" for x in err))
