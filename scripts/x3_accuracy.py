#!/usr/bin/env python3
"""Accuracy of the fp32 mode's two product forms against float64: loss sums and the head gradient of one cfg2-size step
(d = 512, C = 1000, 4096 + 4096 rows, logit scale 100) through umlh_grad_step.  The switch UMLH_F32_X3 is read once per process:
    UMLH_F32_X3=1 python scripts/x3_accuracy.py     (default: three-way bf16 split on v_mfma_f32_32x32x16_bf16)
    UMLH_F32_X3=0 python scripts/x3_accuracy.py     (v_mfma_f32_32x32x2_f32)
The float64 reference is formed here in numpy (no oracle import: this is a measurement script, not a test)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch  # noqa: E402
import umlh  # noqa: E402

d, C, B, scale = 512, 1000, 4096, 100.0
rng = np.random.default_rng(5)
x = rng.standard_normal((2 * B, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
w = rng.standard_normal((C, d)).astype(np.float32); w /= np.linalg.norm(w, axis=1, keepdims=True)
# labels correlated with the logits so that the loss is not dominated by noise
y = np.argmax(x.astype(np.float64) @ w.T.astype(np.float64) + rng.gumbel(size=(2 * B, C)) * 0.02, axis=1)
e = umlh.HeadEngine(d, d, C, optimizer="sgd", max_rows_img=B, max_rows_txt=B, precision="fp32", device="cuda:0")
e.w_head.copy_(torch.from_numpy(w)); e.scales.fill_(scale)
X = torch.from_numpy(x).cuda(); Y = torch.from_numpy(y).cuda()
flat = e.grad_step(umlh.RowBatch(X[:B], Y[:B]), umlh.RowBatch(X[B:], Y[B:]), alpha=1.0).cpu().numpy().astype(np.float64)
torch.cuda.synchronize()
g = flat[:C * d].reshape(C, d)
sc = flat[-umlh.N_SCALARS:]
# float64 reference
z = scale * (x.astype(np.float64) @ w.T.astype(np.float64))
z -= z.max(axis=1, keepdims=True)
lse = np.log(np.exp(z).sum(axis=1))
p = np.exp(z - lse[:, None])
ce = -(z[np.arange(2 * B), y] - lse)
oh = np.zeros_like(p); oh[np.arange(2 * B), y] = 1
gref = scale * ((p[:B] - oh[:B]).T @ x[:B].astype(np.float64) / B + (p[B:] - oh[B:]).T @ x[B:].astype(np.float64) / B)
mode = "x3 (bf16 MFMA, 6 piece products)" if os.environ.get("UMLH_F32_X3", "1") != "0" else "fp32 MFMA"
print(f"{mode}: loss img {sc[umlh.S_LOSS_IMG]:.7f} (ref {ce[:B].mean():.7f}, err {abs(sc[umlh.S_LOSS_IMG] - ce[:B].mean()):.2e})  "
      f"txt {sc[umlh.S_LOSS_TXT]:.7f} (ref {ce[B:].mean():.7f}, err {abs(sc[umlh.S_LOSS_TXT] - ce[B:].mean()):.2e})")
err = np.abs(g - gref)
print(f"{mode}: head gradient max |err| {err.max():.3e}  rms err {np.sqrt((err ** 2).mean()):.3e}  (rms |g| {np.sqrt((gref ** 2).mean()):.3e}, max |g| {np.abs(gref).max():.3e})")
