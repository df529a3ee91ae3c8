#!/usr/bin/env python3
"""Determinism probe of the micro-step kernel: the same head stepped alone twice and inside a grouped launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import numpy as np, torch, umlh
DEV = "cuda:0"
d, C, B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 12, 16, 200
rng = np.random.default_rng(0)
T = lambda a, t=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(DEV, t).contiguous()
x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
y = rng.integers(0, C, n)
X, Y = T(x), T(y, torch.int64)
w0 = (0.1 * rng.standard_normal((C, d))).astype(np.float32)
for steps in (1, 2, 101):
    bi = [T(rng.permutation(n)[:B], torch.int64) for _ in range(steps)]
    bt = [T(rng.permutation(n)[:B], torch.int64) for _ in range(steps)]
    def mk(wd):
        e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=wd, max_rows_img=B, max_rows_txt=B, device=DEV)
        e.w_head.copy_(T(w0)); e.scales.fill_(100.0)
        return e
    outs = []
    for rep in range(2):
        e = mk(0.01)
        e.train_steps((X, Y), bi, (X, Y), bt, [1e-5] * steps, first_step=1)
        torch.cuda.synchronize()
        outs.append(e.w_head.cpu().numpy().copy())
    es = [mk(0.01 if j == 3 else 0.001 * j) for j in range(6)]
    umlh.train_steps_grouped([dict(engine=e, img_table=(X, Y), img_index_batches=bi, txt_table=(X, Y), txt_index_batches=bt,
                                   lrs=[1e-5] * steps, first_step=1) for e in es], steps)
    torch.cuda.synchronize()
    g = es[3].w_head.cpu().numpy()
    print(f"steps {steps}: solo-vs-solo max diff {np.abs(outs[0] - outs[1]).max():.3e}  solo-vs-grouped {np.abs(outs[0] - g).max():.3e}  "
          f"moved {np.abs(outs[0] - w0).max():.3e}  launches {es[3].micro_launches()} status {es[3].micro_status()}")
