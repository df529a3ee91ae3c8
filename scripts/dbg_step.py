"""debug helper: one bf16 grad/train step at a small shape under a given UMLH_BF16_FUSE (argv: fuse d C bi bt mode)"""
import os, sys
fuse, d, C, bi, bt, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
os.environ["UMLH_BF16_FUSE"] = fuse
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unpaired-multimodal-learning_amd"))
import numpy as np, torch, umlh
rng = np.random.default_rng(1)
x = rng.standard_normal((400, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
y = rng.integers(0, C, 400)
w = rng.standard_normal((C, d)).astype(np.float32); w /= np.linalg.norm(w, axis=1, keepdims=True)
e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=512, max_rows_txt=512, precision="bf16", device="cuda:0")
e.w_head.copy_(torch.from_numpy(w)); e.scales.fill_(30.0)
X = torch.from_numpy(x).cuda(); Y = torch.from_numpy(y).cuda()
ii = torch.arange(bi).cuda(); ti = torch.arange(bt).cuda() + 10
for k in range(3):
    if mode == "grad":
        e.grad_step(umlh.RowBatch(X, Y, ii) if bi else None, umlh.RowBatch(X, Y, ti) if bt else None, alpha=0.7)
    else:
        e.train_step(umlh.RowBatch(X, Y, ii) if bi else None, umlh.RowBatch(X, Y, ti) if bt else None, lr=1e-3, step=k + 1)
    torch.cuda.synchronize()
    print("step", k, "ok", e.step_status(), e.step_launches(), flush=True)
print("W sum", float(e.w_head.sum()), flush=True)
