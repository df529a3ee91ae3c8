#!/usr/bin/env python3
"""Analysis build only (UMLH_BUILD_ABLATIONS=1): cycle stamps of workgroup (0,0,0) of one gemm_enc launch of the encoder layer
probe.  scripts/gemm_stamps.py <launch index 0..10 within forward(4)+backward(7... in call order)>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402

buf = torch.zeros(16, dtype=torch.int64, device="cuda:0")
os.environ["UMLH_DBG_GEMM_STAMPS"] = hex(buf.data_ptr())
import enc_layer_probe  # noqa: E402
names = ["qkv fwd", "out-proj fwd", "lin1 fwd", "lin2 fwd (split)", "dW2 (TN split)", "dhid", "dW1 (TN split)", "dx1 (split)", "dWo", "datt", "dWin", "dh_in"]
for sel in range(12):
    # calls are counted from process start: iteration `it` makes launches 12*it .. 12*it+11
    it = 3 + sel                       # stamp launch `sel` of iteration `it`
    os.environ["UMLH_DBG_GEMM_CALL"] = str(12 * it + sel)
os.environ["UMLH_DBG_GEMM_CALL"] = sys.argv[1] if len(sys.argv) > 1 else "38"
enc_layer_probe.main(40, 0.1, 8)
torch.cuda.synchronize()
v = buf.cpu().tolist()
n = v[15]
sel = int(os.environ["UMLH_DBG_GEMM_CALL"]) % 12
print(names[sel], "stamps:", n, "deltas (cycles):", [v[i + 1] - v[i] for i in range(n - 1)], "total", v[n - 1] - v[0])
