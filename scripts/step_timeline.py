#!/usr/bin/env python3
"""Diagnostic: phase timeline of the one-launch step (step_bf16) from per-block s_memrealtime stamps (UMLH_DBG_STEP=1):
when the forward / dW / update blocks start, pass their gate and end, relative to the launch's first block (100 MHz clock)."""
import ctypes as C
import os
import sys

os.environ["UMLH_DBG_STEP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch
import umlh

DEV = "cuda:0"
d, Cn, B = 512, 1000, 4096
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(3 * B, d, generator=g, device=DEV), dim=1)
y = torch.randint(0, Cn, (3 * B,), generator=g, device=DEV)
e = umlh.HeadEngine(d, d, Cn, optimizer="adamw", max_rows_img=B, max_rows_txt=B, precision="bf16", device=DEV)
e.w_head.normal_(0, 0.05)
e.scales.fill_(100.0)
x16 = umlh.to_bf16(x)
for it in range(30):
    ii = torch.randint(0, 3 * B, (B,), generator=g, device=DEV)
    ti = torch.randint(0, 3 * B, (B,), generator=g, device=DEV)
    e.train_step(umlh.RowBatch(x, y, ii, feats_bf16=x16), umlh.RowBatch(x, y, ti, feats_bf16=x16), lr=1e-3, step=it + 1)
torch.cuda.synchronize()
p, n = C.c_void_p(), C.c_uint64()
umlh._lib.check(e.lib.umlh_debug_buffer(e.handle, C.byref(p), C.byref(n)), "dbg")
off = (p.value - e.workspace.data_ptr()) // 4
nfwd, ndw = 2 * B // 32, 4 * 8 * 8
nupd = (Cn * d // 4 + 255) // 256
nhead = (nupd + 1) // 2
nb = nfwd + ndw + nhead + 1
st = e.workspace[off:off + nb * 8].view(torch.int64).reshape(nb, 4).cpu().double()
t0 = st[:nfwd, 0].min()
us = (st - t0) / 100.0                                   # 100 MHz -> microseconds
roles = [("forward blocks", 0, nfwd), ("dW blocks", nfwd, nfwd + ndw), ("update blocks", nfwd + ndw, nfwd + ndw + nhead),
         ("finalize block", nb - 1, nb)]
print("one-launch step (step_bf16), cfg2 shape; microseconds since the first forward block started; last of 30 steps")
print(f"{'role':16s} {'blocks':>6s}  {'start min':>9s} {'start max':>9s}  {'gate open mean':>14s} {'gate open max':>13s}  {'end min':>8s} {'end mean':>8s} {'end max':>8s}")
for name, a, b in roles:
    s_, g_, e_ = us[a:b, 0], us[a:b, 1], us[a:b, 2]
    gm = "-" if name.startswith("forward") else f"{g_.mean():.2f}"
    gx = "-" if name.startswith("forward") else f"{g_.max():.2f}"
    print(f"{name:16s} {b - a:6d}  {s_.min():9.2f} {s_.max():9.2f}  {gm:>14s} {gx:>13s}  {e_.min():8.2f} {e_.mean():8.2f} {e_.max():8.2f}")
print("gate wait of a dW block (gate open - start): mean %.2f us; of an update block: mean %.2f us" % (
    (us[nfwd:nfwd + ndw, 1] - us[nfwd:nfwd + ndw, 0]).mean(), (us[nfwd + ndw:nb - 1, 1] - us[nfwd + ndw:nb - 1, 0]).mean()))
