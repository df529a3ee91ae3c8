#!/usr/bin/env python3
"""Diagnostic: per-phase cycle stamps of dw_bf16 (UMLH_DBG_DW=16+ablation bits; argv[1] = bits:
1 = no dZ^T traffic, 2 = no feature traffic; the bits need a library built with UMLH_BUILD_ABLATIONS=1)."""
import ctypes as C
import os
import sys

os.environ["UMLH_DBG_DW"] = str(16 + int(sys.argv[1]) if len(sys.argv) > 1 else 16)
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
for it in range(5):
    ii = torch.randint(0, 3 * B, (B,), generator=g, device=DEV)
    ti = torch.randint(0, 3 * B, (B,), generator=g, device=DEV)
    e.grad_step(umlh.RowBatch(x, y, ii, feats_bf16=x16), umlh.RowBatch(x, y, ti, feats_bf16=x16))
torch.cuda.synchronize()
p, n = C.c_void_p(), C.c_uint64()
umlh._lib.check(e.lib.umlh_debug_buffer(e.handle, C.byref(p), C.byref(n)), "dbg")
off = (p.value - e.workspace.data_ptr()) // 4
st = e.workspace[off:off + 256 * 128].view(torch.int64).reshape(256, 8, 8).cpu().double()
t0 = st[:, :, 0].min(dim=1, keepdim=True).values
rel = st - t0.unsqueeze(2)
print("dw_bf16 per-wave times (cycles since the workgroup's first wave started), mean over 256 workgroups")
print("wave   start   ids_done  loop_begin  +4 chunks  +8 chunks  +12 chunks  loop_end      end")
for w in range(8):
    r = rel[:, w, :].mean(dim=0)
    print(f"  {w}  " + "  ".join(f"{v:9.0f}" for v in r.tolist()))
print("workgroup duration: mean %.0f  min %.0f  max %.0f cycles" % (rel[:, :, 7].max(dim=1).values.mean(), rel[:, :, 7].max(dim=1).values.min(), rel[:, :, 7].max(dim=1).values.max()))
span = st[:, :, 7].max() - st[:, :, 0].min()
print("first start -> last end over the grid: %.0f cycles" % span)
