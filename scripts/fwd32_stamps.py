#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of fwd_ce_f32 from in-kernel stamps (UMLH_DBG_FWD=9), cfg2 shape."""
import ctypes as C
import os
import sys

os.environ.setdefault("UMLH_DBG_FWD", "9")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch
import umlh

DEV = "cuda:0"
d, Cn, B = 512, 1000, 4096
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(3 * B, d, generator=g, device=DEV), dim=1)
y = torch.randint(0, Cn, (3 * B,), generator=g, device=DEV)
e = umlh.HeadEngine(d, d, Cn, optimizer="adamw", max_rows_img=B, max_rows_txt=B, precision="fp32", device=DEV)
e.w_head.normal_(0, 0.05)
e.scales.fill_(100.0)
for it in range(5):
    ii = torch.randint(0, 3 * B, (B,), generator=g, device=DEV)
    ti = torch.randint(0, 3 * B, (B,), generator=g, device=DEV)
    e.grad_step(umlh.RowBatch(x, y, ii), umlh.RowBatch(x, y, ti))
torch.cuda.synchronize()
p, n = C.c_void_p(), C.c_uint64()
umlh._lib.check(e.lib.umlh_debug_buffer(e.handle, C.byref(p), C.byref(n)), "dbg")
off = (p.value - e.workspace.data_ptr()) // 4
st = e.workspace[off:off + 256 * 128].view(torch.int64).reshape(256, 8, 8).cpu().double()
t0 = st[:, :, 0].min(dim=1, keepdim=True).values
rel = st - t0.unsqueeze(2)
print("fwd_ce_f32, cfg2: per-wave cycles since the workgroup's first wave started, mean over 256 workgroups")
print("wave   start  loop_begin   loop_end   max+argmax   exp+sums   dz_stored      end")
for w in range(8):
    r = rel[:, w, [0, 1, 2, 3, 4, 5, 6]].mean(dim=0)
    print(f"  {w}  " + "  ".join(f"{v:10.0f}" for v in r.tolist()))
print("workgroup duration: mean %.0f max %.0f cycles; main loop mean %.0f" % (
    rel[:, :, 6].max(dim=1).values.mean(), rel[:, :, 6].max(), (rel[:, :, 2] - rel[:, :, 1]).mean()))
