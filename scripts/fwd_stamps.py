#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of fwd_ce_bf16 from in-kernel stamps (UMLH_DBG_FWD=9)."""
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
if os.environ.get("UMLH_BF16_FWD2D", "") != "0":
    # fwd_ce_bf16_q stamps: 0 start, 1 loads issued, 2 main loop end, 3 wave records merged in LDS (barrier), 4 cross-workgroup
    # merge done (barrier), 5 dZ stored
    t0 = st[:, :, 0].min(dim=1, keepdim=True).values
    rel = st - t0.unsqueeze(2)
    print("fwd_ce_bf16_q: per-wave cycles since the workgroup's first wave started, mean over 256 workgroups")
    print("wave   start  loads_issued  loop_end  records_done  merge_done  end")
    for w in range(8):
        r = rel[:, w, [0, 1, 2, 3, 4, 5]].mean(dim=0)
        print(f"  {w}  " + "  ".join(f"{v:9.0f}" for v in r.tolist()))
    print("workgroup duration: mean %.0f max %.0f cycles" % (rel[:, :, 5].max(dim=1).values.mean(), rel[:, :, 5].max()))
    for qq in range(4):
        print(f"class group {qq}: loop_end {rel[qq::4, :, 2].mean():.0f}  merge_done {rel[qq::4, :, 4].mean():.0f}  end {rel[qq::4, :, 5].mean():.0f}")
    sys.exit(0)
names = ["prologue (ptrs, labels, ring fill)", "main loop", "pass1+2 (+exchange)", "pass3 dZ + stores", "tail"]
t0 = st[:, :, 0].min(dim=1, keepdim=True).values          # workgroup start = earliest wave start
rel = st - t0.unsqueeze(2)
print("per-wave times (cycles since the workgroup's first wave started), mean over 256 workgroups")
print("wave   start  loop_begin  loop_end  argmax_done  exp_done  exchange_done  dz_done   end")
for w in range(8):
    r = rel[:, w, [0, 1, 2, 6, 7, 3, 4, 5]].mean(dim=0)
    print(f"  {w}  " + "  ".join(f"{v:9.0f}" for v in r.tolist()))
print("slowest wave's loop_end per workgroup: mean %.0f  max %.0f" % (rel[:, :, 2].max(dim=1).values.mean(), rel[:, :, 2].max()))
print("workgroup duration: mean %.0f cycles" % (rel[:, :, 5].max(dim=1).values.mean()))
