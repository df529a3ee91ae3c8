#!/usr/bin/env python3
"""Per-phase cycle sums of the micro-step kernel (UMLH_DBG_MICRO=1): where a step of one class slice spends its time."""
import ctypes as C, os, sys
os.environ["UMLH_DBG_MICRO"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import numpy as np, torch, umlh
DEV = "cuda:0"
d, Cc, B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 32, int(sys.argv[4]) if len(sys.argv) > 4 else 4096
prec = sys.argv[5] if len(sys.argv) > 5 else "fp32"
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=DEV), dim=1)
y = torch.randint(0, Cc, (n,), generator=g, device=DEV)
e = umlh.HeadEngine(d, d, Cc, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B, precision=prec, device=DEV)
tab = (x, y, umlh.to_bf16(x)) if prec == "bf16" else (x, y)
e.w_head.normal_(0, 0.05); e.scales.fill_(100.0)
steps = 200
names = ["step head", "fwd mfma", "stats+publish", "poll", "gather+merge", "dZ+scalars", "dW update", "ring: vmcnt wait", "ring: barrier", "dW mfma", "ring: issue", "(into ring)"]
for rep in range(2):
    bi = [torch.randint(0, n, (B,), generator=g, device=DEV) for _ in range(steps)]
    bt = [torch.randint(0, n, (B,), generator=g, device=DEV) for _ in range(steps)]
    e.train_steps(tab, bi, tab, bt, [1e-3] * steps, first_step=1 + rep * steps)
    torch.cuda.synchronize()
p, nb = C.c_void_p(), C.c_uint64()
umlh._lib.check(e.lib.umlh_debug_buffer(e.handle, C.byref(p), C.byref(nb)), "dbg")
off = (p.value - e.workspace.data_ptr()) // 4
nwg = (Cc + 15) // 16
st = e.workspace[off:off + nwg * 24].view(torch.int64).cpu().numpy().reshape(nwg, 12)
tot = st.sum(1)
print(f"d={d} C={Cc} B={B} {prec}: {nwg} slices, {steps} steps; cycles per step of wave 0 (100 MHz? no: shader clock), slice 0 / median slice")
for i, nm in enumerate(names[:12]):
    print(f"  {nm:14s} {st[0, i] / steps:9.0f}   {np.median(st[:, i]) / steps:9.0f}")
print(f"  {'total':14s} {tot[0] / steps:9.0f}   {np.median(tot) / steps:9.0f}")
