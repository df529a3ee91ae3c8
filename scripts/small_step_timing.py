#!/usr/bin/env python3
"""Wall time of cfg1-shaped (batch 32+32, C=100, d=512) fused steps, enqueued 100 at a time.
small_step_timing.py [batch] [fp32|bf16] [reps] [d] [C]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch
import umlh
DEV = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = int(sys.argv[4]) if len(sys.argv) > 4 else 512          # feature width (cfg1: 512, cfg5 ViT-L/14: 768)
C = int(sys.argv[5]) if len(sys.argv) > 5 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(4096, d, generator=g, device=DEV), dim=1)
y = torch.randint(0, C, (4096,), generator=g, device=DEV)
x16 = umlh.to_bf16(x)
e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B, precision=prec, device=DEV)
e.w_head.normal_(0, 0.05); e.scales.fill_(100.0)
n = 100
print(f"# cfg1 shape d={d} C={C} batch {B}+{B}, precision {prec}, UMLH_MICRO={os.environ.get('UMLH_MICRO', '1')} (micro launches so far: see the last line)")
tab = (x, y, x16) if prec == "bf16" else (x, y)
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    bi = [torch.randint(0, 4096, (B,), generator=g, device=DEV) for _ in range(n)]
    bt = [torch.randint(0, 4096, (B,), generator=g, device=DEV) for _ in range(n)]
    sc = torch.zeros(n, umlh.N_SCALARS, device=DEV)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.train_steps(tab, bi, tab, bt, [1e-3] * n, first_step=1 + rep * n, scalars_out=sc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    w = e.w_head.detach().cpu().clone()
    t3 = time.perf_counter()
    print(f"rep {rep}: enqueue {1e6 * (t1 - t0) / n:7.1f} us/step   drain {1e6 * (t2 - t1) / n:7.1f} us/step   total {1e6 * (t2 - t0) / n:7.1f} us/step   state copy {1e3 * (t3 - t2):.2f} ms")
print(f"# micro launches: {e.micro_launches()}")
