#!/usr/bin/env python3
"""Informational timings of the other BASELINE configs (parity-test cases, not bench lines):
cfg1/cfg5-like small-batch linear heads and the cfg3 two-layer head.  Prints one JSON line each."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch  # noqa: E402
import umlh  # noqa: E402

DEV = "cuda:0"


def run(tag, d_img, d_sh, C, B, proj, precision, n_img=65536, n_txt=8192, steps=200, learn=False):
    g = torch.Generator(device=DEV).manual_seed(0)
    xi = torch.nn.functional.normalize(torch.randn(n_img, d_img, generator=g, device=DEV), dim=1)
    xt = torch.nn.functional.normalize(torch.randn(n_txt, d_sh, generator=g, device=DEV), dim=1)
    yi = torch.randint(0, C, (n_img,), generator=g, device=DEV)
    yt = torch.randint(0, C, (n_txt,), generator=g, device=DEV)
    e = umlh.HeadEngine(d_img, d_sh, C, has_proj=proj, learnable_temp=learn, optimizer="adamw", weight_decay=0.01,
                        max_rows_img=B, max_rows_txt=B, precision=precision, device=DEV)
    e.w_head.normal_(0, 0.02)
    if proj:
        e.w_proj.normal_(0, 0.02)
    ti = (xi, yi, umlh.to_bf16(xi)) if precision == "bf16" else (xi, yi)
    tt = (xt, yt, umlh.to_bf16(xt)) if precision == "bf16" else (xt, yt)

    def draw(n):                                      # index vectors are drawn outside the timed region
        return ([torch.randint(0, n_img, (B,), generator=g, device=DEV) for _ in range(n)],
                [torch.randint(0, n_txt, (B,), generator=g, device=DEV) for _ in range(n)])
    bi, bt = draw(20)
    e.train_steps(ti, bi, tt, bt, [1e-3] * 20, first_step=1)
    bi, bt = draw(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.train_steps(ti, bi, tt, bt, [1e-3] * steps, first_step=21)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": tag, "precision": precision, "d_img": d_img, "d_shared": d_sh, "C": C, "rows_per_step": 2 * B,
                      "us_per_step": round(dt / steps * 1e6, 1), "samples_per_s": round(2 * B * steps / dt, 1)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "half-classes":   # what-if probe for a class-split forward
        run("probe: C=500 linear head bf16, batch 4096+4096", 512, 512, 500, 4096, False, "bf16", steps=60)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg3-bf16":      # single config, e.g. under rocprofv3
        run("cfg3 DINOv2-L + OpenLLaMA two-layer head, batch 4096+4096", 1024, 3200, 1000, 4096, True, "bf16", steps=40)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg3-fp32":
        run("cfg3 DINOv2-L + OpenLLaMA two-layer head, batch 4096+4096", 1024, 3200, 1000, 4096, True, "fp32", steps=20)
        sys.exit(0)
    run("cfg1 Caltech101-like linear head, batch 32+32", 512, 512, 100, 32, False, "fp32")
    run("cfg1 Caltech101-like linear head, batch 32+32", 512, 512, 100, 32, False, "bf16")
    run("cfg5 CLIP-L/14-like d=768 C=397, batch 32+32", 768, 768, 397, 32, False, "fp32")
    run("cfg5 CLIP-L/14-like d=768 C=397, batch 32+32", 768, 768, 397, 32, False, "bf16")
    run("cfg3 DINOv2-L + OpenLLaMA two-layer head, batch 4096+4096", 1024, 3200, 1000, 4096, True, "fp32", steps=20)
    run("cfg3 DINOv2-L + OpenLLaMA two-layer head, batch 4096+4096", 1024, 3200, 1000, 4096, True, "bf16", steps=40)
    run("cfg2 linear head fp32, batch 4096+4096", 512, 512, 1000, 4096, False, "fp32", steps=50)
