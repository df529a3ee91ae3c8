#!/usr/bin/env python3
"""Where the MultiBench alternation step spends its time: CPU enqueue time vs GPU time, per phase (z = 40 by default)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
from bench_multibench import build, DEV  # noqa: E402


def main(z=40, B=32, T=50, steps=20):
    from engine.optimizer.optim import build_optimizer
    torch.manual_seed(0)
    m = build(z)
    m.train()
    opt = build_optimizer(m.parameters(), "adam", 1e-3, 0.0)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, T, 35, generator=g, device=DEV)
    y = torch.randn(B, T, 300, generator=g, device=DEV)
    lx = torch.randint(5, T + 1, (B,), generator=g, device=DEV)
    ly = torch.randint(5, T + 1, (B,), generator=g, device=DEV)
    ph = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}
    phs = dict(ph)

    def step(sync):
        t0 = time.perf_counter()
        out = m(x, y, lx, ly)
        loss = out["loss_x"] + out["loss_y"]
        if sync: torch.cuda.synchronize()
        t1 = time.perf_counter()
        opt.zero_grad()
        loss.backward()
        if sync: torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step()
        if sync: torch.cuda.synchronize()
        t3 = time.perf_counter()
        d = phs if sync else ph
        d["fwd"] += t1 - t0; d["bwd"] += t2 - t1; d["opt"] += t3 - t2
    for _ in range(5):
        step(True)
    for k in phs: phs[k] = 0.0
    for _ in range(steps):
        step(True)
    print("synchronised phases (ms/step):", {k: round(v / steps * 1e3, 3) for k, v in phs.items()}, "sum", round(sum(phs.values()) / steps * 1e3, 3))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    t_cpu = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("enqueue-only phases (ms/step):", {k: round(v / steps * 1e3, 3) for k, v in ph.items()}, "cpu", round(t_cpu / steps * 1e3, 3), "wall", round(t_all / steps * 1e3, 3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step(False)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
