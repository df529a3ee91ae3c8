#!/usr/bin/env python3
"""MultiBench alternation step (BASELINE configs[3] shape: CMU-MOSEI vision 35-d + text 300-d, T = 50, batch 32,
5-layer shared transformer) on the HIP path, train mode (dropout on), HIP Adam; and -- for information only --
the same model's encoder evaluated through torch.nn (rocBLAS / MIOpen ops) as `tests/test_encoder_gpu.py` does."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

DEV = "cuda:0"


def build(z, dx=35, dy=300):
    from multibench.models import UML, Linear, Transformer
    enc = Transformer(z, z, nhead=5, num_layers=5, conv1d=True, out_last=False, pos_embd=True, pos_learnable=False, max_len=128)   # main.py:119
    return UML(Linear(dx, z), Linear(dy, z), enc, [Linear(z, dx), Linear(z, dy)]).to(DEV)


def run(z, steps=30, B=32, T=50, torch_ref=False):
    from engine.optimizer.optim import build_optimizer
    torch.manual_seed(0)
    m = build(z)
    m.train()
    if torch_ref:
        from test_encoder_gpu import _torch_reference
        enc = m.encoder
        enc.forward = lambda x, lengths=None: _torch_reference(enc, x, lengths)
        enc.forward_pair = lambda x, lx, y, ly: (enc.forward(x, lengths=lx), enc.forward(y, lengths=ly))
        for lin in (m.xproj_in, m.yproj_in):
            lin.forward = lin.fc.forward
    opt = build_optimizer(m.parameters(), "adam", 1e-3, 0.0)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, T, 35, generator=g, device=DEV)
    y = torch.randn(B, T, 300, generator=g, device=DEV)
    lx = torch.randint(5, T + 1, (B,), generator=g, device=DEV)
    ly = torch.randint(5, T + 1, (B,), generator=g, device=DEV)

    def step():
        out = m(x, y, lx, ly)
        loss = out["loss_x"] + out["loss_y"]
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 1.0:                  # >= 1 s of warm-up: the first steps run at idle clocks
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"z": z, "path": "torch.nn ops (information)" if torch_ref else "HIP encoder", "ms_per_step": round(dt * 1e3, 3),
                      "sequences_per_s": round(2 * B / dt, 1), "final_loss": round(float(loss), 4)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:                                   # single HIP-path config, e.g. under rocprofv3
        run(int(sys.argv[1]), steps=int(sys.argv[2]) if len(sys.argv) > 2 else 50)
        sys.exit(0)
    for z in (40, 300):
        run(z)
        run(z, torch_ref=True)
