#!/usr/bin/env python3
"""Calibration of bench.py's CPU baseline ("port", oracle/torch_cpu_loop.py) against the REAL reference loop.

TEST/BENCH INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference, which never travels to the
GPU box).  Times, on the same host cores and the same synthetic cfg2-shaped tensors (N_img = 32768, N_txt = 29940,
d = 512, C = 1000, 4096 + 4096 rows per step, AdamW):
  * the reference's own ``finetune.train()`` (imported with oracle/make_golden.py's stubs; identity backbone; a bare
    ``UMLClip`` given the ``extract_features`` its class lacks) -- step times taken between successive
    ``optimizer.step()`` calls, so the validation pass at iteration 0 is excluded;
  * ``torch_cpu_loop.reference_shaped_steps`` (what bench.py times on the GPU box) and ``bare_math_steps``.
Prints one JSON line; the figures are recorded in DESIGN.md section 7 and in ``torch_cpu_loop.CALIBRATION``.
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG            # noqa: E402  (imports the reference with stubs)
import torch_cpu_loop as port       # noqa: E402


def main(threads=8, steps=14):
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    n_img, n_txt, d, C, B = 32768, 29940, 512, 1000, 4096
    xi = torch.nn.functional.normalize(torch.randn(n_img, d, generator=g), dim=1)
    yi = torch.randint(0, C, (n_img,), generator=g)
    xt = torch.nn.functional.normalize(torch.randn(n_txt, d, generator=g), dim=1)
    yt = torch.randint(0, C, (n_txt,), generator=g)
    from torch.utils.data import DataLoader
    model = MG.make_umlclip(d, C, 4.60517)
    model.extract_features = lambda images: images          # the method train() calls and UMLClip lacks
    opt = MG.ref_build_optimizer(model.parameters(), "adamw", 1e-3, 0.01)
    sched = MG.ref_build_sched(opt, "cosine", 50, 12800, warmup_type="linear", warmup_lr=1e-5)
    il = DataLoader(MG.RecordingImageDS(xi, yi), batch_size=B, shuffle=True, num_workers=0, drop_last=False)
    tl = DataLoader(MG.RefTextDS(xt, yt, torch.zeros(n_txt, dtype=torch.long), n_shots=None), batch_size=B, shuffle=True,
                    num_workers=0, drop_last=False)
    vl = DataLoader(MG.RecordingImageDS(xi[:64], yi[:64]), batch_size=B, shuffle=False)
    stamps = []
    orig = opt.step

    def timed_step(*a, **k):
        r = orig(*a, **k)
        stamps.append(time.perf_counter())
        return r
    opt.step = timed_step
    MG.quiet(MG.ref_finetune.train, model, il, tl, vl, None, opt, sched, device="cpu", max_iters=steps, alpha=1.0,
             eval_freq=10 ** 6, patience=5, capture_features_during_training=False, args=None, logger=None)
    ref_sps = (len(stamps) - 2) * 2 * B / (stamps[-1] - stamps[1])     # first interval dropped (warm-up)
    v_port, _, _ = port.reference_shaped_steps(xi, yi, xt, yt, C, B, steps=steps - 2, warmup=2, threads=threads)
    v_bare, _, _ = port.bare_math_steps(xi, yi, xt, yt, C, B, steps=steps - 2, warmup=2, threads=threads)
    print(json.dumps({"threads": threads, "timed_steps": steps - 2, "reference_train_samples_per_s": round(ref_sps, 1),
                      "port_samples_per_s": round(v_port, 1), "bare_math_samples_per_s": round(v_bare, 1),
                      "port_over_reference": round(v_port / ref_sps, 3)}))


if __name__ == "__main__":
    main()
