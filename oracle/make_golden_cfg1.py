#!/usr/bin/env python3
"""tests/golden/train_cfg1.npz: a complete cfg1-shaped ``finetune.train()`` run OF THE REFERENCE.

Build container only (imports /root/reference through oracle/make_golden.py's stubs).  The model is the
reference's ``UMLClip`` (fixed logit scale 100, ``logit_scale = 4.60517``) built without its network-fetching
``__init__`` and given the ``extract_features`` that ``train()`` calls but the class lacks; zero-shot
init through ``get_zero_shot_weights(..., device="cpu")``; ``clip_linear`` grid point lr 1e-3 /
wd 0.01, batch 32, evaluation every 100 iterations, patience 5, at most 1500 iterations
(vision_language/finetune.py:120-288).  Inputs come from oracle/fixtures_cfg1.py (numpy seed): only OUTPUTS are
stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG            # noqa: E402
import fixtures_cfg1 as FX          # noqa: E402


def main():
    from torch.utils.data import DataLoader
    torch.set_num_threads(4)
    inp = {k: torch.from_numpy(v) for k, v in FX.cfg1_inputs().items()}
    eot = torch.zeros(inp["y_txt"].shape[0], dtype=torch.long)
    MG.set_random_seed(FX.SEED)
    img_log, txt_log = [], []
    text_ds = MG.quiet(MG.RecordingTextDS, inp["x_txt"], inp["y_txt"], eot, n_shots=None, log=txt_log)
    model = MG.make_umlclip(FX.D, FX.C, FX.SCALE_LOG)
    model.extract_features = lambda images: images
    w_init = model.head.weight.detach().clone()
    model.head.weight.data = MG.quiet(MG.get_zero_shot_weights, MG.RefTextDS(inp["x_txt"], inp["y_txt"], eot), FX.C, FX.D,
                                      device="cpu")
    optimizer = MG.ref_build_optimizer(model.parameters(), "adamw", FX.LR, FX.WD)
    scheduler = MG.ref_build_sched(optimizer, "cosine", 50, 12800, warmup_type="linear", warmup_lr=1e-5)   # HYPER_DICT['clip_linear']
    B = FX.BATCH
    il = DataLoader(MG.RecordingImageDS(inp["x_img"], inp["y_img"], img_log), batch_size=B, shuffle=True, num_workers=0, drop_last=False)
    tl = DataLoader(text_ds, batch_size=B, shuffle=True, num_workers=0, drop_last=False)
    vl = DataLoader(MG.RecordingImageDS(inp["x_val"], inp["y_val"]), batch_size=B, shuffle=False)
    te = DataLoader(MG.RecordingImageDS(inp["x_test"], inp["y_test"]), batch_size=B, shuffle=False)
    F = torch.nn.functional
    orig_ce, train_ce = F.cross_entropy, []

    def rec_ce(a, b, *x, **k):
        v = orig_ce(a, b, *x, **k)
        if torch.is_grad_enabled():
            train_ce.append(float(v))
        return v
    F.cross_entropy = rec_ce
    orig_validate, val_calls = MG.ref_finetune.validate, []

    def rec_validate(m, loader, device="cpu"):
        r = orig_validate(m, loader, device=device)
        val_calls.append((len(loader.dataset), r[0], r[1]))
        return r
    MG.ref_finetune.validate = rec_validate
    try:
        out = MG.quiet(MG.ref_finetune.train, model, il, tl, vl, te, optimizer, scheduler, device="cpu",
                       max_iters=FX.MAX_ITERS, alpha=FX.ALPHA, eval_freq=FX.EVAL_FREQ, patience=FX.PATIENCE,
                       capture_features_during_training=False, args=None, logger=None)
    finally:
        F.cross_entropy = orig_ce
        MG.ref_finetune.validate = orig_validate
    test_loss, test_acc = MG.ref_finetune.validate(model, te, device="cpu")
    n_steps = len(train_ce) // 2
    n_val = inp["y_val"].shape[0]
    vals = [(l, a) for (n, l, a) in val_calls if n == n_val]
    tests = [(l, a) for (n, l, a) in val_calls if n == FX.N_TEST]
    il_ = np.asarray(img_log, dtype=np.int64)
    tl_ = np.asarray(txt_log, dtype=np.int64)
    MG.npz("train_cfg1", w_head_init=w_init, n_steps=n_steps, train_ce=np.asarray(train_ce, dtype=np.float32),
           idx_img_head=il_[:10 * B].astype(np.int16), idx_txt_head=tl_[:10 * B].astype(np.int16),
           idx_img_checksum=np.asarray([int((il_ * (np.arange(il_.size) % 977 + 1)).sum())]),
           idx_txt_checksum=np.asarray([int((tl_ * (np.arange(tl_.size) % 977 + 1)).sum())]),
           n_idx=np.asarray([il_.size, tl_.size]),
           val_loss=np.asarray([v[0] for v in vals]), val_acc=np.asarray([v[1] for v in vals]),
           test_acc_trace=np.asarray([v[1] for v in tests]),
           best_iter=out["iter"], best_val_acc=out["val_acc"], best_val_loss=out["val_loss"],
           w_head_best=out["model"]["head.weight"].to(torch.float32), test_loss=test_loss, test_acc=test_acc)
    print(f"   cfg1: steps={n_steps} best_iter={out['iter']} best_val_acc={out['val_acc']:.4f} test_acc={test_acc:.4f}"
          f" val_acc trace={[round(v[1], 4) for v in vals]}")


if __name__ == "__main__":
    main()
