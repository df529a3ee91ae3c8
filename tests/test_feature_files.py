"""Feature-file reader + path scheme (CPU) and an end-to-end sweep from synthetic feature files (GPU)."""
import os
import types

import numpy as np
import pytest
import torch


def _write_files(root, d=64, C=6, seed=0):
    import features as F_
    g = torch.Generator().manual_seed(seed)
    proto = torch.randn(C, d, generator=g)

    def draw(n):
        y = torch.randint(0, C, (n,), generator=g)
        x = torch.nn.functional.normalize(proto[y] + 0.8 * torch.randn(n, d, generator=g), dim=1)
        return x, y
    xtr, ytr = draw(96)
    xva, yva = draw(48)
    xte, yte = draw(60)
    xt, yt = draw(30)
    lab2cname = {i: f"class_{i}" for i in range(C)}
    p_tr = F_.img_outdir(root, "ViT-B/16", "toyset", "crop", 16, 1, "train")
    p_te = F_.img_outdir(root, "ViT-B/16", "toyset", "crop", 16, 1, "test")
    p_tx = F_.text_outdir(root, "ViT-B/16", "toyset", "gpt3_cupl")
    for p in (p_tr, p_te, p_tx):
        os.makedirs(os.path.dirname(p), exist_ok=True)
    torch.save({"train": {"features": xtr, "labels": ytr, "paths": ["a"] * 96},
                "val": {"features": xva, "labels": yva, "paths": ["b"] * 48}, "lab2cname": lab2cname}, p_tr)
    torch.save({"features": xte, "labels": yte, "paths": ["c"] * 60, "lab2cname": lab2cname}, p_te)
    torch.save({"features": xt, "labels": yt, "eot_indices": torch.zeros(30, dtype=torch.long),
                "prompts": {i: ["p"] for i in range(C)}, "lab2cname": lab2cname}, p_tx)
    return p_tr, p_te, p_tx


def test_path_scheme_and_reader_roundtrip(tmp_path):
    import features as F_
    assert F_.img_outdir("/f", "ViT-B/16", "dtd", "crop", 16, 1) == "/f/image/ViT-B-16/dtd/crop/shot_16-seed_1.pth"
    assert F_.img_outdir("/f", "ViT-B/16", "dtd", "crop", 16, 1, mode="test") == "/f/image/ViT-B-16/dtd/test.pth"
    assert F_.text_outdir("/f", "RN50", "dtd", "gpt3_cupl") == "/f/text/RN50/dtd/gpt3_cupl.pth"
    assert F_.img_outdir("/f", "RN50", "dtd", "flip", -1, 2, return_tokens=True).startswith("/f/patch-token/image/RN50/dtd/flip/")
    p_tr, p_te, p_tx = _write_files(str(tmp_path))
    tr, te, tx = F_.load_image_train_features(p_tr), F_.load_image_test_features(p_te), F_.load_text_features(p_tx)
    assert tr["train"][0].shape == (96, 64) and tr["val"][1].dtype == torch.int64 and te["test"][0].shape == (60, 64)
    assert tx["features"].shape == (30, 64) and tx["eot_indices"].shape == (30,) and len(tr["lab2cname"]) == 6
    torch.save({"features": torch.zeros(3, 4, 5), "labels": torch.zeros(3)}, tmp_path / "bad.pth")
    with pytest.raises(ValueError):
        F_.load_image_test_features(str(tmp_path / "bad.pth"))


@pytest.mark.gpu
def test_main_sweeps_feature_files_end_to_end(tmp_path):
    import finetune as ft
    root = str(tmp_path / "features")
    _write_files(root)
    args = types.SimpleNamespace(seed=1, dataset="toyset", train_shot=16, clip_encoder="ViT-B/16", vision_model="",
                                 language_model="", feature_dir=root, result_dir=str(tmp_path / "results"),
                                 text_type="gpt3_cupl", text_shot=None, image_augmentation="crop", modality="crossmodal",
                                 classifier_init="zeroshot", alpha=1.0, logit=4.60517, custom_name="", device="cuda:0",
                                 hyperparams={"optim": "adamw", "lr": [1e-3, 1e-4], "weight_decay": [0.0], "lr_scheduler": "cosine",
                                              "batch_size": [32], "max_iter": [120], "warmup_iter": 50, "warmup_type": "linear",
                                              "warmup_min_lr": 1e-5, "dropout": [0.0], "learnable_temp": [False], "patience": [5]})
    results, best_val, best_test = ft.main(args)
    assert len(results["hparams"]) == 2 and best_val > 0.8 and best_test > 0.8
    p = os.path.join(args.savepath, "optim_adamw-lr_0.001-wd_0.0-bs_32-iters_120-dropout_0.0", "test_result.pth")
    assert os.path.exists(p) and os.path.exists(os.path.join(args.savepath, "results.pth"))
    saved = torch.load(p, weights_only=True)
    assert set(saved) == {"test_acc", "val_acc", "model", "iter"} and saved["model"]["head.weight"].shape == (6, 64)
    # second call: every point is skipped because its result file exists (reference :330-333)
    results2, _, _ = ft.main(args)
    assert results2["test_acc"] == results["test_acc"]
