"""Inputs of the cfg1-shaped full ``finetune.train()`` fixture (tests/golden/train_cfg1.npz)  --  TEST INFRASTRUCTURE ONLY.

BASELINE config 1 / SURVEY 8(d) cfg1 (Caltech101 16-shot, CLIP ViT-B/16): d = 512, C = 100, N_img = 1600 (16 per class),
N_txt = 3000 (30 prompts per class), batch 32, the ``clip_linear`` grid point lr 1e-3 / wd 0.01, evaluation every 100
iterations.  The inputs are regenerated from a numpy seed by BOTH oracle/make_golden_cfg1.py (which runs the
reference on them) and the tests (which replay the run), so the fixture stores outputs only.  Pure numpy; nothing
of the reference is imported here.
"""
import numpy as np

D, C, N_IMG_PER_CLASS, N_TXT_PER_CLASS, N_VAL_PER_CLASS, N_TEST = 512, 100, 16, 30, 4, 2000
BATCH, LR, WD, ALPHA, SCALE_LOG = 32, 1e-3, 0.01, 1.0, 4.60517
MAX_ITERS, EVAL_FREQ, PATIENCE, SEED = 1500, 100, 5, 1
NOISE_IMG, NOISE_TXT, GAP = 5.5, 4.0, 0.7


def _rows(rng, proto, y, noise):
    x = proto[y] + noise * rng.standard_normal((y.shape[0], proto.shape[1]))
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def cfg1_inputs(seed=20260401):
    """dict of float32 feature matrices / int64 labels: x_img,y_img (class-sorted, as a few-shot split is),
    x_txt,y_txt, x_val,y_val, x_test,y_test."""
    rng = np.random.default_rng(seed)
    proto_i = rng.standard_normal((C, D))
    proto_t = proto_i + GAP * rng.standard_normal((C, D))        # the modality gap: text prototypes are shifted
    y_img = np.repeat(np.arange(C), N_IMG_PER_CLASS).astype(np.int64)
    y_txt = np.repeat(np.arange(C), N_TXT_PER_CLASS).astype(np.int64)
    y_val = np.repeat(np.arange(C), N_VAL_PER_CLASS).astype(np.int64)
    y_test = rng.integers(0, C, N_TEST).astype(np.int64)
    return {"x_img": _rows(rng, proto_i, y_img, NOISE_IMG), "y_img": y_img,
            "x_txt": _rows(rng, proto_t, y_txt, NOISE_TXT), "y_txt": y_txt,
            "x_val": _rows(rng, proto_i, y_val, NOISE_IMG), "y_val": y_val,
            "x_test": _rows(rng, proto_i, y_test, NOISE_IMG), "y_test": y_test}
