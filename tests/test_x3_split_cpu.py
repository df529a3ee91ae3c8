"""Numerics of the fp32 mode's product form (csrc/umlh_common.h: split3_pair, umlh_f32_x3), restated in numpy: an fp32 value is
split into three bf16 pieces -- hi and mid by truncation of the running residual, lo by round-to-nearest-even -- and a product
is the six piece products of order <= 2.  The kernels form exactly these pieces on the GPU (tests/test_hip_parity.py compares
their results with the fp32 MFMA's and with the reference's goldens); this file pins the arithmetic claim itself on the CPU."""
import numpy as np


def _trunc_bf16(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def _rne_bf16(x):
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = _trunc_bf16(x)
    r1 = (x - hi).astype(np.float32)                 # exact: hi shares x's leading bits
    mid = _trunc_bf16(r1)
    r2 = (r1 - mid).astype(np.float32)               # exact
    lo = _rne_bf16(r2)
    return hi, mid, lo


def test_three_bf16_pieces_reconstruct_an_fp32_value_to_its_last_bit():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 6, 200000).astype(np.float32),
                        np.asarray([0.0, 1.0, -1.0, 1.0e-30, 1.0e38, np.float32(1) + np.float32(2) ** -23], dtype=np.float32)])   # (normal range)
    hi, mid, lo = split3(x)
    for p in (hi, mid, lo):                          # every piece is a bf16 number: its low 16 bits are zero
        assert not (p.view(np.uint32) & np.uint32(0xFFFF)).any()
    rec = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    err = np.abs(rec - x.astype(np.float64))
    # 8 + 8 + 8 significant bits cover fp32's 24: what is left after hi and mid has at most 8 significant bits (+ rounding of lo)
    assert (err <= np.abs(x.astype(np.float64)) * 2.0 ** -24 + 1e-45).all()
    assert (err == 0).mean() > 0.99


def test_six_piece_products_track_the_exact_product_at_the_fp32_rounding_level():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(100000).astype(np.float32)
    w = rng.standard_normal(100000).astype(np.float32)
    xs, ws = split3(x), split3(w)
    acc = np.zeros(x.shape, dtype=np.float64)
    for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)):        # the order the kernels issue them in (small terms first)
        acc += xs[i].astype(np.float64) * ws[j].astype(np.float64)       # each piece product has <= 16 significant bits: exact in fp32
    exact = x.astype(np.float64) * w.astype(np.float64)
    rel = np.abs(acc - exact) / np.maximum(np.abs(exact), 1e-300)
    # dropped: mid*lo, lo*mid, lo*lo.  A truncated bf16 piece leaves a residual below 2^-7 of the value, so |mid| < 2^-7 |x| and
    # |lo| < 2^-14 |x|: the bound is 2 * 2^-21 of a product; over random operands the maximum is 2^-21.2 and the rms 2^-24.0 -- a
    # single fp32 rounding has maximum 2^-24 and rms 2^-25.2, and a K-long fp32 fma chain makes K roundings of its running sum
    assert rel.max() < 2.0 ** -20
    assert np.sqrt((rel ** 2).mean()) < 2.0 ** -23.5
    # a K = 512 dot product at the cfg2 scale: the x3 sum is as close to float64 as the plain fp32 sum is
    X = rng.standard_normal((64, 512)).astype(np.float32); X /= np.linalg.norm(X, axis=1, keepdims=True)
    W = rng.standard_normal((32, 512)).astype(np.float32); W /= np.linalg.norm(W, axis=1, keepdims=True)
    Xs, Ws = split3(X), split3(W)
    dot = sum((Xs[i].astype(np.float64) @ Ws[j].astype(np.float64).T) for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)))
    ref = X.astype(np.float64) @ W.astype(np.float64).T
    fp32 = (X @ W.T).astype(np.float64)
    assert np.abs(dot - ref).max() <= max(np.abs(fp32 - ref).max(), 2e-8) * 1.5
    assert 100.0 * np.abs(dot - ref).max() < 1e-5                        # logits at scale 100: far inside the 1e-4 the mode carries
