"""GPU: the bf16 throughput mode.  Operands are rounded to bf16 (8 significant bits), so
it is checked (a) tightly against the oracle fed the SAME bf16-rounded operands (isolates
kernel correctness from operand rounding), (b) loosely against the exact fp32 oracle,
(c) on accuracy parity with the fp32 mode over a short training run."""
import numpy as np
import pytest
import torch

from oracle import uml_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf16_round(a):
    return torch.as_tensor(a).to(torch.bfloat16).to(torch.float32).numpy()


def _case(rng, d, C, n_img, n_txt, scale):
    xi = rng.standard_normal((n_img, d)).astype(np.float32)
    xt = rng.standard_normal((n_txt, d)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    w = rng.standard_normal((C, d)).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    return xi, rng.integers(0, C, n_img), xt, rng.integers(0, C, n_txt), w


def _engine(w, scale, cap_i, cap_t, precision, optimizer="adamw", wd=0.01):
    import umlh
    C, d = w.shape
    e = umlh.HeadEngine(d, d, C, optimizer=optimizer, weight_decay=wd, max_rows_img=cap_i, max_rows_txt=cap_t,
                        precision=precision, device=DEV)
    e.w_head.copy_(torch.from_numpy(w))
    e.scales.fill_(scale)
    return e


def _rb(x, y, idx=None):
    import umlh
    T = lambda a, t: torch.as_tensor(np.ascontiguousarray(a)).to(DEV, t).contiguous()
    return umlh.RowBatch(T(x, torch.float32), T(y, torch.int64), None if idx is None else T(idx, torch.int64))


# stw 2: two sample tiles per wave of the 1-D forward; stw "q": the 2-D forward (128-row tiles x groups of 256 classes with
# the cross-workgroup softmax merge) forced on at sizes far below the ones that select it by default
@pytest.mark.parametrize("d,C,bi,bt,scale,stw", [(128, 10, 70, 33, 30.0, 1), (128, 100, 50, 64, 100.0, 1),
                                                 (512, 1000, 300, 257, 100.0, 1), (512, 1000, 300, 257, 100.0, 2), (1024, 1000, 70, 300, 100.0, 1),
                                                 (384, 397, 40, 0, 50.0, 1), (256, 37, 0, 90, 20.0, 1),
                                                 (512, 1000, 300, 257, 100.0, "q"), (512, 1000, 0, 129, -100.0, "q"), (256, 600, 260, 200, 50.0, "q"),
                                                 (512, 300, 130, 0, 100.0, "q"), (256, 1024, 128, 128, 30.0, "q"),
                                                 (512, 1000, 300, 257, 100.0, "f"), (128, 100, 50, 64, 100.0, "f"), (256, 37, 0, 90, 20.0, "f"),
                                                 (384, 397, 40, 0, 50.0, "f"), (1024, 1000, 70, 300, -100.0, "f")])
def test_bf16_grad_step_vs_oracle_on_rounded_operands(d, C, bi, bt, scale, stw, monkeypatch):
    import umlh
    monkeypatch.setenv("UMLH_BF16_STW", "1" if stw in ("q", "f") else str(stw))
    monkeypatch.setenv("UMLH_BF16_FWD2D", "1" if stw == "q" else "0")
    monkeypatch.setenv("UMLH_BF16_FUSE", "1" if stw == "f" else "0")      # "f": forward and dW as ONE launch (granule-gated dW blocks)
    rng = np.random.default_rng(d + C)
    xi, yi, xt, yt, w = _case(rng, d, C, 400, 350, scale)
    ii = rng.permutation(400)[:bi] if bi else None
    ti = rng.permutation(350)[:bt] if bt else None
    e = _engine(w, scale, 512, 512, "bf16")
    flat = e.grad_step(_rb(xi, yi, ii) if bi else None, _rb(xt, yt, ti) if bt else None, alpha=0.7)
    torch.cuda.synchronize()
    f = flat.cpu().numpy()
    gh, sc = f[:C * d].reshape(C, d), f[C * d + 2:]
    st = O.HeadState(_bf16_round(w), None, scale, scale, False)
    so = O.step_grads(st, _bf16_round(xi[ii]) if bi else None, yi[ii] if bi else None,
                      _bf16_round(xt[ti]) if bt else None, yt[ti] if bt else None, 0.7)
    if bi:
        assert abs(sc[umlh.S_LOSS_IMG] - so.loss_img) < 2e-3 and abs(sc[umlh.S_ACC_IMG] - so.acc_img) < 1e-6 + 2.0 / bi
    if bt:
        assert abs(sc[umlh.S_LOSS_TXT] - so.loss_txt) < 2e-3 and abs(sc[umlh.S_ACC_TXT] - so.acc_txt) < 1e-6 + 2.0 / bt
    # dZ is rounded to bf16 before the dW product: 2^-9 relative per element
    s = np.abs(so.grads["w_head"]).max()
    np.testing.assert_allclose(gh, so.grads["w_head"], atol=8e-3 * s, rtol=2e-2)
    # exact-fp32 oracle, loose
    ex = O.step_grads(O.HeadState(w, None, scale, scale, False), xi[ii] if bi else None, yi[ii] if bi else None,
                      xt[ti] if bt else None, yt[ti] if bt else None, 0.7)
    if bi:
        assert abs(sc[umlh.S_LOSS_IMG] - ex.loss_img) < 5e-2
    assert np.abs(gh - ex.grads["w_head"]).max() < 5e-2 * np.abs(ex.grads["w_head"]).max()


@pytest.mark.parametrize("d,C,bi,bt,alpha", [(512, 1000, 300, 257, 0.7), (128, 100, 70, 33, 1.0), (256, 37, 90, 0, 1.0)])
def test_bf16_train_step_gradient_diagnostics(d, C, bi, bt, alpha):
    """The per-modality gradient sums that ride on the slab reduction (finetune.py:190-191,203-206)
    against the oracle's gradients on the same bf16-rounded operands."""
    import umlh
    rng = np.random.default_rng(7 * d + C)
    xi, yi, xt, yt, w = _case(rng, d, C, 400, 350, 50.0)
    ii = rng.permutation(400)[:bi]
    ti = rng.permutation(350)[:bt] if bt else None
    e = _engine(w, 50.0, 512, 512, "bf16")
    e.enable_diagnostics()
    sc = torch.zeros(umlh.N_SCALARS, device=DEV)
    e.train_step(_rb(xi, yi, ii), _rb(xt, yt, ti) if bt else None, lr=1e-3, step=1, alpha=alpha, scalars_out=sc)
    torch.cuda.synchronize()
    got = umlh.grad_diagnostics(sc.cpu(), C * d, bi, bt)
    st = O.HeadState(_bf16_round(w), None, 50.0, 50.0, False)
    so = O.step_grads(st, _bf16_round(xi[ii]), yi[ii], _bf16_round(xt[ti]) if bt else None, yt[ti] if bt else None, alpha)
    ref = O.grad_diagnostics(so.g_head_img, so.g_head_txt)
    assert abs(got["grad_direction_sim"] - ref["grad_direction_sim"]) < 5e-3
    assert abs(got["grad_agreement_rate"] - ref["grad_agreement_rate"]) < 2e-2
    assert abs(got["img_grad_norm"] - ref["img_grad_norm"]) < 1e-2 * ref["img_grad_norm"]
    assert abs(got["txt_grad_norm"] - ref["txt_grad_norm"]) <= 1e-2 * ref["txt_grad_norm"]


def test_bf16_eval_batch():
    import umlh
    rng = np.random.default_rng(2)
    xi, yi, _, _, w = _case(rng, 128, 100, 300, 10, 100.0)
    e = _engine(w, 100.0, 512, 32, "bf16")
    sc = e.eval_batch(_rb(xi, yi)).cpu().numpy()
    st = O.HeadState(_bf16_round(w), None, 100.0, 100.0, False)
    z, _ = O.forward(st, _bf16_round(xi), None)
    assert abs(sc[umlh.S_LOSS_SUM] / 300 - O.cross_entropy_mean(z, yi)) < 2e-3
    assert abs(sc[umlh.S_CORRECT] - O.top1_correct(z, yi).sum()) <= 2


def test_bf16_shape_requirements():
    import umlh
    with pytest.raises(umlh.UmlhError):
        umlh.HeadEngine(48, 128, 10, has_proj=True, precision="bf16", device=DEV)      # d_img % 64
    with pytest.raises(umlh.UmlhError):
        umlh.HeadEngine(96, 96, 10, precision="bf16", device=DEV)                       # d_shared % 128


@pytest.mark.parametrize("d_img,d_sh,C,bi,bt,learn,use_idx", [(64, 128, 10, 70, 33, False, True), (128, 256, 100, 300, 257, True, True),
                                                              (192, 128, 37, 64, 0, False, False), (1024, 384, 1000, 130, 90, False, True)])
def test_bf16_two_layer_head_grad_step(d_img, d_sh, C, bi, bt, learn, use_idx):
    """bf16 mode with img_proj (head.py:64-66,79): H = X W_proj^T, fused forward on H, dW_head, dH^T, dW_proj all
    through the bf16 GEMM kernels; against the oracle on bf16-rounded operands (H, dZ and dH are additionally
    rounded on the device) and loosely against the exact fp32 oracle."""
    import umlh
    rng = np.random.default_rng(d_img + d_sh + C)
    n_i, n_t = 400, 350
    xi = rng.standard_normal((n_i, d_img)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt = rng.standard_normal((n_t, d_sh)).astype(np.float32)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    wp = (rng.standard_normal((d_sh, d_img)) / np.sqrt(d_img)).astype(np.float32)
    wh = rng.standard_normal((C, d_sh)).astype(np.float32)
    wh /= np.linalg.norm(wh, axis=1, keepdims=True)
    yi, yt = rng.integers(0, C, n_i), rng.integers(0, C, n_t)
    si, stx = 8.0, 5.0
    e = umlh.HeadEngine(d_img, d_sh, C, has_proj=True, learnable_temp=learn, optimizer="adamw", max_rows_img=512,
                        max_rows_txt=512, precision="bf16", device=DEV)
    e.w_head.copy_(torch.from_numpy(wh)); e.w_proj.copy_(torch.from_numpy(wp))
    e.scales.copy_(torch.tensor([si, stx]))
    if use_idx:
        ii, ti = rng.permutation(n_i)[:bi], (rng.permutation(n_t)[:bt] if bt else None)
        bi_rb, bt_rb = _rb(xi, yi, ii), (_rb(xt, yt, ti) if bt else None)
        xi_b, yi_b = xi[ii], yi[ii]
        xt_b, yt_b = (xt[ti], yt[ti]) if bt else (None, None)
    else:
        bi_rb, bt_rb = _rb(xi[:bi], yi[:bi]), (_rb(xt[:bt], yt[:bt]) if bt else None)
        xi_b, yi_b = xi[:bi], yi[:bi]
        xt_b, yt_b = (xt[:bt], yt[:bt]) if bt else (None, None)
    flat = e.grad_step(bi_rb, bt_rb, alpha=0.7)
    torch.cuda.synchronize()
    f = flat.cpu().numpy()
    nh, npj = C * d_sh, d_sh * d_img
    gh, gp, gs, sc = f[:nh].reshape(C, d_sh), f[nh:nh + npj].reshape(d_sh, d_img), f[nh + npj:nh + npj + 2], f[nh + npj + 2:]
    for rounded, tol_l, tol_g in ((True, 2e-2, 4e-2), (False, 6e-2, 8e-2)):
        R = _bf16_round if rounded else (lambda a: a)
        st = O.HeadState(R(wh), R(wp), si, stx, learn)
        so = O.step_grads(st, R(xi_b), yi_b, R(xt_b) if bt else None, yt_b, 0.7)
        assert abs(sc[umlh.S_LOSS_IMG] - so.loss_img) < tol_l * max(1.0, so.loss_img)
        if bt:
            assert abs(sc[umlh.S_LOSS_TXT] - so.loss_txt) < tol_l * max(1.0, so.loss_txt)
        for got, key in ((gh, "w_head"), (gp, "w_proj")):
            ref = so.grads[key]
            assert np.abs(got - ref).max() < tol_g * np.abs(ref).max(), key
        if learn:
            assert abs(gs[0] - float(so.grads["img_scale"])) < tol_g * max(1e-3, abs(float(so.grads["img_scale"])))


def test_bf16_two_layer_head_training_tracks_fp32():
    """60 AdamW steps of the 2-layer head in bf16 mode vs fp32 mode on the same batches: same loss curve
    (2e-2) and weights within bf16 noise."""
    import umlh
    rng = np.random.default_rng(3)
    d_img, d_sh, C, n = 128, 256, 50, 2000
    proto_i, proto_t = rng.standard_normal((C, d_img)), rng.standard_normal((C, d_sh))
    yi, yt = rng.integers(0, C, n), rng.integers(0, C, n)
    xi = (proto_i[yi] + 2.0 * rng.standard_normal((n, d_img))).astype(np.float32)
    xt = (proto_t[yt] + 2.0 * rng.standard_normal((n, d_sh))).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True); xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    wp = (rng.standard_normal((d_sh, d_img)) / np.sqrt(d_img)).astype(np.float32)
    wh = (0.05 * rng.standard_normal((C, d_sh))).astype(np.float32)
    T = lambda a, t=torch.float32: torch.as_tensor(a).to(DEV, t).contiguous()
    Xi, Yi, Xt, Yt = T(xi), T(yi, torch.int64), T(xt), T(yt, torch.int64)
    tabs = {"fp32": ((Xi, Yi), (Xt, Yt)), "bf16": ((Xi, Yi, umlh.to_bf16(Xi)), (Xt, Yt, umlh.to_bf16(Xt)))}
    steps, B = 60, 128
    g = torch.Generator().manual_seed(0)
    bi = [torch.randint(0, n, (B,), generator=g).to(DEV) for _ in range(steps)]
    bt = [torch.randint(0, n, (B,), generator=g).to(DEV) for _ in range(steps)]
    out = {}
    for prec in ("fp32", "bf16"):
        e = umlh.HeadEngine(d_img, d_sh, C, has_proj=True, optimizer="adamw", weight_decay=0.01, max_rows_img=B,
                            max_rows_txt=B, precision=prec, device=DEV)
        e.w_head.copy_(T(wh)); e.w_proj.copy_(T(wp)); e.scales.fill_(10.0)
        sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
        e.train_steps(tabs[prec][0], bi, tabs[prec][1], bt, [2e-3] * steps, first_step=1, scalars_out=sc)
        torch.cuda.synchronize()
        out[prec] = (sc.cpu().numpy(), e.w_head.cpu().numpy().copy(), e.w_proj.cpu().numpy().copy())
    l32, l16 = out["fp32"][0], out["bf16"][0]
    assert l32[-1, umlh.S_LOSS_IMG] < 0.8 * l32[0, umlh.S_LOSS_IMG]                    # it trains
    np.testing.assert_allclose(l16[:, umlh.S_LOSS_IMG], l32[:, umlh.S_LOSS_IMG], atol=3e-2)
    np.testing.assert_allclose(l16[:, umlh.S_LOSS_TXT], l32[:, umlh.S_LOSS_TXT], atol=3e-2)
    for k in (1, 2):
        assert np.abs(out["bf16"][k] - out["fp32"][k]).max() < 0.1 * np.abs(out["fp32"][k]).max()


def test_bf16_training_accuracy_parity_with_fp32():
    """Same seed, same batches: 300 AdamW steps in bf16 mode vs fp32 mode; final top-1 on
    20k held-out rows within +-0.1 pp (north_star)."""
    import umlh
    rng = np.random.default_rng(11)
    d, C, n = 128, 100, 20000
    proto = rng.standard_normal((C, d)).astype(np.float32)

    def draw(m):
        y = rng.integers(0, C, m)
        x = proto[y] + 2.5 * rng.standard_normal((m, d)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32), y
    xi, yi = draw(n)
    xt, yt = draw(3000)
    xe, ye = draw(20000)
    w0 = O.zero_shot_weights(xt, yt, C)
    accs = {}
    for prec in ("fp32", "bf16"):
        e = _engine(w0.copy(), 30.0, 256, 256, prec)
        bi_t, bt_t = _rb(xi, yi), _rb(xt, yt)
        g = torch.Generator().manual_seed(5)
        for k in range(300):
            ii = torch.randint(0, n, (256,), generator=g).to(DEV)
            ti = torch.randint(0, 3000, (256,), generator=g).to(DEV)
            e.train_step(umlh.RowBatch(bi_t.feats, bi_t.labels, ii, feats_bf16=bi_t.feats_bf16),
                         umlh.RowBatch(bt_t.feats, bt_t.labels, ti, feats_bf16=bt_t.feats_bf16), lr=2e-3, step=k + 1)
            if prec == "bf16" and bi_t.feats_bf16 is None:
                bi_t.feats_bf16, bt_t.feats_bf16 = umlh.to_bf16(bi_t.feats), umlh.to_bf16(bt_t.feats)
        ev = _engine(e.w_head.cpu().numpy(), 30.0, 20000, 32, "fp32")
        sc = ev.eval_batch(_rb(xe, ye)).cpu().numpy()
        accs[prec] = sc[umlh.S_CORRECT] / 20000
    print(f"top-1 fp32 {accs['fp32']:.4f}  bf16 {accs['bf16']:.4f}  diff {100 * (accs['bf16'] - accs['fp32']):+.3f} pp")
    assert accs["fp32"] > 0.5
    assert abs(accs["bf16"] - accs["fp32"]) <= 0.001


def test_bf16_single_launch_forward_dw_equals_two_launches_bit_for_bit(monkeypatch):
    """UMLH_BF16_FUSE=1 (forward and dW as one launch, dW blocks gated on the forward blocks' granules, dZ^T read with
    device-coherent loads): the arithmetic and its order are those of the two-launch step, so 40 AdamW steps on changing
    batches (the same dZ^T addresses rewritten every step -- a stale L2 line would show) end in identical weights, moments
    and scalars."""
    import umlh
    rng = np.random.default_rng(3)
    d, C, n = 512, 1000, 6000
    xi, yi, xt, yt, w = _case(rng, d, C, n, 3000, 100.0)
    out = {}
    for mode in ("0", "1", "2"):       # separate launches / forward + dW as one / the whole step (update + scalars too) as one
        monkeypatch.setenv("UMLH_BF16_FUSE", mode)
        e = _engine(w.copy(), 100.0, 1024, 1024, "bf16")
        bi_t, bt_t = _rb(xi, yi), _rb(xt, yt)
        bi_t.feats_bf16, bt_t.feats_bf16 = umlh.to_bf16(bi_t.feats), umlh.to_bf16(bt_t.feats)
        g = torch.Generator().manual_seed(9)
        scal = torch.zeros(40, umlh.N_SCALARS, device=DEV) if hasattr(umlh, "N_SCALARS") else None
        for k in range(40):
            ii = torch.randint(0, n, (1024 if k % 3 else 700,), generator=g).to(DEV)
            ti = torch.randint(0, 3000, (1024 if k % 4 else 333,), generator=g).to(DEV)
            bi = umlh.RowBatch(bi_t.feats, bi_t.labels, ii, feats_bf16=bi_t.feats_bf16) if k % 10 != 5 else None     # text-only steps
            bt = umlh.RowBatch(bt_t.feats, bt_t.labels, ti, feats_bf16=bt_t.feats_bf16) if k % 10 != 7 else None     # image-only steps
            e.train_step(bi, bt, lr=1e-3, step=k + 1, scalars_out=None if scal is None else scal[k])
        torch.cuda.synchronize()
        out[mode] = (e.w_head.clone(), e.m_head.clone(), e.v_head.clone(), None if scal is None else scal.clone())
    for mode in ("1", "2"):
        for a, b in zip(out["0"], out[mode]):
            if a is not None:
                assert torch.equal(a, b), mode
    assert torch.isfinite(out["2"][0]).all()


@pytest.mark.parametrize("opt,learn", [("sgd", True), ("adam", False), ("adamw", True)])
def test_bf16_single_launch_step_optimizers_and_learnable_temperature(opt, learn, monkeypatch):
    """The one-launch step against the three-launch step for SGD-momentum / Adam / AdamW and with learnable logit scales
    (their update rides in the finalize block): weights, moments and scales bit for bit after 12 steps."""
    import umlh
    rng = np.random.default_rng(21)
    d, C, n = 256, 300, 3000
    xi, yi, xt, yt, w = _case(rng, d, C, n, 2000, 20.0)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("UMLH_BF16_FUSE", mode)
        e = umlh.HeadEngine(d, d, C, optimizer=opt, weight_decay=0.01, learnable_temp=learn, max_rows_img=512, max_rows_txt=512,
                            precision="bf16", device=DEV)
        e.w_head.copy_(torch.from_numpy(w)); e.scales.fill_(20.0)
        bi_t, bt_t = _rb(xi, yi), _rb(xt, yt)
        bi_t.feats_bf16, bt_t.feats_bf16 = umlh.to_bf16(bi_t.feats), umlh.to_bf16(bt_t.feats)
        g = torch.Generator().manual_seed(4)
        for k in range(12):
            ii = torch.randint(0, n, (512 if k % 2 else 321,), generator=g).to(DEV)
            ti = torch.randint(0, 2000, (384,), generator=g).to(DEV)
            e.train_step(umlh.RowBatch(bi_t.feats, bi_t.labels, ii, feats_bf16=bi_t.feats_bf16),
                         umlh.RowBatch(bt_t.feats, bt_t.labels, ti, feats_bf16=bt_t.feats_bf16), lr=1e-2, step=k + 1)
        torch.cuda.synchronize()
        out[mode] = (e.w_head.clone(), e.m_head.clone(), e.v_head.clone(), e.scales.clone())
    for a, b in zip(out["0"], out["2"]):
        assert torch.equal(a, b)
    if learn:
        assert float((out["2"][3] - 20.0).abs().max()) > 0.0      # the scales did move


def test_bf16_train_step_captured_in_a_graph_replays_correctly():
    """A bf16 train step captured into a HIP graph (torch.cuda.graph) and replayed equals the same step run eagerly the same
    number of times: on a capturing stream the engine takes the launch-per-kernel form (the one-launch step's hand-off tags are
    launch arguments and would be stale in a replay)."""
    import umlh
    rng = np.random.default_rng(8)
    d, C, n = 256, 300, 2000
    xi, yi, xt, yt, w = _case(rng, d, C, n, 1500, 20.0)
    ii = torch.as_tensor(rng.permutation(n)[:512]).to(DEV)
    ti = torch.as_tensor(rng.permutation(1500)[:384]).to(DEV)

    def make():
        e = _engine(w.copy(), 20.0, 512, 512, "bf16")
        bi_t, bt_t = _rb(xi, yi), _rb(xt, yt)
        bi_t.feats_bf16, bt_t.feats_bf16 = umlh.to_bf16(bi_t.feats), umlh.to_bf16(bt_t.feats)
        bi = umlh.RowBatch(bi_t.feats, bi_t.labels, ii, feats_bf16=bi_t.feats_bf16)
        bt = umlh.RowBatch(bt_t.feats, bt_t.labels, ti, feats_bf16=bt_t.feats_bf16)
        return e, bi, bt
    e1, bi, bt = make()
    for _ in range(4):
        e1.train_step(bi, bt, lr=1e-2, step=3)
    torch.cuda.synchronize()
    e2, bi2, bt2 = make()
    e2.train_step(bi2, bt2, lr=1e-2, step=3)                # eager once (also warms every lazily set attribute)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e2.train_step(bi2, bt2, lr=1e-2, step=3)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(e1.w_head, e2.w_head) and torch.equal(e1.m_head, e2.m_head) and torch.equal(e1.v_head, e2.v_head)


_WT_SCRIPT = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import umlh
rng = np.random.default_rng(5)
d, C, n = 256, 300, 2000
x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
y = rng.integers(0, C, n)
w = rng.standard_normal((C, d)).astype(np.float32); w /= np.linalg.norm(w, axis=1, keepdims=True)
out = []
for prec in ("bf16", "fp32"):
    e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=512, max_rows_txt=512, precision=prec, device="cuda:0")
    e.w_head.copy_(torch.from_numpy(w)); e.scales.fill_(30.0)
    X = torch.from_numpy(x).cuda(); Y = torch.from_numpy(y).cuda(); X16 = umlh.to_bf16(X)
    g = torch.Generator().manual_seed(2)
    for k in range(6):
        ii = torch.randint(0, n, (512,), generator=g).cuda(); ti = torch.randint(0, n, (300,), generator=g).cuda()
        e.train_step(umlh.RowBatch(X, Y, ii, feats_bf16=X16), umlh.RowBatch(X, Y, ti, feats_bf16=X16), lr=1e-2, step=k + 1)
    torch.cuda.synchronize()
    out.append(hashlib.sha256(e.w_head.cpu().numpy().tobytes() + e.v_head.cpu().numpy().tobytes()).hexdigest())
print("DIGEST", *out)
"""


def test_plain_stores_switch_gives_identical_results(tmp_path):
    """UMLH_WT=0 (plain instead of write-through stores, launch-per-kernel step) is read once per process: two child processes,
    one per setting, must end six steps (bf16 and fp32) with bit-identical weights and second moments."""
    import os, subprocess, sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unpaired-multimodal-learning_amd")
    script = tmp_path / "wt_case.py"
    script.write_text(_WT_SCRIPT)
    digests = {}
    for wt in ("1", "0"):
        env = dict(os.environ, UMLH_WT=wt)
        r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[wt] = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1]
    assert digests["1"] == digests["0"]


def test_bf16_two_layer_head_split_step_equals_fused_step():
    """Data-parallel split (grad_step -> apply_update) == fused train_step for the bf16 2-layer head: same weights,
    optimizer state and scalars after 3 steps (single rank: the all-reduce is the identity)."""
    import umlh
    rng = np.random.default_rng(9)
    d_img, d_sh, C, n, B = 128, 128, 20, 500, 96
    xi = rng.standard_normal((n, d_img)).astype(np.float32)
    xt = rng.standard_normal((n, d_sh)).astype(np.float32)
    yi, yt = rng.integers(0, C, n), rng.integers(0, C, n)
    wp = (rng.standard_normal((d_sh, d_img)) / np.sqrt(d_img)).astype(np.float32)
    wh = (0.1 * rng.standard_normal((C, d_sh))).astype(np.float32)
    T = lambda a, t=torch.float32: torch.as_tensor(a).to(DEV, t).contiguous()
    Xi, Yi, Xt, Yt = T(xi), T(yi, torch.int64), T(xt), T(yt, torch.int64)
    Xi16, Xt16 = umlh.to_bf16(Xi), umlh.to_bf16(Xt)
    engines = []
    for _ in range(2):
        e = umlh.HeadEngine(d_img, d_sh, C, has_proj=True, learnable_temp=True, optimizer="adamw", weight_decay=0.01,
                            max_rows_img=B, max_rows_txt=B, precision="bf16", device=DEV)
        e.w_head.copy_(T(wh)); e.w_proj.copy_(T(wp)); e.scales.fill_(5.0)
        engines.append(e)
    g = torch.Generator().manual_seed(1)
    for step in range(1, 4):
        ii = torch.randint(0, n, (B,), generator=g).to(DEV)
        ti = torch.randint(0, n, (70,), generator=g).to(DEV)
        bi = umlh.RowBatch(Xi, Yi, ii, feats_bf16=Xi16)
        bt = umlh.RowBatch(Xt, Yt, ti, feats_bf16=Xt16)
        s1 = engines[0].train_step(bi, bt, lr=2e-3, step=step, alpha=0.5).clone()
        engines[1].grad_step(bi, bt, alpha=0.5)
        s2 = engines[1].apply_update(lr=2e-3, step=step).clone()
        torch.cuda.synchronize()
        np.testing.assert_allclose(s1[:8].cpu().numpy(), s2[:8].cpu().numpy(), rtol=1e-5, atol=1e-6)
    for name in ("w_head", "w_proj", "m_head", "v_head", "m_proj", "v_proj", "scales"):
        a, b = getattr(engines[0], name).cpu().numpy(), getattr(engines[1], name).cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7, err_msg=name)


# ------------------------------------------------------------------------------------------------------------------- #
# round 3: the one-launch step runs CLAIMED TASKS on persistent workgroups (csrc/umlh_common.h, StepCtl): no wait may depend on
# dispatch order, on the width of the grid or on every workgroup being resident
# ------------------------------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("variant", ["grid24", "grid7", "lazy", "lazy_grid24", "grid24_dp"])
def test_bf16_one_launch_step_is_independent_of_grid_width_and_of_who_runs_a_task(variant, monkeypatch):
    """The step with (i) far fewer workgroups than tasks (every workgroup then walks several home tasks per phase through the
    cold path), (ii) every fourth workgroup leaving its forward and dW home tasks alone, as if it had not been dispatched yet
    (UMLH_STEP_LAZY: whoever needs a tile finds it untaken, takes it and runs it first -- update -> dW -> forward, the full
    depth of tasks_wait's hand-back), and both together, ends 24 AdamW steps on changing batches with weights, moments and step
    scalars BIT-identical to the launch-per-kernel step; the status word stays 0.  "_dp": the same for the data-parallel split
    step (gradient message written by the update tasks, update as its own launch)."""
    import umlh
    rng = np.random.default_rng(33)
    d, C, n = 512, 1000, 6000
    xi, yi, xt, yt, w = _case(rng, d, C, n, 3000, 100.0)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("UMLH_BF16_FUSE", mode)
        monkeypatch.delenv("UMLH_STEP_GRID", raising=False)
        monkeypatch.delenv("UMLH_STEP_LAZY", raising=False)
        monkeypatch.delenv("UMLH_FORCE_DP", raising=False)
        if mode == "2":
            if "grid24" in variant: monkeypatch.setenv("UMLH_STEP_GRID", "24")
            if "grid7" in variant: monkeypatch.setenv("UMLH_STEP_GRID", "7")
            if "lazy" in variant: monkeypatch.setenv("UMLH_STEP_LAZY", "1")
        e = _engine(w.copy(), 100.0, 1024, 1024, "bf16")
        bi_t, bt_t = _rb(xi, yi), _rb(xt, yt)
        bi_t.feats_bf16, bt_t.feats_bf16 = umlh.to_bf16(bi_t.feats), umlh.to_bf16(bt_t.feats)
        g = torch.Generator().manual_seed(19)
        scal = torch.zeros(24, umlh.N_SCALARS, device=DEV)
        for k in range(24):
            ii = torch.randint(0, n, (1024 if k % 3 else 700,), generator=g).to(DEV)
            ti = torch.randint(0, 3000, (1024 if k % 4 else 333,), generator=g).to(DEV)
            bi = umlh.RowBatch(bi_t.feats, bi_t.labels, ii, feats_bf16=bi_t.feats_bf16) if k % 10 != 5 else None
            bt = umlh.RowBatch(bt_t.feats, bt_t.labels, ti, feats_bf16=bt_t.feats_bf16) if k % 10 != 7 else None
            if variant.endswith("_dp"):
                e.grad_step(bi, bt, weights_unchanged=k > 0)
                e.apply_update(lr=1e-3, step=k + 1, scalars_out=scal[k])
            else:
                e.train_step(bi, bt, lr=1e-3, step=k + 1, scalars_out=scal[k])
        torch.cuda.synchronize()
        assert e.step_status() == (0, 0, 0, 0)
        assert (e.step_launches() > 0) == (mode == "2")
        out[mode] = (e.w_head.clone(), e.m_head.clone(), e.v_head.clone(), scal.clone())
    for a, b in zip(out["0"], out["2"]):
        assert torch.equal(a, b)
    assert torch.isfinite(out["2"][0]).all() and torch.isfinite(out["2"][3][:, :4]).all()


_SHARED_GPU_SCRIPT = r"""
import sys, time, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import umlh
steps, tag = int(sys.argv[2]), sys.argv[3]
d, C, n_i, n_t, B = 512, 1000, 65536, 29940, 4096
gen = torch.Generator(device="cuda").manual_seed(5)
def rows(n):
    x = torch.randn(n, d, generator=gen, device="cuda"); x /= x.norm(dim=1, keepdim=True)
    return x, torch.randint(0, C, (n,), generator=gen, device="cuda")
xi, yi = rows(n_i); xt, yt = rows(n_t)
w = torch.randn(C, d, generator=gen, device="cuda"); w /= w.norm(dim=1, keepdim=True)
e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B, precision="bf16", device="cuda:0")
e.w_head.copy_(w); e.scales.fill_(100.0)
ti_tab, tt_tab = (xi, yi, umlh.to_bf16(xi)), (xt, yt, umlh.to_bf16(xt))
g = torch.Generator().manual_seed(11)
scal = torch.zeros(steps, umlh.N_SCALARS, device="cuda")
blk = 20
if tag != "solo":                 # rendezvous: both children are past their set-up before either starts stepping
    import os
    open(sys.argv[4] + "." + tag, "w").close()
    t0 = time.time()
    while not all(os.path.exists(sys.argv[4] + "." + t) for t in ("a", "b")) and time.time() - t0 < 120: time.sleep(0.01)
worst = 0.0
for k0 in range(0, steps, blk):
    ib = [torch.randint(0, n_i, (B,), generator=g).cuda() for _ in range(blk)]
    tb = [torch.randint(0, n_t, (B,), generator=g).cuda() for _ in range(blk)]
    torch.cuda.synchronize(); t1 = time.time()
    e.train_steps(ti_tab, ib, tt_tab, tb, [1e-3] * blk, first_step=k0 + 1, scalars_out=scal[k0:k0 + blk])
    torch.cuda.synchronize(); worst = max(worst, (time.time() - t1) / blk)
e.check_status()
assert e.step_launches() == steps, e.step_launches()
s = scal.cpu().numpy()
print("RESULT", hashlib.sha256(e.w_head.cpu().numpy().tobytes() + s.tobytes()).hexdigest(), bool(np.isfinite(s[:, :4]).all()),
      float(s[-1, 0]), "%.3f" % (worst * 1e3))
"""


def test_bf16_two_processes_step_cfg2_size_engines_concurrently_on_one_gpu(tmp_path):
    """two processes driving cfg2-size bf16 engines (d = 512, C = 1000, 4096 + 4096 rows per step, the
    one-launch step) at the same time on cuda:0.  With round 2's waits on lower block ids this timed out (2 s per wait) and
    ended in NaN; claimed tasks do not care who holds the CUs: both children end 120 steps with finite losses, status 0 and
    weights + per-step scalars bit-identical to a solo run of the same script, and no block of 20 steps comes near a wait's
    50 ms bound per step."""
    import os, subprocess, sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unpaired-multimodal-learning_amd")
    script = tmp_path / "shared_gpu_case.py"
    script.write_text(_SHARED_GPU_SCRIPT)
    rv = str(tmp_path / "rendezvous")
    def parse(r):
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        f = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
        return f[1], f[2] == "True", float(f[3]), float(f[4])
    solo = parse(subprocess.run([sys.executable, str(script), root, "120", "solo", rv], capture_output=True, text=True, timeout=600))
    procs = [subprocess.Popen([sys.executable, str(script), root, "120", t, rv], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for t in ("a", "b")]
    res = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        res.append(parse(subprocess.CompletedProcess(p.args, p.returncode, o, e)))
    assert solo[1] and np.isfinite(solo[2])
    for r in res:
        assert r[1] and np.isfinite(r[2])
        assert r[0] == solo[0]                   # same bits as the run that had the GPU to itself
        assert r[3] < 5.0, r                     # ms per step of the slowest 20-step block (solo: ~0.05; a timed-out wait: >= 2.5)
