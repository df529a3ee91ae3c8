"""GPU parity: the HIP path (through the C ABI) against the committed golden
vectors of the reference and against the CPU oracle on seeded inputs.

Tolerances: logits / loss 1e-4 absolute (north_star, fp32); gradients and
weights relative to their scale (fp32 reduction-order noise)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import uml_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _t(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def _engine(state: O.HeadState, optimizer="adamw", wd=0.0, max_img=64, max_txt=64, **kw):
    import umlh
    C, ds = state.w_head.shape
    di = state.w_proj.shape[1] if state.w_proj is not None else ds
    e = umlh.HeadEngine(di, ds, C, has_proj=state.w_proj is not None, learnable_temp=state.learnable_temp,
                        optimizer=optimizer, weight_decay=wd, max_rows_img=max_img, max_rows_txt=max_txt,
                        device=DEV, **kw)
    e.w_head.copy_(_t(state.w_head))
    if state.w_proj is not None:
        e.w_proj.copy_(_t(state.w_proj))
    e.scales.copy_(torch.tensor([state.img_scale, state.txt_scale], dtype=torch.float32))
    return e


def _rb(x, y, index=None):
    import umlh
    return umlh.RowBatch(_t(x, torch.float32), _t(y, torch.int64), None if index is None else _t(index, torch.int64))


def _unpack_grads(e, flat):
    flat = flat.detach().cpu().numpy()
    nh = e.num_classes * e.d_shared
    npj = e.d_shared * e.d_img if e.has_proj else 0
    gh = flat[:nh].reshape(e.num_classes, e.d_shared)
    gp = flat[nh:nh + npj].reshape(e.d_shared, e.d_img) if npj else None
    gs = flat[nh + npj:nh + npj + 2]
    sc = flat[nh + npj + 2:]
    return gh, gp, gs, sc


STEP_CASES = ["clip_d64_c10", "clip_d512_c100", "clip_d128_c1000", "lin_d96_c37",
              "mlp_d48_t64_c10", "mlp_d96_t160_c20"]


@pytest.mark.parametrize("case", STEP_CASES)
def test_step_against_reference_golden(case):
    import umlh
    g = load_golden("step_" + case)
    learn = "g_img_scale" in g.files
    st = O.HeadState(g["w_head"].copy(), g["w_proj"].copy() if "w_proj" in g.files else None,
                     float(g["scale_img"]), float(g["scale_txt"]), learn)
    e = _engine(st)
    bi, bt = _rb(g["x_img"], g["y_img"]), _rb(g["x_txt"], g["y_txt"])
    zi = e.logits(bi, 0).cpu().numpy()
    zt = e.logits(bt, 1).cpu().numpy()
    np.testing.assert_allclose(zi, g["img_logits"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(zt, g["txt_logits"], atol=1e-4, rtol=0)
    flat = e.grad_step(bi, bt, alpha=float(g["alpha"]))
    torch.cuda.synchronize()
    gh, gp, gs, sc = _unpack_grads(e, flat)
    assert abs(sc[umlh.S_LOSS_IMG] - float(g["loss_img"])) < 1e-4
    assert abs(sc[umlh.S_LOSS_TXT] - float(g["loss_txt"])) < 1e-4
    assert abs(sc[umlh.S_ACC_IMG] - float(g["acc_img"])) < 1e-6
    assert abs(sc[umlh.S_ACC_TXT] - float(g["acc_txt"])) < 1e-6
    s = np.abs(g["g_head"]).max()
    np.testing.assert_allclose(gh, g["g_head"], atol=3e-5 * s, rtol=1e-4)
    if gp is not None:
        np.testing.assert_allclose(gp, g["g_proj"], atol=3e-5 * np.abs(g["g_proj"]).max(), rtol=1e-4)
    if learn:
        assert abs(gs[0] - float(g["g_img_scale"])) < 2e-5
        assert abs(gs[1] - float(g["g_txt_scale"])) < 2e-5


@pytest.mark.parametrize("tag,optim", [("adamw_lin", "adamw"), ("sgd_lin", "sgd"), ("adam_lin", "adam"),
                                       ("adamw_mlp", "adamw")])
def test_trajectory_against_reference_golden(tag, optim):
    import umlh
    g = load_golden("traj_" + tag)
    lr, wd, warm, max_iter, wlr, alpha = g["hyper"]
    learn = tag.endswith("mlp")
    st = O.HeadState(g["w_head0"].copy(), g["w_proj0"].copy() if "w_proj0" in g.files else None, 1.0, 1.0, learn)
    e = _engine(st, optimizer=optim, wd=float(wd))
    steps = g["x_img"].shape[0]
    scal = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
    for k in range(steps):
        e.train_step(_rb(g["x_img"][k], g["y_img"][k]), _rb(g["x_txt"][k], g["y_txt"][k]),
                     lr=float(g["lr"][k]), step=k + 1, alpha=float(alpha), scalars_out=scal[k])
        key = f"w_head_after_{k}"
        if key in g.files:
            np.testing.assert_allclose(e.w_head.cpu().numpy(), g[key], atol=3e-6, rtol=3e-5, err_msg=key)
            if e.has_proj:
                np.testing.assert_allclose(e.w_proj.cpu().numpy(), g[f"w_proj_after_{k}"], atol=3e-6, rtol=3e-5)
    sc = scal.cpu().numpy()
    np.testing.assert_allclose(sc[:, umlh.S_LOSS_IMG], g["loss_img"], atol=1e-4)
    np.testing.assert_allclose(sc[:, umlh.S_LOSS_TXT], g["loss_txt"], atol=1e-4)
    np.testing.assert_allclose(e.m_head.cpu().numpy(), g["m_head_final"], atol=2e-7, rtol=2e-3)
    if "v_head_final" in g.files:
        np.testing.assert_allclose(e.v_head.cpu().numpy(), g["v_head_final"], atol=1e-10, rtol=2e-3)
    if learn:
        s = e.scales.cpu().numpy()
        assert abs(s[0] - float(g["img_scale_final"])) < 2e-5 and abs(s[1] - float(g["txt_scale_final"])) < 2e-5


def test_zero_shot_init_golden():
    g = load_golden("text_side")
    C = int(g["num_classes"])
    st = O.HeadState(np.zeros((C, g["feats"].shape[1]), np.float32))
    e = _engine(st)
    e.zero_shot_init(torch.as_tensor(g["feats"]), torch.as_tensor(g["labels"]))
    w = e.w_head.cpu().numpy()
    np.testing.assert_allclose(w, g["zero_shot_w"], atol=1e-6)
    assert np.all(w[1] == 0) and np.all(w[7] == 0)


def _random_case(rng, di, ds, C, n_img, n_txt, bi, bt, proj, learn, scale):
    xi = rng.standard_normal((n_img, di)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt = rng.standard_normal((n_txt, ds)).astype(np.float32)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    yi = rng.integers(0, C, n_img)
    yt = rng.integers(0, C, n_txt)
    w = rng.standard_normal((C, ds)).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    wp = (rng.standard_normal((ds, di)) / np.sqrt(di)).astype(np.float32) if proj else None
    ii = rng.permutation(n_img)[:bi] if bi else None
    ti = rng.permutation(n_txt)[:bt] if bt else None
    st = O.HeadState(w, wp, scale, scale * 0.5 if learn else scale, learn)
    return st, xi, yi, xt, yt, ii, ti


@pytest.mark.parametrize("di,ds,C,bi,bt,proj,learn,scale", [
    (96, 96, 37, 50, 33, False, False, 100.0),      # ragged rows, C not a tile multiple
    (64, 64, 10, 300, 7, False, False, 30.0),       # several sample tiles / one tiny segment
    (40, 56, 12, 21, 40, True, True, 1.3),          # img_proj + learnable temperature
    (128, 128, 397, 64, 64, False, False, 100.0),   # CTW=2 path (SUN397)
    (70, 70, 200, 45, 0, False, False, 20.0),       # image-only modality (finetune.py:373-376)
    (32, 48, 1000, 33, 65, True, False, 1.0),       # C=1000 with projection
    (30, 30, 101, 17, 19, False, True, 2.0),        # d not a multiple of 4 -> scalar load path
])
def test_gathered_step_against_oracle(di, ds, C, bi, bt, proj, learn, scale):
    import umlh
    rng = np.random.default_rng(di * 1000 + C)
    st, xi, yi, xt, yt, ii, ti = _random_case(rng, di, ds, C, 400, 350, bi, bt, proj, learn, scale)
    e = _engine(st, max_img=512, max_txt=128)
    b_img = _rb(xi, yi, ii) if bi else None
    b_txt = _rb(xt, yt, ti) if bt else None
    flat = e.grad_step(b_img, b_txt, alpha=0.7)
    torch.cuda.synchronize()
    gh, gp, gs, sc = _unpack_grads(e, flat)
    so = O.step_grads(st, xi[ii] if bi else None, yi[ii] if bi else None,
                      xt[ti] if bt else None, yt[ti] if bt else None, 0.7)
    if bi:
        np.testing.assert_allclose(e.logits(b_img, 0).cpu().numpy(), so.zi, atol=1e-4, rtol=0)
        assert abs(sc[umlh.S_LOSS_IMG] - so.loss_img) < 1e-4 and abs(sc[umlh.S_ACC_IMG] - so.acc_img) < 1e-6
    if bt:
        np.testing.assert_allclose(e.logits(b_txt, 1).cpu().numpy(), so.zt, atol=1e-4, rtol=0)
        assert abs(sc[umlh.S_LOSS_TXT] - so.loss_txt) < 1e-4 and abs(sc[umlh.S_ACC_TXT] - so.acc_txt) < 1e-6
    s = np.abs(so.grads["w_head"]).max()
    np.testing.assert_allclose(gh, so.grads["w_head"], atol=3e-5 * s, rtol=2e-4)
    if proj and bi:
        np.testing.assert_allclose(gp, so.grads["w_proj"], atol=3e-5 * np.abs(so.grads["w_proj"]).max(), rtol=2e-4)
    if learn:
        if bi:
            assert abs(gs[0] - float(so.grads["img_scale"])) < 3e-5
        if bt:
            assert abs(gs[1] - float(so.grads["txt_scale"])) < 3e-5


def test_eval_batch_matches_validate():
    import umlh
    rng = np.random.default_rng(5)
    st, xi, yi, *_ = _random_case(rng, 64, 64, 100, 300, 10, 0, 0, False, False, 100.0)
    e = _engine(st, max_img=128)
    losses, correct = [], 0
    sc = torch.zeros(3, umlh.N_SCALARS, device=DEV)
    for j, s in enumerate(range(0, 300, 128)):
        e.eval_batch(_rb(xi[s:s + 128], yi[s:s + 128]), sc[j])
    sc = sc.cpu().numpy()
    rows = [128, 128, 44]
    val_loss = np.mean([sc[j, umlh.S_LOSS_SUM] / rows[j] for j in range(3)])
    val_acc = sc[:, umlh.S_CORRECT].sum() / 300
    ol, oa = O.validate(st, xi, yi, 128, rng_draw=False)
    assert abs(val_loss - ol) < 1e-4 and abs(val_acc - oa) < 1e-6


def test_split_grad_then_update_equals_fused_step():
    """umlh_grad_step + umlh_apply_update (the data-parallel split) == umlh_train_step."""
    rng = np.random.default_rng(9)
    st, xi, yi, xt, yt, ii, ti = _random_case(rng, 48, 64, 50, 200, 200, 96, 80, True, True, 3.0)
    e1 = _engine(st, wd=0.01, max_img=128, max_txt=128)
    e2 = _engine(st, wd=0.01, max_img=128, max_txt=128)
    for k in range(3):
        e1.train_step(_rb(xi, yi, ii), _rb(xt, yt, ti), lr=1e-3, step=k + 1, alpha=0.5)
        e2.grad_step(_rb(xi, yi, ii), _rb(xt, yt, ti), alpha=0.5)
        e2.apply_update(lr=1e-3, step=k + 1)
    torch.cuda.synchronize()
    for a, b in [(e1.w_head, e2.w_head), (e1.w_proj, e2.w_proj), (e1.scales, e2.scales), (e1.m_head, e2.m_head)]:
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-7, rtol=1e-6)


def test_full_size_cfg2_step_against_oracle():
    """BASELINE config 2 shape: d=512, C=1000, 4096 image + 4096 text rows gathered from
    larger tables, AdamW; two steps vs the oracle, plus the CE-gradient property
    sum_c dW[c,:] = 0."""
    import umlh
    rng = np.random.default_rng(0)
    d, C, B = 512, 1000, 4096
    st, xi, yi, xt, yt, ii, ti = _random_case(rng, d, d, C, 3 * B, 2 * B, B, B, False, False, 100.0)
    e = _engine(st, optimizer="adamw", wd=0.01, max_img=B, max_txt=B)
    flat = e.grad_step(_rb(xi, yi, ii), _rb(xt, yt, ti), alpha=1.0)
    torch.cuda.synchronize()
    gh, _, _, sc = _unpack_grads(e, flat)
    so = O.step_grads(st, xi[ii], yi[ii], xt[ti], yt[ti], 1.0)
    assert abs(sc[umlh.S_LOSS_IMG] - so.loss_img) < 1e-4 and abs(sc[umlh.S_LOSS_TXT] - so.loss_txt) < 1e-4
    s = np.abs(so.grads["w_head"]).max()
    np.testing.assert_allclose(gh, so.grads["w_head"], atol=5e-5 * s, rtol=5e-4)
    assert np.abs(gh.sum(axis=0)).max() < 1e-4 * np.abs(gh).sum(axis=0).max()
    opt = O.OptState("adamw", 0.01)
    for k in range(2):
        so = O.step_grads(st, xi[ii], yi[ii], xt[ti], yt[ti], 1.0)
        O.optimizer_step(st, so.grads, opt, 1e-3)
        e.train_step(_rb(xi, yi, ii), _rb(xt, yt, ti), lr=1e-3, step=k + 1, alpha=1.0)
    # Adam's first steps move every weight by ~lr*sign(g): the handful of elements whose
    # gradient is at fp32-noise level (|g| ~ eps) may legitimately take the other sign.
    diff = np.abs(e.w_head.cpu().numpy() - st.w_head)
    assert (diff > 2e-5 + 1e-4 * np.abs(st.w_head)).mean() < 1e-4
    assert diff.max() <= 2 * 2 * 1e-3 + 1e-6


def test_argument_errors_are_loud():
    import umlh
    with pytest.raises(AssertionError):
        umlh.HeadEngine(8, 8, 4, optimizer="lion", device=DEV)
    with pytest.raises(umlh.UmlhError):
        umlh.HeadEngine(8, 8, 5000, device=DEV)                 # C > 1024 unsupported
    st = O.HeadState(np.zeros((4, 8), np.float32))
    e = _engine(st, max_img=16, max_txt=16)
    x = np.zeros((32, 8), np.float32)
    y = np.zeros(32, np.int64)
    with pytest.raises(umlh.UmlhError):
        e.train_step(_rb(x, y), None, lr=1e-3, step=1)          # rows > capacity
    with pytest.raises(umlh.UmlhError):
        e.train_step(None, None, lr=1e-3, step=1)               # finetune.py:123


def test_device_permutation_is_a_bijection_and_seed_dependent():
    import umlh
    for n in (1, 2, 7, 4096, 29940, 1281167):
        p = umlh.random_permutation(n, 12345, DEV)
        assert torch.equal(torch.sort(p).values, torch.arange(n, device=DEV))
    a, b = umlh.random_permutation(29940, 1, DEV), umlh.random_permutation(29940, 2, DEV)
    assert not torch.equal(a, b)
    assert (a == torch.arange(29940, device=DEV)).float().mean() < 0.01       # not the identity
    # epoch coverage through the loader: every row exactly once per epoch
    from engine.datasets.utils import FeatureLoader, FeatureTable
    ld = FeatureLoader(FeatureTable(torch.zeros(1000, 4), torch.zeros(1000, dtype=torch.long), DEV), 96, shuffle=True,
                       order_rng="device")
    seen = torch.cat(list(ld.iter_index()))
    assert seen.numel() == 1000 and torch.equal(torch.sort(seen).values, torch.arange(1000, device=DEV))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_eval_rows_per_row_stats(precision):
    """umlh_eval_rows: per-row CE and top-1 flag of a whole slab in one launch == the oracle's per-row values; the
    per-batch means validate() forms from them == O.validate (finetune.py:291-315) for a batch size that divides
    nothing (37)."""
    import umlh
    rng = np.random.default_rng(5)
    d, C, n = 128, 100, 333
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rng.integers(0, C, n)
    w = rng.standard_normal((C, d)).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    e = umlh.HeadEngine(d, d, C, max_rows_img=512, max_rows_txt=32, precision=precision, device=DEV)
    e.w_head.copy_(torch.from_numpy(w))
    e.scales.fill_(30.0)
    T = lambda a, t: torch.as_tensor(np.ascontiguousarray(a)).to(DEV, t).contiguous()
    st = e.eval_rows(umlh.RowBatch(T(x, torch.float32), T(y, torch.int64))).cpu().numpy()
    rnd = (lambda a: torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()) if precision == "bf16" else (lambda a: a)
    z = (rnd(x) @ rnd(w).T) * 30.0
    zmax = z.max(1, keepdims=True)
    ce = (np.log(np.exp(z - zmax).sum(1)) + zmax[:, 0] - z[np.arange(n), y])
    tol = 2e-3 if precision == "bf16" else 1e-4
    np.testing.assert_allclose(st[:, 0], ce, atol=tol)
    assert (st[:, 1] == (z.argmax(1) == y)).mean() > (0.99 if precision == "bf16" else 0.999999)
    if precision == "fp32":
        bs = 37
        loss = np.mean([st[s:s + bs, 0].mean() for s in range(0, n, bs)])
        ol, oa = O.validate(O.HeadState(w, None, 30.0, 30.0, False), x, y, bs, rng_draw=False)
        assert abs(loss - ol) < 1e-4 and abs(st[:, 1].mean() - oa) < 1e-6


def test_gradient_diagnostics_are_bitwise_reproducible():
    """The per-modality gradient diagnostics (finetune.py:190-191,203-206) are reduced in a fixed order (per-workgroup
    partials, summed by the workgroup that takes the last ticket): the same step from the same state gives bit-identical
    dot / norms / agreement, whatever order the workgroups finished in.  C x d = 512 000 elements = 500 workgroups."""
    rng = np.random.default_rng(21)
    st, xi, yi, xt, yt, ii, ti = _random_case(rng, 512, 512, 1000, 600, 600, 256, 256, False, False, 100.0)
    rows = []
    for rep in range(6):
        e = _engine(st, wd=0.01, max_img=256, max_txt=256)
        e.enable_diagnostics(True)
        out = torch.zeros(12, dtype=torch.float32, device=DEV)
        for k in range(2):
            e.train_step(_rb(xi, yi, ii), _rb(xt, yt, ti), lr=1e-3, step=k + 1, alpha=0.7, scalars_out=out)
        torch.cuda.synchronize()
        rows.append(out.cpu().numpy().copy())
    assert np.isfinite(rows[0]).all() and abs(rows[0][8:]).sum() > 0
    for r in rows[1:]:
        assert np.array_equal(r, rows[0])


@pytest.mark.parametrize("tag,kind", [("clip_d64_c10_adamw", "clip"), ("uml_d96_c37_sgd", "uml"), ("clip_d512_c100_adam", "clip"),
                                      ("mlp_d48_t64_c10_adamw", "uml")])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_head_with_bias_matches_reference(tag, kind, precision):
    """bias=True heads (engine/models/head.py:65,68,122) against the reference's own classes, autograd and optimizers
    (tests/golden/bias_heads.npz, oracle/make_golden_bias.py): logits x W^T + b (1e-4 in fp32), per-step losses of a
    6-8 step trajectory through build_optimizer / build_lr_scheduler, final weight and bias.  The kernels run the bias as
    column d of a packed [weight | bias | padding] head over rows [x | 1 | 0...]."""
    import umlh
    from engine.models.head import UML, UMLClip
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    g = load_golden("bias_heads")
    d, C, Bi, Bt, steps, alpha, wd, oid, t_in = g[f"{tag}::cfg"]
    d, C, steps, t_in = int(d), int(C), int(steps), int(t_in)
    optim = {2: "adamw", 1: "adam", 0: "sgd"}[int(oid)]
    if precision == "bf16" and optim == "sgd":
        pytest.skip("one bf16 trajectory per optimizer family is enough")
    m = (UMLClip(d, C, logit_scale_init=4.60517, bias=True) if kind == "clip" else UML(d, t_in, C, bias=True)).to(DEV)
    sd = {"head.weight": torch.as_tensor(g[f"{tag}::w0"]), "head.bias": torch.as_tensor(g[f"{tag}::b0"])}
    if t_in:                                                               # 2-layer head: img_proj carries a bias too
        sd.update({"img_proj.weight": torch.as_tensor(g[f"{tag}::pw0"]), "img_proj.bias": torch.as_tensor(g[f"{tag}::pb0"])})
    m.load_state_dict(sd)
    xi, yi, xt, yt = (_t(g[f"{tag}::{k}"], dt) for k, dt in (("xi", torch.float32), ("yi", torch.int64), ("xt", torch.float32), ("yt", torch.int64)))
    ii, ti = g[f"{tag}::idx_i"], g[f"{tag}::idx_t"]
    li, lt = m(xi[_t(ii[0], torch.int64)], xt[_t(ti[0], torch.int64)])
    np.testing.assert_allclose(li.cpu().numpy(), g[f"{tag}::logits_img0"], atol=1e-4, rtol=1e-5)
    np.testing.assert_allclose(lt.cpu().numpy(), g[f"{tag}::logits_txt0"], atol=1e-4, rtol=1e-5)
    opt = build_optimizer(m.parameters(), optim, 1e-3, float(wd))
    sch = build_lr_scheduler(opt, "cosine", 2, 100, warmup_type="linear", warmup_lr=1e-5)
    eng = m.fused_engine(opt, 64, 64, precision=precision)
    sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
    for k in range(steps):
        assert abs(opt.param_groups[0]["lr"] - float(g[f"{tag}::lrs"][k])) < 1e-12
        eng.train_step(umlh.RowBatch(xi, yi, _t(ii[k], torch.int64)), umlh.RowBatch(xt, yt, _t(ti[k], torch.int64)),
                       lr=opt.param_groups[0]["lr"], step=k + 1, alpha=float(alpha), scalars_out=sc[k])
        opt.step_count += 1
        sch.step()
    torch.cuda.synchronize()
    got = sc.cpu().numpy()[:, [umlh.S_LOSS_IMG, umlh.S_LOSS_TXT]]
    tol = 1e-4 if precision == "fp32" else 0.05
    np.testing.assert_allclose(got, g[f"{tag}::losses"], atol=tol * max(1.0, np.abs(g[f"{tag}::losses"]).max() if precision == "bf16" else 1.0), rtol=1e-5 if precision == "fp32" else 5e-3)
    w1, b1 = m.head.weight.detach().cpu().numpy(), m.head.bias.detach().cpu().numpy()
    lim = 2 * float(np.sum(g[f"{tag}::lrs"])) + 1e-6                       # Adam's first steps move a weight by ~lr * sign(g)
    pairs = [(w1, g[f"{tag}::w1"]), (b1, g[f"{tag}::b1"])]
    if t_in:
        pairs += [(m.img_proj.weight.detach().cpu().numpy(), g[f"{tag}::pw1"]), (m.img_proj.bias.detach().cpu().numpy(), g[f"{tag}::pb1"])]
        row = m._packed_proj[t_in].cpu().numpy()                           # the constant row never moves
        assert row[d] == 1.0 and np.abs(row).sum() == 1.0
        feats = m.extract_features(xi[:5])
        ref_h = xi[:5].cpu().numpy() @ g[f"{tag}::pw1"].T + g[f"{tag}::pb1"]
        assert tuple(feats.shape) == (5, t_in)
        if precision == "fp32":
            np.testing.assert_allclose(feats.cpu().numpy(), ref_h, atol=5e-3)
    for got_p, ref in pairs:
        diff = np.abs(got_p - ref)
        if precision == "fp32":
            assert diff.max() <= lim and (diff > 2e-6 + 1e-4 * np.abs(ref)).mean() < 5e-3
        else:
            assert diff.max() <= lim
    assert float(m._packed[:, (t_in or d) + 1:].abs().max()) == 0.0        # the padding columns never move
    # the optimizer's state is visible under the reference's parameter names
    st = opt.state[m.head.bias]
    assert any(v.shape == (C,) for v in st.values())


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["uml_d96_c37_adamw_learn", "mlp_d48_t64_c10_adamw_learn", "uml_d128_c20_sgd_fixed"])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_bias_head_with_learnable_temperature_and_diagnostics_matches_reference(tag, precision):
    """Round 3: bias=True TOGETHER with learnable_temp=True (head.py:65-70) and the per-step gradient diagnostics of a bias head
    (finetune.py:190-191,203-206: autograd.grad with respect to head.weight ONLY, so the bias column and the padding of the packed
    [weight | bias | 0...] rows must not enter the dot product, the norms or the sign-agreement rate).  Golden vectors from the
    reference's own UML class, autograd, build_optimizer and build_lr_scheduler (tests/golden/bias_heads_r3.npz,
    oracle/make_golden_bias_r3.py): per-step losses, diagnostics, logit scales; final weights and biases."""
    import umlh
    from engine.models.head import UML
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    g = load_golden("bias_heads_r3")
    d, C, Bi, Bt, steps, alpha, wd, oid, t_in, learn = g[f"{tag}::cfg"]
    d, C, steps, t_in, learn = int(d), int(C), int(steps), int(t_in), bool(learn)
    optim = {2: "adamw", 1: "adam", 0: "sgd"}[int(oid)]
    m = UML(d, t_in, C, bias=True, learnable_temp=learn).to(DEV)
    sd = {"head.weight": torch.as_tensor(g[f"{tag}::w0"]), "head.bias": torch.as_tensor(g[f"{tag}::b0"])}
    if t_in:
        sd.update({"img_proj.weight": torch.as_tensor(g[f"{tag}::pw0"]), "img_proj.bias": torch.as_tensor(g[f"{tag}::pb0"])})
    if learn:
        sd.update({"img_scale": torch.tensor(2.0), "txt_scale": torch.tensor(1.5)})
    m.load_state_dict(sd)
    xi, yi, xt, yt = (_t(g[f"{tag}::{k}"], dt) for k, dt in (("xi", torch.float32), ("yi", torch.int64), ("xt", torch.float32), ("yt", torch.int64)))
    ii, ti = g[f"{tag}::idx_i"], g[f"{tag}::idx_t"]
    opt = build_optimizer(m.parameters(), optim, 1e-3, float(wd))
    sch = build_lr_scheduler(opt, "cosine", 2, 100, warmup_type="linear", warmup_lr=1e-5)
    eng = m.fused_engine(opt, 64, 64, precision=precision)
    eng.enable_diagnostics(True)
    sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
    scales = []
    for k in range(steps):
        assert abs(opt.param_groups[0]["lr"] - float(g[f"{tag}::lrs"][k])) < 1e-12
        eng.train_step(umlh.RowBatch(xi, yi, _t(ii[k], torch.int64)), umlh.RowBatch(xt, yt, _t(ti[k], torch.int64)),
                       lr=opt.param_groups[0]["lr"], step=k + 1, alpha=float(alpha), scalars_out=sc[k])
        opt.step_count += 1
        sch.step()
        scales.append([float(m.img_scale.detach()), float(m.txt_scale.detach())])
    torch.cuda.synchronize()
    eng.check_status()
    rows = sc.cpu()
    got = rows.numpy()[:, [umlh.S_LOSS_IMG, umlh.S_LOSS_TXT]]
    f32 = precision == "fp32"
    np.testing.assert_allclose(got, g[f"{tag}::losses"], atol=1e-4 if f32 else 0.05, rtol=1e-5 if f32 else 5e-3)
    ref_d = g[f"{tag}::diag"]                                   # [steps][cos, |g_img|, |g_txt|, agreement]
    for k in range(steps):
        dg = umlh.grad_diagnostics(rows[k], m.head.weight.numel(), 1, 1)
        tol = 2e-4 if f32 else 3e-2
        assert abs(dg["grad_direction_sim"] - ref_d[k, 0]) < tol + (0 if f32 else 0.05 * abs(ref_d[k, 0]))
        np.testing.assert_allclose([dg["img_grad_norm"], dg["txt_grad_norm"]], ref_d[k, 1:3], rtol=1e-4 if f32 else 3e-2)
        # sign flips of near-zero entries: a handful of the C * d elements in fp32, a few percent with bf16 operands
        assert abs(dg["grad_agreement_rate"] - ref_d[k, 3]) < (2e-3 if f32 else 4e-2)
    if learn:
        np.testing.assert_allclose(np.asarray(scales), g[f"{tag}::scales"], atol=2e-6 if f32 else 2e-3)
    lim = 2 * float(np.sum(g[f"{tag}::lrs"])) + 1e-6
    pairs = [(m.head.weight.detach().cpu().numpy(), g[f"{tag}::w1"]), (m.head.bias.detach().cpu().numpy(), g[f"{tag}::b1"])]
    if t_in:
        pairs += [(m.img_proj.weight.detach().cpu().numpy(), g[f"{tag}::pw1"]), (m.img_proj.bias.detach().cpu().numpy(), g[f"{tag}::pb1"])]
    for got_p, ref in pairs:
        diff = np.abs(got_p - ref)
        assert diff.max() <= lim
        if f32:
            assert (diff > 2e-6 + 1e-4 * np.abs(ref)).mean() < 5e-3
    assert float(m._packed[:, (t_in or d) + 1:].abs().max()) == 0.0        # the padding columns never move


_F32_FWD_SCRIPT = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import umlh
rng = np.random.default_rng(17)
out = []
vals = []
for (d, C, n, bi, bt, proj) in ((512, 1000, 3000, 1024, 700, 0), (64, 100, 900, 300, 257, 0), (96, 37, 500, 40, 90, 0), (128, 300, 800, 256, 100, 64)):
    d_img = proj if proj else d
    x = rng.standard_normal((n, d_img)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    xt = rng.standard_normal((n, d)).astype(np.float32); xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    y = rng.integers(0, C, n)
    w = rng.standard_normal((C, d)).astype(np.float32); w /= np.linalg.norm(w, axis=1, keepdims=True)
    e = umlh.HeadEngine(d_img, d, C, has_proj=bool(proj), optimizer="adamw", weight_decay=0.01, max_rows_img=1024, max_rows_txt=1024,
                        precision="fp32", device="cuda:0")
    e.w_head.copy_(torch.from_numpy(w)); e.scales.fill_(50.0)
    if proj: e.w_proj.normal_(0, 0.1, generator=torch.Generator(device="cuda").manual_seed(3))
    X = torch.from_numpy(x).cuda(); XT = torch.from_numpy(xt).cuda(); Y = torch.from_numpy(y).cuda()
    g = torch.Generator().manual_seed(2)
    sc = torch.zeros(8, umlh.N_SCALARS, device="cuda")
    for k in range(8):
        ii = torch.randint(0, n, (bi if k % 3 else bi // 2 + 1,), generator=g).cuda(); ti = torch.randint(0, n, (bt,), generator=g).cuda()
        e.train_step(umlh.RowBatch(X, Y, ii), umlh.RowBatch(XT, Y, ti) if k != 5 else None, lr=1e-2, step=k + 1, scalars_out=sc[k])
    ev = e.eval_batch(umlh.RowBatch(X, Y, torch.arange(0, min(n, 1024)).cuda()))
    torch.cuda.synchronize()
    out.append(hashlib.sha256(e.w_head.cpu().numpy().tobytes() + e.v_head.cpu().numpy().tobytes() + sc.cpu().numpy().tobytes()
                              + np.asarray([float(v) for v in ev], dtype=np.float64).tobytes()).hexdigest()[:16])
    vals.append(np.concatenate([e.w_head.cpu().numpy().ravel(), sc.cpu().numpy()[:, :8].ravel(), np.asarray([float(v) for v in ev], dtype=np.float32)]))
print("DIGEST", *out)
if len(sys.argv) > 2:
    np.save(sys.argv[2], np.concatenate(vals))
"""


@pytest.mark.gpu
def test_fp32_forward_streamed_weights_equals_lds_staged_forward_bit_for_bit(tmp_path):
    """Round 3: fwd_ce_f32 streams W from a fragment-major fp32 shadow (kept current by the update kernel) instead of staging it
    through LDS chunk by chunk.  Same products in the same order: eight AdamW steps (changing batches, a text-less step, the
    2-layer head, a K that needs several X blocks' worth of chunks) and an evaluation end in bit-identical weights, moments and
    scalars under UMLH_F32_FWD=1 (LDS-staged kernel) and the default.  The switch is read once per process: child processes."""
    import os, subprocess, sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unpaired-multimodal-learning_amd")
    script = tmp_path / "f32_fwd_case.py"
    script.write_text(_F32_FWD_SCRIPT)
    digests = {}
    for mode in ("1", "2"):
        env = dict(os.environ, UMLH_F32_FWD=mode, UMLH_F32_X3="0")      # (both on the fp32 MFMA: the x3 products are a different rounding)
        r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        digests[mode] = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1]
    assert digests["1"] == digests["2"]


@pytest.mark.gpu
def test_fp32_mode_on_the_bf16_matrix_pipe_matches_the_fp32_mfma_path(tmp_path):
    """Round 3: fp32 mode forms its products on v_mfma_f32_32x32x16_bf16 from three-way bf16 splits of both operands (six piece
    products per product, fp32 accumulation: umlh_f32_x3 in csrc/umlh_common.h) -- every piece product is exact, what is dropped is
    <= 2^-23 of a product, i.e. one fp32 rounding.  The same eight AdamW steps + evaluation as the test above under UMLH_F32_X3=1
    (default) and =0 (fp32 MFMA): step scalars (losses at logit scale 50, accuracies) within 2e-5 / exact, weights after 8 steps
    within 1e-5 (Adam's normalised steps amplify last-bit gradient differences by at most lr per step)."""
    import os, subprocess, sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unpaired-multimodal-learning_amd")
    script = tmp_path / "f32_fwd_case.py"
    script.write_text(_F32_FWD_SCRIPT)
    vals = {}
    for x3 in ("0", "1"):
        env = dict(os.environ, UMLH_F32_X3=x3)
        out = tmp_path / f"vals_{x3}.npy"
        r = subprocess.run([sys.executable, str(script), root, str(out)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        vals[x3] = np.load(out)
    a, b = vals["0"], vals["1"]
    assert a.shape == b.shape and np.isfinite(b).all()
    assert not np.array_equal(a, b)                                  # (the switch did switch)
    diff = np.abs(b - a)
    # (a weight whose gradient is at the rounding-noise level can take a different Adam step: a handful of elements, bounded by 8 x lr)
    assert (diff > 2e-5 + 2e-5 * np.abs(a)).mean() < 1e-3 and diff.max() < 0.1, (float((diff > 2e-5 + 2e-5 * np.abs(a)).mean()), float(diff.max()))
