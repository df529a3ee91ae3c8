"""GPU parity at the BASELINE configurations' real shapes:
  * cfg1: the reference's own cfg1-shaped finetune.train() run (tests/golden/train_cfg1.npz) replayed through this
    build's finetune.train() in fp32 (losses 1e-4, same best iteration, final top-1 +-0.1 pp) and in bf16 mode
    (final top-1 +-0.1 pp of the REFERENCE's);
  * cfg2: bf16 grad step at 4096 + 4096 rows against the oracle; bf16-vs-fp32 training accuracy at the cfg2 shape +-0.1 pp;
  * cfg3: full-size 2-layer step (1024 -> 3200, C = 1000, 4096 + 4096 rows) against the oracle, fp32 and bf16;
  * cfg5: d = 768 16-shot farm shapes, C in {10, 37, 47, 397, 1000}, batch 32, fp32 and bf16.
Tolerances: fp32 logits / loss 1e-4 absolute (north_star); bf16 against the oracle on bf16-rounded operands."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import uml_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _T(a, t=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV, t).contiguous()


def _r16(a):
    return torch.as_tensor(a).to(torch.bfloat16).to(torch.float32).numpy()


def _rb(x, y, idx=None):
    import umlh
    return umlh.RowBatch(_T(x), _T(y, torch.int64), None if idx is None else _T(idx, torch.int64))


# --------------------------------------------------------------------------------------------- cfg1
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cfg1_train_replays_reference_run(precision):
    import finetune as ft
    import umlh
    from engine.datasets.utils import FeatureLoader, FeatureTable, TextTensorDataset
    from engine.models.head import UMLClip
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    from engine.tools.utils import set_random_seed
    from oracle import fixtures_cfg1 as FX
    g = load_golden("train_cfg1")
    inp = {k: torch.from_numpy(v) for k, v in FX.cfg1_inputs().items()}
    set_random_seed(FX.SEED)
    text_ds = TextTensorDataset(inp["x_txt"], inp["y_txt"], torch.zeros(len(inp["y_txt"]), dtype=torch.long))
    model = UMLClip(FX.D, FX.C, logit_scale_init=FX.SCALE_LOG, bias=False)
    np.testing.assert_array_equal(model.head.weight.detach().numpy(), g["w_head_init"])
    model.to(DEV)
    model.zero_shot_init(text_ds)
    optimizer = build_optimizer(model.parameters(), "adamw", FX.LR, FX.WD)
    scheduler = build_lr_scheduler(optimizer, "cosine", 50, 12800, warmup_type="linear", warmup_lr=1e-5)
    B = FX.BATCH
    image_loader = FeatureLoader(FeatureTable(inp["x_img"], inp["y_img"], DEV), B, shuffle=True, kind="image")
    text_loader = FeatureLoader(FeatureTable(text_ds.input_tensor, text_ds.label_tensor, DEV), B, shuffle=True, kind="text")
    val_loader = FeatureLoader(FeatureTable(inp["x_val"], inp["y_val"], DEV), B, shuffle=False)
    test_loader = FeatureLoader(FeatureTable(inp["x_test"], inp["y_test"], DEV), B, shuffle=False)
    out = ft.train(model, image_loader, text_loader, val_loader, test_loader, optimizer, scheduler, device=DEV,
                   max_iters=FX.MAX_ITERS, alpha=FX.ALPHA, eval_freq=FX.EVAL_FREQ, patience=FX.PATIENCE, precision=precision)
    test_loss, test_acc = ft.validate(model, test_loader, device=DEV)
    sc = out["train_scalars"].numpy()
    ce = g["train_ce"]
    n = int(g["n_steps"])
    print(f"cfg1 {precision}: steps {sc.shape[0]} (ref {n}) best iter {out['iter']} (ref {int(g['best_iter'])}) "
          f"test top-1 {test_acc:.4f} (ref {float(g['test_acc']):.4f}) max|dloss| "
          f"{np.abs(sc[:min(n, len(sc)), umlh.S_LOSS_IMG] - ce[0::2][:len(sc)]).max():.2e}")
    assert abs(test_acc - float(g["test_acc"])) <= 1e-3                      # north_star: final top-1 within +-0.1 pp
    if precision == "fp32":
        assert sc.shape[0] == n
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_IMG], ce[0::2], atol=1e-4)
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_TXT], ce[1::2], atol=1e-4)
        assert out["iter"] == int(g["best_iter"])
        assert abs(out["val_acc"] - float(g["best_val_acc"])) < 1e-6 and abs(out["val_loss"] - float(g["best_val_loss"])) < 1e-4
        np.testing.assert_allclose(out["model"]["head.weight"].numpy(), g["w_head_best"], atol=2e-5, rtol=1e-3)
        assert abs(test_loss - float(g["test_loss"])) < 1e-4
    else:
        m = min(n, sc.shape[0])
        np.testing.assert_allclose(sc[:m, umlh.S_LOSS_IMG], ce[0::2][:m], atol=3e-2)
        np.testing.assert_allclose(sc[:m, umlh.S_LOSS_TXT], ce[1::2][:m], atol=3e-2)
        assert abs(out["val_acc"] - float(g["best_val_acc"])) <= 2.5e-3 + 1e-9      # 400 validation rows: one row


# --------------------------------------------------------------------------------------------- cfg2
def _cfg2_case(rng, n_img, n_txt, d=512, C=1000):
    xi = rng.standard_normal((n_img, d)).astype(np.float32)
    xt = rng.standard_normal((n_txt, d)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    w = rng.standard_normal((C, d)).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    return xi, rng.integers(0, C, n_img), xt, rng.integers(0, C, n_txt), w


def test_bf16_full_size_cfg2_grad_step_against_oracle():
    """BASELINE config 2 at its benchmarked size in the benchmarked (bf16) mode: 4096 image + 4096 text rows gathered
    from larger tables, d = 512, C = 1000, scale 100 -- loss / accuracy / dW against the oracle fed the same
    bf16-rounded operands, loosely against the exact oracle, plus the CE property sum_c dW[c,:] = 0."""
    import umlh
    rng = np.random.default_rng(2)
    d, C, B = 512, 1000, 4096
    xi, yi, xt, yt, w = _cfg2_case(rng, 3 * B, 2 * B)
    ii, ti = rng.permutation(3 * B)[:B], rng.permutation(2 * B)[:B]
    e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B, precision="bf16", device=DEV)
    e.w_head.copy_(_T(w)); e.scales.fill_(100.0)
    flat = e.grad_step(_rb(xi, yi, ii), _rb(xt, yt, ti), alpha=1.0)
    torch.cuda.synchronize()
    f = flat.cpu().numpy()
    gh, sc = f[:C * d].reshape(C, d), f[C * d + 2:]
    so = O.step_grads(O.HeadState(_r16(w), None, 100.0, 100.0, False), _r16(xi[ii]), yi[ii], _r16(xt[ti]), yt[ti], 1.0)
    assert abs(sc[umlh.S_LOSS_IMG] - so.loss_img) < 2e-3 and abs(sc[umlh.S_LOSS_TXT] - so.loss_txt) < 2e-3
    assert abs(sc[umlh.S_ACC_IMG] - so.acc_img) <= 2.0 / B and abs(sc[umlh.S_ACC_TXT] - so.acc_txt) <= 2.0 / B
    s = np.abs(so.grads["w_head"]).max()
    np.testing.assert_allclose(gh, so.grads["w_head"], atol=8e-3 * s, rtol=2e-2)
    assert np.abs(gh.sum(axis=0)).max() < 2e-2 * np.abs(gh).sum(axis=0).max()
    ex = O.step_grads(O.HeadState(w, None, 100.0, 100.0, False), xi[ii], yi[ii], xt[ti], yt[ti], 1.0)
    assert abs(sc[umlh.S_LOSS_IMG] - ex.loss_img) < 5e-2 and abs(sc[umlh.S_LOSS_TXT] - ex.loss_txt) < 5e-2
    assert np.abs(gh - ex.grads["w_head"]).max() < 5e-2 * np.abs(ex.grads["w_head"]).max()


def test_bf16_training_accuracy_parity_cfg2_shape():
    """Same seed, same batches, the cfg2 shape (d = 512, C = 1000, 4096 + 4096 rows per step, scale 100, zero-shot init,
    AdamW): 200 steps in bf16 mode vs the fp32 parity mode; final top-1 on 50 000 held-out rows within +-0.1 pp
    (north_star) and the loss curves agree."""
    import umlh
    rng = np.random.default_rng(17)
    d, C, n, n_txt, n_eval, B, steps = 512, 1000, 65536, 29940, 50000, 4096, 200
    proto = rng.standard_normal((C, d)).astype(np.float32)
    proto_t = proto + 0.7 * rng.standard_normal((C, d)).astype(np.float32)

    def draw(m, p, noise):
        y = rng.integers(0, C, m)
        x = p[y] + noise * rng.standard_normal((m, d)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32), y
    xi, yi = draw(n, proto, 6.5)
    xt, yt = draw(n_txt, proto_t, 4.0)
    xe, ye = draw(n_eval, proto, 6.5)
    w0 = O.zero_shot_weights(xt, yt, C)
    Xi, Yi, Xt, Yt = _T(xi), _T(yi, torch.int64), _T(xt), _T(yt, torch.int64)
    tabs = {"fp32": ((Xi, Yi), (Xt, Yt)), "bf16": ((Xi, Yi, umlh.to_bf16(Xi)), (Xt, Yt, umlh.to_bf16(Xt)))}
    g = torch.Generator().manual_seed(5)
    bi = [torch.randint(0, n, (B,), generator=g).to(DEV) for _ in range(steps)]
    bt = [torch.randint(0, n_txt, (B,), generator=g).to(DEV) for _ in range(steps)]
    lrs = [1e-3 * min(1.0, (k + 1) / 50) for k in range(steps)]
    accs, curves = {}, {}
    Xe, Ye = _T(xe), _T(ye, torch.int64)
    for prec in ("fp32", "bf16"):
        e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B, precision=prec, device=DEV)
        e.w_head.copy_(_T(w0)); e.scales.fill_(100.0)
        sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
        e.train_steps(tabs[prec][0], bi, tabs[prec][1], bt, lrs, first_step=1, scalars_out=sc)
        torch.cuda.synchronize()
        curves[prec] = sc.cpu().numpy()
        ev = umlh.HeadEngine(d, d, C, max_rows_img=4096, max_rows_txt=32, precision="fp32", device=DEV)
        ev.w_head.copy_(e.w_head); ev.scales.fill_(100.0)
        correct = 0.0
        for s0 in range(0, n_eval, 4096):
            st = ev.eval_rows(umlh.RowBatch(Xe[s0:s0 + 4096], Ye[s0:s0 + 4096]))
            correct += float(st[:, 1].sum())
        accs[prec] = correct / n_eval
    print(f"cfg2-shape top-1 fp32 {accs['fp32']:.4f}  bf16 {accs['bf16']:.4f}  diff {100 * (accs['bf16'] - accs['fp32']):+.3f} pp; "
          f"final loss img fp32 {curves['fp32'][-1, 0]:.4f} bf16 {curves['bf16'][-1, 0]:.4f}")
    assert 0.2 < accs["fp32"] < 0.98                                           # a regime where 0.1 pp means something
    assert abs(accs["bf16"] - accs["fp32"]) <= 1e-3
    np.testing.assert_allclose(curves["bf16"][:, umlh.S_LOSS_IMG], curves["fp32"][:, umlh.S_LOSS_IMG], atol=3e-2)
    np.testing.assert_allclose(curves["bf16"][:, umlh.S_LOSS_TXT], curves["fp32"][:, umlh.S_LOSS_TXT], atol=3e-2)


# --------------------------------------------------------------------------------------------- cfg3
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cfg3_full_size_step_against_oracle(precision):
    """BASELINE config 3 at its real size: DINOv2-L/14 rows (d_v = 1024) -> img_proj -> 3200 (OpenLLaMA width), C = 1000,
    4096 image + 4096 text rows, logit scale 1 -- losses, dW_head, dW_proj against the oracle (fp32: exact operands,
    1e-4 on the losses; bf16: the oracle on bf16-rounded operands)."""
    import umlh
    rng = np.random.default_rng(33)
    dv, dt, C, B = 1024, 3200, 1000, 4096
    n_i, n_t = B + 512, B + 256
    xi = rng.standard_normal((n_i, dv)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt = rng.standard_normal((n_t, dt)).astype(np.float32)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    yi, yt = rng.integers(0, C, n_i), rng.integers(0, C, n_t)
    wp = (rng.standard_normal((dt, dv)) / np.sqrt(dv)).astype(np.float32)
    wh = rng.standard_normal((C, dt)).astype(np.float32)          # |logit| ~ N(0, 1) at scale 1
    ii, ti = rng.permutation(n_i)[:B], rng.permutation(n_t)[:B]
    e = umlh.HeadEngine(dv, dt, C, has_proj=True, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B,
                        precision=precision, device=DEV)
    e.w_head.copy_(_T(wh)); e.w_proj.copy_(_T(wp)); e.scales.fill_(1.0)
    flat = e.grad_step(_rb(xi, yi, ii), _rb(xt, yt, ti), alpha=1.0)
    torch.cuda.synchronize()
    f = flat.cpu().numpy()
    nh, npj = C * dt, dt * dv
    gh, gp, sc = f[:nh].reshape(C, dt), f[nh:nh + npj].reshape(dt, dv), f[nh + npj + 2:]
    R = _r16 if precision == "bf16" else (lambda a: a)
    so = O.step_grads(O.HeadState(R(wh), R(wp), 1.0, 1.0, False), R(xi[ii]), yi[ii], R(xt[ti]), yt[ti], 1.0)
    tol_l, tol_g = (1e-4, 2e-4) if precision == "fp32" else (2e-2, 4e-2)
    print(f"cfg3 {precision}: loss img {sc[umlh.S_LOSS_IMG]:.5f} (oracle {so.loss_img:.5f}) txt {sc[umlh.S_LOSS_TXT]:.5f} ({so.loss_txt:.5f})")
    assert abs(sc[umlh.S_LOSS_IMG] - so.loss_img) < tol_l * max(1.0, so.loss_img if precision == "bf16" else 1.0)
    assert abs(sc[umlh.S_LOSS_TXT] - so.loss_txt) < tol_l * max(1.0, so.loss_txt if precision == "bf16" else 1.0)
    for got, key in ((gh, "w_head"), (gp, "w_proj")):
        ref = so.grads[key]
        assert np.abs(got - ref).max() < tol_g * np.abs(ref).max(), key
    assert np.abs(gh.sum(axis=0)).max() < 1e-2 * np.abs(gh).sum(axis=0).max()


# --------------------------------------------------------------------------------------------- cfg5
@pytest.mark.parametrize("C", [10, 37, 47, 397, 1000])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cfg5_d768_few_shot_step_against_oracle(C, precision):
    """BASELINE config 5 shapes (CLIP ViT-L/14 width d = 768, 16-shot: N_img = 16 C, batch 32, scale 100): logits and
    one AdamW step at every class count of the 11-dataset sweep's distinct tile regimes."""
    import umlh
    rng = np.random.default_rng(768 + C)
    d, B = 768, 32
    n_i, n_t = 16 * C, 30 * C
    xi = rng.standard_normal((n_i, d)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt = rng.standard_normal((n_t, d)).astype(np.float32)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    yi, yt = np.repeat(np.arange(C), 16), np.repeat(np.arange(C), 30)
    w = O.zero_shot_weights(xt, yt, C)
    ii, ti = rng.permutation(n_i)[:B], rng.permutation(n_t)[:B]
    e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B, precision=precision, device=DEV)
    e.w_head.copy_(_T(w)); e.scales.fill_(100.0)
    R = _r16 if precision == "bf16" else (lambda a: a)
    st = O.HeadState(R(w), None, 100.0, 100.0, False)
    so = O.step_grads(st, R(xi[ii]), yi[ii], R(xt[ti]), yt[ti], 1.0)
    if precision == "fp32":
        np.testing.assert_allclose(e.logits(_rb(xi, yi, ii), 0).cpu().numpy(), so.zi, atol=1e-4, rtol=0)
        np.testing.assert_allclose(e.logits(_rb(xt, yt, ti), 1).cpu().numpy(), so.zt, atol=1e-4, rtol=0)
    flat = e.grad_step(_rb(xi, yi, ii), _rb(xt, yt, ti), alpha=1.0)
    torch.cuda.synchronize()
    f = flat.cpu().numpy()
    gh, sc = f[:C * d].reshape(C, d), f[C * d + 2:]
    tol_l = 1e-4 if precision == "fp32" else 2e-3
    assert abs(sc[umlh.S_LOSS_IMG] - so.loss_img) < tol_l and abs(sc[umlh.S_LOSS_TXT] - so.loss_txt) < tol_l
    assert abs(sc[umlh.S_ACC_IMG] - so.acc_img) < 1e-6 + (0 if precision == "fp32" else 1.0 / B)
    s = np.abs(so.grads["w_head"]).max()
    if precision == "fp32":
        np.testing.assert_allclose(gh, so.grads["w_head"], atol=3e-5 * s, rtol=2e-4)
    else:
        np.testing.assert_allclose(gh, so.grads["w_head"], atol=8e-3 * s, rtol=2e-2)
    # one fused AdamW step == oracle update (fp32 mode; Adam's sign flips at |g| ~ eps aside)
    if precision == "fp32":
        e.train_step(_rb(xi, yi, ii), _rb(xt, yt, ti), lr=1e-3, step=1, alpha=1.0)
        O.optimizer_step(st, so.grads, O.OptState("adamw", 0.01), 1e-3)
        diff = np.abs(e.w_head.cpu().numpy() - st.w_head)
        assert (diff > 2e-6 + 1e-5 * np.abs(st.w_head)).mean() < 1e-3 and diff.max() <= 2e-3 + 1e-6
