"""GPU: the single-launch micro step (umlh_kernels_micro.hip) -- the reference's own operating point (batch 8 / 32 / 64,
engine/optimizer/default.py:3-45) -- against the CPU oracle, against the general three-kernel path, and the grouped
multi-head launch (finetune.py:406-448 sweep) against isolated runs, bit for bit."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import uml_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _T(a, t=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV, t).contiguous()


def _tables(rng, d, C, n_img, n_txt):
    xi = rng.standard_normal((n_img, d)).astype(np.float32)
    xt = rng.standard_normal((n_txt, d)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    return xi, rng.integers(0, C, n_img), xt, rng.integers(0, C, n_txt)


def _run(monkeypatch, micro, d, C, w0, scales, learn, optim, wd, xi, yi, xt, yt, bi, bt, lrs, alpha):
    import umlh
    monkeypatch.setenv("UMLH_MICRO", "1" if micro else "0")
    e = umlh.HeadEngine(d, d, C, learnable_temp=learn, optimizer=optim, weight_decay=wd, max_rows_img=64, max_rows_txt=64, device=DEV)
    e.w_head.copy_(_T(w0)); e.scales.copy_(torch.tensor(scales, dtype=torch.float32))
    n = len(lrs)
    sc = torch.zeros(n, umlh.N_SCALARS, device=DEV)
    ti = (_T(xi), _T(yi, torch.int64)) if bi is not None else None
    tt = (_T(xt), _T(yt, torch.int64)) if bt is not None else None
    e.train_steps(ti, [_T(b, torch.int64) for b in bi] if bi is not None else None,
                  tt, [_T(b, torch.int64) for b in bt] if bt is not None else None, lrs, first_step=1, alpha=alpha, scalars_out=sc)
    torch.cuda.synchronize()
    assert e.micro_status() == 0
    assert (e.micro_launches() > 0) == micro
    return e, sc.cpu().numpy()


CASES = [
    # d, C, rows_img, rows_txt, optimizer, learnable_temp, scale
    (512, 100, 32, 32, "adamw", False, 100.0),     # cfg1: clip_linear point
    (768, 1000, 32, 32, "adamw", False, 100.0),    # cfg5 ImageNet head, 63 class slices
    (768, 37, 32, 24, "adamw", False, 100.0),      # ragged last text batch of an epoch
    (512, 397, 8, 8, "adamw", True, 3.0),          # 'linear' grid: batch 8, learnable temperature
    (96, 10, 64, 0, "sgd", False, 20.0),           # image-only, SGD (finetune.py:373-376), 3 x 32-wide chunks
    (32, 10, 20, 30, "adam", True, 1.5),           # toy width of the reference-run goldens
    (1024, 47, 16, 48, "adamw", False, 50.0),      # d = 1024: two-buffer ring
    (640, 101, 32, 32, "adamw", False, 100.0),     # RN50x4 width
]


@pytest.mark.parametrize("d,C,ri,rt,optim,learn,scale", CASES)
def test_micro_steps_match_oracle_and_general_path(d, C, ri, rt, optim, learn, scale, monkeypatch):
    import umlh
    rng = np.random.default_rng(d + 7 * C + ri)
    n_img, n_txt, steps = 300, 260, 12
    xi, yi, xt, yt = _tables(rng, d, C, n_img, n_txt)
    w0 = rng.standard_normal((C, d)).astype(np.float32)
    w0 /= np.linalg.norm(w0, axis=1, keepdims=True)
    scales = [scale, scale * 0.5] if learn else [scale, scale]
    bi = [rng.permutation(n_img)[:ri if k != 5 else max(1, ri - 5)] for k in range(steps)] if ri else None
    bt = [rng.permutation(n_txt)[:rt if k != 7 else max(1, rt - 9)] for k in range(steps)] if rt else None
    lrs = [1e-3 * (k + 1) / steps for k in range(steps)]
    wd = 0.01 if optim != "sgd" else 1e-3
    em, scm = _run(monkeypatch, True, d, C, w0, scales, learn, optim, wd, xi, yi, xt, yt, bi, bt, lrs, 0.7)
    eg, scg = _run(monkeypatch, False, d, C, w0, scales, learn, optim, wd, xi, yi, xt, yt, bi, bt, lrs, 0.7)
    # oracle
    st = O.HeadState(w0.copy(), None, scales[0], scales[1], learn)
    opt = O.OptState(optim, wd)
    for k in range(steps):
        so = O.step_grads(st, xi[bi[k]] if ri else None, yi[bi[k]] if ri else None, xt[bt[k]] if rt else None,
                          yt[bt[k]] if rt else None, 0.7)
        if ri:
            assert abs(scm[k, umlh.S_LOSS_IMG] - so.loss_img) < 1e-4, (k, scm[k], so.loss_img)
            assert abs(scm[k, umlh.S_ACC_IMG] - so.acc_img) < 1e-6
        if rt:
            assert abs(scm[k, umlh.S_LOSS_TXT] - so.loss_txt) < 1e-4, (k, scm[k], so.loss_txt)
            assert abs(scm[k, umlh.S_ACC_TXT] - so.acc_txt) < 1e-6
        O.optimizer_step(st, so.grads, opt, lrs[k])
    cols = list(range(8)) if learn else [0, 1, 2, 3, 6, 7]          # d loss / d scale is only formed for learnable scales
    np.testing.assert_allclose(scm[:, cols], scg[:, cols], atol=2e-5, rtol=1e-4)       # == the general path
    wm, wg = em.w_head.cpu().numpy(), eg.w_head.cpu().numpy()
    # Adam's early steps move each weight by ~lr*sign(g): elements whose gradient is at fp32-noise level may take the other sign
    lim = 2 * sum(lrs) + 1e-6
    for ref in (st.w_head, wg):
        diff = np.abs(wm - ref)
        assert (diff > 5e-6 + 1e-4 * np.abs(ref)).mean() < 2e-3 and diff.max() <= lim
    np.testing.assert_allclose(em.m_head.cpu().numpy(), eg.m_head.cpu().numpy(), atol=1e-6 * max(1.0, np.abs(eg.m_head.cpu().numpy()).max()), rtol=2e-3)
    if learn:
        np.testing.assert_allclose(em.scales.cpu().numpy(), [st.img_scale, st.txt_scale], atol=5e-5)
        np.testing.assert_allclose(em.scales.cpu().numpy(), eg.scales.cpu().numpy(), atol=5e-5)


def test_micro_path_continues_across_calls_and_evals():
    """Two calls (evaluation intervals) == one call; eval_rows between the calls sees the updated weights."""
    import umlh
    rng = np.random.default_rng(3)
    d, C, n = 512, 100, 400
    xi, yi, xt, yt = _tables(rng, d, C, n, n)
    w0 = (0.05 * rng.standard_normal((C, d))).astype(np.float32)
    Xi, Yi, Xt, Yt = _T(xi), _T(yi, torch.int64), _T(xt), _T(yt, torch.int64)
    bi = [_T(rng.permutation(n)[:32], torch.int64) for _ in range(30)]
    bt = [_T(rng.permutation(n)[:32], torch.int64) for _ in range(30)]
    outs = []
    for split in (None, 11):
        e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=64, max_rows_txt=64, device=DEV)
        e.w_head.copy_(_T(w0)); e.scales.fill_(30.0)
        sc = torch.zeros(30, umlh.N_SCALARS, device=DEV)
        if split is None:
            e.train_steps((Xi, Yi), bi, (Xt, Yt), bt, [1e-3] * 30, first_step=1, scalars_out=sc)
        else:
            e.train_steps((Xi, Yi), bi[:split], (Xt, Yt), bt[:split], [1e-3] * split, first_step=1, scalars_out=sc[:split])
            mid = e.eval_rows(umlh.RowBatch(Xi[:64], Yi[:64])).cpu().numpy()
            e.train_steps((Xi, Yi), bi[split:], (Xt, Yt), bt[split:], [1e-3] * (30 - split), first_step=split + 1, scalars_out=sc[split:])
            assert np.isfinite(mid).all()
        torch.cuda.synchronize()
        assert e.micro_launches() == (1 if split is None else 2) and e.micro_status() == 0
        outs.append((e.w_head.cpu().numpy().copy(), e.v_head.cpu().numpy().copy(), sc.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_grouped_launch_equals_isolated_runs_bit_for_bit():
    """18 heads of the clip_linear grid (lr x wd x 3 seeds, engine/optimizer/default.py:17-30) over the SAME tables, own
    index streams: one grouped launch == 18 separate umlh_train_steps calls, bit for bit (weights, moments, scalars)."""
    import umlh
    rng = np.random.default_rng(11)
    d, C, n_img, n_txt, steps = 512, 100, 1600, 3000, 25
    xi, yi, xt, yt = _tables(rng, d, C, n_img, n_txt)
    Xi, Yi, Xt, Yt = _T(xi), _T(yi, torch.int64), _T(xt), _T(yt, torch.int64)
    grid = [(lr, wd, seed) for lr in (1e-3, 1e-4) for wd in (0.0, 0.01, 0.001) for seed in (1, 2, 3)]
    w0 = O.zero_shot_weights(xt, yt, C)

    def make(lr, wd, seed):
        e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=wd, max_rows_img=32, max_rows_txt=32, device=DEV)
        e.w_head.copy_(_T(w0)); e.scales.fill_(100.0)
        g = np.random.default_rng(seed)
        bi = [_T(g.permutation(n_img)[:32], torch.int64) for _ in range(steps)]
        bt = [_T(g.permutation(n_txt)[:32], torch.int64) for _ in range(steps)]
        lrs = [lr * min(1.0, (k + 1) / 10) for k in range(steps)]
        return e, bi, bt, lrs

    iso = []
    for lr, wd, seed in grid:
        e, bi, bt, lrs = make(lr, wd, seed)
        sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
        e.train_steps((Xi, Yi), bi, (Xt, Yt), bt, lrs, first_step=1, scalars_out=sc)
        iso.append((e, sc))
    jobs, grp = [], []
    for lr, wd, seed in grid:
        e, bi, bt, lrs = make(lr, wd, seed)
        sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
        jobs.append(dict(engine=e, img_table=(Xi, Yi), img_index_batches=bi, txt_table=(Xt, Yt), txt_index_batches=bt,
                         lrs=lrs, first_step=1, alpha=1.0, scalars_out=sc))
        grp.append((e, sc))
    umlh.train_steps_grouped(jobs, steps)
    torch.cuda.synchronize()
    for (ea, sa), (eb, sb) in zip(iso, grp):
        assert ea.micro_status() == 0 and eb.micro_status() == 0 and eb.micro_launches() == 1
        for name in ("w_head", "m_head", "v_head"):
            np.testing.assert_array_equal(getattr(ea, name).cpu().numpy(), getattr(eb, name).cpu().numpy(), err_msg=name)
        np.testing.assert_array_equal(sa.cpu().numpy(), sb.cpu().numpy())
    # the heads differ from each other (lr / wd / batches really are per head)
    assert not np.array_equal(grp[0][0].w_head.cpu().numpy(), grp[1][0].w_head.cpu().numpy())
    assert not np.array_equal(grp[0][0].w_head.cpu().numpy(), grp[9][0].w_head.cpu().numpy())


@pytest.mark.parametrize("tag", ["lin_zs", "lin_imgonly"])
def test_train_replays_reference_run_on_the_micro_path(tag):
    """The reference's own finetune.train() runs (golden fixtures) replayed with the per-step diagnostics off, i.e.
    through the micro-step kernel: same batches, per-step losses 1e-4, evaluation trace, early-stop step, best weights."""
    import finetune as ft
    import umlh
    from engine.datasets.utils import FeatureLoader, FeatureTable, TextTensorDataset
    from engine.models.head import UML
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    from engine.tools.utils import set_random_seed
    g = load_golden("train_" + tag)
    (d_img, text_indim, C, B, max_iters, eval_freq, patience, lr, wd, alpha, learnable, zeroshot, seed) = g["cfg"]
    d_img, text_indim, C, B, max_iters, eval_freq, patience, seed = map(int, (d_img, text_indim, C, B, max_iters, eval_freq, patience, seed))
    modality = str(g["modality"])
    T = torch.as_tensor
    set_random_seed(seed)
    text_ds = TextTensorDataset(T(g["x_txt"]), T(g["y_txt"]), torch.zeros(len(g["y_txt"]), dtype=torch.long))
    model = UML(d_img, text_indim, C, bias=False, learnable_temp=bool(learnable)).to(DEV)
    if zeroshot:
        model.zero_shot_init(text_ds)
    optimizer = build_optimizer(model.parameters(), str(g["optim"]), float(lr), float(wd))
    scheduler = build_lr_scheduler(optimizer, "cosine", 50, max_iters, warmup_type="linear", warmup_lr=1e-5)
    image_loader = FeatureLoader(FeatureTable(T(g["x_img"]), T(g["y_img"]), DEV), B, shuffle=True, kind="image")
    text_loader = None if modality == "image" else FeatureLoader(FeatureTable(text_ds.input_tensor, text_ds.label_tensor, DEV), B, shuffle=True, kind="text")
    val_loader = FeatureLoader(FeatureTable(T(g["x_val"]), T(g["y_val"]), DEV), B, shuffle=False)
    test_loader = FeatureLoader(FeatureTable(T(g["x_test"]), T(g["y_test"]), DEV), B, shuffle=False)
    out = ft.train(model, image_loader, text_loader, val_loader, test_loader, optimizer, scheduler, device=DEV,
                   max_iters=max_iters, alpha=float(alpha), eval_freq=eval_freq, patience=patience)
    assert any(e.micro_launches() > 0 for e in model._engines.values())
    test_loss, test_acc = ft.validate(model, test_loader, device=DEV)
    n = int(g["n_steps"])
    sc = out["train_scalars"].numpy()
    assert sc.shape[0] == n
    ce = g["train_ce"]
    if modality == "image":
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_IMG], ce, atol=1e-4)
    else:
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_IMG], ce[0::2], atol=1e-4)
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_TXT], ce[1::2], atol=1e-4)
    assert out["iter"] == int(g["best_iter"]) and abs(out["val_acc"] - float(g["best_val_acc"])) < 1e-6
    np.testing.assert_allclose(out["model"]["head.weight"].numpy(), g["w_head_best"], atol=1e-5, rtol=1e-4)
    assert abs(test_acc - float(g["test_acc"])) <= 1e-3 and abs(test_loss - float(g["test_loss"])) < 1e-4


@pytest.mark.parametrize("d,C,ri,rt", [(512, 100, 32, 32), (768, 1000, 32, 24), (1024, 47, 16, 48), (128, 10, 64, 0)])
def test_micro_steps_bf16_operand_mode(d, C, ri, rt, monkeypatch):
    """bf16 engines take the micro path too (operands rounded to bf16 at use, fp32 accumulation, fp32 master weights): the
    per-step losses follow the general bf16 path (three kernels, bf16 shadows) within the precision mode's own noise and the
    fp32 path loosely; accuracies and the final weights agree."""
    import umlh
    rng = np.random.default_rng(d + C)
    n_img, n_txt, steps = 300, 260, 10
    xi, yi, xt, yt = _tables(rng, d, C, n_img, n_txt)
    w0 = rng.standard_normal((C, d)).astype(np.float32)
    w0 /= np.linalg.norm(w0, axis=1, keepdims=True)
    bi = [rng.permutation(n_img)[:ri] for _ in range(steps)] if ri else None
    bt = [rng.permutation(n_txt)[:rt] for _ in range(steps)] if rt else None
    lrs = [1e-3] * steps

    def run(micro, precision):
        monkeypatch.setenv("UMLH_MICRO", "1" if micro else "0")
        e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=64, max_rows_txt=64, precision=precision, device=DEV)
        e.w_head.copy_(_T(w0)); e.scales.fill_(30.0)
        sc = torch.zeros(steps, umlh.N_SCALARS, device=DEV)
        tab = lambda x, y: (_T(x), _T(y, torch.int64)) + ((umlh.to_bf16(_T(x)),) if precision == "bf16" else ())
        e.train_steps(tab(xi, yi) if ri else None, [_T(b, torch.int64) for b in bi] if ri else None,
                      tab(xt, yt) if rt else None, [_T(b, torch.int64) for b in bt] if rt else None, lrs, first_step=1, alpha=0.7, scalars_out=sc)
        torch.cuda.synchronize()
        assert e.micro_status() == 0 and (e.micro_launches() > 0) == micro
        return e.w_head.cpu().numpy(), sc.cpu().numpy()
    wm, sm = run(True, "bf16")
    wg, sg = run(False, "bf16")
    wf, sf = run(True, "fp32")
    cols = [c for c, live in ((umlh.S_LOSS_IMG, ri), (umlh.S_LOSS_TXT, rt)) if live]
    np.testing.assert_allclose(sm[:, cols], sg[:, cols], atol=3e-3, rtol=2e-3)          # two bf16 implementations
    np.testing.assert_allclose(sm[:, cols], sf[:, cols], atol=2e-2, rtol=1e-2)          # bf16 operands vs fp32
    acc = [c for c, live in ((umlh.S_ACC_IMG, ri), (umlh.S_ACC_TXT, rt)) if live]
    assert np.abs(sm[:, acc] - sg[:, acc]).max() <= 2.0 / 16 + 1e-6                     # at most a couple of near-tie rows flip
    lim = 2 * sum(lrs) + 1e-6
    for ref in (wg, wf):
        diff = np.abs(wm - ref)
        assert diff.max() <= lim and (diff > 2e-4 + 1e-3 * np.abs(ref)).mean() < 0.05
