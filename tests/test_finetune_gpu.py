"""GPU: the reference-shaped train()/validate() on the fused HIP step replays the
reference's own finetune.train() runs (golden fixtures) under identical seeds:
same batches, per-step losses, eval accuracies, early-stop iteration, best weights,
final top-1 within +-0.1 pp (north_star)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag", ["lin_zs", "mlp_lt", "lin_imgonly"])
def test_train_replays_reference_run(tag):
    import finetune as ft
    from engine.datasets.utils import FeatureLoader, FeatureTable, TextTensorDataset
    from engine.models.head import UML
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    from engine.tools.utils import set_random_seed
    import umlh

    g = load_golden("train_" + tag)
    (d_img, text_indim, C, B, max_iters, eval_freq, patience, lr, wd, alpha, learnable, zeroshot, seed) = g["cfg"]
    d_img, text_indim, C, B, max_iters, eval_freq, patience, seed = map(
        int, (d_img, text_indim, C, B, max_iters, eval_freq, patience, seed))
    modality = str(g["modality"])
    T = torch.as_tensor
    set_random_seed(seed)                                                    # finetune.py:452-454
    text_ds = TextTensorDataset(T(g["x_txt"]), T(g["y_txt"]), torch.zeros(len(g["y_txt"]), dtype=torch.long))
    model = UML(d_img, text_indim, C, bias=False, learnable_temp=bool(learnable))
    np.testing.assert_array_equal(model.head.weight.detach().numpy(), g["w_head_init"])
    model.to(DEV)
    if zeroshot:
        model.zero_shot_init(text_ds)
    optimizer = build_optimizer(model.parameters(), str(g["optim"]), float(lr), float(wd))
    scheduler = build_lr_scheduler(optimizer, "cosine", 50, max_iters, warmup_type="linear", warmup_lr=1e-5)
    image_loader = FeatureLoader(FeatureTable(T(g["x_img"]), T(g["y_img"]), DEV), B, shuffle=True, kind="image")
    text_loader = FeatureLoader(FeatureTable(text_ds.input_tensor, text_ds.label_tensor, DEV), B, shuffle=True, kind="text")
    if modality == "image":
        text_loader = None
    val_loader = FeatureLoader(FeatureTable(T(g["x_val"]), T(g["y_val"]), DEV), B, shuffle=False)
    test_loader = FeatureLoader(FeatureTable(T(g["x_test"]), T(g["y_test"]), DEV), B, shuffle=False)
    out = ft.train(model, image_loader, text_loader, val_loader, test_loader, optimizer, scheduler, device=DEV,
                   max_iters=max_iters, alpha=float(alpha), eval_freq=eval_freq, patience=patience, diagnostics=True)
    test_loss, test_acc = ft.validate(model, test_loader, device=DEV)

    n = int(g["n_steps"])
    sc = out["train_scalars"].numpy()
    assert sc.shape[0] == n                                                  # same early-stop step
    ce = g["train_ce"]
    if modality == "image":
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_IMG], ce, atol=1e-4)
    else:
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_IMG], ce[0::2], atol=1e-4)
        np.testing.assert_allclose(sc[:, umlh.S_LOSS_TXT], ce[1::2], atol=1e-4)
    # per-step gradient diagnostics (by-products of the slab reduction) vs the reference's own gradients
    gd = g["grad_diag"]
    n_el = model.head.weight.numel()
    has_txt = modality != "image"
    dg = [umlh.grad_diagnostics(row, n_el, 1, 1 if has_txt else 0) for row in out["train_scalars"]]
    got = np.asarray([[r["grad_direction_sim"], r["grad_agreement_rate"], r["img_grad_norm"], r["txt_grad_norm"]] for r in dg])
    np.testing.assert_allclose(got[:, 0], gd[:, 0], atol=5e-4)
    np.testing.assert_allclose(got[:, 1], gd[:, 1], atol=5e-3)
    np.testing.assert_allclose(got[:, 2:], gd[:, 2:], rtol=2e-3, atol=1e-7)
    assert out["iter"] == int(g["best_iter"])
    assert abs(out["val_acc"] - float(g["best_val_acc"])) < 1e-6
    assert abs(out["val_loss"] - float(g["best_val_loss"])) < 1e-4
    np.testing.assert_allclose(out["model"]["head.weight"].numpy(), g["w_head_best"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(model.head.weight.detach().cpu().numpy(), g["w_head_best"], atol=1e-5, rtol=1e-4)
    if "w_proj_best" in g.files:
        np.testing.assert_allclose(out["model"]["img_proj.weight"].numpy(), g["w_proj_best"], atol=1e-5, rtol=1e-4)
    if bool(learnable):
        assert abs(float(out["model"]["img_scale"]) - float(g["img_scale_best"])) < 1e-5
        assert abs(float(out["model"]["txt_scale"]) - float(g["txt_scale_best"])) < 1e-5
    assert abs(test_acc - float(g["test_acc"])) <= 1e-3                      # +-0.1 pp
    assert abs(test_loss - float(g["test_loss"])) < 1e-4
    assert set(out) >= {"iter", "val_acc", "model", "val_classwise", "val_loss", "model_records"}


def test_model_forward_and_optimizer_step_unfused():
    """model(images, text) returns the reference's logits pair; optimizer.step() applies the
    HIP update kernel from .grad (the unfused surface of the drop-in)."""
    from engine.models.head import UML, UMLClip
    from engine.optimizer.optim import build_optimizer
    from oracle import uml_oracle as O
    g = load_golden("step_mlp_d48_t64_c10")
    m = UML(48, 64, 10, learnable_temp=True).to(DEV)
    with torch.no_grad():
        m.head.weight.copy_(torch.as_tensor(g["w_head"]))
        m.img_proj.weight.copy_(torch.as_tensor(g["w_proj"]))
        m.img_scale.fill_(float(g["scale_img"]))
        m.txt_scale.fill_(float(g["scale_txt"]))
    zi, zt = m(torch.as_tensor(g["x_img"]).to(DEV), torch.as_tensor(g["x_txt"]).to(DEV))
    np.testing.assert_allclose(zi.cpu().numpy(), g["img_logits"], atol=1e-4)
    np.testing.assert_allclose(zt.cpu().numpy(), g["txt_logits"], atol=1e-4)
    zi2, none = m(torch.as_tensor(g["x_img"]).to(DEV))
    assert none is None and torch.equal(zi, zi2)
    f = m.extract_features(torch.as_tensor(g["x_img"]).to(DEV))
    np.testing.assert_allclose(f.cpu().numpy(), g["x_img"] @ g["w_proj"].T, atol=1e-5)
    # optimizer.step() from a caller-provided gradient
    opt = build_optimizer([m.head.weight], "adamw", 1e-3, 0.01)
    m.head.weight.grad = torch.as_tensor(g["g_head"]).to(DEV)
    st = O.HeadState(g["w_head"].copy())
    O.optimizer_step(st, {"w_head": g["g_head"]}, O.OptState("adamw", 0.01), 1e-3)
    opt.step()
    np.testing.assert_allclose(m.head.weight.detach().cpu().numpy(), st.w_head, atol=1e-7, rtol=1e-6)
    opt.zero_grad()
    assert m.head.weight.grad is None
    c = UMLClip("ViT-B/16", 7).to(DEV)
    x = torch.nn.functional.normalize(torch.randn(5, 512), dim=1).to(DEV)
    z, _ = c(x)
    ref = (x.cpu().numpy() @ c.head.weight.detach().cpu().numpy().T) * np.float32(np.exp(np.log(1 / 0.07)))
    np.testing.assert_allclose(z.cpu().numpy(), ref, atol=1e-4)


def _toy_dataset(seed=5, C=12, d=64, n_per=10, n_txt=80, n_val=60):
    g = torch.Generator().manual_seed(seed)
    proto = torch.randn(C, d, generator=g)

    def draw(n):
        y = torch.randint(0, C, (n,), generator=g)
        x = torch.nn.functional.normalize(proto[y] + 1.5 * torch.randn(n, d, generator=g), dim=1)
        return x, y
    xi = torch.nn.functional.normalize(torch.cat([proto[c] + 1.5 * torch.randn(n_per, d, generator=g) for c in range(C)]), dim=1)
    yi = torch.arange(C).repeat_interleave(n_per)
    return (xi, yi), draw(n_val), draw(n_val + 5), draw(n_txt), C


def test_sweep_farm_equals_isolated_runs(tmp_path):
    """Grid points run concurrently (one engine + HIP stream + host thread each, shared feature
    tables) give exactly what each point gives when run alone with the same private generator."""
    import types
    import finetune as ft
    from engine.datasets.utils import TextTensorDataset
    from engine.models.head import UMLClip
    tr, va, te, (xt, yt), C = _toy_dataset()
    text_ds = TextTensorDataset(xt, yt, torch.zeros(len(yt), dtype=torch.long))
    grid = {"optim": "adamw", "lr": [1e-3, 1e-4], "weight_decay": [0.0, 0.01, 0.001], "lr_scheduler": "cosine",
            "batch_size": 16, "max_iter": 240, "warmup_iter": 50, "warmup_type": "linear", "warmup_min_lr": 1e-5,
            "patience": 50, "dropout": None, "learnable_temp": False}
    datasets = {"img_tr": tr, "img_val": va, "img_te": te, "text_ds": text_ds}

    def args_for(path, workers):
        os_path = str(path)
        return types.SimpleNamespace(savepath=os_path, device=DEV, modality="crossmodal", alpha=1.0,
                                     classifier_init="zeroshot", use_clip=True, logit=4.60517, nclasses=C, seed=3,
                                     precision="fp32", sweep_workers=workers, eval_test=True)
    (tmp_path / "farm").mkdir()
    res, best_val, best_test = ft.sweep(datasets, grid, args_for(tmp_path / "farm", 4))
    assert len(res["val_acc"]) == 6 and best_val == max(res["val_acc"])
    for idx, hp in enumerate(ft._grid(grid)):
        gen = torch.Generator()
        gen.manual_seed(ft.farm_seed(3, idx))
        torch.manual_seed(ft.farm_seed(3, idx))
        model = UMLClip(tr[0].shape[1], C, logit_scale_init=4.60517, bias=False, learnable_temp=False)
        solo = ft.setup_feature_run(tr, va, te, text_ds, hp, num_classes=C, use_clip=True, device=DEV, generator=gen,
                                    model=model)
        saved = torch.load(tmp_path / "farm" / ft.hparam_str(hp["optim"], hp["lr"], hp["weight_decay"], 16, 240, None, False)
                           / "test_result.pth", weights_only=True)
        assert saved["iter"] == solo["iter"]
        assert saved["val_acc"] == solo["val_acc"] == res["val_acc"][idx]
        assert saved["test_acc"] == solo["test_acc"] == res["test_acc"][idx]
        assert torch.equal(saved["model"]["head.weight"], solo["model"]["head.weight"])
    # a second sweep finds every result file and skips the work (reference finetune.py:330-333)
    res2, _, _ = ft.sweep(datasets, grid, args_for(tmp_path / "farm", 4))
    assert res2["val_acc"] == res["val_acc"]


def test_grouped_sweep_equals_isolated_runs(tmp_path):
    """The sweep as ONE grouped job (sweep_mode="grouped": every grid point a head of the same persistent launches,
    umlh_train_steps_grouped) gives exactly what each point gives when run alone with the same private generator:
    early-stop iteration, accuracies and best weights, bit for bit (finetune.py:406-448)."""
    import types
    import finetune as ft
    from engine.datasets.utils import TextTensorDataset
    from engine.models.head import UMLClip
    tr, va, te, (xt, yt), C = _toy_dataset()
    text_ds = TextTensorDataset(xt, yt, torch.zeros(len(yt), dtype=torch.long))
    grid = {"optim": "adamw", "lr": [1e-3, 1e-4], "weight_decay": [0.0, 0.01, 0.001], "lr_scheduler": "cosine",
            "batch_size": 16, "max_iter": [240, 130], "warmup_iter": 50, "warmup_type": "linear", "warmup_min_lr": 1e-5,
            "patience": 2, "dropout": None, "learnable_temp": False}
    datasets = {"img_tr": tr, "img_val": va, "img_te": te, "text_ds": text_ds}
    args = types.SimpleNamespace(savepath=str(tmp_path), device=DEV, modality="crossmodal", alpha=1.0,
                                 classifier_init="zeroshot", use_clip=True, logit=4.60517, nclasses=C, seed=3,
                                 precision="fp32", sweep_mode="grouped", eval_test=True)
    res, best_val, best_test = ft.sweep(datasets, grid, args)
    points = ft._grid(grid)
    assert len(res["val_acc"]) == len(points) == 12 and best_val == max(res["val_acc"])
    iters = set()
    for idx, hp in enumerate(points):
        gen = torch.Generator()
        gen.manual_seed(ft.farm_seed(3, idx))
        torch.manual_seed(ft.farm_seed(3, idx))
        model = UMLClip(tr[0].shape[1], C, logit_scale_init=4.60517, bias=False, learnable_temp=False)
        solo = ft.setup_feature_run(tr, va, te, text_ds, hp, num_classes=C, use_clip=True, device=DEV, generator=gen,
                                    model=model)
        saved = torch.load(tmp_path / ft.hparam_str(hp["optim"], hp["lr"], hp["weight_decay"], 16, hp["max_iter"], None, False)
                           / "test_result.pth", weights_only=True)
        assert saved["iter"] == solo["iter"]
        assert saved["val_acc"] == solo["val_acc"] == res["val_acc"][idx]
        assert saved["test_acc"] == solo["test_acc"] == res["test_acc"][idx]
        assert torch.equal(saved["model"]["head.weight"], solo["model"]["head.weight"])
        iters.add(saved["iter"])


@pytest.mark.parametrize("tag", ["lin_zs", "mlp_lt"])
def test_logger_receives_reference_diagnostics(tag):
    """With a logger, train() logs per step what the reference logs (finetune.py:236-240): losses, accuracies, lr, the
    gradient diagnostics and feature_direction_sim (checked against numpy on the reference's recorded batches)."""
    import finetune as ft
    from engine.datasets.utils import FeatureLoader, FeatureTable, TextTensorDataset
    from engine.models.head import UML
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    from engine.tools.utils import set_random_seed
    g = load_golden("train_" + tag)
    (d_img, text_indim, C, B, max_iters, eval_freq, patience, lr, wd, alpha, learnable, zeroshot, seed) = g["cfg"]
    d_img, text_indim, C, B, seed = map(int, (d_img, text_indim, C, B, seed))
    T = torch.as_tensor
    set_random_seed(seed)
    text_ds = TextTensorDataset(T(g["x_txt"]), T(g["y_txt"]), torch.zeros(len(g["y_txt"]), dtype=torch.long))
    model = UML(d_img, text_indim, C, bias=False, learnable_temp=bool(learnable)).to(DEV)
    if zeroshot:
        model.zero_shot_init(text_ds)
    optimizer = build_optimizer(model.parameters(), str(g["optim"]), float(lr), float(wd))
    scheduler = build_lr_scheduler(optimizer, "cosine", 50, int(max_iters), warmup_type="linear", warmup_lr=1e-5)
    il = FeatureLoader(FeatureTable(T(g["x_img"]), T(g["y_img"]), DEV), B, shuffle=True, kind="image")
    tl = FeatureLoader(FeatureTable(text_ds.input_tensor, text_ds.label_tensor, DEV), B, shuffle=True, kind="text")
    vl = FeatureLoader(FeatureTable(T(g["x_val"]), T(g["y_val"]), DEV), B, shuffle=False)
    tel = FeatureLoader(FeatureTable(T(g["x_test"]), T(g["y_test"]), DEV), B, shuffle=False)   # its iter() draws RNG too

    class Log:
        def __init__(self):
            self.rows = []

        def log(self, d):
            self.rows.append(d)
    lg = Log()
    n = 12
    ft.train(model, il, tl, vl, tel, optimizer, scheduler, device=DEV, max_iters=n, alpha=float(alpha), eval_freq=int(eval_freq),
             patience=100, logger=lg)
    steps = [r for r in lg.rows if "train/image_loss" in r]
    assert len(steps) == n
    gd, ce = g["grad_diag"], g["train_ce"]
    n_img = len(g["y_img"])
    for k, r in enumerate(steps):
        assert abs(r["train/image_loss"] - ce[2 * k]) < 1e-4 and abs(r["train/text_loss"] - ce[2 * k + 1]) < 1e-4
        assert abs(r["train/grad_direction_sim"] - gd[k, 0]) < 5e-4 and abs(r["train/grad_agreement_rate"] - gd[k, 1]) < 5e-3
        assert abs(r["train/img_grad_norm"] - gd[k, 2]) < 2e-3 * gd[k, 2] + 1e-7
    # feature_direction_sim: batches in the reference's order (the recorded sampler indices); with img_proj only step 0
    # uses known (initial) projection weights
    def batches(idx, n_rows):                     # DataLoader(drop_last=False): every epoch ends with the short batch
        out, pos = [], 0
        while len(out) < n:
            for s in range(0, n_rows, B):
                m = min(B, n_rows - s)
                out.append(idx[pos:pos + m])
                pos += m
        return out
    bi_all, bt_all = batches(g["idx_img"], n_img), batches(g["idx_txt"], len(g["y_txt"]))
    for k in range(n if text_indim == 0 else 1):
        fi = g["x_img"][bi_all[k]]
        if text_indim > 0:
            fi = fi @ g["w_proj_init"].T
        fim, ftm = fi.mean(0), g["x_txt"][bt_all[k]].mean(0)
        ref = float(fim @ ftm / (np.linalg.norm(fim) * np.linalg.norm(ftm)))
        assert abs(steps[k]["train/feature_direction_sim"] - ref) < 1e-4, k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_loop_with_bias_head(precision, monkeypatch):
    """finetune.train() end to end with a bias=True head (head.py:65,68,122): the single-launch micro path (d = 512 packs
    to a 640-wide head) and the general three-kernel path train the same model (same batches, per-step losses within the
    arithmetic's noise), the returned best state carries head.weight AND head.bias, and the padding columns stay zero."""
    import finetune as ft
    from engine.datasets.utils import TextTensorDataset
    from engine.models.head import UMLClip
    g = torch.Generator().manual_seed(5)
    C, d = 12, 512
    proto = torch.nn.functional.normalize(torch.randn(C, d, generator=g), dim=1)

    def draw(n):
        y = torch.randint(0, C, (n,), generator=g)
        return torch.nn.functional.normalize(proto[y] + 0.08 * torch.randn(n, d, generator=g), dim=1), y
    tr, va, te, (xt, yt) = draw(192), draw(200), draw(300), draw(120)
    text_ds = TextTensorDataset(xt, yt, torch.zeros(len(yt), dtype=torch.long))
    hp = {"optim": "adamw", "lr": 1e-3, "weight_decay": 0.01, "lr_scheduler": "cosine", "batch_size": 16, "max_iter": 150,
          "warmup_iter": 10, "warmup_type": "linear", "warmup_min_lr": 1e-5, "patience": 5, "dropout": None, "learnable_temp": False}
    outs = []
    for micro in ("1", "0"):
        monkeypatch.setenv("UMLH_MICRO", micro)
        gen = torch.Generator()
        gen.manual_seed(11)
        torch.manual_seed(11)
        model = UMLClip(d, C, logit_scale_init=4.60517, bias=True)
        out = ft.setup_feature_run(tr, va, te, text_ds, hp, num_classes=C, use_clip=True, device=DEV, generator=gen, model=model,
                                   precision=precision)
        assert set(out["model"]) == {"head.weight", "head.bias"} and tuple(out["model"]["head.bias"].shape) == (C,)
        assert out["val_acc"] > 0.9 and float(out["model"]["head.bias"].abs().max()) > 0
        assert float(model._packed[:, d + 1:].abs().max()) == 0.0
        outs.append(out)
    a, b = outs
    assert a["iter"] == b["iter"] and abs(a["val_acc"] - b["val_acc"]) <= 0.01
    tol = 2e-4 if precision == "fp32" else 2e-2
    np.testing.assert_allclose(a["train_scalars"][:, :2].numpy(), b["train_scalars"][:, :2].numpy(), atol=tol, rtol=tol)
