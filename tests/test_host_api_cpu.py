"""CPU tests of the reference-shaped host API (no GPU compute): scheduler values
against the reference's golden LR traces, loader order against torch's own
DataLoader under the same seed, TextTensorDataset reductions, factory errors."""
import numpy as np
import pytest
import torch

from conftest import load_golden


class _FakeOpt:
    def __init__(self, lr):
        self.param_groups = [{"lr": lr, "initial_lr": lr}]


@pytest.mark.parametrize("tag,kind,wtype", [("cos_lin", "cosine", "linear"), ("cos_const", "cosine", "constant"),
                                            ("lin_lin", "linear", "linear"), ("cos_nowarm", "cosine", None)])
def test_scheduler_matches_reference_traces(tag, kind, wtype):
    from engine.optimizer.scheduler import build_lr_scheduler
    g = load_golden("lr_traces")
    lr, warm, max_iter, wlr = g[tag + "_cfg"]
    opt = _FakeOpt(float(lr))
    s = build_lr_scheduler(opt, kind, int(warm), int(max_iter), warmup_type=wtype, warmup_lr=None if wlr < 0 else float(wlr))
    ref = g[tag]
    got = []
    for _ in range(len(ref)):
        got.append(opt.param_groups[0]["lr"])
        assert s.get_last_lr()[0] == opt.param_groups[0]["lr"]
        s.step()
    # closed form vs torch's recursive cosine: agree to fp64 round-off
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-16)
    np.testing.assert_allclose(s.__class__(opt, kind, max_iter, int(warm), wtype, None if wlr < 0 else float(wlr)).lr_table(8, 0),
                               ref[:8], rtol=1e-9, atol=1e-16)


def test_scheduler_and_optimizer_argument_errors():
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    with pytest.raises(ValueError):
        build_lr_scheduler(_FakeOpt(1e-3), "step", 0, 10)
    with pytest.raises(ValueError):
        build_lr_scheduler(_FakeOpt(1e-3), "cosine", 5, 10, warmup_type="exp", warmup_lr=1e-5)
    with pytest.raises(AssertionError):
        build_optimizer([torch.nn.Parameter(torch.zeros(2))], "lion", 1e-3, 0.0)
    o = build_optimizer([torch.nn.Parameter(torch.zeros(2))], "adamw", 1e-3, 0.01)
    assert o.param_groups[0]["lr"] == 1e-3 and o.param_groups[0]["weight_decay"] == 0.01 and o.step_count == 0


def test_hyper_dict_grids():
    from engine.optimizer.default import HYPER_DICT
    g = HYPER_DICT["clip_linear"]
    assert g["optim"] == "adamw" and g["lr"] == [0.001, 0.0001] and g["weight_decay"] == [0.0, 0.01, 0.001]
    assert g["batch_size"] == [32] and g["max_iter"] == [12800] and g["warmup_iter"] == 50 and g["patience"] == [5]
    assert HYPER_DICT["linear"]["learnable_temp"] == [True] and HYPER_DICT["linear"]["batch_size"] == [8, 32]
    assert set(HYPER_DICT) == {"full_ds_full_model_finetune", "clip_linear", "linear", "audio"}


class _DictDS(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"i": i}


def test_feature_loader_order_equals_torch_dataloader_under_same_seed():
    """Two interleaved loaders cycled with fetch_next + an unshuffled loader iterated in
    between (validate) -- the global-RNG protocol of finetune.train."""
    from engine.datasets.utils import FeatureLoader, FeatureTable
    from finetune import fetch_next
    n_a, n_b, bs = 50, 23, 8

    def run(make_a, make_b, make_v, get):
        torch.manual_seed(1234)
        la, lb, lv = make_a(), make_b(), make_v()
        ia, ib = iter(la) if not hasattr(la, "iter_index") else la.iter_index(), \
            iter(lb) if not hasattr(lb, "iter_index") else lb.iter_index()
        seq = []
        for step in range(25):
            if hasattr(la, "iter_index"):
                try:
                    a = next(ia)
                except StopIteration:
                    ia = la.iter_index(); a = next(ia)
                try:
                    b = next(ib)
                except StopIteration:
                    ib = lb.iter_index(); b = next(ib)
            else:
                a, ia = fetch_next(la, ia)
                b, ib = fetch_next(lb, ib)
            seq.append((get(a), get(b)))
            if step % 10 == 0:
                for _ in (lv.iter_index() if hasattr(lv, "iter_index") else lv):
                    pass
        return seq

    from torch.utils.data import DataLoader
    ref = run(lambda: DataLoader(_DictDS(n_a), batch_size=bs, shuffle=True), lambda: DataLoader(_DictDS(n_b), batch_size=bs, shuffle=True),
              lambda: DataLoader(_DictDS(9), batch_size=4, shuffle=False), lambda b: b["i"].tolist())
    tab = lambda n: FeatureTable(torch.zeros(n, 2), torch.zeros(n, dtype=torch.long), "cpu")
    got = run(lambda: FeatureLoader(tab(n_a), bs, shuffle=True), lambda: FeatureLoader(tab(n_b), bs, shuffle=True),
              lambda: FeatureLoader(tab(9), 4, shuffle=False), lambda b: b.tolist())
    assert got == ref


def test_text_tensor_dataset_reductions_golden():
    from engine.datasets.utils import TextTensorDataset
    g = load_golden("text_side")
    f, l, e = torch.as_tensor(g["feats"]), torch.as_tensor(g["labels"]), torch.as_tensor(g["eot"])
    avg = TextTensorDataset(f, l, e, n_shots="average")
    np.testing.assert_allclose(avg.input_tensor.numpy(), g["avg_feats"], atol=1e-6)
    np.testing.assert_array_equal(avg.label_tensor.numpy(), g["avg_labels"])
    np.testing.assert_array_equal(avg.eot_indices.numpy(), g["avg_eot"])
    torch.manual_seed(int(g["shot_seed"]))
    shot = TextTensorDataset(f, l, e, n_shots=3)
    np.testing.assert_array_equal(shot.input_tensor.numpy(), g["shot_feats"])
    np.testing.assert_array_equal(shot.label_tensor.numpy(), g["shot_labels"])
    np.testing.assert_array_equal(shot.eot_indices.numpy(), g["shot_eot"])
    assert len(shot) == len(g["shot_labels"]) and len(shot[0]) == 3
    with pytest.raises(ValueError):
        TextTensorDataset(f, l, e, n_shots=2.5)


def test_model_surface_and_seed_parity_of_init():
    """UML/UMLClip expose the reference attributes; nn.Linear init order consumes the
    global RNG like the reference (w_head_init of the golden run)."""
    from engine.models.head import UML, UMLClip
    from engine.tools.utils import set_random_seed
    g = load_golden("train_mlp_lt")
    d_img, text_indim, C = int(g["cfg"][0]), int(g["cfg"][1]), int(g["cfg"][2])
    set_random_seed(int(g["cfg"][12]))
    m = UML(d_img, text_indim, C, bias=False, learnable_temp=True)
    np.testing.assert_array_equal(m.head.weight.detach().numpy(), g["w_head_init"])
    np.testing.assert_array_equal(m.img_proj.weight.detach().numpy(), g["w_proj_init"])
    assert m.num_classes == C and m.shared_dim == text_indim and m.img_proj is not None
    assert list(dict(m.named_parameters())) == ["img_scale", "txt_scale", "img_proj.weight", "head.weight"] or \
        set(dict(m.named_parameters())) == {"img_proj.weight", "head.weight", "img_scale", "txt_scale"}
    c = UMLClip("ViT-B/16", 10)
    assert c.shared_dim == 512 and c.img_proj is None and abs(float(c._scales[0]) - 1 / 0.07) < 1e-3
    assert hasattr(c, "extract_features") and hasattr(m, "extract_raw_features") and hasattr(m, "zero_shot_init")
    with pytest.raises(ValueError):
        UML("vit_base_patch16_224", 0, 3)


def test_host_cpu_budget_and_thread_fit():
    """umlh fits torch's intra-op pool to the CPUs the process may really use (affinity and cgroup quota)."""
    import os
    import torch
    import umlh
    b = umlh.host_cpu_budget()
    assert 1 <= b <= (os.cpu_count() or 1)
    if "OMP_NUM_THREADS" not in os.environ:
        assert torch.get_num_threads() <= b


def test_zero_shot_init_condition_mirrors_reference_setup():
    """reference finetune.py:362-363: zero-shot head for crossmodal runs, or image-only runs whose common_dim equals the
    text width; default unimodal runs (common_dim 0) and every text-only run keep the random nn.Linear init."""
    from finetune import wants_zero_shot_init as w
    assert w("zeroshot", "crossmodal", 0, 512)
    assert w("zeroshot", "image", 512, 512)
    assert not w("zeroshot", "image", 0, 512)           # CLIP features: d_img == d_txt must NOT trigger it
    assert not w("zeroshot", "image", 768, 512)
    assert not w("zeroshot", "text", 512, 512)
    assert not w("zeroshot", "text", 0, 512)
    assert not w("random", "crossmodal", 0, 512)


def test_multibench_dropout_seeds_do_not_consume_the_global_cpu_generator():
    """The loaders' shuffles draw from the global CPU generator (as the reference's DataLoaders do); dropout mask
    seeds must come from a module-private generator so batch orders stay those of the reference."""
    from multibench.models import Transformer
    torch.manual_seed(5)
    enc = Transformer(8, 10, nhead=5, num_layers=1)
    state = torch.get_rng_state()
    a, b = enc._dropout_seed(), enc._dropout_seed()
    assert a != b and torch.equal(torch.get_rng_state(), state)
    torch.manual_seed(5)
    enc2 = Transformer(8, 10, nhead=5, num_layers=1)
    assert enc2._dropout_seed() == a                     # still fixed by torch.manual_seed


def test_bias_head_keeps_reference_parameter_surface():
    """bias=True (engine/models/head.py:65,68,122): head.weight / head.bias keep the reference's names and shapes while
    living as views of one packed [C, d_aug] tensor; state_dict round-trips; with img_proj both layers are packed."""
    import pytest
    import torch
    from engine.models.head import UML, UMLClip
    torch.manual_seed(0)
    m = UMLClip("ViT-B/16", 10, bias=True)
    assert tuple(m.head.weight.shape) == (10, 512) and tuple(m.head.bias.shape) == (10,)
    assert set(m.state_dict()) == {"head.weight", "head.bias"}
    assert m._packed.shape == (10, 640) and float(m._packed[:, 513:].abs().max()) == 0.0
    assert m.head.weight.data_ptr() == m._packed.data_ptr()                      # views, not copies
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = UMLClip("ViT-B/16", 10, bias=True)
    m2.load_state_dict(sd)
    assert torch.equal(m2._packed[:, :512], sd["head.weight"]) and torch.equal(m2._packed[:, 512], sd["head.bias"])
    u = UML(96, 0, 7, bias=True)
    assert u._packed.shape == (7, 128) and torch.equal(u._packed[:, 96], u.head.bias.data)
    # 2-layer head: both layers packed; row d_sh of the packed projection is the constant row (1 at the ones column)
    p2 = UML(96, 64, 7, bias=True)
    assert set(p2.state_dict()) == {"img_proj.weight", "img_proj.bias", "head.weight", "head.bias"}
    assert p2._packed.shape == (7, 128) and p2._packed_proj.shape == (128, 128)
    assert tuple(p2.img_proj.weight.shape) == (64, 96) and tuple(p2.img_proj.bias.shape) == (64,)
    assert float(p2._packed_proj[64, 96]) == 1.0 and float(p2._packed_proj[64].sum()) == 1.0
    assert torch.equal(p2._packed_proj[:64, 96], p2.img_proj.bias.data) and float(p2._packed_proj[65:].abs().max()) == 0.0


def test_encoder_plan_pool_leases_and_eviction():
    """multibench.encoder's plan pool (host logic only, fake plans): a busy plan is never handed out twice, an idle one is
    reused, releasing is tied to the lease's lifetime, and least-recently-used idle configurations are dropped."""
    import gc
    from multibench import encoder as E

    class Fake:
        made = 0

        def __init__(self):
            Fake.made += 1
            self.busy = False
    saved = dict(E._PLANS)
    E._PLANS.clear()
    try:
        a = E._lease_plan("k", Fake)
        b = E._lease_plan("k", Fake)
        assert a.plan is not b.plan and a.plan.busy and b.plan.busy and Fake.made == 2
        pa = a.plan
        del a
        gc.collect()
        assert not pa.busy
        c = E._lease_plan("k", Fake)
        assert c.plan is pa and Fake.made == 2
        del b, c
        gc.collect()
        held = E._lease_plan("held", Fake)                       # a busy configuration survives eviction
        for i in range(E._PLAN_KEYS_MAX + 5):
            E._lease_plan(("cfg", i), Fake)
        assert len(E._PLANS) <= E._PLAN_KEYS_MAX + 1 and "held" in E._PLANS and "k" not in E._PLANS
        del held
    finally:
        E._PLANS.clear()
        E._PLANS.update(saved)


def test_lr_table_equals_stepwise_schedule():
    """StepLR.lr_table == lr_at step by step, BIT for bit (blockwise runs use lr_table, stepwise runs lr_at, and the two
    are claimed bit-identical), for every schedule / warm-up pair, across the warm-up boundary, past max_iter and over the whole
    12 800-step schedule of HYPER_DICT."""
    import numpy as np
    from engine.optimizer.scheduler import StepLR

    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 1e-3}]
    for kind in ("cosine", "linear"):
        for wt, wi in (("linear", 50), ("constant", 7), (None, 0)):
            s = StepLR(Opt(), kind, 300, warmup_iter=wi, warmup_type=wt, warmup_lr=1e-5)
            assert s.base_lrs == [1e-3]
            for start, n in ((0, 120), (40, 30), (wi, 1), (290, 40)):
                tab = np.asarray(s.lr_table(n, start=start))
                ref = np.asarray([s.lr_at(start + i, 1e-3) for i in range(n)])
                assert tab.shape == ref.shape
                assert tab.tobytes() == ref.tobytes()
        s = StepLR(Opt(), kind, 12800, warmup_iter=50, warmup_type="linear", warmup_lr=1e-5)
        tab = np.asarray(s.lr_table(12800, start=0))
        assert tab.tobytes() == np.asarray([s.lr_at(i, 1e-3) for i in range(12800)]).tobytes()


def test_head_optimizer_step_refuses_packed_bias_views_with_a_clear_error():
    """with bias=True the layer's weight / bias are strided views of one packed tensor; the unfused
    HeadOptimizer.step() cannot update them (detached moments, non-contiguous operands) and says so instead of failing deep in the
    multi-tensor launch."""
    import torch
    import umlh
    from engine.optimizer.optim import build_optimizer
    packed = torch.zeros(6, 8)
    w = torch.nn.Parameter(packed[:, :5])          # a strided view, as head.py's _pack_linear makes them
    w.grad = torch.ones_like(w)
    opt = build_optimizer([w], "adamw", 1e-3, 0.01)
    with pytest.raises(umlh.UmlhError, match="packed"):
        opt.step()
