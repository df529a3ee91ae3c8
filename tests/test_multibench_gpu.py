"""GPU: the MultiBench mirror (fused HIP decoder + masked MSE, torch encoder) against golden
vectors of the reference's models.UML (eval mode): losses, reconstructions, gradients, and a
4-step Adam alternation trajectory with step_k gating."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import multibench_oracle as MO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEEDS = {"z20_sin": 3, "z40_learn": 4, "z20_nopos": 5}


def _build(g):
    from multibench.models import Linear, Transformer, UML
    z, dx, dy, B, T, pe, pl = (int(v) for v in g["cfg"])
    m = UML(Linear(dx, z), Linear(dy, z), Transformer(z, z, nhead=5, num_layers=5, conv1d=True, out_last=False,
                                                       pos_embd=bool(pe), pos_learnable=bool(pl), max_len=128),
            [Linear(z, dx), Linear(z, dy)], modality="xy")
    sd = {k[4:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd::")}
    assert set(sd) == set(m.state_dict())                       # same parameter / buffer names as the reference
    m.load_state_dict(sd)
    return m.to(DEV).eval()


@pytest.mark.parametrize("tag", ["z20_sin", "z40_learn", "z20_nopos"])
def test_forward_loss_and_grads_match_reference(tag):
    g = load_golden("mb_" + tag)
    m = _build(g)
    T = lambda a: torch.as_tensor(a).to(DEV)
    out = m(T(g["x"]), T(g["y"]), T(g["lx"]), T(g["ly"]))
    assert abs(float(out["loss_x"]) - float(g["loss_x"])) < 1e-4
    assert abs(float(out["loss_y"]) - float(g["loss_y"])) < 1e-4
    np.testing.assert_allclose(out["zx"].detach().cpu().numpy(), g["zx"], atol=2e-4)
    np.testing.assert_allclose(out["x_recon"].detach().cpu().numpy(), g["x_recon"], atol=2e-4)
    np.testing.assert_allclose(out["y_recon"].detach().cpu().numpy(), g["y_recon"], atol=2e-4)
    assert abs(float(out["loss_private"]) - float(g["loss_private"])) < 1e-4
    ax, ay = g["alpha"]
    (float(ax) * out["loss_x"] + float(ay) * out["loss_y"]).backward()
    for k, p in m.named_parameters():
        gn = float(g["gn::" + k])
        assert abs(float(p.grad.norm()) - gn) <= 2e-3 * max(gn, 1e-3), k
        if "g::" + k in g.files:
            ref = g["g::" + k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=2e-4 * max(np.abs(ref).max(), 1e-3), rtol=2e-3, err_msg=k)


def test_fused_decoder_kernel_against_oracle_incl_edge_shapes():
    from multibench.models import _DecoderNextStepMSE
    rng = np.random.default_rng(0)
    for (B, Tn, Z, D, with_len) in [(3, 9, 20, 35, True), (2, 50, 40, 300, True), (4, 6, 17, 5, False), (5, 1, 8, 6, True),
                                    (2, 4, 130, 200, True)]:
        z = rng.standard_normal((B, Tn, Z)).astype(np.float32)
        w = (rng.standard_normal((D, Z)) * 0.2).astype(np.float32)
        b = rng.standard_normal(D).astype(np.float32)
        x = rng.standard_normal((B, Tn, D)).astype(np.float32)
        lens = rng.integers(1, Tn + 1, B) if with_len else None
        tz, tw, tb = (torch.as_tensor(a).to(DEV).requires_grad_(True) for a in (z, w, b))
        loss, recon = _DecoderNextStepMSE.apply(tz, tw, tb, torch.as_tensor(x).to(DEV),
                                                None if lens is None else torch.as_tensor(lens).to(DEV))
        (3.0 * loss).backward()
        if Tn == 1:      # models.py:209-210: plain reconstruction error of the single position
            ref = float(((z @ w.T + b - x) ** 2).mean())
            assert abs(float(loss) - ref) < 1e-5
            continue
        ol, orecon, odz, odw, odb = MO.decoder_next_step_loss(z, w, b, x, lens)
        assert abs(float(loss) - ol) < 1e-5 * max(1.0, ol)
        np.testing.assert_allclose(recon.cpu().numpy(), orecon, atol=1e-4)
        np.testing.assert_allclose(tz.grad.cpu().numpy(), 3 * odz, atol=1e-5, rtol=1e-3)
        np.testing.assert_allclose(tw.grad.cpu().numpy(), 3 * odw, atol=1e-5, rtol=1e-3)
        np.testing.assert_allclose(tb.grad.cpu().numpy(), 3 * odb, atol=1e-5, rtol=1e-3)


class _EpochLoader:
    """Two batches per epoch, the same tensors oracle/make_golden_multibench.py drew."""

    def __init__(self, seed, B, T, dx, dy, lx, ly):
        self.seed, self.shape, self.l, self.epoch = seed, (B, T, dx, dy), (lx, ly), -1

    def __iter__(self):
        B, T, dx, dy = self.shape
        self.epoch += 1          # (zip() never resumes the second loader's generator past its last batch)
        for bidx in range(2):
            gb = torch.Generator().manual_seed(1000 * self.seed + 10 * self.epoch + bidx)
            xb = torch.randn(B, T, dx, generator=gb)
            yb = torch.randn(B, T, dy, generator=gb)
            yield [[xb, None, yb], [self.l[0], None, self.l[1]]]


@pytest.mark.parametrize("tag", ["z20_sin", "z20_nopos"])
def test_alternation_loop_trajectory_matches_reference(tag):
    """train.py:354-398 with step_k = 0: epoch 0 trains on y only, epoch 1 on both; Adam lr 1e-3
    through the HIP optimizer kernel; two loaders zipped."""
    from engine.optimizer.optim import build_optimizer
    from multibench import train as mbt
    g = load_golden("mb_" + tag)
    m = _build(g)
    m.train = lambda *a, **k: m          # stay in eval mode (dropout off) like the fixture
    z, dx, dy, B, T, pe, pl = (int(v) for v in g["cfg"])
    ax, ay = (float(v) for v in g["alpha"])
    lx, ly = torch.as_tensor(g["lx"]), torch.as_tensor(g["ly"])
    l1 = _EpochLoader(SEEDS[tag], B, T, dx, dy, lx, ly)
    l2 = _EpochLoader(SEEDS[tag], B, T, dx, dy, lx, ly)
    opt = build_optimizer(m.parameters(), "adam", 1e-3, 0.0)
    r = mbt.train(m, "xy", l1, l2, opt, modalities=[0, 2], num_epoch=2, step_k=0, alpha_x=ax, alpha_y=ay, device=DEV)
    ref = g["traj_losses"]
    np.testing.assert_allclose(r["loss_x"], ref[:, 0], atol=3e-4)
    np.testing.assert_allclose(r["loss_y"], ref[:, 1], atol=3e-4)
    np.testing.assert_allclose(r["loss"], ref[:, 2], atol=5e-4)
    np.testing.assert_allclose(m.decoders[0].fc.weight.detach().cpu().numpy(), g["traj_dec0_w"], atol=3e-4)
    np.testing.assert_allclose(m.xproj_in.fc.weight.detach().cpu().numpy(), g["traj_xproj_w"], atol=3e-4)


def test_get_embedding_is_time_mean_of_encoder_outputs():
    from multibench.models import UML, Linear, Transformer
    torch.manual_seed(0)
    z = 20
    enc = Transformer(z, z, nhead=5, num_layers=1, conv1d=True, out_last=False, pos_embd=True, pos_learnable=False, max_len=128)
    m = UML(Linear(7, z), Linear(9, z), enc, [Linear(z, 7), Linear(z, 9)]).to(DEV).eval()
    x, y = torch.randn(4, 11, 7, device=DEV), torch.randn(4, 11, 9, device=DEV)
    ex, ey = m.get_embedding(x, y)
    with torch.no_grad():
        rx = m.encoder(m.xproj_in(x)).mean(dim=1)
        ry = m.encoder(m.yproj_in(y)).mean(dim=1)
    np.testing.assert_allclose(ex.cpu().numpy(), rx.cpu().numpy(), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(ey.cpu().numpy(), ry.cpu().numpy(), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("tag", ["small", "masked", "wide", "one_row_seqs"])
def test_infonce_hip_op_matches_reference_golden(tag):
    """SequenceInfoNCELoss on the HIP op (umlh_infonce_forward / _backward) against the reference's loss and autograd
    gradient (MultiBench/models.py:145-175; tests/golden/infonce.npz)."""
    from multibench.models import SequenceInfoNCELoss
    g = load_golden("infonce")
    pred = torch.as_tensor(g[f"{tag}::pred"]).to(DEV).requires_grad_(True)
    tgt = torch.as_tensor(g[f"{tag}::tgt"]).to(DEV)
    mask = torch.as_tensor(g[f"{tag}::mask"]).to(DEV) if int(g[f"{tag}::masked"]) else None
    loss = SequenceInfoNCELoss(float(g[f"{tag}::temperature"]))(pred, tgt, mask=mask)
    loss.backward()
    assert abs(float(loss) - float(g[f"{tag}::loss"])) < 1e-4
    np.testing.assert_allclose(pred.grad.cpu().numpy(), g[f"{tag}::dpred"], atol=2e-6, rtol=1e-3)


def test_infonce_hip_op_at_mosei_size_against_oracle():
    """n = 32 x 49 valid rows of width 300 (the MOSEI text modality): loss and gradient against the pinned oracle."""
    from multibench.models import SequenceInfoNCELoss
    rng = np.random.default_rng(5)
    B, Tn, D = 32, 49, 300
    pred = rng.standard_normal((B, Tn, D)).astype(np.float32)
    tgt = (0.5 * pred + rng.standard_normal((B, Tn, D))).astype(np.float32)
    lens = rng.integers(1, Tn + 1, B)
    lens[0] = Tn
    mask = np.arange(Tn)[None, :] < lens[:, None]
    ref_loss, ref_dp = MO.infonce_loss(pred, tgt, mask, 0.07)
    tp = torch.as_tensor(pred).to(DEV).requires_grad_(True)
    loss = SequenceInfoNCELoss(0.07)(tp, torch.as_tensor(tgt).to(DEV), mask=torch.as_tensor(mask).to(DEV))
    (3.0 * loss).backward()
    assert abs(float(loss) - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    np.testing.assert_allclose(tp.grad.cpu().numpy(), 3.0 * ref_dp, atol=1e-6 * np.abs(ref_dp).max() * 50, rtol=2e-3)


def test_uml_step_with_infonce_critic_matches_reference():
    """UML(infoNCE_loss=True): loss_x (next-step MSE), loss_y (InfoNCE) and parameter gradients against the reference model."""
    from multibench.models import Linear, Transformer, UML
    g = load_golden("infonce")
    z, dx, dy, B, Tn = (int(v) for v in g["model::cfg"])
    m = UML(Linear(dx, z), Linear(dy, z), Transformer(z, z, nhead=5, num_layers=2, conv1d=True, out_last=False, pos_embd=True,
                                                       pos_learnable=False, max_len=128),
            [Linear(z, dx), Linear(z, dy)], modality="xy", infoNCE_loss=True)
    sd = {k[len("model::sd::"):]: torch.as_tensor(g[k]) for k in g.files if k.startswith("model::sd::")}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    T = lambda a: torch.as_tensor(a).to(DEV)
    out = m(T(g["model::x"]), T(g["model::y"]), T(g["model::lx"]), T(g["model::ly"]))
    assert abs(float(out["loss_x"]) - float(g["model::loss_x"])) < 1e-4
    assert abs(float(out["loss_y"]) - float(g["model::loss_y"])) < 2e-4
    (out["loss_x"] + out["loss_y"]).backward()
    n = 0
    for k, p in m.named_parameters():
        if "model::g::" + k in g.files:
            ref = g["model::g::" + k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=3e-4 * max(np.abs(ref).max(), 1e-3), rtol=3e-3, err_msg=k)
            n += 1
    assert n >= 5
