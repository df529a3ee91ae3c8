"""CPU: the MultiBench oracle (masked next-step MSE + decoder grads, alternation schedule)
against golden vectors of the reference's own models.py forward/backward."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import multibench_oracle as MO


@pytest.mark.parametrize("tag", ["z20_sin", "z40_learn", "z20_nopos"])
def test_decoder_loss_and_grads_from_reference_latents(tag):
    g = load_golden("mb_" + tag)
    ax, ay = g["alpha"]
    for mod, dec, a in (("x", 0, ax), ("y", 1, ay)):
        z, x, lens = g["z" + mod], g[mod], g["l" + mod]
        w, b = g[f"sd::decoders.{dec}.fc.weight"], g[f"sd::decoders.{dec}.fc.bias"]
        loss, recon, dz, dw, db = MO.decoder_next_step_loss(z, w, b, x, lens)
        assert abs(loss - float(g["loss_" + mod])) < 1e-5
        np.testing.assert_allclose(recon, g[mod + "_recon"], atol=1e-5)
        np.testing.assert_allclose(a * dw, g[f"g::decoders.{dec}.fc.weight"], atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(a * db, g[f"g::decoders.{dec}.fc.bias"], atol=1e-6, rtol=1e-4)


def test_masked_mse_edge_cases():
    p = np.arange(12, dtype=np.float32).reshape(1, 4, 3)
    t = np.zeros_like(p)
    assert abs(MO.masked_mse(p, t) - float((p.astype(np.float64) ** 2).mean())) < 1e-9
    m = np.array([[True, False, False, False]])
    assert abs(MO.masked_mse(p, t, m) - float((p[0, 0] ** 2).sum() / 3)) < 1e-6
    assert MO.masked_mse(p, t, np.zeros((1, 4), bool)) == 0.0            # empty mask: 0 / 1e-8


def test_alternation_schedule():
    assert MO.alternation_alphas(0, 10, "xy", 1.0, 2.0) == (0.0, 2.0)
    assert MO.alternation_alphas(10, 10, "xy", 1.0, 2.0) == (0.0, 2.0)    # <= step_k
    assert MO.alternation_alphas(11, 10, "xy", 1.0, 2.0) == (1.0, 2.0)
    assert MO.alternation_alphas(0, 10, "x", 1.0, 2.0) == (1.0, 2.0)      # gating only in 'xy' mode
    assert MO.alternation_alphas(0, -1, "xy", 0.5, 1.0) == (0.5, 1.0)     # step_k = -1: never gated
    from multibench.train import alternation_alphas
    for e, k, mode in [(0, 10, "xy"), (11, 10, "xy"), (3, 2, "y"), (0, -1, "xy")]:
        assert tuple(alternation_alphas(e, k, mode, 0.7, 1.3)) == MO.alternation_alphas(e, k, mode, 0.7, 1.3)


def test_infonce_oracle_matches_reference_golden():
    """oracle.multibench_oracle.infonce_loss (restating MultiBench/models.py:145-175) against the reference's own
    SequenceInfoNCELoss outputs and autograd gradient (tests/golden/infonce.npz, oracle/make_golden_infonce.py)."""
    from conftest import load_golden
    g = load_golden("infonce")
    for tag in ("small", "masked", "wide", "one_row_seqs"):
        mask = g[f"{tag}::mask"] if int(g[f"{tag}::masked"]) else None
        loss, dp = MO.infonce_loss(g[f"{tag}::pred"], g[f"{tag}::tgt"], mask, float(g[f"{tag}::temperature"]))
        assert abs(loss - float(g[f"{tag}::loss"])) < 2e-5, tag
        np.testing.assert_allclose(dp, g[f"{tag}::dpred"], atol=2e-6, rtol=2e-4, err_msg=tag)
