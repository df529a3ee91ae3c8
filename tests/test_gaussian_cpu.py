"""CPU: the Gaussian toy's host-side data generation reproduces the reference's tensors under the same seeds
(golden: oracle/make_golden_gaussian.py ran Gaussian_experiment/data.py itself)."""
import numpy as np

from conftest import load_golden


def test_generate_data_matches_reference():
    from gaussian.data import UnpairedDataset, generate_data
    g = load_golden("gaussian_toy")
    cfg = {"seed": 42, "num_samples": 600, "dim_c": 10, "dim_x": 5, "dim_y": 5, "dim_obs": 50, "noise_std": 0.09,
           "attenuate_x": True, "attenuation": 0.05, "shared_latent_distribution_type": "gaussian"}
    d = generate_data(cfg)
    np.testing.assert_array_equal(d["x"].numpy(), g["data_x"])
    np.testing.assert_array_equal(d["y"].numpy(), g["data_y"])
    v = generate_data(dict(cfg, seed=43, num_samples=64, attenuate_x=False))
    np.testing.assert_array_equal(v["x"].numpy(), g["val_x"])
    lap = generate_data(dict(cfg, seed=44, num_samples=8, shared_latent_distribution_type="laplace"))
    np.testing.assert_array_equal(lap["y"].numpy(), g["laplace_y"])
    ds = UnpairedDataset(d["x"][:300], d["y"][:250])
    assert len(ds) == 300 and np.array_equal(ds[299]["y"].numpy(), g["data_y"][299 % 250])
