"""Synthetic unpaired two-view data of the Gaussian experiment (reference Gaussian_experiment/data.py:7-61,
dataset.py:3-17, utils.py:5-11): host-side generation with torch's CPU generator, so the same seeds give the
reference's tensors."""
import random

import numpy as np
import torch
import torch.distributions as dist
from torch.utils.data import Dataset


def make_reproducible(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def sample_latent(num_samples, dim, dist_type="gaussian", **kwargs):
    if dist_type == "gaussian":
        z = torch.randn(num_samples, dim)
        return z - z.mean(0)
    if dist_type == "gmm":
        k = kwargs.get("n_clusters", 10)
        centroids = torch.randn(k, dim) * 5.0
        ids = torch.randint(0, k, (num_samples,))
        z = centroids[ids] + torch.randn(num_samples, dim) * 0.5
        return z - z.mean(0)
    if dist_type == "laplace":
        return dist.Laplace(torch.tensor([0.0]), torch.tensor([1.0])).sample((num_samples, dim)).squeeze(-1)
    raise ValueError(f"Unsupported distribution type: {dist_type}")


def generate_data(cfg):
    """x = A_c (w * theta_c) + A_x theta_x + eps;  y = B_c theta_c + B_y theta_y + eps  (data.py:29-61; draw order kept)."""
    make_reproducible(cfg["seed"])
    n = cfg["num_samples"]
    theta_c = sample_latent(n, cfg["dim_c"], dist_type=cfg["shared_latent_distribution_type"], n_clusters=10)
    theta_x, theta_y = torch.randn(n, cfg["dim_x"]), torch.randn(n, cfg["dim_y"])
    noise_x = torch.randn(n, cfg["dim_obs"]) * cfg["noise_std"]
    noise_y = torch.randn(n, cfg["dim_obs"]) * cfg["noise_std"]
    a_c, a_x = torch.randn(cfg["dim_obs"], cfg["dim_c"]), torch.randn(cfg["dim_obs"], cfg["dim_x"])
    b_c, b_y = torch.randn(cfg["dim_obs"], cfg["dim_c"]), torch.randn(cfg["dim_obs"], cfg["dim_y"])
    if cfg["attenuate_x"]:
        att = torch.full((cfg["dim_c"],), cfg["attenuation"])
        att[:int(cfg["dim_c"] * 0.1)] = 1.0
        theta_c_x = theta_c * att
    else:
        theta_c_x = theta_c
    return {"x": theta_c_x @ a_c.T + theta_x @ a_x.T + noise_x, "y": theta_c @ b_c.T + theta_y @ b_y.T + noise_y}


class UnpairedDataset(Dataset):
    def __init__(self, data_x, data_y):
        self.data_x, self.data_y = data_x, data_y
        self.len_x, self.len_y = len(data_x), len(data_y)
        self.length = max(self.len_x, self.len_y)

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return {"x": self.data_x[idx % self.len_x], "y": self.data_y[idx % self.len_y]}
