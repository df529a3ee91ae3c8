"""GPU: the Gaussian toy (shared autoencoder over two unpaired views) on the HIP ops replays the reference's own
training steps: same batches (seeded loader), per-step losses, first-step gradients of all 16 parameter tensors,
weights after 6 Adam steps, validation losses and embeddings."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_training_steps_replay_reference():
    from gaussian.train import build_run, train_model_steps
    g = load_golden("gaussian_toy")
    data = {"x": torch.from_numpy(g["data_x"]), "y": torch.from_numpy(g["data_y"])}
    loader, model, opt = build_run(data, data, mode="xy", train_num_samples=600, batch_size=128, seed=0, device=DEV)
    for k, v in model.state_dict().items():                       # same init as the reference under make_reproducible(0)
        np.testing.assert_array_equal(v.cpu().numpy(), g["init/" + k])
    grads1 = {}

    def on_step(step, lx, ly, loss):
        if step == 0:
            grads1.update({k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()})
    vx, vy = torch.from_numpy(g["val_x"]).to(DEV), torch.from_numpy(g["val_y"]).to(DEV)
    log = train_model_steps(model, loader, opt, 6, vx, vy, DEV, mode="xy", alpha_x=1.0, alpha_y=0.5, eval_every=6, on_step=on_step)
    np.testing.assert_allclose(log["loss_x"], g["loss_x"], rtol=2e-5)
    np.testing.assert_allclose(log["loss_y"], g["loss_y"], rtol=2e-5)
    for k, v in grads1.items():
        ref = g["grad1/" + k]
        np.testing.assert_allclose(v, ref, atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=1e-3, err_msg=k)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["final/" + k], atol=2e-5, rtol=1e-4, err_msg=k)
    assert abs(log["val_loss_x"][-1] - float(g["val_loss_x"])) < 2e-5 * float(g["val_loss_x"])
    assert abs(log["val_loss_y"][-1] - float(g["val_loss_y"])) < 2e-5 * float(g["val_loss_y"])
    model.eval()
    ex, ey = model.get_embeddings(x=vx, y=vy)
    np.testing.assert_allclose(ex.detach().cpu().numpy(), g["emb_x"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(ey.detach().cpu().numpy(), g["emb_y"], atol=2e-4, rtol=1e-4)


def test_x_only_mode_trains():
    from gaussian.data import generate_data
    from gaussian.train import build_run, train_model_steps
    cfg = {"seed": 1, "num_samples": 512, "dim_c": 10, "dim_x": 5, "dim_y": 5, "dim_obs": 50, "noise_std": 0.09,
           "attenuate_x": True, "attenuation": 0.05, "shared_latent_distribution_type": "gaussian"}
    d = generate_data(cfg)
    loader, model, opt = build_run(d, d, mode="x", train_num_samples=512, batch_size=128, seed=0, lr=3e-3, device=DEV)
    log = train_model_steps(model, loader, opt, 60, d["x"][:64].to(DEV), d["y"][:64].to(DEV), DEV, mode="x", eval_every=0)
    assert log["loss_x"][-1] < 0.7 * log["loss_x"][0]
    assert log["loss"] == log["loss_x"]
