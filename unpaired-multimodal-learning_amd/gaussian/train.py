"""Training loop of the Gaussian experiment (reference Gaussian_experiment/main.py:33-91,132-151): infinite cycling
over one shuffled unpaired loader, ``loss = alpha_x*loss_x + alpha_y*loss_y`` ('xy') or ``loss_x`` ('x'), Adam through the
HIP optimizer kernel, validation reconstruction losses every ``eval_every`` steps.  wandb and the CKA / mutual-kNN
alignment metrics are outside the path."""
import torch
from torch.utils.data import DataLoader

from .data import UnpairedDataset, make_reproducible
from .model import SharedAutoencoder


def train_model_steps(model, data_loader, optimizer, num_steps, val_data_x, val_data_y, device, mode="xy", alpha_x=1.0,
                      alpha_y=1.0, eval_every=1, on_step=None):
    model.train()
    data_iter = iter(data_loader)
    log = {"loss_x": [], "loss_y": [], "loss": [], "val_loss_x": [], "val_loss_y": []}
    for step in range(num_steps):
        try:
            batch = next(data_iter)
        except StopIteration:
            data_iter = iter(data_loader)
            batch = next(data_iter)
        optimizer.zero_grad()
        x, y = batch["x"].to(device), batch["y"].to(device)
        loss_x, loss_y, _, _ = model(x, y)
        loss = alpha_x * loss_x + alpha_y * loss_y if mode == "xy" else loss_x
        loss.backward()
        optimizer.step()
        log["loss_x"].append(loss_x.detach()); log["loss_y"].append(loss_y.detach()); log["loss"].append(loss.detach())
        if eval_every and (step + 1) % eval_every == 0:
            model.eval()
            with torch.no_grad():
                vx, vy, _, _ = model(x=val_data_x, y=val_data_y)      # MSELoss(recon, data) of each view
                log["val_loss_x"].append(vx.detach()); log["val_loss_y"].append(vy.detach())
            model.train()
        if on_step is not None:
            on_step(step, loss_x, loss_y, loss)
    return {k: [float(t) for t in torch.stack(v).cpu()] if v else [] for k, v in log.items()}


def build_run(train_data, train_data2, *, mode="xy", unrelated_info=False, train_num_samples=10000, batch_size=512, seed=0,
              dim_obs=50, dim_common=128, dim_latent=10, lr=1e-3, device="cuda:0"):
    """Dataset / loader / model / optimizer wiring of main.py:132-148."""
    from engine.optimizer.optim import build_optimizer
    n = train_num_samples
    if mode == "xy":
        ysrc = train_data2 if unrelated_info else train_data
        ds = UnpairedDataset(train_data["x"][:n // 2], ysrc["y"][:n - n // 2])
    else:
        ds = UnpairedDataset(train_data["x"], train_data2["y"])
    g = torch.Generator()
    g.manual_seed(42)
    loader = DataLoader(ds, batch_size=batch_size, shuffle=True, drop_last=True, generator=g)
    make_reproducible(seed)
    model = SharedAutoencoder(dim_obs=dim_obs, dim_common=dim_common, dim_latent=dim_latent).to(device)
    return loader, model, build_optimizer(model.parameters(), "adam", lr, 0.0)
