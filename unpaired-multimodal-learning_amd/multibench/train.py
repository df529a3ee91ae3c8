"""The MultiBench unpaired alternation loop (reference: MultiBench/train.py:354-399): two
INDEPENDENTLY shuffled loaders zipped, x loss switched off while epoch <= step_k in 'xy' mode,
loss = alpha_x*loss_x + alpha_y*loss_y, one optimizer step per batch pair.  The per-batch
diagnostics of the reference (covariance / svdvals effective rank, wandb, sklearn probes:
train.py:386-389,428-443) are outside the hot path."""
from __future__ import annotations

import torch


def alternation_alphas(epoch, step_k, train_mode, alpha_x=1.0, alpha_y=1.0):
    """[alpha_x, alpha_y] for this epoch (train.py:355-358)."""
    alphas = [alpha_x, alpha_y]
    if epoch <= step_k and train_mode == "xy":
        alphas[0] = 0.0
    return alphas


def _unpack(batch, modality, ds_name, which):
    if ds_name != "mimic":
        return batch[0][modality].float(), batch[1][modality]          # _process_1 layout (get_data.py:418-444)
    return (batch[0].float(), batch[2]) if which == 0 else (batch[1].float(), batch[3])


def train(model, train_mode, train_loader_1, train_loader_2, optimizer, modalities=[0, 2], num_epoch=100, step_k=30,
          ds_name="mosi", eval_config={}, alpha_x=1.0, alpha_y=1.0, capture_embeddings_during_training=False, augment=False,
          debug=False, args=None, device="cuda:0", on_step=None):
    """Returns {'loss_x': [...], 'loss_y': [...], 'loss': [...]} with one entry per batch pair
    (device tensors are read back once at the end)."""
    model.train()
    dev = torch.device(device)
    rec_x, rec_y, rec_l = [], [], []
    for epoch in range(num_epoch):
        alphas = alternation_alphas(epoch, step_k, train_mode, alpha_x, alpha_y)
        for i_batch, (b1, b2) in enumerate(zip(train_loader_1, train_loader_2)):
            x1, l1 = _unpack(b1, modalities[0], ds_name, 0)
            x2, l2 = _unpack(b2, modalities[1], ds_name, 1)
            x1, x2, l1, l2 = x1.to(dev), x2.to(dev), l1.to(dev), l2.to(dev)
            if "x" not in train_mode:
                x1 = None
            if "y" not in train_mode:
                x2 = None
            out = model(x1, x2, l1, l2)
            loss = alphas[0] * out["loss_x"] + alphas[1] * out["loss_y"]
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            rec_x.append(out["loss_x"].detach())
            rec_y.append(out["loss_y"].detach())
            rec_l.append(loss.detach())
            if on_step is not None:
                on_step(epoch, i_batch, out, loss)
    stack = lambda v: torch.stack([t.reshape(()) for t in v]).cpu().tolist() if v else []
    return {"loss_x": stack(rec_x), "loss_y": stack(rec_y), "loss": stack(rec_l)}
