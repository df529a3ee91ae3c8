"""CPU ORACLE for the MultiBench alternation step  --  TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference-owned arithmetic of `MultiBench/` on the hot path
(SURVEY.md section 8(a14)); the shared transformer encoder itself is third-party
torch.nn and is pinned through golden vectors of the reference's own forward
(oracle/make_golden_multibench.py), not restated here.
"""
from __future__ import annotations

import numpy as np


def length_mask(T: int, lengths) -> np.ndarray:
    """mask[b, t] = t < lengths[b]   (MultiBench/models.py:206,238)."""
    return np.arange(T)[None, :] < np.asarray(lengths)[:, None]


def masked_mse(pred: np.ndarray, target: np.ndarray, mask=None) -> float:
    """models.MSE.forward (MultiBench/models.py:129-143): mean over all elements, or
    sum(sq * mask) / (sum(mask expanded) + 1e-8)."""
    sq = (pred.astype(np.float64) - target.astype(np.float64)) ** 2
    if mask is None:
        return float(sq.mean())
    m = np.broadcast_to(mask[..., None], sq.shape).astype(np.float64)
    return float((sq * m).sum() / (m.sum() + 1e-8))


def decoder_next_step_loss(z: np.ndarray, w: np.ndarray, b: np.ndarray, x: np.ndarray, lengths):
    """x_recon = Linear(z) (models.py:202,234); loss = MSE(x_recon[:, :-1], x[:, 1:], mask[:, 1:])
    (models.py:213,243).  Returns (loss, x_recon, d loss/d z, d loss/d w, d loss/d b)."""
    B, T, _ = z.shape
    recon = z @ w.T + b
    mask = length_mask(T, lengths)[:, 1:] if lengths is not None else np.ones((B, T - 1), bool)
    loss = masked_mse(recon[:, :-1], x[:, 1:], mask)
    m = np.broadcast_to(mask[..., None], recon[:, :-1].shape).astype(np.float64)
    d_recon = np.zeros(recon.shape, np.float64)
    d_recon[:, :-1] = 2.0 * (recon[:, :-1].astype(np.float64) - x[:, 1:]) * m / (m.sum() + 1e-8)
    dz = d_recon @ w.astype(np.float64)
    dw = d_recon.reshape(-1, w.shape[0]).T @ z.reshape(-1, z.shape[2]).astype(np.float64)
    db = d_recon.reshape(-1, w.shape[0]).sum(0)
    return loss, recon, dz, dw, db


def alternation_alphas(epoch: int, step_k: int, train_mode: str, alpha_x: float, alpha_y: float):
    """alphas of MultiBench/train.py:355-358: in 'xy' mode the x loss is switched off while
    epoch <= step_k (train only on y first)."""
    ax = 0.0 if (epoch <= step_k and train_mode == "xy") else alpha_x
    return ax, alpha_y


def infonce_loss(pred: np.ndarray, target: np.ndarray, mask=None, temperature: float = 0.07):
    """models.SequenceInfoNCELoss.forward (MultiBench/models.py:145-175): the valid (batch, time) rows of predictions and
    targets are L2-normalised (F.normalize: x / max(|x|, 1e-12)), logits = p t^T / temperature, labels = arange(n), loss =
    mean cross-entropy.  Returns (loss, d loss / d predictions) with the gradient scattered back to the [B, T, D] layout."""
    sel = np.ones(pred.shape[:2], bool) if mask is None else np.asarray(mask).astype(bool)
    p, t = pred[sel].astype(np.float64), target[sel].astype(np.float64)
    n = p.shape[0]
    pn = np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-12)
    tn = np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-12)
    ph, th = p / pn, t / tn
    logits = ph @ th.T / temperature
    mx = logits.max(1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(logits - mx).sum(1))
    loss = float((lse - np.diag(logits)).mean())
    dl = (np.exp(logits - lse[:, None]) - np.eye(n)) / n
    dph = dl @ th / temperature
    dp = (dph - ph * (ph * dph).sum(1, keepdims=True)) / pn
    out = np.zeros(pred.shape, np.float64)
    out[sel] = dp
    return loss, out
