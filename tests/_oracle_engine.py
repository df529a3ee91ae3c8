"""Test double: an engine with HeadEngine's grad_step / apply_update / train_step
surface computed by the CPU oracle, so the data-parallel layer (umlh.dp) can be
exercised with gloo on machines without a GPU.  Lives under tests/ only."""
import numpy as np
import torch

from oracle import uml_oracle as O


class OracleEngine:
    def __init__(self, state: O.HeadState, optimizer="adamw", wd=0.0):
        self.state = state
        self.opt = O.OptState(optimizer, wd)
        nh = state.w_head.size
        npj = state.w_proj.size if state.w_proj is not None else 0
        self.flat = torch.zeros(nh + npj + 2 + 12, dtype=torch.float32)   # [.. | g_scales(2) | scalars(UMLH_N_SCALARS)]
        self.nh, self.npj = nh, npj

    @staticmethod
    def _np(b):
        if b is None:
            return None, None, 0, 0
        x, y = b.feats.numpy(), b.labels.numpy()
        if b.index is not None:
            i = b.index.numpy()
            x, y = x[i], y[i]
        n = x.shape[0]
        return x, y, n, int(b.global_rows or n)

    def grad_step(self, img, txt, alpha=1.0, img_alpha=1.0):
        xi, yi, ni, gi = self._np(img)
        xt, yt, nt, gt = self._np(txt)
        # local partial of the GLOBAL mean: weight each modality by local_rows / global_rows
        so = O.step_grads(self.state, xi, yi, xt, yt, alpha * (nt / gt if nt else 0.0), img_alpha * (ni / gi if ni else 0.0))
        f = self.flat
        f.zero_()
        f[:self.nh] = torch.from_numpy(so.grads["w_head"].ravel())
        if self.npj:
            f[self.nh:self.nh + self.npj] = torch.from_numpy(so.grads["w_proj"].ravel())
        o = self.nh + self.npj
        if self.state.learnable_temp:
            f[o] = float(so.grads.get("img_scale", 0.0))
            f[o + 1] = float(so.grads.get("txt_scale", 0.0))
        f[o + 2 + 0] = so.loss_img * (ni / gi if ni else 0.0)
        f[o + 2 + 1] = so.loss_txt * (nt / gt if nt else 0.0)
        f[o + 2 + 2] = so.acc_img * (ni / gi if ni else 0.0)
        f[o + 2 + 3] = so.acc_txt * (nt / gt if nt else 0.0)
        return f

    def apply_update(self, lr, step, scalars_out=None):
        f = self.flat.numpy()
        grads = {"w_head": f[:self.nh].reshape(self.state.w_head.shape).copy()}
        if self.npj:
            grads["w_proj"] = f[self.nh:self.nh + self.npj].reshape(self.state.w_proj.shape).copy()
        o = self.nh + self.npj
        if self.state.learnable_temp:
            grads["img_scale"] = np.float32(f[o])
            grads["txt_scale"] = np.float32(f[o + 1])
        O.optimizer_step(self.state, grads, self.opt, lr)
        if scalars_out is not None:
            scalars_out.copy_(self.flat[o + 2:])
        return self.flat[o + 2:]

    def train_step(self, img, txt, lr, step, alpha=1.0, img_alpha=1.0, scalars_out=None):
        self.grad_step(img, txt, alpha, img_alpha)
        return self.apply_update(lr, step, scalars_out)
