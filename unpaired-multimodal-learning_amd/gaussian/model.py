"""SharedAutoencoder of the Gaussian experiment (reference Gaussian_experiment/model.py:5-59): per-view input heads,
a SHARED encoder/decoder MLP, per-view output heads, reconstruction MSE per view.  Same attribute names and
``state_dict`` keys as the reference; every Linear (with its ReLU) runs through ``LinearFn`` (fp32 MFMA GEMM + bias/ReLU
kernels) and each output head is fused with its MSE (``umlh_seq_mse_forward/_backward`` at T = 1)."""
import torch
from torch import nn

from multibench.encoder import LinearFn
from multibench.models import _DecoderNextStepMSE


class SharedAutoencoder(nn.Module):
    def __init__(self, dim_obs, dim_common, dim_latent):
        super().__init__()
        self.in_head_x = nn.Linear(dim_obs, dim_common)
        self.in_head_y = nn.Linear(dim_obs, dim_common)
        self.shared_encoder = nn.Sequential(nn.Linear(dim_common, dim_latent), nn.ReLU(), nn.Linear(dim_latent, dim_latent))
        self.shared_decoder = nn.Sequential(nn.Linear(dim_latent, dim_latent), nn.ReLU(), nn.Linear(dim_latent, dim_common))
        self.out_head_x = nn.Linear(dim_common, dim_obs)
        self.out_head_y = nn.Linear(dim_common, dim_obs)
        self.loss_fn = nn.MSELoss()

    @staticmethod
    def _lin(x, layer, relu=False):
        return LinearFn.apply(x, layer.weight, layer.bias, relu)

    def _encode(self, v, head):
        z = self._lin(v, head)
        return self._lin(self._lin(z, self.shared_encoder[0], True), self.shared_encoder[2])

    def _view(self, v, in_head, out_head):
        latent = self._encode(v, in_head)
        common = self._lin(self._lin(latent, self.shared_decoder[0], True), self.shared_decoder[2])
        loss, recon = _DecoderNextStepMSE.apply(common.unsqueeze(1), out_head.weight, out_head.bias, v.unsqueeze(1), None)
        return loss, recon.squeeze(1)

    def forward(self, x=None, y=None):
        dev = (x if x is not None else y).device
        if dev.type != "cuda":
            raise RuntimeError("gaussian.SharedAutoencoder runs on the HIP kernels only: move the model and data to the GPU")
        loss_x = loss_y = torch.tensor(0.0, device=dev)
        recon_x = recon_y = None
        if x is not None:
            loss_x, recon_x = self._view(x, self.in_head_x, self.out_head_x)
        if y is not None:
            loss_y, recon_y = self._view(y, self.in_head_y, self.out_head_y)
        return loss_x, loss_y, recon_x, recon_y

    def get_embeddings(self, x=None, y=None):
        ex = self._encode(x, self.in_head_x) if x is not None else None
        ey = self._encode(y, self.in_head_y) if y is not None else None
        return ex, ey
