"""MultiBench UML model: per-modality Linear in/out projections around a shared causal
transformer encoder, next-step prediction loss (reference: MultiBench/models.py:7-277,
assembly MultiBench/main.py:117-121)."""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn.functional as F
from torch import nn


class Linear(nn.Module):
    """nn.Linear wrapper with optional Xavier init and zero bias (models.py:7-35)."""

    def __init__(self, indim, outdim, xavier_init=False):
        super().__init__()
        self.fc = nn.Linear(indim, outdim)
        if xavier_init:
            nn.init.xavier_normal_(self.fc.weight)
            self.fc.bias.data.fill_(0.0)

    def forward(self, x):
        from .encoder import LinearFn
        return LinearFn.apply(x, self.fc.weight, self.fc.bias)


class Transformer(nn.Module):
    """Shared encoder: 1x1 conv (no bias) -> optional positions -> causal TransformerEncoder with
    key-padding mask (models.py:39-127).  The nn modules are parameter containers (same init, same
    ``state_dict`` keys as the reference); the arithmetic runs on the HIP kernels of encoder.py."""

    def __init__(self, n_features, dim, nhead=5, num_layers=5, conv1d=True, out_last=True, pos_embd=False,
                 pos_learnable=False, max_len=128):
        super().__init__()
        self.embed_dim, self.conv1d, self.out_last = dim, conv1d, out_last
        self.pos_embd, self.pos_learnable, self.max_len = pos_embd, pos_learnable, max_len
        if conv1d:
            self.conv = nn.Conv1d(n_features, dim, kernel_size=1, padding=0, bias=False)
        self.transformer = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=dim, nhead=nhead), num_layers=num_layers)
        if pos_embd:
            if pos_learnable:
                self.pos_embedding = nn.Embedding(max_len, dim)
            else:
                pos = torch.arange(max_len).unsqueeze(1)
                div = torch.exp(torch.arange(0, dim, 2) * (-math.log(10000.0) / dim))
                table = torch.zeros(max_len, dim)
                table[:, 0::2] = torch.sin(pos * div)
                table[:, 1::2] = torch.cos(pos * div)
                self.register_buffer("pos_table", table)

    def _dropout_seed(self):
        """Seed of this forward's counter-based dropout masks, drawn from a generator PRIVATE to the module.  The
        reference's dropout consumes the CUDA generator and leaves the global CPU generator to the DataLoader
        samplers (MultiBench/train.py's two shuffled loaders): drawing mask seeds from the global CPU stream
        would shift every batch order after the first epoch.  Seeded once from the process seed, so
        ``torch.manual_seed`` still fixes the masks."""
        g = getattr(self, "_drop_gen", None)
        if g is None:
            g = self._drop_gen = torch.Generator()
            g.manual_seed((torch.initial_seed() * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) % (2 ** 63))
        return int(torch.randint(0, 2 ** 62, (1,), generator=g).item())

    def _prepare(self, x, lengths):
        """Arguments of one encoder pass (models.py:75-127): truncated input, positions, dropout / mask configuration."""
        if type(x) is list:
            x = x[0]
        if not x.is_cuda:
            raise RuntimeError("multibench.Transformer runs on the HIP kernels only: move the model and inputs to the GPU")
        if self.pos_embd and x.size(1) > self.max_len:
            x = x[:, :self.max_len]                                           # models.py:104-106
        T = x.size(1)
        pos = None
        if self.pos_embd:
            pos = self.pos_embedding.weight[:T] if self.pos_learnable else self.pos_table[:T]
        l0 = self.transformer.layers[0]
        p = float(l0.dropout.p) if self.training else 0.0
        seed = self._dropout_seed() if p > 0.0 else 0
        cfg = {"H": l0.self_attn.num_heads, "p": p, "eps": float(l0.norm1.eps), "seed": seed,
               "out_mode": ("last_len" if lengths is not None else "last") if self.out_last else "all"}
        return x, pos, cfg

    def _params(self):
        from .encoder import layer_params
        return [t for layer in self.transformer.layers for t in layer_params(layer)]

    def forward(self, x, lengths=None):
        """models.py:75-127 on the HIP encoder (multibench/encoder.py): conv1d -> positions -> causal
        transformer layers under the key-padding mask -> last valid token / all tokens."""
        from .encoder import EncoderFn
        x, pos, cfg = self._prepare(x, lengths)
        return EncoderFn.apply(x, lengths, cfg, self.conv.weight if self.conv1d else None, pos, *self._params())

    def forward_pair(self, x, x_lengths, y, y_lengths):
        """The two calls `encoder(x_proj, lengths=x_lengths)`, `encoder(y_proj, lengths=y_lengths)` of the alternation step
        (models.py:200,232) as one autograd node: same arithmetic and dropout streams as two `forward` calls in this order,
        the shared parameters' gradients summed once (encoder.EncoderPairFn)."""
        from .encoder import EncoderPairFn
        if self.pos_embd and self.pos_learnable:                              # (a learnable table is a parameter too: keep its
            return self.forward(x, x_lengths), self.forward(y, y_lengths)     #  two gradient contributions on autograd's path)
        x, pos_x, cfg_x = self._prepare(x, x_lengths)
        y, pos_y, cfg_y = self._prepare(y, y_lengths)
        return EncoderPairFn.apply(x, x_lengths, cfg_x, y, y_lengths, cfg_y, self.conv.weight if self.conv1d else None,
                                   pos_x, pos_y, *self._params())


class MSE(nn.Module):
    """Masked mean squared error (models.py:129-143) -- torch-op form, used for shapes the fused
    decoder kernel does not cover."""

    def forward(self, predictions, targets, mask=None):
        if mask is None:
            return (predictions - targets).pow(2).mean()
        m = mask.unsqueeze(-1).expand_as(targets).float()
        return ((predictions - targets) ** 2 * m).sum() / (m.sum() + 1e-8)


class _InfoNCE(torch.autograd.Function):
    """mean CE of normalize(p) normalize(t)^T / temperature against the diagonal (umlh_infonce_forward / _backward)."""

    @staticmethod
    def forward(ctx, p, t, temperature):
        import umlh
        lib = umlh.load_library()
        p, t = (v.detach().to(torch.float32).contiguous() for v in (p, t))
        n, D = p.shape
        dev = p.device
        phat, that = torch.empty_like(p), torch.empty_like(t)
        pnorm, row_loss = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
        probs = torch.empty(n, n, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        vp = lambda v: C.c_void_p(v.data_ptr())
        umlh._lib.check(lib.umlh_infonce_forward(vp(p), vp(t), n, D, float(temperature), vp(phat), vp(that), vp(pnorm), vp(probs), vp(row_loss),
                                                 vp(loss), st), "umlh_infonce_forward")
        ctx.save_for_backward(phat, that, pnorm, probs)
        ctx.temperature = float(temperature)
        return loss

    @staticmethod
    def backward(ctx, g):
        import umlh
        lib = umlh.load_library()
        phat, that, pnorm, probs = ctx.saved_tensors
        n, D = phat.shape
        g = g.detach().to(torch.float32).reshape(1).contiguous()
        dhat, dp = torch.empty_like(phat), torch.empty_like(phat)
        st = C.c_void_p(torch.cuda.current_stream(phat.device).cuda_stream)
        vp = lambda v: C.c_void_p(v.data_ptr())
        umlh._lib.check(lib.umlh_infonce_backward(vp(phat), vp(that), vp(pnorm), vp(probs), vp(g), n, D, ctx.temperature, vp(dhat), vp(dp), st),
                        "umlh_infonce_backward")
        return dp, None, None


class SequenceInfoNCELoss(nn.Module):
    """Contrastive alternative to the MSE critic (models.py:145-175).  The valid rows are selected as the reference does
    (boolean indexing: data formatting, torch); normalisation, the n x n logits, the cross-entropy and their backward run on
    the HIP op.  Targets are model inputs: no gradient flows to them."""

    def __init__(self, temperature=0.07):
        super().__init__()
        self.temperature = temperature

    def forward(self, predictions, targets, mask=None):
        if mask is not None:
            p, t = predictions[mask.bool()], targets[mask.bool()]
        else:
            p, t = predictions.flatten(0, 1), targets.flatten(0, 1)
        if not p.is_cuda:
            raise RuntimeError("SequenceInfoNCELoss runs on the HIP kernels only: move the inputs to the GPU")
        return _InfoNCE.apply(p, t.detach(), self.temperature)


class _DecoderNextStepMSE(torch.autograd.Function):
    """recon = Linear(z); loss = masked MSE(recon[:, :-1], x[:, 1:])  -- one HIP forward + one HIP
    backward (umlh_seq_mse_forward / _backward)."""

    @staticmethod
    def forward(ctx, z, w, b, x, lengths):
        import umlh
        lib = umlh.load_library()
        B, T, Z = z.shape
        D = w.shape[0]
        z, w, b, x = (t.detach().to(torch.float32).contiguous() for t in (z, w, b, x))
        lens = None if lengths is None else lengths.to(device=z.device, dtype=torch.int64).contiguous()
        recon = torch.empty(B, T, D, dtype=torch.float32, device=z.device)
        dres = torch.empty(B * T * D, dtype=torch.float32, device=z.device)
        part = torch.empty(B * T, dtype=torch.float32, device=z.device)
        loss_cnt = torch.empty(2, dtype=torch.float32, device=z.device)
        st = C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        umlh._lib.check(lib.umlh_seq_mse_forward(p(z), p(w), p(b), p(x), p(lens), B, T, Z, D, p(recon), p(dres), p(part),
                                                 p(loss_cnt), st), "umlh_seq_mse_forward")
        ctx.save_for_backward(z, w, dres, loss_cnt)
        ctx.shape = (B, T, Z, D)
        ctx.mark_non_differentiable(recon)
        return loss_cnt[0].clone(), recon

    @staticmethod
    def backward(ctx, g_loss, _g_recon):
        import umlh
        lib = umlh.load_library()
        z, w, dres, loss_cnt = ctx.saved_tensors
        B, T, Z, D = ctx.shape
        g = g_loss.detach().to(torch.float32).reshape(1).contiguous()
        dz = torch.empty(B, T, Z, dtype=torch.float32, device=z.device)
        dw = torch.empty(D, Z, dtype=torch.float32, device=z.device)
        db = torch.empty(D, dtype=torch.float32, device=z.device)
        st = C.c_void_p(torch.cuda.current_stream(z.device).cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        scratch = torch.empty(int(lib.umlh_seq_mse_backward_scratch_floats(B, T, Z, D)), dtype=torch.float32, device=z.device)
        umlh._lib.check(lib.umlh_seq_mse_backward(p(z), p(w), p(dres), p(loss_cnt), p(g), B, T, Z, D, p(dz), p(dw), p(db), p(scratch), st),
                        "umlh_seq_mse_backward")
        return dz, dw, db, None, None


class UML(nn.Module):
    """x -> xproj_in -> shared encoder -> decoders[0] -> next-step loss; same for y with decoders[1]
    (models.py:178-277).  forward returns the reference's dict."""

    def __init__(self, xproj_in, yproj_in, shared_encoder, decoders, modality="x", infoNCE_loss=False):
        super().__init__()
        self.xproj_in, self.yproj_in, self.encoder = xproj_in, yproj_in, shared_encoder
        self.decoders = nn.ModuleList(decoders)
        self.modality = modality
        self.critic = MSE()
        self.infoNCE_loss = infoNCE_loss
        self.y_critic = SequenceInfoNCELoss() if infoNCE_loss else MSE()

    def _critic(self, x, x_proj, z, dec, lengths, use_nce):
        """Decoder + next-step loss of one modality (models.py:202-215 / 234-244) and the logged `diff_next`."""
        if use_nce and x.shape[1] > 1:
            recon = dec(z)
            mask = None
            if lengths is not None:
                mask = torch.arange(x.shape[1], device=x.device).unsqueeze(0) < lengths.unsqueeze(1)
            loss = self.y_critic(recon[:, :-1, :], x[:, 1:, :], mask=mask[:, 1:] if mask is not None else None)
        else:
            loss, recon = _DecoderNextStepMSE.apply(z, dec.fc.weight, dec.fc.bias, x, lengths)
        return recon, loss, (x_proj - z).pow(2).mean()

    def forward(self, x, y, x_lengths=None, y_lengths=None):
        dev = (x if x is not None else y).device
        loss_x = loss_y = torch.zeros((), device=dev)          # a fill kernel: torch.tensor(0.0, device=...) is a blocking host copy
        x_proj = y_proj = zx = zy = x_recon = y_recon = diff_next_x = diff_next_y = None
        if x is not None:
            x = x.unsqueeze(1).float() if x.ndim == 2 else x
            x_proj = self.xproj_in(x)
        if y is not None:
            y = y.unsqueeze(1).float() if y.ndim == 2 else y
            y_proj = self.yproj_in(y)
        # the reference encodes y WITHOUT a key-padding mask (models.py:233) but masks its loss
        if x is not None and y is not None and hasattr(self.encoder, "forward_pair"):
            zx, zy = self.encoder.forward_pair(x_proj, x_lengths, y_proj, None)     # both passes, one autograd node
        else:
            zx = self.encoder(x_proj, lengths=x_lengths) if x is not None else None
            zy = self.encoder(y_proj) if y is not None else None
        if x is not None:
            x_recon, loss_x, diff_next_x = self._critic(x, x_proj, zx, self.decoders[0], x_lengths, False)
        if y is not None:
            y_recon, loss_y, diff_next_y = self._critic(y, y_proj, zy, self.decoders[1], y_lengths, self.infoNCE_loss)
        loss_private = torch.zeros((), device=dev)
        x_private = y_private = None
        if x is not None and y is not None:
            x_private, y_private = x_proj - zx, y_proj - zy
            if x_private.shape == y_private.shape:
                loss_private = ((x_private * y_private).mean([1, 2]) ** 2).sum()
        return {"loss_x": loss_x, "loss_y": loss_y, "loss_private": loss_private, "x_proj": x_proj, "y_proj": y_proj,
                "zx": zx, "zy": zy, "x_recon": x_recon, "y_recon": y_recon, "x_private": x_private, "y_private": y_private,
                "diff_next_x": diff_next_x, "diff_next_y": diff_next_y}

    def get_embedding(self, x, y):
        """Time-averaged encoder outputs (models.py:273-277); the mean over the sequence runs on the HIP reduction
        (``umlh_positions_backward`` sums the middle dimension of a [B,T,Z] block)."""
        x = x.unsqueeze(1).float() if x.ndim == 2 else x
        y = y.unsqueeze(1).float() if y.ndim == 2 else y
        return self._time_mean(self.encoder(self.xproj_in(x))), self._time_mean(self.encoder(self.yproj_in(y)))

    @staticmethod
    def _time_mean(h):
        import umlh
        from umlh._lib import check
        h = h.detach().to(torch.float32).contiguous()
        B, T, Z = h.shape
        out = torch.empty(B, Z, dtype=torch.float32, device=h.device)
        st = C.c_void_p(torch.cuda.current_stream(h.device).cuda_stream)
        check(umlh.load_library().umlh_positions_backward(C.c_void_p(h.data_ptr()), B, T, Z, C.c_void_p(out.data_ptr()), st),
              "umlh_positions_backward")
        return out / T
