"""The MultiBench shared encoder on HIP kernels: forward AND backward of
``Conv1d(k=1, no bias) -> positions -> N x nn.TransformerEncoderLayer (post-norm, relu, dropout) -> last
valid token`` (reference MultiBench/models.py:39-127) as one ``torch.autograd.Function`` over the C ABI of
``include/umlh.h`` (umlh_gemm_f32, umlh_attention_*, umlh_add_layernorm_*, ...).  torch supplies device
memory and the autograd plumbing around the function; none of the arithmetic.

Activations are token-row matrices ``[T*B, *]`` with row ``m = t*B + b`` (torch's sequence-first layout),
so the math is line for line that of ``torch.nn.TransformerEncoderLayer.forward`` (norm_first=False):

    x = norm1(x + dropout1(self_attn(x)));  x = norm2(x + dropout2(linear2(dropout(relu(linear1(x))))))

Dropout masks are counter-based (a pure function of a per-call seed and the element index), so the
backward regenerates them instead of storing them; train-mode trajectories are therefore not
bit-comparable with torch's Philox masks (the reference's train mode is unpinnable anyway: SURVEY 8(a14))
while eval mode is compared with torch.nn to 1e-5.
"""
from __future__ import annotations

import ctypes as C

import torch

import umlh
from umlh._lib import check


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _st(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _splits(m, n, k):
    """Split-K factor (same rule as csrc/umlh_encoder.cpp: splits_for): these GEMMs have few 64x64 output tiles
    (z <= 300) and a long reduction (1600 token rows or the 2048-wide FFN) whose walk is latency-bound, so K is cut
    until ~768 workgroups with at least 64 reduction rows each."""
    tiles = ((m + 63) // 64) * ((n + 63) // 64)
    if k < 128 or tiles >= 768:
        return 1
    return max(1, min(32, k // 64, -(-768 // tiles)))


def gemm(a, b, m, n, k, lda, ldb, ta, tb, a_rows=None, k_rows=None, alpha=1.0, out=None):
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    sp = _splits(m, n, k)
    slabs = torch.empty(sp, m, n, dtype=torch.float32, device=a.device) if sp > 1 else None
    check(umlh.load_library().umlh_gemm_f32(_p(a), _p(b), _p(out), m, n, k, lda, ldb, n, ta, tb, _p(a_rows), _p(k_rows),
                                            float(alpha), sp, _p(slabs), _st(a.device)), "umlh_gemm_f32")
    return out


def linear_forward(x, w, b, relu=False, a_rows=None, rows=None, out=None):
    """y = act(x w^T + b); x [M,K] (rows optionally gathered by a_rows -> `rows` output rows), w [N,K]."""
    m = x.shape[0] if rows is None else rows
    y = gemm(x, w, m, w.shape[0], w.shape[1], x.shape[1], w.shape[1], 0, 0, a_rows=a_rows, out=out)
    if b is not None or relu:
        check(umlh.load_library().umlh_bias_act(_p(y), _p(b), m, w.shape[0], int(relu), _st(x.device)), "umlh_bias_act")
    return y


def linear_backward(x, w, dy, need_dx=True, has_bias=True, x_rows=None, dx_rows=None, n_dx_rows=None):
    """(dx, dw, db) of y = x w^T + b.  x_rows: y row m used x row x_rows[m]; dx_rows: dx row r takes dy row dx_rows[r]."""
    lib, st = umlh.load_library(), _st(dy.device)
    m, n, k = dy.shape[0], w.shape[0], w.shape[1]
    dw = gemm(dy, x, n, k, m, n, k, 1, 1, k_rows=x_rows)                       # dw[n][k] = sum_m dy[m][n] x[m][k]
    db = None
    if has_bias:
        db = torch.empty(n, dtype=torch.float32, device=dy.device)
        check(lib.umlh_colsum(_p(dy), m, n, _p(db), st), "umlh_colsum")
    dx = None
    if need_dx:
        rows = m if n_dx_rows is None else n_dx_rows
        dx = gemm(dy, w, rows, k, n, n, k, 0, 1, a_rows=dx_rows)               # dx[m][k] = sum_n dy[m][n] w[n][k]
    return dx, dw, db


def _dropout_(x, p, seed):
    if p > 0.0:
        check(umlh.load_library().umlh_dropout(_p(x), x.numel(), float(p), C.c_uint64(seed & (2 ** 64 - 1)), _st(x.device)), "umlh_dropout")
    return x


def _add_(y, x):
    check(umlh.load_library().umlh_add_inplace(_p(y), _p(x), y.numel(), _st(y.device)), "umlh_add_inplace")
    return y


def _add_ln(x, r, gamma, beta, eps):
    m, n = x.shape
    s, y = torch.empty_like(x), torch.empty_like(x)
    mean = torch.empty(m, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(umlh.load_library().umlh_add_layernorm_forward(_p(x), _p(r), _p(gamma), _p(beta), m, n, float(eps), _p(s), _p(y),
                                                         _p(mean), _p(rstd), _st(x.device)), "umlh_add_layernorm_forward")
    return s, y, mean, rstd


def _ln_bwd(dy, s, gamma, mean, rstd):
    m, n = dy.shape
    ds = torch.empty_like(dy)
    dg = torch.empty(n, dtype=torch.float32, device=dy.device)
    db = torch.empty_like(dg)
    check(umlh.load_library().umlh_layernorm_backward(_p(dy), _p(s), _p(gamma), _p(mean), _p(rstd), m, n, _p(ds), _p(dg), _p(db),
                                                      _st(dy.device)), "umlh_layernorm_backward")
    return ds, dg, db


N_LAYER_PARAMS = 12   # in_w, in_b, out_w, out_b, w1, b1, w2, b2, g1, be1, g2, be2


def _layer_cfg(T, B, Z, H, dff, p, eps, seed):
    from umlh._lib import EncLayer
    return EncLayer(int(T), int(B), int(Z), int(H), int(dff), float(p), float(eps), int(seed) & (2 ** 64 - 1), None)


class _Plan:
    """umlh_encoder_plan_*: the layer stack bound to fixed buffers (one torch allocation) whose launch sequences replay from
    HIP graphs.  A plan serves one forward/backward pair at a time: EncoderFn leases it for the lifetime of the autograd
    node (the saved activations live in the plan) and the pool hands out / creates another one meanwhile."""

    def __init__(self, lc, n_layers, lp, has_lens, dev):
        lib = umlh.load_library()
        n = int(lib.umlh_encoder_plan_floats(C.byref(lc), n_layers))
        if n == 0:
            raise umlh.UmlhError(f"encoder layer shape outside the kernels' envelope: T={lc.T} Z={lc.Z} H={lc.H}")
        self.ws = torch.empty(n, dtype=torch.float32, device=dev)
        self.params = lp                                   # keeps the tensors whose addresses the graphs hold alive
        h = C.c_void_p()
        check(lib.umlh_encoder_plan_create(C.byref(lc), n_layers, _ptr_array(lp), int(has_lens), _p(self.ws), C.byref(h)), "umlh_encoder_plan_create")
        self.handle = h
        offs = (C.c_uint64 * 6)()
        check(lib.umlh_encoder_plan_offsets(h, offs), "umlh_encoder_plan_offsets")
        M, Z, B = lc.T * lc.B, lc.Z, lc.B
        view = lambda o, k: self.ws[int(o):int(o) + k]
        self.h0, self.h_last = view(offs[0], M * Z).view(M, Z), view(offs[2], M * Z).view(M, Z)
        self.lens = view(offs[1], 2 * B).view(torch.int64) if has_lens else None
        self.dh_out, self.dh0 = view(offs[3], M * Z).view(M, Z), view(offs[4], M * Z).view(M, Z)
        self.grads = view(offs[5], sum(t.numel() for t in lp))
        self.busy = False

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            umlh.load_library().umlh_encoder_plan_destroy(h)


class _Lease:
    def __init__(self, plan):
        self.plan = plan
        plan.busy = True

    def __del__(self):
        self.plan.busy = False


_PLANS = {}                 # configuration key -> plans (dict order = least recently used first)
_PLAN_POOL_MAX = 8          # plans kept per configuration (each holds the stack's saved activations: ~0.1 GB at MOSEI sizes)
_PLAN_KEYS_MAX = 16         # configurations kept: beyond that the least recently used idle ones are dropped


def _lease_plan(key, make):
    pool = _PLANS.pop(key, [])
    _PLANS[key] = pool                                     # most recently used
    if len(_PLANS) > _PLAN_KEYS_MAX:
        for k in [k for k, v in _PLANS.items() if k != key and not any(pl.busy for pl in v)][:len(_PLANS) - _PLAN_KEYS_MAX]:
            del _PLANS[k]
    for pl in pool:
        if not pl.busy:
            return _Lease(pl)
    pl = make()
    if len(pool) < _PLAN_POOL_MAX:
        pool.append(pl)
    return _Lease(pl)


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _enc_forward(x, lengths, cfg, conv_w, pos, lp):
    """One pass of the shared encoder: x [B,T,F] -> conv -> (+pos) -> layer stack -> output rows.  Returns (out, state);
    `state` is what `_enc_backward` needs (it holds the plan lease: the saved activations live in the plan)."""
    lib = umlh.load_library()
    dev = x.device
    B, T, F = x.shape
    H, p, eps, seed = cfg["H"], cfg["p"], cfg["eps"], cfg["seed"]
    st = _st(dev)
    x2d = _f32(x).reshape(B * T, F)
    lens = None if lengths is None else lengths.to(device=dev, dtype=torch.int64).contiguous()
    ar_b = torch.arange(B, device=dev, dtype=torch.int64)
    ar_t = torch.arange(T, device=dev, dtype=torch.int64)
    rows_tb = (ar_b.unsqueeze(0) * T + ar_t.unsqueeze(1)).reshape(-1).contiguous()      # token row m=(t,b) -> source row b*T+t
    rows_bt = (ar_t.unsqueeze(0) * B + ar_b.unsqueeze(1)).reshape(-1).contiguous()      # source row (b,t) -> token row t*B+b
    M = T * B
    n_layers = len(lp) // N_LAYER_PARAMS
    Z = conv_w.shape[0] if conv_w is not None else F
    dff = lp[4].shape[0] if n_layers else 0
    lease = plan = None
    if n_layers:
        # the layer stack runs on a leased plan: fixed buffers, one HIP-graph launch per direction
        key = (dev.index, T, B, Z, H, dff, float(p), float(eps), n_layers, lens is not None, tuple(t.data_ptr() for t in lp))
        lease = _lease_plan(key, lambda: _Plan(_layer_cfg(T, B, Z, H, dff, p, eps, 0), n_layers, lp, lens is not None, dev))
        plan = lease.plan
    if conv_w is not None:
        cw = _f32(conv_w).reshape(conv_w.shape[0], -1)
        h = linear_forward(x2d, cw, None, a_rows=rows_tb, rows=M, out=None if plan is None else plan.h0)     # [M, Z]
    else:
        cw = None
        h = torch.empty(M, F, dtype=torch.float32, device=dev) if plan is None else plan.h0
        check(lib.umlh_gather_rows(_p(x2d), _p(rows_tb), M, F, _p(h), 0, st), "umlh_gather_rows")
    if pos is not None:
        check(lib.umlh_add_positions(_p(h), _p(_f32(pos)), T, B, Z, st), "umlh_add_positions")
    if plan is not None:
        if lens is not None:
            plan.lens.copy_(lens)
        check(lib.umlh_encoder_plan_forward(plan.handle, C.c_uint64(int(seed) & (2 ** 64 - 1)), st), "umlh_encoder_plan_forward")
        h = plan.h_last
    mode = cfg["out_mode"]
    if mode == "all":
        idx, n_out = rows_bt, M
    elif mode == "last_len":
        idx, n_out = ((lens - 1) * B + ar_b).contiguous(), B
    else:
        idx, n_out = ((T - 1) * B + ar_b).contiguous(), B
    out = torch.empty(n_out, Z, dtype=torch.float32, device=dev)
    check(lib.umlh_gather_rows(_p(h), _p(idx), n_out, Z, _p(out), 0, st), "umlh_gather_rows")
    state = ((B, T, F, Z, M), x2d, rows_tb, rows_bt, idx, cw, pos is not None and pos.requires_grad, lease, x.requires_grad)
    return (out.reshape(B, T, Z) if mode == "all" else out), state


def _enc_backward(state, g_out):
    """Backward of one pass.  Returns (dx, dconv, dpos, flat) -- `flat` is the PLAN's flat gradient buffer of all layer
    parameters (rewritten by the plan's next backward: the caller copies or sums it), None without layers."""
    lib = umlh.load_library()
    (B, T, F, Z, M), x2d, rows_tb, rows_bt, idx, cw, need_dpos, lease, need_dx = state
    dev = g_out.device
    st = _st(dev)
    g = _f32(g_out).reshape(-1, Z)
    plan = None if lease is None else lease.plan
    dh = torch.zeros(M, Z, dtype=torch.float32, device=dev) if plan is None else plan.dh_out.zero_()
    check(lib.umlh_gather_rows(_p(g), _p(idx), g.shape[0], Z, _p(dh), 1, st), "umlh_gather_rows(scatter)")
    flat = None
    if plan is not None:
        check(lib.umlh_encoder_plan_backward(plan.handle, st), "umlh_encoder_plan_backward")
        flat, dh = plan.grads, plan.dh0
    dpos = None
    if need_dpos:                                        # learnable position table only
        dpos = torch.empty(T, Z, dtype=torch.float32, device=dev)
        check(lib.umlh_positions_backward(_p(dh), T, B, Z, _p(dpos), st), "umlh_positions_backward")
    dconv = dx = None
    if cw is not None:
        dx, dconv, _ = linear_backward(x2d, cw, dh, need_dx=need_dx, has_bias=False, x_rows=rows_tb, dx_rows=rows_bt, n_dx_rows=B * T)
        dconv = dconv.reshape(cw.shape[0], cw.shape[1], 1)
    elif need_dx:
        dx = torch.empty(B * T, F, dtype=torch.float32, device=dev)
        check(lib.umlh_gather_rows(_p(dh), _p(rows_bt), B * T, F, _p(dx), 0, st), "umlh_gather_rows")
    if dx is not None:
        dx = dx.reshape(B, T, F)
    return dx, dconv, dpos, flat


def _split_flat(flat, lp):
    grads, o = [], 0
    for t in lp:
        grads.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    return grads


class EncoderFn(torch.autograd.Function):
    """x [B,T,F] -> conv -> (+pos) -> layers -> output rows.  ``cfg`` = dict(H, p, eps, seed, out_mode)
    with out_mode 'last_len' | 'last' | 'all'; ``params`` = conv_w | None, pos [T,Z] | None, then 12 tensors per layer."""

    @staticmethod
    def forward(ctx, x, lengths, cfg, conv_w, pos, *layer_params):
        lp = [_f32(t) for t in layer_params]
        out, ctx.state = _enc_forward(x, lengths, cfg, conv_w, pos, lp)
        ctx.lp = lp
        return out

    @staticmethod
    def backward(ctx, g_out):
        dx, dconv, dpos, flat = _enc_backward(ctx.state, g_out)
        grads = _split_flat(flat.clone(), ctx.lp) if flat is not None else []     # the plan's buffer is rewritten by its next backward
        return (dx, None, None, dconv, dpos, *grads)


class EncoderPairFn(torch.autograd.Function):
    """Both modality passes of the alternation step through the SHARED encoder as one autograd node (MultiBench/models.py:
    200,232 call the same module twice): the two passes run exactly as two EncoderFn calls would, but the gradients of the
    shared parameters are summed here by one kernel over the two flat buffers instead of 62 per-tensor accumulations in
    autograd's AccumulateGrad."""

    @staticmethod
    def forward(ctx, x, lx, cfg_x, y, ly, cfg_y, conv_w, pos_x, pos_y, *layer_params):
        lp = [_f32(t) for t in layer_params]
        ox, ctx.sx = _enc_forward(x, lx, cfg_x, conv_w, pos_x, lp)
        oy, ctx.sy = _enc_forward(y, ly, cfg_y, conv_w, pos_y, lp)
        ctx.lp = lp
        return ox, oy

    @staticmethod
    def backward(ctx, gx, gy):
        dx, dcx, dpx, fx = _enc_backward(ctx.sx, gx)
        dy, dcy, dpy, fy = _enc_backward(ctx.sy, gy)
        grads = _split_flat(fx + fy, ctx.lp) if fx is not None else []
        dconv = None if dcx is None else dcx + dcy
        return (dx, None, None, dy, None, None, dconv, dpx, dpy, *grads)


class LinearFn(torch.autograd.Function):
    """y = act(x w^T + b) over the trailing dimension (the per-modality in/out projections, models.py:7-35; with
    ``relu`` the Linear+ReLU pairs of the Gaussian toy's shared encoder/decoder)."""

    @staticmethod
    def forward(ctx, x, w, b, relu=False):
        x2 = _f32(x).reshape(-1, x.shape[-1])
        w2, b2 = _f32(w), (None if b is None else _f32(b))
        y = linear_forward(x2, w2, b2, relu=relu)
        ctx.save_for_backward(x2, w2, y if relu else None)
        ctx.has_bias, ctx.shape, ctx.need_dx, ctx.relu = b is not None, x.shape, x.requires_grad, relu
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w2, y = ctx.saved_tensors
        dy = _f32(gy).reshape(-1, w2.shape[0])
        if ctx.relu:
            dy = dy.clone()
            check(umlh.load_library().umlh_relu_backward(_p(y), _p(dy), dy.numel(), _st(dy.device)), "umlh_relu_backward")
        dx, dw, db = linear_backward(x2, w2, dy, need_dx=ctx.need_dx, has_bias=ctx.has_bias)
        return (None if dx is None else dx.reshape(ctx.shape)), dw, db, None


def layer_params(layer):
    """The 12 tensors of one nn.TransformerEncoderLayer in EncoderFn's order."""
    a = layer.self_attn
    return [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, layer.linear1.weight, layer.linear1.bias,
            layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias]
