// One post-norm nn.TransformerEncoderLayer of the MultiBench shared encoder (MultiBench/models.py:39-127), forward and
// backward, as ONE C-ABI call each: the whole launch sequence (4 dense layers, attention, 2 x add+LayerNorm, 3 dropouts,
// their backward) is enqueued from C.  The host mirror (multibench/encoder.py) used to make each of the ~25 forward and
// ~35 backward launches of a layer through its own ctypes call: at MOSEI sizes (z = 40, 1600 token rows) the step was
// bound by the Python call rate (7.7 ms for ~1000 launches), not by the GPU.
#include <hip/hip_runtime.h>
#include <cstring>
#include "umlh.h"

namespace {

inline long long ru64(long long x) { return (x + 63) / 64 * 64; }

// Split-K factor of the layer's GEMMs: few 64x64 output tiles and a long reduction -- one tile per CU walking K alone is
// latency-bound, so K is cut until ~2 tiles per CU (same rule as multibench/encoder.py: _splits).
int splits_for(int m, int n, int k) {
    const long long tiles = (long long)((m + 63) / 64) * ((n + 63) / 64);
    if (k < 512 || tiles >= 512) return 1;
    long long s = (512 + tiles - 1) / tiles;
    if (s > 8) s = 8;
    if (s > k / 256) s = k / 256;
    return (int)(s < 1 ? 1 : s);
}

struct Dims { int T, B, Z, H, F; long long M; };

struct Saved {           // per-layer activations kept for the backward (offsets in floats inside `saved`)
    long long qkv, lse, att, s1, mean1, rstd1, x1, hid, s2, mean2, rstd2, total;
};
Saved saved_layout(const Dims& d) {
    Saved s;
    long long o = 0;
    auto take = [&](long long n) { long long r = o; o += ru64(n); return r; };
    s.qkv = take(d.M * 3 * d.Z); s.lse = take((long long)d.B * d.H * d.T); s.att = take(d.M * d.Z);
    s.s1 = take(d.M * d.Z); s.mean1 = take(d.M); s.rstd1 = take(d.M); s.x1 = take(d.M * d.Z);
    s.hid = take(d.M * d.F); s.s2 = take(d.M * d.Z); s.mean2 = take(d.M); s.rstd2 = take(d.M);
    s.total = o;
    return s;
}

long long slab_floats(const Dims& d) {      // largest split-K slab set any GEMM of the layer needs (<= 8 slabs of m*n)
    long long mx = 0;
    auto need = [&](long long m, long long n, long long k) { long long s = splits_for((int)m, (int)n, (int)k); if (s > 1 && s * m * n > mx) mx = s * m * n; };
    need(d.M, 3 * d.Z, d.Z); need(d.M, d.Z, d.Z); need(d.M, d.F, d.Z); need(d.M, d.Z, d.F);          // forward
    need(d.Z, d.F, d.M); need(d.F, d.Z, d.M); need(3 * d.Z, d.Z, d.M); need(d.Z, d.Z, d.M);          // dW
    need(d.M, d.F, d.Z); need(d.M, d.Z, d.F); need(d.M, d.Z, 3 * d.Z);                               // dx
    return mx;
}

struct Scratch { long long a, b, c, big, qkv, slabs, total; };   // a,b,c: [M,Z]; big: [M,F]; qkv: [M,3Z]
Scratch scratch_layout(const Dims& d) {
    Scratch s;
    long long o = 0;
    auto take = [&](long long n) { long long r = o; o += ru64(n); return r; };
    s.a = take(d.M * d.Z); s.b = take(d.M * d.Z); s.c = take(d.M * d.Z); s.big = take(d.M * d.F); s.qkv = take(d.M * 3 * d.Z);
    s.slabs = take(slab_floats(d));
    s.total = o;
    return s;
}

bool dims_ok(const umlh_enc_layer_t* c, Dims& d) {
    if (!c || c->T < 1 || c->T > 128 || c->B < 1 || c->Z < 1 || c->H < 1 || c->Z % c->H != 0 || c->Z / c->H > 64 || c->d_ff < 1) return false;
    if (!(c->p >= 0.f && c->p < 1.f)) return false;
    d.T = c->T; d.B = c->B; d.Z = c->Z; d.H = c->H; d.F = c->d_ff; d.M = (long long)c->T * c->B;
    return true;
}

#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// y[M,N] = act(x[M,K] w[N,K]^T + b)
int linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int relu, float* slabs, void* st) {
    RC(umlh_gemm_f32(x, w, y, M, N, K, K, K, N, 0, 0, nullptr, nullptr, 1.f, splits_for(M, N, K), slabs, st));
    return umlh_bias_act(y, b, M, N, relu, st);
}
// dx[M,K] = dy[M,N] w[N,K];  dw[N,K] = dy^T x;  db[N] = colsum(dy)
int linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int M, int N, int K, float* slabs, void* st) {
    RC(umlh_gemm_f32(dy, x, dw, N, K, M, N, K, K, 1, 1, nullptr, nullptr, 1.f, splits_for(N, K, M), slabs, st));
    RC(umlh_colsum(dy, M, N, db, st));
    return umlh_gemm_f32(dy, w, dx, M, K, N, N, K, K, 0, 1, nullptr, nullptr, 1.f, splits_for(M, K, N), slabs, st);
}

}  // namespace

extern "C" {

uint64_t umlh_encoder_layer_saved_floats(const umlh_enc_layer_t* cfg) {
    Dims d;
    return dims_ok(cfg, d) ? (uint64_t)saved_layout(d).total : 0;
}

uint64_t umlh_encoder_layer_scratch_floats(const umlh_enc_layer_t* cfg) {
    Dims d;
    return dims_ok(cfg, d) ? (uint64_t)scratch_layout(d).total : 0;
}

int umlh_encoder_layer_forward(const umlh_enc_layer_t* cfg, const float* const* P, const float* h_in, const int64_t* lengths,
                               float* saved, float* scratch, float* h_out, void* stream) {
    Dims d;
    if (!dims_ok(cfg, d) || !P || !h_in || !saved || !scratch || !h_out) return UMLH_E_INVALID;
    const Saved S = saved_layout(d);
    const Scratch X = scratch_layout(d);
    const int M = (int)d.M, Z = d.Z, F = d.F;
    const float *in_w = P[0], *in_b = P[1], *out_w = P[2], *out_b = P[3], *w1 = P[4], *b1 = P[5], *w2 = P[6], *b2 = P[7],
                *g1 = P[8], *be1 = P[9], *g2 = P[10], *be2 = P[11];
    float* slabs = scratch + X.slabs;
    const uint64_t sd = cfg->seed;
    // x = norm1(x + dropout1(self_attn(x)))
    RC(linear_fwd(h_in, in_w, in_b, saved + S.qkv, M, 3 * Z, Z, 0, slabs, stream));
    RC(umlh_attention_forward(saved + S.qkv, lengths, d.T, d.B, Z, d.H, cfg->p, sd, saved + S.att, saved + S.lse, stream));
    float* a = scratch + X.a;
    RC(linear_fwd(saved + S.att, out_w, out_b, a, M, Z, Z, 0, slabs, stream));
    RC(umlh_dropout(a, (int64_t)M * Z, cfg->p, sd + 1, stream));
    RC(umlh_add_layernorm_forward(h_in, a, g1, be1, M, Z, cfg->eps, saved + S.s1, saved + S.x1, saved + S.mean1, saved + S.rstd1, stream));
    // x = norm2(x + dropout2(linear2(dropout(relu(linear1(x))))))
    RC(linear_fwd(saved + S.x1, w1, b1, saved + S.hid, M, F, Z, 1, slabs, stream));
    RC(umlh_dropout(saved + S.hid, (int64_t)M * F, cfg->p, sd + 2, stream));
    float* f = scratch + X.b;
    RC(linear_fwd(saved + S.hid, w2, b2, f, M, Z, F, 0, slabs, stream));
    RC(umlh_dropout(f, (int64_t)M * Z, cfg->p, sd + 3, stream));
    return umlh_add_layernorm_forward(saved + S.x1, f, g2, be2, M, Z, cfg->eps, saved + S.s2, h_out, saved + S.mean2, saved + S.rstd2, stream);
}

int umlh_encoder_layer_backward(const umlh_enc_layer_t* cfg, const float* const* P, const float* h_in, const int64_t* lengths,
                                const float* saved, const float* dh_out, float* scratch, float* const* G, float* dh_in, void* stream) {
    Dims d;
    if (!dims_ok(cfg, d) || !P || !h_in || !saved || !dh_out || !scratch || !G || !dh_in) return UMLH_E_INVALID;
    const Saved S = saved_layout(d);
    const Scratch X = scratch_layout(d);
    const int M = (int)d.M, Z = d.Z, F = d.F;
    const float *in_w = P[0], *out_w = P[2], *w1 = P[4], *w2 = P[6], *g1 = P[8], *g2 = P[10];
    float *dinw = G[0], *dinb = G[1], *dow = G[2], *dob = G[3], *dw1 = G[4], *db1 = G[5], *dw2 = G[6], *db2 = G[7],
          *dg1 = G[8], *dbe1 = G[9], *dg2 = G[10], *dbe2 = G[11];
    float* slabs = scratch + X.slabs;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t sd = cfg->seed;
    const size_t mz = sizeof(float) * (size_t)M * Z;
    float *ds2 = scratch + X.a, *df = scratch + X.b, *dx1 = scratch + X.c, *dhid = scratch + X.big, *dqkv = scratch + X.qkv;
    // s2 = x1 + f
    RC(umlh_layernorm_backward(dh_out, saved + S.s2, g2, saved + S.mean2, saved + S.rstd2, M, Z, ds2, dg2, dbe2, stream));
    if (hipMemcpyAsync(df, ds2, mz, hipMemcpyDeviceToDevice, st) != hipSuccess) return UMLH_E_HIP;
    RC(umlh_dropout(df, (int64_t)M * Z, cfg->p, sd + 3, stream));
    RC(linear_bwd(saved + S.hid, w2, df, dhid, dw2, db2, M, Z, F, slabs, stream));
    RC(umlh_dropout(dhid, (int64_t)M * F, cfg->p, sd + 2, stream));
    RC(umlh_relu_backward(saved + S.hid, dhid, (int64_t)M * F, stream));
    RC(linear_bwd(saved + S.x1, w1, dhid, dx1, dw1, db1, M, F, Z, slabs, stream));
    RC(umlh_add_inplace(dx1, ds2, (int64_t)M * Z, stream));                      // residual fan-in at x1
    // s1 = h_in + a
    float *ds1 = scratch + X.a, *da = scratch + X.b, *datt = dhid;               // ds2 / df / dhid are dead from here on
    RC(umlh_layernorm_backward(dx1, saved + S.s1, g1, saved + S.mean1, saved + S.rstd1, M, Z, ds1, dg1, dbe1, stream));
    if (hipMemcpyAsync(da, ds1, mz, hipMemcpyDeviceToDevice, st) != hipSuccess) return UMLH_E_HIP;
    RC(umlh_dropout(da, (int64_t)M * Z, cfg->p, sd + 1, stream));
    RC(linear_bwd(saved + S.att, out_w, da, datt, dow, dob, M, Z, Z, slabs, stream));
    RC(umlh_attention_backward(saved + S.qkv, lengths, saved + S.lse, datt, d.T, d.B, Z, d.H, cfg->p, sd, dqkv, stream));
    RC(linear_bwd(h_in, in_w, dqkv, dh_in, dinw, dinb, M, 3 * Z, Z, slabs, stream));
    return umlh_add_inplace(dh_in, ds1, (int64_t)M * Z, stream);                 // residual fan-in at the layer input
}

}  // extern "C"
