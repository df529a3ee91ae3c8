// One post-norm nn.TransformerEncoderLayer of the MultiBench shared encoder (MultiBench/models.py:39-127), forward and
// backward, as ONE C-ABI call each, 7 + 16 launches per layer:
//   forward   qkv GEMM (+bias) | attention | out-proj GEMM | [bias, dropout, +residual, LayerNorm] | linear1 GEMM (+bias, relu,
//             dropout) | linear2 GEMM (split-K slabs) | [slab sum, bias, dropout, +residual, LayerNorm]
//   backward  LayerNorm rows (ds, dropout(ds)) | LayerNorm column partials (+ db of the next dense layer) | dW slabs | dx GEMM with
//             the relu / dropout mask in the epilogue | ... | ONE multi-reduce of every dW slab set and column partial.
// Round-1 form was ~16 + ~40 launches (separate bias / dropout / relu / add / copy kernels, a reduce per split-K GEMM, and
// single-block column sums of 10-30 us each): 7.0 ms per alternation step at MOSEI sizes, 5.6 ms of it kernel time
// (profiles/r02_multibench_kernel_stats.md).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <new>
#include "umlh.h"
#include "umlh_enc.h"

namespace {

inline long long ru64(long long x) { return (x + 63) / 64 * 64; }

// Split-K factor: these GEMMs have few 64x64 output tiles and are bound by the latency of the K walk (one staged chunk per
// iteration), so K is cut until ~768 workgroups (3 per CU, co-resident) with at least 64 reduction rows each.
int splits_for(int m, int n, int k) {
    static const int target = [] { const char* e = getenv("UMLH_ENC_SPLIT_WGS"); return e ? atoi(e) : 768; }();
    const long long tiles = (long long)((m + 63) / 64) * ((n + 63) / 64);
    if (k < 128 || tiles >= target) return 1;
    long long s = (target + tiles - 1) / tiles;
    if (s > k / 64) s = k / 64;
    if (s > 32) s = 32;
    return (int)(s < 1 ? 1 : s);
}
inline int row_chunk(long long M) { long long c = (M + 63) / 64; return (int)(c < 64 ? 64 : c); }   // <= 64 row chunks of >= 64 rows

struct Dims { int T, B, Z, H, F, R, chunk; long long M; };

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

long long slab_need(long long m, long long n, long long k) { return (long long)splits_for((int)m, (int)n, (int)k) * m * n; }

// a..f: [M,Z]; big: [M,F]; qkv: [M,3Z]; slabs: split-K slabs of the activation GEMMs (consumed by the next launch);
// pw*: split-K slabs of the four weight gradients and pcol: row-chunk partials of the column sums (all live until the
// layer's multi-reduce)
struct Scratch { long long a, b, c, e, f, big, qkv, slabs, pw2, pw1, pwo, pwin, pcol, total; };
Scratch scratch_layout(const Dims& d) {
    Scratch s;
    long long o = 0;
    auto take = [&](long long n) { long long r = o; o += ru64(n); return r; };
    const long long M = d.M, Z = d.Z, F = d.F;
    s.a = take(M * Z); s.b = take(M * Z); s.c = take(M * Z); s.e = take(M * Z); s.f = take(M * Z); s.big = take(M * F); s.qkv = take(M * 3 * Z);
    long long mx = 0;
    for (long long v : {slab_need(M, 3 * Z, Z), slab_need(M, Z, Z), slab_need(M, F, Z), slab_need(M, Z, F), slab_need(M, Z, 3 * Z)}) if (v > mx) mx = v;
    s.slabs = take(mx);
    s.pw2 = take(slab_need(Z, F, M)); s.pw1 = take(slab_need(F, Z, M)); s.pwo = take(slab_need(Z, Z, M)); s.pwin = take(slab_need(3 * Z, Z, M));
    s.pcol = take((long long)d.R * (6 * Z + F + 3 * Z));
    s.total = o;
    return s;
}

bool dims_ok(const umlh_enc_layer_t* c, Dims& d) {
    if (!c || c->T < 1 || c->T > 128 || c->B < 1 || c->Z < 1 || c->H < 1 || c->Z % c->H != 0 || c->Z / c->H > 64 || c->d_ff < 1) return false;
    if (!(c->p >= 0.f && c->p < 1.f)) return false;
    d.T = c->T; d.B = c->B; d.Z = c->Z; d.H = c->H; d.F = c->d_ff; d.M = (long long)c->T * c->B;
    d.chunk = row_chunk(d.M); d.R = (int)((d.M + d.chunk - 1) / d.chunk);
    return true;
}

#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
#define HC(expr) do { if ((expr) != 0) return UMLH_E_HIP; } while (0)

Epilogue epi_none() { Epilogue e; memset(&e, 0, sizeof(e)); return e; }
void epi_dropout(Epilogue& e, float p, uint64_t seed, const unsigned long long* seed_ptr) {
    e.thresh = umlh_enc_drop_thresh(p); e.inv_keep = umlh_enc_drop_inv_keep(p); e.seed = seed; e.seed_ptr = seed_ptr; if (e.thresh) e.on = 1;
}

// y[M,N] = epilogue(x[M,K] w[N,K]^T)           (defer: raw slabs left in `slabs`, *ns of them)
int dense_fwd(const float* x, const float* w, float* y, int M, int N, int K, const Epilogue& e, float* slabs, int defer, int* ns, hipStream_t st) {
    return umlh_gemm_f32_epi(x, w, y, M, N, K, K, K, 0, 0, &e, splits_for(M, N, K), slabs, defer, ns, st);
}
// dx[M,K] = epilogue(dy[M,N] w[N,K])
int dense_bwd_x(const float* dy, const float* w, float* dx, int M, int N, int K, const Epilogue& e, float* slabs, int defer, int* ns, hipStream_t st) {
    return umlh_gemm_f32_epi(dy, w, dx, M, K, N, N, K, 0, 1, &e, splits_for(M, K, N), slabs, defer, ns, st);
}
// dw[N,K] = dy[M,N]^T x[M,K] as split-K slabs (one slab: written to dw directly); appends the reduction to `red`
int dense_bwd_w(const float* dy, const float* x, float* dw, int M, int N, int K, float* slabs, MultiReduceArgs& red, hipStream_t st) {
    int ns = 1;
    const int sp = splits_for(N, K, M);
    if (sp == 1) return umlh_gemm_f32_epi(dy, x, dw, N, K, M, N, K, 1, 1, nullptr, 1, nullptr, 0, &ns, st);
    RC(umlh_gemm_f32_epi(dy, x, nullptr, N, K, M, N, K, 1, 1, nullptr, sp, slabs, 1, &ns, st));
    red.d[red.count++] = ReduceDesc{slabs, dw, (long long)N * K, (long long)N * K, ns, 0};
    return UMLH_OK;
}

// Second stream of a layer's backward: the weight-gradient work (dW slabs, column partials, the multi-reduce) does not feed
// the activation-gradient chain, so it runs beside it -- forked after each producer, joined at the end of the layer.  In a
// captured sequence the fork/join become parallel branches of the graph.
struct SideCtx { hipStream_t s2; hipEvent_t ev[5]; };

int layer_backward_impl(const umlh_enc_layer_t* cfg, const float* const* P, const float* h_in, const int64_t* lengths, const float* saved,
                        const float* dh_out, float* scratch, float* const* G, float* dh_in, hipStream_t st, const SideCtx* side);

int layer_backward_impl(const umlh_enc_layer_t* cfg, const float* const* P, const float* h_in, const int64_t* lengths, const float* saved,
                        const float* dh_out, float* scratch, float* const* G, float* dh_in, hipStream_t st, const SideCtx* side) {
    Dims d;
    if (!dims_ok(cfg, d) || !P || !h_in || !saved || !dh_out || !scratch || !G || !dh_in) return UMLH_E_INVALID;
    const Saved S = saved_layout(d);
    const Scratch X = scratch_layout(d);
    const int M = (int)d.M, Z = d.Z, F = d.F, R = d.R;
    const float *in_w = P[0], *out_w = P[2], *w1 = P[4], *w2 = P[6], *g1 = P[8], *g2 = P[10];
    float *dinw = G[0], *dinb = G[1], *dow = G[2], *dob = G[3], *dw1 = G[4], *db1 = G[5], *dw2 = G[6], *db2 = G[7],
          *dg1 = G[8], *dbe1 = G[9], *dg2 = G[10], *dbe2 = G[11];
    float* slabs = scratch + X.slabs;
    hipStream_t sw = side ? side->s2 : st;          // stream of the weight-gradient work
    auto fork = [&](int i) -> int {
        if (!side) return 0;
        if (hipEventRecord(side->ev[i], st) != hipSuccess || hipStreamWaitEvent(side->s2, side->ev[i], 0) != hipSuccess) return UMLH_E_HIP;
        return 0;
    };
    const uint64_t sd = cfg->seed;
    const unsigned long long* sp = reinterpret_cast<const unsigned long long*>(cfg->seed_device);
    const long long mz = (long long)M * Z;
    float *ds2 = scratch + X.a, *df = scratch + X.b, *dx1 = scratch + X.c, *ds1 = scratch + X.e, *dhid = scratch + X.big,
          *dqkv = scratch + X.qkv;
    float* pc = scratch + X.pcol;                      // column partials [R][width] each
    float *p_g2 = pc, *p_be2 = p_g2 + (long long)R * Z, *p_b2 = p_be2 + (long long)R * Z, *p_g1 = p_b2 + (long long)R * Z,
          *p_be1 = p_g1 + (long long)R * Z, *p_ob = p_be1 + (long long)R * Z, *p_b1 = p_ob + (long long)R * Z,
          *p_inb = p_b1 + (long long)R * F;
    MultiReduceArgs red;
    memset(&red, 0, sizeof(red));
    auto col = [&](float* part, float* dst, int width) { red.d[red.count++] = ReduceDesc{part, dst, (long long)width, (long long)width, R, 0}; };
    int ns = 1;
    // h_out = norm2(s2), s2 = x1 + dropout3(f): ds2, df = dropout3(ds2)
    HC(umlh_enc_launch_layernorm_bwd_rows_fused(dh_out, 1, 0, nullptr, nullptr, saved + S.s2, g2, saved + S.mean2, saved + S.rstd2, M, Z,
                                                ds2, df, cfg->p, sd + 3, sp, st));
    RC(fork(0));
    HC(umlh_enc_launch_ln_cols_partial(dh_out, saved + S.s2, saved + S.mean2, saved + S.rstd2, df, M, Z, d.chunk, p_g2, p_be2, p_b2, sw));
    col(p_g2, dg2, Z); col(p_be2, dbe2, Z); col(p_b2, db2, Z);
    // f = hid w2^T + b2, hid = dropout2(relu(x1 w1^T + b1))
    RC(dense_bwd_w(df, saved + S.hid, dw2, M, Z, F, scratch + X.pw2, red, sw));
    Epilogue e = epi_none();
    e.gate = saved + S.hid; e.on = 1;
    epi_dropout(e, cfg->p, sd + 2, sp);
    RC(dense_bwd_x(df, w2, dhid, M, Z, F, e, slabs, 0, &ns, st));
    RC(fork(1));
    RC(dense_bwd_w(dhid, saved + S.x1, dw1, M, F, Z, scratch + X.pw1, red, sw));
    HC(umlh_enc_launch_colsum_partial(dhid, M, F, d.chunk, p_b1, sw));
    col(p_b1, db1, F);
    RC(dense_bwd_x(dhid, w1, nullptr, M, F, Z, epi_none(), slabs, 1, &ns, st));
    // x1 = norm1(s1), s1 = h_in + dropout1(a): dx1 = slabs + ds2 (residual fan-in), ds1, da = dropout1(ds1)
    float* da = scratch + X.f;                         // (not df's buffer: the side stream may still be reading df)
    HC(umlh_enc_launch_layernorm_bwd_rows_fused(slabs, ns, mz, ds2, dx1, saved + S.s1, g1, saved + S.mean1, saved + S.rstd1, M, Z,
                                                ds1, da, cfg->p, sd + 1, sp, st));
    RC(fork(2));
    HC(umlh_enc_launch_ln_cols_partial(dx1, saved + S.s1, saved + S.mean1, saved + S.rstd1, da, M, Z, d.chunk, p_g1, p_be1, p_ob, sw));
    col(p_g1, dg1, Z); col(p_be1, dbe1, Z); col(p_ob, dob, Z);
    // a = att out_w^T + out_b
    RC(dense_bwd_w(da, saved + S.att, dow, M, Z, Z, scratch + X.pwo, red, sw));
    float* datt = ds2;                                 // ds2 is dead
    RC(dense_bwd_x(da, out_w, datt, M, Z, Z, epi_none(), slabs, 0, &ns, st));
    HC(umlh_enc_launch_attention_bwd(saved + S.qkv, lengths, saved + S.lse, datt, d.T, d.B, Z, d.H, cfg->p, sd, sp, dqkv, st));
    // qkv = h_in in_w^T + in_b
    RC(fork(3));
    RC(dense_bwd_w(dqkv, h_in, dinw, M, 3 * Z, Z, scratch + X.pwin, red, sw));
    HC(umlh_enc_launch_colsum_partial(dqkv, M, 3 * Z, d.chunk, p_inb, sw));
    col(p_inb, dinb, 3 * Z);
    e = epi_none();
    e.add = ds1; e.on = 1;                             // residual fan-in at the layer input
    RC(dense_bwd_x(dqkv, in_w, dh_in, M, 3 * Z, Z, e, slabs, 0, &ns, st));
    HC(umlh_enc_launch_multi_reduce(&red, sw));
    if (side && (hipEventRecord(side->ev[4], side->s2) != hipSuccess || hipStreamWaitEvent(st, side->ev[4], 0) != hipSuccess)) return UMLH_E_HIP;   // join
    return UMLH_OK;
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
    hipStream_t st = (hipStream_t)stream;
    const uint64_t sd = cfg->seed;
    const unsigned long long* sp = reinterpret_cast<const unsigned long long*>(cfg->seed_device);
    const long long mz = (long long)M * Z;
    int ns = 1;
    // x1 = norm1(x + dropout1(self_attn(x)))
    Epilogue e = epi_none();
    e.bias = in_b; e.on = 1;
    RC(dense_fwd(h_in, in_w, saved + S.qkv, M, 3 * Z, Z, e, slabs, 0, &ns, st));
    HC(umlh_enc_launch_attention_fwd(saved + S.qkv, lengths, d.T, d.B, Z, d.H, cfg->p, sd, sp, saved + S.att, saved + S.lse, st));
    RC(dense_fwd(saved + S.att, out_w, nullptr, M, Z, Z, epi_none(), slabs, 1, &ns, st));
    e = epi_none();
    e.bias = out_b; e.add = h_in; e.on = 1;
    epi_dropout(e, cfg->p, sd + 1, sp);
    HC(umlh_enc_launch_add_layernorm_fused(slabs, ns, mz, &e, g1, be1, M, Z, cfg->eps, saved + S.s1, saved + S.x1, saved + S.mean1,
                                           saved + S.rstd1, st));
    // h_out = norm2(x1 + dropout2(linear2(dropout(relu(linear1(x1))))))
    e = epi_none();
    e.bias = b1; e.relu = 1; e.on = 1;
    epi_dropout(e, cfg->p, sd + 2, sp);
    RC(dense_fwd(saved + S.x1, w1, saved + S.hid, M, F, Z, e, slabs, 0, &ns, st));
    RC(dense_fwd(saved + S.hid, w2, nullptr, M, Z, F, epi_none(), slabs, 1, &ns, st));
    e = epi_none();
    e.bias = b2; e.add = saved + S.x1; e.on = 1;
    epi_dropout(e, cfg->p, sd + 3, sp);
    HC(umlh_enc_launch_add_layernorm_fused(slabs, ns, mz, &e, g2, be2, M, Z, cfg->eps, saved + S.s2, h_out, saved + S.mean2,
                                           saved + S.rstd2, st));
    return UMLH_OK;
}

int umlh_encoder_layer_backward(const umlh_enc_layer_t* cfg, const float* const* P, const float* h_in, const int64_t* lengths,
                                const float* saved, const float* dh_out, float* scratch, float* const* G, float* dh_in, void* stream) {
    return layer_backward_impl(cfg, P, h_in, lengths, saved, dh_out, scratch, G, dh_in, (hipStream_t)stream, nullptr);
}

// the whole layer stack in one call: layer li reads (li ? h + (li-1)*M*Z : h0), writes h + li*M*Z and saved + li*saved_floats;
// its dropout streams start at cfg->seed + 7919*li
int umlh_encoder_stack_forward(const umlh_enc_layer_t* cfg, int32_t n_layers, const float* const* P, const float* h0,
                               const int64_t* lengths, float* saved, float* scratch, float* h, void* stream) {
    Dims d;
    if (!dims_ok(cfg, d) || n_layers < 0 || !P || !h0 || !saved || !scratch || !h) return UMLH_E_INVALID;
    const long long nsv = saved_layout(d).total, mz = d.M * d.Z;
    umlh_enc_layer_t lc = *cfg;
    for (int li = 0; li < n_layers; ++li) {
        lc.seed = cfg->seed + 7919ull * (uint64_t)li;
        RC(umlh_encoder_layer_forward(&lc, P + 12 * li, li ? h + (li - 1) * mz : h0, lengths, saved + li * nsv, scratch, h + li * mz, stream));
    }
    return UMLH_OK;
}

// G: 12 gradient pointers per layer; dh_out: gradient of the last layer's output; dh: two [M,Z] ping-pong buffers; the gradient
// of h0 is left in dh0
static int stack_backward_impl(const umlh_enc_layer_t* cfg, int32_t n_layers, const float* const* P, const float* h0,
                               const int64_t* lengths, const float* saved, const float* h, const float* dh_out, float* scratch,
                               float* const* G, float* dh, float* dh0, hipStream_t st, const SideCtx* side) {
    Dims d;
    if (!dims_ok(cfg, d) || n_layers < 1 || !P || !h0 || !saved || !h || !dh_out || !scratch || !G || !dh || !dh0) return UMLH_E_INVALID;
    const long long nsv = saved_layout(d).total, mz = d.M * d.Z;
    umlh_enc_layer_t lc = *cfg;
    const float* g = dh_out;
    for (int li = n_layers - 1; li >= 0; --li) {
        lc.seed = cfg->seed + 7919ull * (uint64_t)li;
        float* out = li == 0 ? dh0 : dh + (li & 1) * mz;
        RC(layer_backward_impl(&lc, P + 12 * li, li ? h + (li - 1) * mz : h0, lengths, saved + li * nsv, g, scratch, G + 12 * li, out, st, side));
        g = out;
    }
    return UMLH_OK;
}

int umlh_encoder_stack_backward(const umlh_enc_layer_t* cfg, int32_t n_layers, const float* const* P, const float* h0,
                                const int64_t* lengths, const float* saved, const float* h, const float* dh_out, float* scratch,
                                float* const* G, float* dh, float* dh0, void* stream) {
    return stack_backward_impl(cfg, n_layers, P, h0, lengths, saved, h, dh_out, scratch, G, dh, dh0, (hipStream_t)stream, nullptr);
}

// ---- encoder plan: the stack on fixed buffers, replayed from HIP graphs ----
struct umlh_enc_plan_s {
    umlh_enc_layer_t cfg;
    int n_layers, has_lengths, device;
    const float** P;
    float** G;
    float* ws;
    long long o_h0, o_lens, o_seed, o_hs, o_saved, o_scratch, o_dhout, o_dhtmp, o_dh0, o_grads, total;
    hipStream_t cap;
    SideCtx side;                // second stream + fork/join events of the backward
    bool has_side;
    hipGraphExec_t exec[2];
    int calls[2];
};

static void plan_layout(const Dims& d, int n, umlh_enc_plan_s& p) {
    long long o = 0;
    auto take = [&](long long k) { long long r = o; o += ru64(k); return r; };
    const long long mz = d.M * d.Z, Z = d.Z, F = d.F;
    p.o_h0 = take(mz); p.o_lens = take(2LL * d.B); p.o_seed = take(2);
    p.o_hs = take((long long)n * mz); p.o_saved = take((long long)n * saved_layout(d).total); p.o_scratch = take(scratch_layout(d).total);
    p.o_dhout = take(mz); p.o_dhtmp = take(2 * mz); p.o_dh0 = take(mz);
    p.o_grads = take((long long)n * (3 * Z * Z + 3 * Z + Z * Z + Z + F * Z + F + Z * F + Z + 4 * Z));
    p.total = o;
}

uint64_t umlh_encoder_plan_floats(const umlh_enc_layer_t* cfg, int32_t n_layers) {
    Dims d;
    if (!dims_ok(cfg, d) || n_layers < 1) return 0;
    umlh_enc_plan_s p;
    plan_layout(d, n_layers, p);
    return (uint64_t)p.total;
}

int umlh_encoder_plan_create(const umlh_enc_layer_t* cfg, int32_t n_layers, const float* const* P, int32_t has_lengths,
                             float* workspace, umlh_enc_plan_t* out) {
    Dims d;
    if (!dims_ok(cfg, d) || n_layers < 1 || !P || !workspace || !out) return UMLH_E_INVALID;
    umlh_enc_plan_s* p = new (std::nothrow) umlh_enc_plan_s();
    if (!p) return UMLH_E_INVALID;
    plan_layout(d, n_layers, *p);
    p->cfg = *cfg;
    p->cfg.seed = 0;
    p->n_layers = n_layers; p->has_lengths = has_lengths; p->ws = workspace;
    p->cfg.seed_device = reinterpret_cast<const uint64_t*>(workspace + p->o_seed);
    p->P = new const float*[12 * n_layers];
    p->G = new float*[12 * n_layers];
    const long long Z = d.Z, F = d.F;
    const long long sizes[12] = {3 * Z * Z, 3 * Z, Z * Z, Z, F * Z, F, Z * F, Z, Z, Z, Z, Z};
    long long o = p->o_grads;
    for (int i = 0; i < 12 * n_layers; ++i) { p->P[i] = P[i]; p->G[i] = workspace + o; o += sizes[i % 12]; }
    p->exec[0] = p->exec[1] = nullptr;
    p->calls[0] = p->calls[1] = 0;
    if (hipGetDevice(&p->device) != hipSuccess || hipStreamCreateWithFlags(&p->cap, hipStreamNonBlocking) != hipSuccess) {
        delete[] p->P; delete[] p->G; delete p;
        return UMLH_E_HIP;
    }
    p->has_side = hipStreamCreateWithFlags(&p->side.s2, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 5 && p->has_side; ++i)
        if (hipEventCreateWithFlags(&p->side.ev[i], hipEventDisableTiming) != hipSuccess) p->has_side = false;   // (falls back to one stream)
    *out = p;
    return UMLH_OK;
}

int umlh_encoder_plan_offsets(umlh_enc_plan_t p, uint64_t offsets[6]) {
    if (!p || !offsets) return UMLH_E_INVALID;
    Dims d;
    dims_ok(&p->cfg, d);
    offsets[0] = (uint64_t)p->o_h0; offsets[1] = (uint64_t)p->o_lens; offsets[2] = (uint64_t)(p->o_hs + (long long)(p->n_layers - 1) * d.M * d.Z);
    offsets[3] = (uint64_t)p->o_dhout; offsets[4] = (uint64_t)p->o_dh0; offsets[5] = (uint64_t)p->o_grads;
    return UMLH_OK;
}

static int plan_enqueue(umlh_enc_plan_t p, int dir, hipStream_t st) {
    float* w = p->ws;
    const int64_t* lens = p->has_lengths ? reinterpret_cast<const int64_t*>(w + p->o_lens) : nullptr;
    if (dir == 0) return umlh_encoder_stack_forward(&p->cfg, p->n_layers, p->P, w + p->o_h0, lens, w + p->o_saved, w + p->o_scratch, w + p->o_hs, st);
    // Forking the weight-gradient work onto the side stream is OFF by default: measured on MI355X / ROCm 7.2 the forked graph
    // is slower than the single chain (3.15 vs 2.90 ms per step at z = 40; its launch also costs ~1 ms more host time).
    static const bool no_fork = [] { const char* e = getenv("UMLH_ENC_FORK"); return !(e && atoi(e) == 1); }();
    return stack_backward_impl(&p->cfg, p->n_layers, p->P, w + p->o_h0, lens, w + p->o_saved, w + p->o_hs, w + p->o_dhout,
                               w + p->o_scratch, p->G, w + p->o_dhtmp, w + p->o_dh0, st, (p->has_side && !no_fork) ? &p->side : nullptr);
}

// call 0: plain launches (also sets the kernels' attributes, which a capture cannot); call 1: capture + instantiate; then replay
static int plan_run(umlh_enc_plan_t p, int dir, hipStream_t st) {
    static const bool no_graph = [] { const char* e = getenv("UMLH_ENC_GRAPH"); return e && atoi(e) == 0; }();
    const int call = p->calls[dir]++;
    if (call == 0 || no_graph) return plan_enqueue(p, dir, st);
    if (!p->exec[dir]) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(p->cap, hipStreamCaptureModeThreadLocal) != hipSuccess) return UMLH_E_HIP;
        const int rc = plan_enqueue(p, dir, p->cap);
        const hipError_t e = hipStreamEndCapture(p->cap, &graph);
        if (rc || e != hipSuccess || !graph) { if (graph) hipGraphDestroy(graph); return rc ? rc : UMLH_E_HIP; }
        const hipError_t ei = hipGraphInstantiate(&p->exec[dir], graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ei != hipSuccess) { p->exec[dir] = nullptr; return UMLH_E_HIP; }
    }
    return hipGraphLaunch(p->exec[dir], st) == hipSuccess ? UMLH_OK : UMLH_E_HIP;
}

int umlh_encoder_plan_forward(umlh_enc_plan_t p, uint64_t seed, void* stream) {
    if (!p) return UMLH_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    HC(umlh_enc_launch_set_u64(reinterpret_cast<unsigned long long*>(p->ws + p->o_seed), (unsigned long long)seed, st));
    return plan_run(p, 0, st);
}

int umlh_encoder_plan_backward(umlh_enc_plan_t p, void* stream) {
    if (!p) return UMLH_E_INVALID;
    return plan_run(p, 1, (hipStream_t)stream);
}

void umlh_encoder_plan_destroy(umlh_enc_plan_t p) {
    if (!p) return;
    for (int i = 0; i < 2; ++i) if (p->exec[i]) hipGraphExecDestroy(p->exec[i]);
    hipStreamDestroy(p->cap);
    if (p->has_side) { for (int i = 0; i < 5; ++i) hipEventDestroy(p->side.ev[i]); hipStreamDestroy(p->side.s2); }
    delete[] p->P; delete[] p->G;
    delete p;
}

}  // extern "C"
