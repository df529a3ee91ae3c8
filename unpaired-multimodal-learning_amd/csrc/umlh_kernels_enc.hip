// MultiBench shared encoder (MultiBench/models.py:39-127: conv1d k=1 -> positions -> 5 x post-norm
// nn.TransformerEncoderLayer(d_model=z, nhead=5, dim_feedforward=2048, relu, dropout 0.1) with a
// causal + key-padding mask), forward AND backward, as fp32 HIP kernels for gfx950.
//
// The dense layers go through gemm_f32 (fp32 MFMA, umlh_kernels_f32.hip); this file holds what sits
// between the GEMMs.  Sizes are tiny (z <= 300, T <= 128, B = 32: <= 4096 token rows), so these are
// latency/HBM-bound row kernels -- one wave per token row, LDS only where rows are shared (attention);
// nothing here is reshaped to reach MFMA.
#include "umlh_common.h"
#include <atomic>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// sum_s p[s*stride], s = 0 .. ns-1 in that order; the loads go out 8 at a time (a plain `v += p[s*stride]` loop waits for
// each load before issuing the next: 25 slabs cost 25 memory latencies)
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, long long stride, int ns) {
    float v = 0.f;
    for (int s = 0; s < ns; s += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (s + u < ns) ? p[(size_t)(s + u) * stride] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    return v;
}

// y[m][n] = act(y[m][n] + b[n])
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ y, const float* __restrict__ b, long long total,
                                                       int N, int relu) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float v = y[i] + (b ? b[i % N] : 0.f);
    y[i] = relu ? fmaxf(v, 0.f) : v;
}

// dy[i] = y[i] > 0 ? dy[i] : 0
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, float* __restrict__ dy, long long total) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total && !(y[i] > 0.f)) dy[i] = 0.f;
}

// x[i] = keep ? x[i] / (1 - p) : 0   (same call on the gradient in the backward pass: same mask)
__global__ __launch_bounds__(256) void dropout_kernel(float* __restrict__ x, long long total, unsigned thresh, float inv_keep,
                                                      unsigned long long seed) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) x[i] = keep_elem(seed, (unsigned long long)i, thresh) ? x[i] * inv_keep : 0.f;
}

// y[i] += x[i]   (gradient fan-in of a residual branch)
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long long total) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < total) y[i] += x[i];
}

// out[n] = sum_m x[m][n]      one block per 64 columns, 4 row groups, fixed summation order
constexpr int CG = 16;     // row groups of the column reductions (1024 threads: these are latency-bound row walks)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, int M, int N, float* __restrict__ out) {
    __shared__ float sh[CG][64];
    const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, g = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N) s = slab_sum(x + (size_t)g * N + c, (long long)CG * N, (M - g + CG - 1) / CG);
    sh[g][l] = s;
    __syncthreads();
    if (g == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CG; ++k) t += sh[k][l];
        out[c] = t;
    }
}

// s = x + r (r may be NULL); y = LayerNorm(s) * gamma + beta; one wave per row
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                int M, int N, float eps, float* __restrict__ s_out,
                                                                float* __restrict__ y, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const size_t o = (size_t)m * N;
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) {
        float v = x[o + n] + (r ? r[o + n] : 0.f);
        s_out[o + n] = v;
        sum += v;
    }
    const float mean = wave_sum(sum) / (float)N;
    float var = 0.f;
    for (int n = lane; n < N; n += 64) {
        float d = s_out[o + n] - mean;          // written by this lane above
        var += d * d;
    }
    const float rstd = rsqrtf(wave_sum(var) / (float)N + eps);
    for (int n = lane; n < N; n += 64) y[o + n] = (s_out[o + n] - mean) * rstd * gamma[n] + beta[n];
    if (lane == 0) { mean_out[m] = mean; rstd_out[m] = rstd; }
}

// ds = rstd * (g*dy - mean_n(g*dy) - xhat * mean_n(g*dy*xhat)),  xhat = (s - mean) * rstd; one wave per row
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int M, int N,
                                                                 float* __restrict__ ds) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const size_t o = (size_t)m * N;
    const float mu = mean[m], rs = rstd[m];
    float a = 0.f, b = 0.f;
    for (int n = lane; n < N; n += 64) {
        float g = gamma[n] * dy[o + n], xh = (s[o + n] - mu) * rs;
        a += g;
        b += g * xh;
    }
    a = wave_sum(a) / (float)N;
    b = wave_sum(b) / (float)N;
    for (int n = lane; n < N; n += 64) {
        float g = gamma[n] * dy[o + n], xh = (s[o + n] - mu) * rs;
        ds[o + n] = rs * (g - a - xh * b);
    }
}

// dgamma[n] = sum_m dy*xhat, dbeta[n] = sum_m dy    (column kernel, fixed order)
__global__ __launch_bounds__(1024) void layernorm_bwd_cols_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 int M, int N, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
    __shared__ float sh[2][CG][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, l = threadIdx.x & 63;
    float a = 0.f, b = 0.f;
    if (c < N)
        for (int m = g; m < M; m += CG) {
            float d = dy[(size_t)m * N + c];
            a += d * (s[(size_t)m * N + c] - mean[m]) * rstd[m];
            b += d;
        }
    sh[0][g][l] = a; sh[1][g][l] = b;
    __syncthreads();
    if (g == 0 && c < N) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int k = 0; k < CG; ++k) { ta += sh[0][k][l]; tb += sh[1][k][l]; }
        dgamma[c] = ta;
        dbeta[c] = tb;
    }
}

// x[(t*B + b)*Z + z] += pos[t*Z + z]
__global__ __launch_bounds__(256) void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos, int T, int B, int Z) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * B * Z) return;
    const int z = (int)(i % Z), t = (int)(i / ((long long)B * Z));
    x[i] += pos[(size_t)t * Z + z];
}

// dpos[t*Z + z] = sum_b dx[(t*B + b)*Z + z]
__global__ __launch_bounds__(256) void pos_grad_kernel(const float* __restrict__ dx, int T, int B, int Z, float* __restrict__ dpos) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)T * Z) return;
    const int z = (int)(i % Z), t = (int)(i / Z);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[((size_t)t * B + b) * Z + z];
    dpos[i] = s;
}

// out[j][:] = x[idx[j]][:]   /   dx[idx[j]][:] = dout[j][:]  (dx zero-filled by the caller; idx unique)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, int n, int Z,
                                                          float* __restrict__ out, int scatter) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * Z) return;
    const int j = (int)(i / Z), z = (int)(i % Z);
    if (scatter) out[(size_t)idx[j] * Z + z] = x[i];
    else out[i] = x[(size_t)idx[j] * Z + z];
}

// --------------------------------------------------------------------------- //
// causal multi-head self-attention with key padding, one workgroup (4 waves) per (batch, head).
// qkv rows are token rows m = t*B + b of [q | k | v] (3Z floats), head h = columns h*dh .. h*dh+dh of
// each third (torch.nn.MultiheadAttention's packed in_proj layout).  Q, K, V (and dO) of the head sit in
// LDS with an odd row stride.  A WAVE owns a query row: lanes are keys for the scores / softmax (wave
// reductions), then lanes are head-dim columns for the P.V product (probabilities broadcast from LDS).
// The backward mirrors it: phase 1 a wave per query (D_t, dS, dq), phase 2 a wave per key (dk, dv) --
// every sum has a fixed order, no atomics.  Attention-probability dropout uses the counter mask at element
// ((b*H + h)*T + t)*T + j.
// --------------------------------------------------------------------------- //
constexpr int ADH = 64;    // max head dim
constexpr int ATM = 128;   // max sequence length
constexpr int AW = 8;      // waves per workgroup

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__global__ __launch_bounds__(64 * AW) void attention_fwd_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ lengths,
                                                            int T, int B, int Z, int H, unsigned thresh, float inv_keep,
                                                            unsigned long long seed0, const unsigned long long* __restrict__ seed_ptr,
                                                            float* __restrict__ ctx, float* __restrict__ lse) {
    extern __shared__ float sm[];
    const unsigned long long seed = seed0 + (seed_ptr ? *seed_ptr : 0ull);
    const int dh = Z / H, rs = dh | 1, b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Qs = sm;
    float* Ks = Qs + (size_t)T * rs;
    float* Vs = Ks + (size_t)T * rs;
    float* Ps = Vs + (size_t)T * rs;                   // [AW][T]
    const float scale = rsqrtf((float)dh);
    int dhp = 1;
    while (dhp < dh) dhp <<= 1;                        // the P.V product spreads the keys over G = 64/dhp lane groups
    const int G = 64 / dhp, dl = lane & (dhp - 1), grp = lane / dhp;
    for (int i = threadIdx.x; i < T * dh; i += 64 * AW) {
        int j = i / dh, d = i % dh;
        const float* row = qkv + ((size_t)j * B + b) * 3 * Z + h * dh + d;
        Qs[j * rs + d] = row[0] * scale;
        Ks[j * rs + d] = row[Z];
        Vs[j * rs + d] = row[2 * Z];
    }
    __syncthreads();
    const int len = lengths ? (int)lengths[b] : T;
    float* P = Ps + wave * T;
    const unsigned long long hb = ((unsigned long long)b * H + h) * T;
    for (int t = wave; t < T; t += AW) {
        const int jmax = min(t + 1, len);             // keys j <= t and j < len
        float sc[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j = ps * 64 + lane;
            float a = -__builtin_huge_valf();
            if (j < jmax) {
                a = 0.f;
                for (int d = 0; d < dh; ++d) a = __builtin_fmaf(Qs[t * rs + d], Ks[j * rs + d], a);
            }
            sc[ps] = a;
        }
        const float mx = wave_max(fmaxf(sc[0], sc[1]));
        float l = 0.f;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j = ps * 64 + lane;
            const float pj = j < jmax ? __expf(sc[ps] - mx) : 0.f;
            l += pj;
            if (j < T) P[j] = (thresh == 0 || keep_elem(seed, (hb + t) * T + j, thresh)) ? pj * inv_keep : 0.f;
        }
        l = wave_sum(l);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float o = 0.f;                                  // lane = (key group, head-dim column): G partial sums per column
        if (dl < dh)
            for (int j = grp; j < jmax; j += G) o = __builtin_fmaf(P[j], Vs[j * rs + dl], o);
        for (int off = dhp; off < 64; off <<= 1) o += __shfl_xor(o, off);
        if (lane < dh) ctx[((size_t)t * B + b) * Z + h * dh + lane] = jmax > 0 ? o / l : 0.f;
        if (lane == 0) lse[hb + t] = jmax > 0 ? mx + __logf(l) : 0.f;
        __builtin_amdgcn_wave_barrier();               // P is reused by this wave's next query
    }
}

// backward: p_tj = exp(s_tj - lse_t), dP_tj = (dO_t . v_j) * mask/(1-p), D_t = sum_j p_tj dP_tj,
// dS_tj = p_tj (dP_tj - D_t) * scale;  dq_t = sum_j dS_tj k_j, dk_j = sum_t dS_tj q_t, dv_j = sum_t pdrop_tj dO_t
__global__ __launch_bounds__(64 * AW) void attention_bwd_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ lengths,
                                                            const float* __restrict__ lse, const float* __restrict__ dctx,
                                                            int T, int B, int Z, int H, unsigned thresh, float inv_keep,
                                                            unsigned long long seed0, const unsigned long long* __restrict__ seed_ptr,
                                                            float* __restrict__ dqkv) {
    extern __shared__ float sm[];
    const unsigned long long seed = seed0 + (seed_ptr ? *seed_ptr : 0ull);
    const int dh = Z / H, rs = dh | 1, b = blockIdx.x / H, h = blockIdx.x % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Qs = sm;
    float* Ks = Qs + (size_t)T * rs;
    float* Vs = Ks + (size_t)T * rs;
    float* Gs = Vs + (size_t)T * rs;                   // dO
    float* Ds = Gs + (size_t)T * rs;                   // [T] D_t
    float* Ls = Ds + T;                                // [T] lse_t
    float* S1 = Ls + T + wave * 2 * T;                 // per wave: [T] dS, [T] dropped probabilities
    float* S2 = S1 + T;
    for (int i = threadIdx.x; i < T * dh; i += 64 * AW) {
        int j = i / dh, d = i % dh;
        const float* row = qkv + ((size_t)j * B + b) * 3 * Z + h * dh + d;
        Qs[j * rs + d] = row[0]; Ks[j * rs + d] = row[Z]; Vs[j * rs + d] = row[2 * Z];
        Gs[j * rs + d] = dctx[((size_t)j * B + b) * Z + h * dh + d];
    }
    const unsigned long long hb = ((unsigned long long)b * H + h) * T;
    for (int i = threadIdx.x; i < T; i += 64 * AW) Ls[i] = lse[hb + i];
    __syncthreads();
    const int len = lengths ? (int)lengths[b] : T;
    const float scale = rsqrtf((float)dh);
    int dhp = 1;
    while (dhp < dh) dhp <<= 1;                        // the dq / dk / dv sums spread their terms over G = 64/dhp lane groups
    const int G = 64 / dhp, dl = lane & (dhp - 1), grp = lane / dhp;
    // ---- phase 1: a wave per query t ----
    for (int t = wave; t < T; t += AW) {
        const int jmax = min(t + 1, len);
        const float lt = Ls[t];
        float pj[2], dp[2], D = 0.f;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j = ps * 64 + lane;
            pj[ps] = dp[ps] = 0.f;
            if (j < jmax) {
                float a = 0.f, c = 0.f;
                for (int d = 0; d < dh; ++d) {
                    a = __builtin_fmaf(Qs[t * rs + d], Ks[j * rs + d], a);
                    c = __builtin_fmaf(Gs[t * rs + d], Vs[j * rs + d], c);
                }
                pj[ps] = __expf(a * scale - lt);
                if (thresh != 0) c = keep_elem(seed, (hb + t) * T + j, thresh) ? c * inv_keep : 0.f;
                dp[ps] = c;
                D = __builtin_fmaf(pj[ps], c, D);
            }
        }
        D = wave_sum(D);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int j = ps * 64 + lane;
            if (j < T) S1[j] = pj[ps] * (dp[ps] - D) * scale;
        }
        if (lane == 0) Ds[t] = D;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float dq = 0.f;
        if (dl < dh)
            for (int j = grp; j < jmax; j += G) dq = __builtin_fmaf(S1[j], Ks[j * rs + dl], dq);
        for (int off = dhp; off < 64; off <<= 1) dq += __shfl_xor(dq, off);
        if (lane < dh) dqkv[((size_t)t * B + b) * 3 * Z + h * dh + lane] = dq;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- phase 2: a wave per key j; lanes are the queries t >= j that see it ----
    for (int j = wave; j < T; j += AW) {
        const bool live = j < len;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int t = j + ps * 64 + lane;
            if (t < T) {
                float dS = 0.f, pd = 0.f;
                if (live) {
                    float a = 0.f, c = 0.f;
                    for (int d = 0; d < dh; ++d) {
                        a = __builtin_fmaf(Qs[t * rs + d], Ks[j * rs + d], a);
                        c = __builtin_fmaf(Gs[t * rs + d], Vs[j * rs + d], c);
                    }
                    const float p = __expf(a * scale - Ls[t]);
                    pd = p;
                    if (thresh != 0) {
                        const bool kp = keep_elem(seed, (hb + t) * T + j, thresh);
                        c = kp ? c * inv_keep : 0.f;
                        pd = kp ? p * inv_keep : 0.f;
                    }
                    dS = p * (c - Ds[t]) * scale;
                }
                S1[t] = dS;
                S2[t] = pd;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float dk = 0.f, dv = 0.f;
        if (dl < dh)
            for (int t = j + grp; t < T; t += G) {
                dk = __builtin_fmaf(S1[t], Qs[t * rs + dl], dk);
                dv = __builtin_fmaf(S2[t], Gs[t * rs + dl], dv);
            }
        for (int off = dhp; off < 64; off <<= 1) { dk += __shfl_xor(dk, off); dv += __shfl_xor(dv, off); }
        if (lane < dh) {
            float* drow = dqkv + ((size_t)j * B + b) * 3 * Z + h * dh + lane;
            drow[Z] = dk;
            drow[2 * Z] = dv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// --------------------------------------------------------------------------- //
// fused kernels of the layer chain (umlh_encoder.cpp): the elementwise tails (bias, relu, dropout, residual) ride in the
// GEMM epilogue or in the row kernel that consumes the GEMM's split-K slabs; column reductions (bias / LayerNorm weight
// gradients) are row-chunk partials that ONE multi_reduce launch per layer sums in a fixed order.
// --------------------------------------------------------------------------- //
// out[i] = epilogue(sum_s slabs[s*stride + i])
__global__ __launch_bounds__(256) void reduce_epilogue_kernel(const float* __restrict__ slabs, int ns, long long stride, long long total,
                                                              int N, Epilogue e, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float v = slab_sum(slabs + i, stride, ns);
    out[i] = e.on ? epilogue_apply(e, v, i, (int)(i % N)) : v;
}

// s = epilogue(sum of ns slabs of x) (bias, dropout, + residual through e.add); y = LayerNorm(s) * gamma + beta; a wave per row
__global__ __launch_bounds__(256) void add_layernorm_fused_kernel(const float* __restrict__ x, int ns, long long stride, Epilogue e,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int M, int N, float eps, float* __restrict__ s_out,
                                                                  float* __restrict__ y, float* __restrict__ mean_out,
                                                                  float* __restrict__ rstd_out) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const size_t o = (size_t)m * N;
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) {
        float v = slab_sum(x + o + n, stride, ns);
        v = epilogue_apply(e, v, (long long)(o + n), n);
        s_out[o + n] = v;
        sum += v;
    }
    const float mean = wave_sum(sum) / (float)N;
    float var = 0.f;
    for (int n = lane; n < N; n += 64) {
        float d = s_out[o + n] - mean;          // written by this lane above
        var += d * d;
    }
    const float rstd = rsqrtf(wave_sum(var) / (float)N + eps);
    for (int n = lane; n < N; n += 64) y[o + n] = (s_out[o + n] - mean) * rstd * gamma[n] + beta[n];
    if (lane == 0) { mean_out[m] = mean; rstd_out[m] = rstd; }
}

// dy = sum of ns slabs (+ add: the gradient arriving over the residual path), kept in dy_out when it had to be formed here;
// ds = LayerNorm backward of the row; dsd = dropout(ds) on stream `seed` (the branch gradient), NULL = not wanted
__global__ __launch_bounds__(256) void layernorm_bwd_rows_fused_kernel(const float* __restrict__ dy, int ns, long long stride,
                                                                       const float* __restrict__ add, float* __restrict__ dy_out,
                                                                       const float* __restrict__ s, const float* __restrict__ gamma,
                                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                       int M, int N, float* __restrict__ ds, float* __restrict__ dsd,
                                                                       unsigned thresh, float inv_keep, unsigned long long seed0,
                                                                       const unsigned long long* __restrict__ seed_ptr) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const unsigned long long seed = seed0 + (seed_ptr ? *seed_ptr : 0ull);
    const size_t o = (size_t)m * N;
    const float mu = mean[m], rs = rstd[m];
    const float* dyr = dy_out ? dy_out : dy;
    float a = 0.f, b = 0.f;
    for (int n = lane; n < N; n += 64) {
        float d;
        if (dy_out) {
            d = slab_sum(dy + o + n, stride, ns);
            if (add) d += add[o + n];
            dy_out[o + n] = d;
        } else d = dy[o + n];
        const float g = gamma[n] * d, xh = (s[o + n] - mu) * rs;
        a += g;
        b += g * xh;
    }
    a = wave_sum(a) / (float)N;
    b = wave_sum(b) / (float)N;
    for (int n = lane; n < N; n += 64) {
        const float g = gamma[n] * dyr[o + n], xh = (s[o + n] - mu) * rs;
        const float v = rs * (g - a - xh * b);
        ds[o + n] = v;
        if (dsd) dsd[o + n] = (thresh == 0 || keep_elem(seed, (unsigned long long)(o + n), thresh)) ? v * inv_keep : 0.f;
    }
}

// part[r][n] = sum of x[m][n] over the rows m of chunk r = blockIdx.y (4 row groups x 64 columns per block, fixed order)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int M, int N, int chunk,
                                                             float* __restrict__ part) {
    __shared__ float sh[4][64];
    const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, g = threadIdx.x >> 6;
    const int m0 = blockIdx.y * chunk, m1 = min(M, m0 + chunk);
    float t = 0.f;
    if (c < N)
        for (int m = m0 + g; m < m1; m += 4) t += x[(size_t)m * N + c];
    sh[g][l] = t;
    __syncthreads();
    if (g == 0 && c < N) part[(size_t)blockIdx.y * N + c] = (sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l]);
}

// row-chunk partials of dgamma = sum dy*xhat, dbeta = sum dy and (dsd != NULL) of the following dense layer's db = sum dsd
__global__ __launch_bounds__(256) void ln_cols_partial_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ dsd, int M, int N, int chunk,
                                                              float* __restrict__ part_g, float* __restrict__ part_b,
                                                              float* __restrict__ part_d) {
    __shared__ float sh[3][4][64];
    const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, g = threadIdx.x >> 6;
    const int m0 = blockIdx.y * chunk, m1 = min(M, m0 + chunk);
    float a = 0.f, b = 0.f, d3 = 0.f;
    if (c < N)
        for (int m = m0 + g; m < m1; m += 4) {
            const float d = dy[(size_t)m * N + c];
            a += d * (s[(size_t)m * N + c] - mean[m]) * rstd[m];
            b += d;
            if (dsd) d3 += dsd[(size_t)m * N + c];
        }
    sh[0][g][l] = a; sh[1][g][l] = b; sh[2][g][l] = d3;
    __syncthreads();
    if (g == 0 && c < N) {
        const size_t o = (size_t)blockIdx.y * N + c;
        part_g[o] = (sh[0][0][l] + sh[0][1][l]) + (sh[0][2][l] + sh[0][3][l]);
        part_b[o] = (sh[1][0][l] + sh[1][1][l]) + (sh[1][2][l] + sh[1][3][l]);
        if (dsd) part_d[o] = (sh[2][0][l] + sh[2][1][l]) + (sh[2][2][l] + sh[2][3][l]);
    }
}

// dst[i] = sum_s src[s*stride + i] for every entry of the table: all partial sums of a layer's backward in one launch
__global__ __launch_bounds__(256) void multi_reduce_kernel(MultiReduceArgs a) {
    int t = 0;
#pragma unroll 1
    while (t + 1 < a.count && (int)blockIdx.x >= a.d[t + 1].blk0) ++t;
    const ReduceDesc d = a.d[t];
    const float scale = a.s_num ? a.s_mul * a.s_num[0] / a.s_den[0] : 1.f;
    const long long base = (long long)((int)blockIdx.x - d.blk0) * 1024 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long i = base + 256 * q;
        if (i < d.n) {
            const float v = slab_sum(d.src + i, d.stride, d.ns);
            d.dst[i] = a.s_num ? v * scale : v;
        }
    }
}

// *dst = v (the dropout seed word a captured launch sequence reads)
__global__ void set_u64_kernel(unsigned long long* dst, unsigned long long v) { *dst = v; }

inline unsigned drop_thresh(float p) { return p <= 0.f ? 0u : (unsigned)((double)p * 4294967296.0); }
inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

int umlh_enc_launch_bias_act(float* y, const float* b, long long M, int N, int relu, hipStream_t st) {
    if (M * N <= 0) return 0;
    hipLaunchKernelGGL(bias_act_kernel, dim3(blocks_for(M * N)), dim3(256), 0, st, y, b, M * N, N, relu);
    return (int)hipGetLastError();
}

int umlh_enc_launch_relu_bwd(const float* y, float* dy, long long n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, st, y, dy, n);
    return (int)hipGetLastError();
}

int umlh_enc_launch_dropout(float* x, long long n, float p, unsigned long long seed, hipStream_t st) {
    if (n <= 0 || p <= 0.f) return 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n)), dim3(256), 0, st, x, n, drop_thresh(p), 1.f / (1.f - p), seed);
    return (int)hipGetLastError();
}

int umlh_enc_launch_add_inplace(float* y, const float* x, long long n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks_for(n)), dim3(256), 0, st, y, x, n);
    return (int)hipGetLastError();
}

int umlh_enc_launch_colsum(const float* x, int M, int N, float* out, hipStream_t st) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(64 * CG), 0, st, x, M, N, out);
    return (int)hipGetLastError();
}

int umlh_enc_launch_add_layernorm(const float* x, const float* r, const float* gamma, const float* beta, int M, int N, float eps,
                                  float* s_out, float* y, float* mean, float* rstd, hipStream_t st) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, r, gamma, beta, M, N, eps, s_out, y, mean, rstd);
    return (int)hipGetLastError();
}

int umlh_enc_launch_layernorm_bwd(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                                  int M, int N, float* ds, float* dgamma, float* dbeta, hipStream_t st) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(layernorm_bwd_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, st, dy, s, gamma, mean, rstd, M, N, ds);
    hipLaunchKernelGGL(layernorm_bwd_cols_kernel, dim3((N + 63) / 64), dim3(64 * CG), 0, st, dy, s, mean, rstd, M, N, dgamma, dbeta);
    return (int)hipGetLastError();
}


int umlh_enc_launch_reduce_epilogue(const float* slabs, int ns, long long stride, long long total, int N, const Epilogue* e, float* out,
                                    hipStream_t st) {
    if (total <= 0) return 0;
    hipLaunchKernelGGL(reduce_epilogue_kernel, dim3(blocks_for(total)), dim3(256), 0, st, slabs, ns, stride, total, N, *e, out);
    return (int)hipGetLastError();
}

int umlh_enc_launch_add_layernorm_fused(const float* x, int ns, long long stride, const Epilogue* e, const float* gamma,
                                        const float* beta, int M, int N, float eps, float* s_out, float* y, float* mean, float* rstd,
                                        hipStream_t st) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(add_layernorm_fused_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, ns, stride, *e, gamma, beta, M, N, eps,
                       s_out, y, mean, rstd);
    return (int)hipGetLastError();
}

int umlh_enc_launch_layernorm_bwd_rows_fused(const float* dy, int ns, long long stride, const float* add, float* dy_out, const float* s,
                                             const float* gamma, const float* mean, const float* rstd, int M, int N, float* ds,
                                             float* dsd, float p, unsigned long long seed, const unsigned long long* seed_ptr,
                                             hipStream_t st) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(layernorm_bwd_rows_fused_kernel, dim3((M + 3) / 4), dim3(256), 0, st, dy, ns, stride, add, dy_out, s, gamma, mean,
                       rstd, M, N, ds, dsd, drop_thresh(p), p > 0.f ? 1.f / (1.f - p) : 1.f, seed, seed_ptr);
    return (int)hipGetLastError();
}

int umlh_enc_launch_colsum_partial(const float* x, int M, int N, int chunk, float* part, hipStream_t st) {
    if (M <= 0 || N <= 0) return 0;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, (M + chunk - 1) / chunk), dim3(256), 0, st, x, M, N, chunk, part);
    return (int)hipGetLastError();
}

int umlh_enc_launch_ln_cols_partial(const float* dy, const float* s, const float* mean, const float* rstd, const float* dsd, int M, int N,
                                    int chunk, float* part_g, float* part_b, float* part_d, hipStream_t st) {
    if (M <= 0 || N <= 0) return 0;
    hipLaunchKernelGGL(ln_cols_partial_kernel, dim3((N + 63) / 64, (M + chunk - 1) / chunk), dim3(256), 0, st, dy, s, mean, rstd, dsd, M, N,
                       chunk, part_g, part_b, part_d);
    return (int)hipGetLastError();
}

// fills blk0 of every entry; entries with n == 0 are allowed
int umlh_enc_launch_multi_reduce(MultiReduceArgs* a, hipStream_t st) {
    if (a->count < 1 || a->count > UMLH_MULTI_REDUCE_MAX) return (int)hipErrorInvalidValue;
    int blocks = 0;
    for (int t = 0; t < a->count; ++t) { a->d[t].blk0 = blocks; blocks += (int)((a->d[t].n + 1023) / 1024); }
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(multi_reduce_kernel, dim3(blocks), dim3(256), 0, st, *a);
    return (int)hipGetLastError();
}

int umlh_enc_launch_set_u64(unsigned long long* dst, unsigned long long v, hipStream_t st) {
    hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, st, dst, v);
    return (int)hipGetLastError();
}

float umlh_enc_drop_inv_keep(float p) { return p > 0.f ? 1.f / (1.f - p) : 1.f; }
unsigned umlh_enc_drop_thresh(float p) { return drop_thresh(p); }

int umlh_enc_launch_add_pos(float* x, const float* pos, int T, int B, int Z, hipStream_t st) {
    hipLaunchKernelGGL(add_pos_kernel, dim3(blocks_for((long long)T * B * Z)), dim3(256), 0, st, x, pos, T, B, Z);
    return (int)hipGetLastError();
}

int umlh_enc_launch_pos_grad(const float* dx, int T, int B, int Z, float* dpos, hipStream_t st) {
    hipLaunchKernelGGL(pos_grad_kernel, dim3(blocks_for((long long)T * Z)), dim3(256), 0, st, dx, T, B, Z, dpos);
    return (int)hipGetLastError();
}

int umlh_enc_launch_gather_rows(const float* x, const int64_t* idx, int n, int Z, float* out, int scatter, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks_for((long long)n * Z)), dim3(256), 0, st, x, idx, n, Z, out, scatter);
    return (int)hipGetLastError();
}

// returns hipErrorInvalidValue for shapes outside the kernel's envelope (T <= 128, head dim <= 64)
int umlh_enc_launch_attention_fwd(const float* qkv, const int64_t* lengths, int T, int B, int Z, int H, float p,
                                  unsigned long long seed, const unsigned long long* seed_ptr, float* ctx, float* lse, hipStream_t st) {
    if (T < 1 || T > ATM || H < 1 || Z % H != 0 || Z / H > ADH) return (int)hipErrorInvalidValue;
    const int rs = (Z / H) | 1;
    const size_t smem = sizeof(float) * (3 * (size_t)T * rs + AW * T);
    static std::atomic<unsigned long long> attr_done{0};      // bit d: done on device d (the attribute is per device)
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ULL)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(1ULL << (dev_ & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(B * H), dim3(64 * AW), smem, st, qkv, lengths, T, B, Z, H, drop_thresh(p),
                       p > 0.f ? 1.f / (1.f - p) : 1.f, seed, seed_ptr, ctx, lse);
    return (int)hipGetLastError();
}

int umlh_enc_launch_attention_bwd(const float* qkv, const int64_t* lengths, const float* lse, const float* dctx, int T, int B,
                                  int Z, int H, float p, unsigned long long seed, const unsigned long long* seed_ptr, float* dqkv,
                                  hipStream_t st) {
    if (T < 1 || T > ATM || H < 1 || Z % H != 0 || Z / H > ADH) return (int)hipErrorInvalidValue;
    const int rs = (Z / H) | 1;
    const size_t smem = sizeof(float) * (4 * (size_t)T * rs + 2 * T + AW * 2 * T);
    static std::atomic<unsigned long long> attr_done{0};      // bit d: done on device d (the attribute is per device)
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ULL)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(1ULL << (dev_ & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * H), dim3(64 * AW), smem, st, qkv, lengths, lse, dctx, T, B, Z, H, drop_thresh(p),
                       p > 0.f ? 1.f / (1.f - p) : 1.f, seed, seed_ptr, dqkv);
    return (int)hipGetLastError();
}

}  // extern "C"
