// MultiBench shared encoder (MultiBench/models.py:39-127: conv1d k=1 -> positions -> 5 x post-norm
// nn.TransformerEncoderLayer(d_model=z, nhead=5, dim_feedforward=2048, relu, dropout 0.1) with a
// causal + key-padding mask), forward AND backward, as fp32 HIP kernels for gfx950.
//
// The dense layers go through gemm_f32 (fp32 MFMA, umlh_kernels_f32.hip); this file holds what sits
// between the GEMMs.  Sizes are tiny (z <= 300, T <= 128, B = 32: <= 4096 token rows), so these are
// latency/HBM-bound row kernels -- one wave per token row, LDS only where rows are shared (attention);
// nothing here is reshaped to reach MFMA.
#include "umlh_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// counter-based dropout mask: keep element i of stream `seed` with probability 1 - p
__device__ __forceinline__ bool keep_elem(unsigned long long seed, unsigned long long i, unsigned thresh) {
    unsigned long long x = seed + i * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (unsigned)(x >> 32) >= thresh;
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
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int M, int N, float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N)
        for (int m = g; m < M; m += 4) s += x[(size_t)m * N + c];
    sh[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < N) out[c] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
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
__global__ __launch_bounds__(256) void layernorm_bwd_cols_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 int M, int N, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
    __shared__ float sh[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, l = threadIdx.x & 63;
    float a = 0.f, b = 0.f;
    if (c < N)
        for (int m = g; m < M; m += 4) {
            float d = dy[(size_t)m * N + c];
            a += d * (s[(size_t)m * N + c] - mean[m]) * rstd[m];
            b += d;
        }
    sh[0][g][l] = a; sh[1][g][l] = b;
    __syncthreads();
    if (g == 0 && c < N) {
        dgamma[c] = sh[0][0][l] + sh[0][1][l] + sh[0][2][l] + sh[0][3][l];
        dbeta[c] = sh[1][0][l] + sh[1][1][l] + sh[1][2][l] + sh[1][3][l];
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
// causal multi-head self-attention with key padding, one workgroup per (batch, head), one thread per
// query row.  qkv rows are token rows m = t*B + b of [q | k | v] (3Z floats), head h = columns
// h*dh .. h*dh+dh of each third (torch.nn.MultiheadAttention's packed in_proj layout).
// K and V of the head sit in LDS; every thread streams over its keys with an online softmax.
// Attention-probability dropout uses the counter mask at element ((b*H + h)*T + t)*T + j.
// --------------------------------------------------------------------------- //
constexpr int ADH = 64;    // max head dim

__global__ __launch_bounds__(128) void attention_fwd_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ lengths,
                                                            int T, int B, int Z, int H, unsigned thresh, float inv_keep,
                                                            unsigned long long seed, float* __restrict__ ctx,
                                                            float* __restrict__ lse) {
    extern __shared__ float sm[];
    const int dh = Z / H, b = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    float* Ks = sm;
    float* Vs = sm + (size_t)T * dh;
    for (int i = threadIdx.x; i < T * dh; i += blockDim.x) {
        int j = i / dh, d = i % dh;
        const float* row = qkv + ((size_t)j * B + b) * 3 * Z + h * dh + d;
        Ks[i] = row[Z];
        Vs[i] = row[2 * Z];
    }
    __syncthreads();
    if (t >= T) return;
    const int len = lengths ? (int)lengths[b] : T;
    const int jmax = min(t + 1, len);                 // keys j <= t and j < len
    const float scale = rsqrtf((float)dh);
    float q[ADH], o[ADH];
    const float* qrow = qkv + ((size_t)t * B + b) * 3 * Z + h * dh;
#pragma unroll
    for (int d = 0; d < ADH; ++d) { q[d] = d < dh ? qrow[d] * scale : 0.f; o[d] = 0.f; }
    float mx = -__builtin_huge_valf(), l = 0.f;
    const unsigned long long mbase = (((unsigned long long)b * H + h) * T + t) * T;
    for (int j = 0; j < jmax; ++j) {
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < ADH; ++d) if (d < dh) sc = __builtin_fmaf(q[d], Ks[j * dh + d], sc);
        const float nm = fmaxf(mx, sc);
        const float corr = __expf(mx - nm), p = __expf(sc - nm);
        l = l * corr + p;
        const float pd = (thresh == 0 || keep_elem(seed, mbase + j, thresh)) ? p * inv_keep : 0.f;
#pragma unroll
        for (int d = 0; d < ADH; ++d) if (d < dh) o[d] = o[d] * corr + pd * Vs[j * dh + d];
        mx = nm;
    }
    // jmax >= 1 whenever len >= 1 (key 0 is always visible)
    const float inv_l = jmax > 0 ? 1.f / l : 0.f;
    float* orow = ctx + ((size_t)t * B + b) * Z + h * dh;
#pragma unroll
    for (int d = 0; d < ADH; ++d) if (d < dh) orow[d] = o[d] * inv_l;
    lse[((size_t)b * H + h) * T + t] = jmax > 0 ? mx + __logf(l) : 0.f;
}

// backward: phase 1 (thread = query t): D_t = sum_j p_tj dP_tj, dq_t = scale * sum_j dS_tj k_j;
// phase 2 (thread = key j): dk_j = scale * sum_{t>=j} dS_tj q_t, dv_j = sum_{t>=j} pdrop_tj dO_t,
// with p_tj = exp(s_tj - lse_t), dP_tj = (dO_t . v_j) * mask/(1-p), dS_tj = p_tj (dP_tj - D_t).
__global__ __launch_bounds__(128) void attention_bwd_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ lengths,
                                                            const float* __restrict__ lse, const float* __restrict__ dctx,
                                                            int T, int B, int Z, int H, unsigned thresh, float inv_keep,
                                                            unsigned long long seed, float* __restrict__ dqkv) {
    extern __shared__ float sm[];
    const int dh = Z / H, b = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    float* Qs = sm;
    float* Ks = Qs + (size_t)T * dh;
    float* Vs = Ks + (size_t)T * dh;
    float* Gs = Vs + (size_t)T * dh;                   // dO
    float* Ds = Gs + (size_t)T * dh;                   // [T] D_t
    float* Ls = Ds + T;                                // [T] lse_t
    for (int i = threadIdx.x; i < T * dh; i += blockDim.x) {
        int j = i / dh, d = i % dh;
        const float* row = qkv + ((size_t)j * B + b) * 3 * Z + h * dh + d;
        Qs[i] = row[0]; Ks[i] = row[Z]; Vs[i] = row[2 * Z];
        Gs[i] = dctx[((size_t)j * B + b) * Z + h * dh + d];
    }
    if (t < T) Ls[t] = lse[((size_t)b * H + h) * T + t];
    __syncthreads();
    const int len = lengths ? (int)lengths[b] : T;
    const float scale = rsqrtf((float)dh);
    const unsigned long long hb = ((unsigned long long)b * H + h) * T;
    if (t < T) {
        const int jmax = min(t + 1, len);
        float q[ADH], g[ADH], dq[ADH];
#pragma unroll
        for (int d = 0; d < ADH; ++d) { q[d] = d < dh ? Qs[t * dh + d] : 0.f; g[d] = d < dh ? Gs[t * dh + d] : 0.f; dq[d] = 0.f; }
        const float lt = Ls[t];
        float D = 0.f;
        for (int j = 0; j < jmax; ++j) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < ADH; ++d) if (d < dh) { sc = __builtin_fmaf(q[d], Ks[j * dh + d], sc); dp = __builtin_fmaf(g[d], Vs[j * dh + d], dp); }
            const float p = __expf(sc * scale - lt);
            if (thresh != 0) dp = keep_elem(seed, (hb + t) * T + j, thresh) ? dp * inv_keep : 0.f;
            D = __builtin_fmaf(p, dp, D);
        }
        Ds[t] = D;
        for (int j = 0; j < jmax; ++j) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < ADH; ++d) if (d < dh) { sc = __builtin_fmaf(q[d], Ks[j * dh + d], sc); dp = __builtin_fmaf(g[d], Vs[j * dh + d], dp); }
            const float p = __expf(sc * scale - lt);
            if (thresh != 0) dp = keep_elem(seed, (hb + t) * T + j, thresh) ? dp * inv_keep : 0.f;
            const float dS = p * (dp - D) * scale;
#pragma unroll
            for (int d = 0; d < ADH; ++d) if (d < dh) dq[d] = __builtin_fmaf(dS, Ks[j * dh + d], dq[d]);
        }
        float* dqrow = dqkv + ((size_t)t * B + b) * 3 * Z + h * dh;
#pragma unroll
        for (int d = 0; d < ADH; ++d) if (d < dh) dqrow[d] = dq[d];
    }
    __syncthreads();
    if (t < T) {
        const int j = t;                                // this thread's key
        float k[ADH], v[ADH], dk[ADH], dv[ADH];
#pragma unroll
        for (int d = 0; d < ADH; ++d) { k[d] = d < dh ? Ks[j * dh + d] : 0.f; v[d] = d < dh ? Vs[j * dh + d] : 0.f; dk[d] = 0.f; dv[d] = 0.f; }
        if (j < len) {
            for (int tq = j; tq < T; ++tq) {            // queries that see key j
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < ADH; ++d) if (d < dh) { sc = __builtin_fmaf(Qs[tq * dh + d], k[d], sc); dp = __builtin_fmaf(Gs[tq * dh + d], v[d], dp); }
                const float p = __expf(sc * scale - Ls[tq]);
                float pd = p;
                if (thresh != 0) {
                    const bool kp = keep_elem(seed, (hb + tq) * T + j, thresh);
                    dp = kp ? dp * inv_keep : 0.f;
                    pd = kp ? p * inv_keep : 0.f;
                }
                const float dS = p * (dp - Ds[tq]) * scale;
#pragma unroll
                for (int d = 0; d < ADH; ++d) if (d < dh) { dk[d] = __builtin_fmaf(dS, Qs[tq * dh + d], dk[d]); dv[d] = __builtin_fmaf(pd, Gs[tq * dh + d], dv[d]); }
            }
        }
        float* drow = dqkv + ((size_t)j * B + b) * 3 * Z + h * dh;
#pragma unroll
        for (int d = 0; d < ADH; ++d) if (d < dh) { drow[Z + d] = dk[d]; drow[2 * Z + d] = dv[d]; }
    }
}

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
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, st, x, M, N, out);
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
    hipLaunchKernelGGL(layernorm_bwd_cols_kernel, dim3((N + 63) / 64), dim3(256), 0, st, dy, s, mean, rstd, M, N, dgamma, dbeta);
    return (int)hipGetLastError();
}

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
                                  unsigned long long seed, float* ctx, float* lse, hipStream_t st) {
    if (T < 1 || T > 128 || H < 1 || Z % H != 0 || Z / H > ADH) return (int)hipErrorInvalidValue;
    const size_t smem = sizeof(float) * 2 * (size_t)T * (Z / H);
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(B * H), dim3(128), smem, st, qkv, lengths, T, B, Z, H, drop_thresh(p),
                       p > 0.f ? 1.f / (1.f - p) : 1.f, seed, ctx, lse);
    return (int)hipGetLastError();
}

int umlh_enc_launch_attention_bwd(const float* qkv, const int64_t* lengths, const float* lse, const float* dctx, int T, int B,
                                  int Z, int H, float p, unsigned long long seed, float* dqkv, hipStream_t st) {
    if (T < 1 || T > 128 || H < 1 || Z % H != 0 || Z / H > ADH) return (int)hipErrorInvalidValue;
    const size_t smem = sizeof(float) * (4 * (size_t)T * (Z / H) + 2 * T);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * H), dim3(128), smem, st, qkv, lengths, lse, dctx, T, B, Z, H, drop_thresh(p),
                       p > 0.f ? 1.f / (1.f - p) : 1.f, seed, dqkv);
    return (int)hipGetLastError();
}

}  // extern "C"
