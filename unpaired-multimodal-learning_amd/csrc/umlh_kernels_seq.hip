// MultiBench alternation step: fused per-modality decoder Linear(z -> D) + masked next-step
// MSE (MultiBench/models.py:202,213,234,243 + MSE :129-143) and its backward.
// Shapes are tiny (B*T <= ~1600 rows, z <= 300, D <= 300): plain fp32, latency-bound, one
// workgroup per sequence position; only the decoder's dW (a [D x rows] x [rows x z] contraction) goes
// through the fp32 MFMA GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "umlh_common.h"

// row r = (b, t): recon[r][d] = bias[d] + sum_k z[r][k] W[d][k];  masked residual vs x[b][t+1].
__global__ __launch_bounds__(128) void seq_mse_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ x,
                                                          const int64_t* __restrict__ lengths, int B, int T, int Z, int D,
                                                          float* __restrict__ recon, float* __restrict__ dres,
                                                          float* __restrict__ row_partial) {
    extern __shared__ float zs[];            // [Z] + [128]
    float* red = zs + Z;
    const int r = blockIdx.x, b = r / T, t = r % T, tid = threadIdx.x;
    for (int k = tid; k < Z; k += 128) zs[k] = z[(size_t)r * Z + k];
    __syncthreads();
    // target position: t+1 (next-step prediction), or t itself for length-1 sequences (models.py:209-210)
    const bool shift = T > 1;
    const int tt = shift ? t + 1 : t;
    bool live = shift ? (t < T - 1) : true;
    if (live && shift && lengths != nullptr) live = (int64_t)tt < lengths[b];
    float sq = 0.f;
    for (int d = tid; d < D; d += 128) {
        float acc = bias[d];
        const float* wr = w + (size_t)d * Z;
        for (int k = 0; k < Z; ++k) acc = fmaf(zs[k], wr[k], acc);
        if (recon) recon[(size_t)r * D + d] = acc;
        float e = 0.f;
        if (live) { e = acc - x[((size_t)b * T + tt) * D + d]; sq += e * e; }
        dres[(size_t)r * D + d] = e;
    }
    red[tid] = sq;
    __syncthreads();
    for (int off = 64; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) row_partial[r] = red[0];
}

// loss_cnt[0] = sum(partials) / (cnt + 1e-8), loss_cnt[1] = cnt = D * #live positions
__global__ __launch_bounds__(256) void seq_mse_finalize_kernel(const float* __restrict__ row_partial,
                                                               const int64_t* __restrict__ lengths, int B, int T, int D,
                                                               float* __restrict__ loss_cnt) {
    __shared__ float red[256];
    __shared__ float cred[256];
    const int tid = threadIdx.x;
    float s = 0.f, c = 0.f;
    for (int r = tid; r < B * T; r += 256) s += row_partial[r];
    for (int b = tid; b < B; b += 256) {
        int live;
        if (T == 1) live = 1;
        else if (lengths == nullptr) live = T - 1;
        else { long long l = lengths[b]; live = (int)(l < 1 ? 0 : (l > T ? T - 1 : l - 1)); }
        c += (float)live;
    }
    red[tid] = s; cred[tid] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { red[tid] += red[tid + off]; cred[tid] += cred[tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        float cnt = cred[0] * (float)D;
        // lengths-less, T>1 case is MSE.mean() over all (B, T-1, D) elements: same count, no epsilon (models.py:140-141)
        float denom = (lengths == nullptr || T == 1) ? cnt : cnt + 1e-8f;
        loss_cnt[0] = red[0] / denom;
        loss_cnt[1] = denom;
    }
}

// dz[r][k] = s * sum_d dres[r][d] W[d][k],  s = 2 * grad_out / denom
__global__ __launch_bounds__(128) void seq_mse_bwd_dz_kernel(const float* __restrict__ dres, const float* __restrict__ w,
                                                             const float* __restrict__ loss_cnt, const float* __restrict__ grad_out,
                                                             int Z, int D, float* __restrict__ dz) {
    extern __shared__ float ds[];            // [D]
    const int r = blockIdx.x, tid = threadIdx.x;
    for (int d = tid; d < D; d += 128) ds[d] = dres[(size_t)r * D + d];
    __syncthreads();
    const float s = 2.f * grad_out[0] / loss_cnt[1];
    for (int k = tid; k < Z; k += 128) {
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc = fmaf(ds[d], w[(size_t)d * Z + k], acc);
        dz[(size_t)r * Z + k] = acc * s;
    }
}

// db[d] = s * sum_r dres[r][d]   (one block per 64 columns, 16 row groups, fixed order);  s = 2 * grad_out / count
__global__ __launch_bounds__(1024) void seq_mse_bwd_db_kernel(const float* __restrict__ dres, const float* __restrict__ loss_cnt,
                                                              const float* __restrict__ grad_out, int R, int D,
                                                              float* __restrict__ db) {
    __shared__ float sh[16][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6, d = blockIdx.x * 64 + l;
    float a = 0.f;
    if (d < D)
        for (int r = g; r < R; r += 16) a += dres[(size_t)r * D + d];
    sh[g][l] = a;
    __syncthreads();
    if (g == 0 && d < D) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][l];
        db[d] = t * (2.f * grad_out[0] / loss_cnt[1]);
    }
}

// dw[i] *= s   (the dW GEMM runs with alpha = 1: s is a device-side scalar)
__global__ __launch_bounds__(256) void seq_mse_scale_kernel(float* __restrict__ dw, long long n, const float* __restrict__ loss_cnt,
                                                            const float* __restrict__ grad_out) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dw[i] *= 2.f * grad_out[0] / loss_cnt[1];
}

extern "C" int umlh_f32_launch_gemm(const GemmArgs* g, int ta, int tb, int splits, hipStream_t stream);

extern "C" {
int umlh_seq_launch_fwd(const float* z, const float* w, const float* b, const float* x, const int64_t* lengths, int B, int T,
                        int Z, int D, float* recon, float* dres, float* row_partial, float* loss_cnt, hipStream_t st) {
    hipLaunchKernelGGL(seq_mse_fwd_kernel, dim3(B * T), dim3(128), sizeof(float) * (Z + 128), st, z, w, b, x, lengths, B, T, Z,
                       D, recon, dres, row_partial);
    hipLaunchKernelGGL(seq_mse_finalize_kernel, dim3(1), dim3(256), 0, st, row_partial, lengths, B, T, D, loss_cnt);
    return (int)hipGetLastError();
}
int umlh_seq_launch_bwd(const float* z, const float* w, const float* dres, const float* loss_cnt, const float* grad_out, int B,
                        int T, int Z, int D, float* dz, float* dw, float* db, int with_params, hipStream_t st) {
    hipLaunchKernelGGL(seq_mse_bwd_dz_kernel, dim3(B * T), dim3(128), sizeof(float) * D, st, dres, w, loss_cnt, grad_out, Z, D, dz);
    if (!with_params) return (int)hipGetLastError();     // dW / db by the caller (split-K slabs + row-chunk partials)
    hipLaunchKernelGGL(seq_mse_bwd_db_kernel, dim3((D + 63) / 64), dim3(1024), 0, st, dres, loss_cnt, grad_out, B * T, D, db);
    // dW[d][k] = s * sum_r dres[r][d] z[r][k]: the transposed-A GEMM of the fp32 MFMA kernel, then the device-side scale
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dres; g.lda = D; g.B = z; g.ldb = Z; g.out = dw; g.ldo = Z;
    g.M = D; g.N = Z; g.K = B * T; g.k_chunk = g.K; g.k_switch = g.K; g.k_valid1 = g.K; g.alpha = 1.f;
    int rc = umlh_f32_launch_gemm(&g, 1, 1, 1, st);
    if (rc) return rc;
    hipLaunchKernelGGL(seq_mse_scale_kernel, dim3((unsigned)(((long long)D * Z + 255) / 256)), dim3(256), 0, st, dw, (long long)D * Z,
                       loss_cnt, grad_out);
    return (int)hipGetLastError();
}
}

// --------------------------------------------------------------------------- //
// SequenceInfoNCELoss (MultiBench/models.py:145-175): rows = the valid (batch, time) positions, logits = normalize(pred) .
// normalize(target)^T / temperature, labels = the diagonal, loss = mean cross-entropy.  The n x n logits come from the fp32 GEMM;
// these are the row kernels around it.
// --------------------------------------------------------------------------- //
__device__ __forceinline__ float nce_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float nce_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// y[r] = x[r] / max(|x[r]|, 1e-12) (F.normalize), norm[r] = that denominator; a wave per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, int n, int D, float* __restrict__ y,
                                                          float* __restrict__ norm) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { const float v = x[(size_t)r * D + d]; s = fmaf(v, v, s); }
    const float den = fmaxf(sqrtf(nce_wave_sum(s)), 1e-12f);
    for (int d = lane; d < D; d += 64) y[(size_t)r * D + d] = x[(size_t)r * D + d] / den;
    if (lane == 0) norm[r] = den;
}

// row r of the raw dot products: l = dots / temperature; row_loss[r] = logsumexp(l) - l[r]; the row is overwritten with
// softmax(l) - onehot(r)  (d loss_r / d l)
__global__ __launch_bounds__(256) void nce_rows_kernel(float* __restrict__ dots, int n, float inv_temp, float* __restrict__ row_loss) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float* row = dots + (size_t)r * n;
    float mx = -__builtin_huge_valf();
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, row[j] * inv_temp);
    mx = nce_wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += __expf(row[j] * inv_temp - mx);
    s = nce_wave_sum(s);
    const float lse = mx + __logf(s), diag = row[r] * inv_temp;
    for (int j = lane; j < n; j += 64) row[j] = __expf(row[j] * inv_temp - lse) - (j == r ? 1.f : 0.f);
    if (lane == 0) row_loss[r] = lse - diag;
}

// loss = mean of row_loss (one block, fixed order)
__global__ __launch_bounds__(256) void nce_finalize_kernel(const float* __restrict__ row_loss, int n, float* __restrict__ loss) {
    __shared__ float sh[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += row_loss[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sh[0] / (float)n;
}

// gradient through F.normalize: dx = c * (dy - y (y . dy)) / norm, c = grad_out * scale (scale = 1 / (n * temperature))
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ norm, const float* __restrict__ grad_out,
                                                         float scale, int n, int D, float* __restrict__ dx) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot = fmaf(y[(size_t)r * D + d], dy[(size_t)r * D + d], dot);
    dot = nce_wave_sum(dot);
    const float c = grad_out[0] * scale / norm[r];
    for (int d = lane; d < D; d += 64) dx[(size_t)r * D + d] = c * (dy[(size_t)r * D + d] - y[(size_t)r * D + d] * dot);
}

extern "C" {
int umlh_seq_launch_l2norm(const float* x, int n, int D, float* y, float* norm, hipStream_t st) {
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, x, n, D, y, norm);
    return (int)hipGetLastError();
}
int umlh_seq_launch_nce_rows(float* dots, int n, float inv_temp, float* row_loss, float* loss, hipStream_t st) {
    hipLaunchKernelGGL(nce_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, dots, n, inv_temp, row_loss);
    hipLaunchKernelGGL(nce_finalize_kernel, dim3(1), dim3(256), 0, st, row_loss, n, loss);
    return (int)hipGetLastError();
}
int umlh_seq_launch_l2norm_bwd(const float* dy, const float* y, const float* norm, const float* grad_out, float scale, int n, int D,
                               float* dx, hipStream_t st) {
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((n + 3) / 4), dim3(256), 0, st, dy, y, norm, grad_out, scale, n, D, dx);
    return (int)hipGetLastError();
}
}
