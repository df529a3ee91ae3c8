// MultiBench alternation step: fused per-modality decoder Linear(z -> D) + masked next-step
// MSE (MultiBench/models.py:202,213,234,243 + MSE :129-143) and its backward.
// Shapes are tiny (B*T <= ~1600 rows, z <= 300, D <= 300): plain fp32, latency-bound, one
// workgroup per sequence position; no MFMA reshaping.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// dW[d][k] = s * sum_r dres[r][d] z[r][k];  db[d] = s * sum_r dres[r][d]     (one block per d)
__global__ __launch_bounds__(128) void seq_mse_bwd_dw_kernel(const float* __restrict__ dres, const float* __restrict__ z,
                                                             const float* __restrict__ loss_cnt, const float* __restrict__ grad_out,
                                                             int R, int Z, int D, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[128];
    const int d = blockIdx.x, tid = threadIdx.x;
    const float s = 2.f * grad_out[0] / loss_cnt[1];
    float bsum = 0.f;
    for (int r = tid; r < R; r += 128) bsum += dres[(size_t)r * D + d];
    for (int k = tid; k < Z; k += 128) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) acc = fmaf(dres[(size_t)r * D + d], z[(size_t)r * Z + k], acc);
        dw[(size_t)d * Z + k] = acc * s;
    }
    red[tid] = bsum;
    __syncthreads();
    for (int off = 64; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    if (tid == 0) db[d] = red[0] * s;
}

extern "C" {
int umlh_seq_launch_fwd(const float* z, const float* w, const float* b, const float* x, const int64_t* lengths, int B, int T,
                        int Z, int D, float* recon, float* dres, float* row_partial, float* loss_cnt, hipStream_t st) {
    hipLaunchKernelGGL(seq_mse_fwd_kernel, dim3(B * T), dim3(128), sizeof(float) * (Z + 128), st, z, w, b, x, lengths, B, T, Z,
                       D, recon, dres, row_partial);
    hipLaunchKernelGGL(seq_mse_finalize_kernel, dim3(1), dim3(256), 0, st, row_partial, lengths, B, T, D, loss_cnt);
    return (int)hipGetLastError();
}
int umlh_seq_launch_bwd(const float* z, const float* w, const float* dres, const float* loss_cnt, const float* grad_out, int B,
                        int T, int Z, int D, float* dz, float* dw, float* db, hipStream_t st) {
    hipLaunchKernelGGL(seq_mse_bwd_dz_kernel, dim3(B * T), dim3(128), sizeof(float) * D, st, dres, w, loss_cnt, grad_out, Z, D, dz);
    hipLaunchKernelGGL(seq_mse_bwd_dw_kernel, dim3(D), dim3(128), 0, st, dres, z, loss_cnt, grad_out, B * T, Z, D, dw, db);
    return (int)hipGetLastError();
}
}
