// Internal launch interface of the fused MultiBench encoder-layer kernels (umlh_kernels_enc.hip) and of the fp32 GEMM with an
// elementwise epilogue (umlh_api.cpp), used by umlh_encoder.cpp.  Not part of the C ABI.
#pragma once
#include "umlh_common.h"

extern "C" {
int umlh_enc_launch_reduce_epilogue(const float* slabs, int ns, long long stride, long long total, int N, const Epilogue* e, float* out,
                                    hipStream_t st);
int umlh_enc_launch_add_layernorm_fused(const float* x, int ns, long long stride, const Epilogue* e, const float* gamma,
                                        const float* beta, int M, int N, float eps, float* s_out, float* y, float* mean, float* rstd,
                                        hipStream_t st);
int umlh_enc_launch_layernorm_bwd_rows_fused(const float* dy, int ns, long long stride, const float* add, float* dy_out, const float* s,
                                             const float* gamma, const float* mean, const float* rstd, int M, int N, float* ds,
                                             float* dsd, float p, unsigned long long seed, const unsigned long long* seed_ptr, hipStream_t st);
int umlh_enc_launch_colsum_partial(const float* x, int M, int N, int chunk, float* part, hipStream_t st);
int umlh_enc_launch_ln_cols_partial(const float* dy, const float* s, const float* mean, const float* rstd, const float* dsd, int M, int N,
                                    int chunk, float* part_g, float* part_b, float* part_d, hipStream_t st);
int umlh_enc_launch_multi_reduce(MultiReduceArgs* a, hipStream_t st);
int umlh_enc_launch_attention_fwd(const float* qkv, const int64_t* lengths, int T, int B, int Z, int H, float p,
                                  unsigned long long seed, const unsigned long long* seed_ptr, float* ctx, float* lse, hipStream_t st);
int umlh_enc_launch_attention_bwd(const float* qkv, const int64_t* lengths, const float* lse, const float* dctx, int T, int B,
                                  int Z, int H, float p, unsigned long long seed, const unsigned long long* seed_ptr, float* dqkv, hipStream_t st);
int umlh_enc_launch_set_u64(unsigned long long* dst, unsigned long long v, hipStream_t st);
float umlh_enc_drop_inv_keep(float p);
unsigned umlh_enc_drop_thresh(float p);

// out[M,N] (ldo == N) = epilogue(A B^T) with the operand layouts of umlh_gemm_f32.  `splits` K-slabs go to `slabs`
// ([ns][M*N], ns returned in *ns_out).  defer != 0: the raw slabs (ns >= 1) are left for the consumer, `out` and `epi` unused;
// else one slab applies `epi` in the GEMM, more launch the slab reduction with `epi`.
int umlh_gemm_f32_epi(const float* A, const float* B, float* out, int M, int N, int K, int lda, int ldb, int ta, int tb,
                      const Epilogue* epi, int splits, float* slabs, int defer, int* ns_out, hipStream_t stream);
}
