// Shared device/host definitions for the UML head kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "umlh.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int UMLH_MULTI_OPT_MAX = 48;     // tensors per multi-tensor optimizer launch (descriptor table travels as a kernel argument)

// K elements staged per LDS chunk in the fp32 kernels (8 MFMA k-steps of 2).
constexpr int KT = 16;

// Row of a 32x32 MFMA accumulator register: C/D layout is
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Write-through (sc1) stores for everything a kernel hands to the NEXT launch (dZ^T, split-K slabs): the bytes leave for
// memory when the instruction retires instead of sitting dirty in the XCD's L2 until the end-of-kernel write-back that the
// dependent launch waits for (microarch guide, rows 'boundary' and 'publish-large').  cfg2 bf16 step 46.7 -> 44.8 us.
// UMLH_WT=0 in the environment selects plain stores (args.plain = 1) for A/B timing.
#if defined(__HIPCC__)
__device__ __forceinline__ void store_wt_f32(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_out_f32x4(float* p, f32x4v v, int plain) {
    if (plain) *reinterpret_cast<f32x4v*>(p) = v;
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_out_f32(float* p, float v, int plain) {
    if (plain) *p = v; else store_wt_f32(p, v);
}
#endif
static inline int umlh_plain_stores() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("UMLH_WT"); v = (e && e[0] == '0') ? 1 : 0; }
    return v;
}

// fp32 mode on the bf16 matrix pipe ("x3", round 3): every fp32 operand is split into three bf16 pieces, x = hi + mid + lo
// (hi, mid by truncation -- each residual is exact in fp32 --, lo rounded to nearest), and a product x*w is formed as the six
// piece products of order <= 2 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: every piece product is exact; the dropped terms (mid*lo, lo*mid, lo*lo) are below 2^-20 |x||w| in the worst
// case and 2^-24 |x||w| rms (tests/test_x3_split_cpu.py) -- the size of the roundings an fp32 fma chain makes on its running sum:
// against float64 the two forms are equally accurate (profiles/r03_x3_accuracy.txt) -- and the 32x32x16 bf16 MFMA does in 32
// cycles what the 32x32x2 fp32 MFMA does in 8 x 64.  UMLH_F32_X3=0 keeps the fp32 MFMA.
static inline int umlh_f32_x3() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("UMLH_F32_X3"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- one modality's rows as the forward kernel sees them ----
struct SegDesc {
    const float*   feats;        // [*, ld] fp32
    const int64_t* feat_index;   // row gather for feats (NULL = identity)
    const int64_t* labels;       // [*] int64
    const int64_t* label_index;  // row gather for labels (NULL = identity)
    const float*   scale_ptr;    // device scalar: logit scale of this modality
    int   rows;                  // valid rows in this step
    int   ld;                    // leading dim of feats
    int   col0;                  // first column of this segment in dZ^T
    int   blk0;                  // first block of this segment in the grid
    float w_over_rows;           // loss weight / global row count
};

struct FwdArgs {
    SegDesc seg[2];
    const float* W;              // [C, K] head weight
    int   C, K;
    float* dzt;                  // [C, ldz] dZ^T (class-major), NULL = eval (no gradient)
    int   ldz;
    float* partials;             // [grid][4] = {loss_sum, correct, gscale_sum, 0}
    int   plain;                 // 1 = plain stores for dZ^T (default 0: write-through, see store_wt_f32)
    float* row_stats;            // optional per-row {CE, top-1 correct} of segment 0 then segment 1 (whole-table evaluation)
    unsigned long long* stamps;  // diagnostics (UMLH_DBG_FWD=9): [grid][8 waves][8] cycle stamps, else NULL
    const float* Ws;             // fp32 fragment-major shadow of W (w_shadow32_kernel), current; NULL = stage W through LDS
    int   dbg;                   // timing-only ablations of the streamed main loop (analysis build -DUMLH_ABLATIONS only): 21 no W refills, 22 no B reads, 23 both
    int   learn;                 // learnable_temp: also reduce sum_c p_c * raw_c (d loss / d scale); 0: the per-element fma is skipped
    int   x3;                    // Ws holds the three bf16 piece planes of W (w_shadow_x3_kernel): fwd_ce_f32 MODE 3
};

// Elementwise tail of a dense layer of the MultiBench encoder, applied to v = alpha * sum (element (m, n), flat index
// i = m*N + n of a dense [M, N] result) in this order:  + bias[n];  relu;  zero where gate[i] <= 0 (relu backward against the
// saved activation);  dropout (counter mask, stream `seed`, element i);  + add[i] (residual fan-in).  All-zero = identity.
struct Epilogue {
    const float* bias;
    const float* gate;
    const float* add;
    unsigned long long seed;
    const unsigned long long* seed_ptr;   // optional device word added to `seed` (launch sequences replayed from a HIP graph)
    unsigned thresh;             // 0 = no dropout; else keep element i iff hash(seed, i) >= thresh
    float inv_keep;
    int   relu;
    int   on;                    // any field set
};

// counter-based dropout mask: keep element i of stream `seed` with probability 1 - p
__device__ __forceinline__ bool keep_elem(unsigned long long seed, unsigned long long i, unsigned thresh) {
    unsigned long long x = seed + i * 0x9E3779B97F4A7C15ULL;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (unsigned)(x >> 32) >= thresh;
}

__device__ __forceinline__ float epilogue_apply(const Epilogue& e, float v, long long i, int n) {
    if (e.bias) v += e.bias[n];
    if (e.relu) v = fmaxf(v, 0.f);
    if (e.gate && !(e.gate[i] > 0.f)) v = 0.f;
    if (e.thresh) v = keep_elem(e.seed + (e.seed_ptr ? *e.seed_ptr : 0ull), (unsigned long long)i, e.thresh) ? v * e.inv_keep : 0.f;
    if (e.add) v += e.add[i];
    return v;
}

// ---- generic fp32 GEMM: out[m][n] = alpha * sum_k A(m,k) * B(n,k) ----
struct GemmArgs {
    const float* A;  const float* B;  float* out;
    int   plain;                 // 1 = plain output stores (default 0: write-through, see store_wt_f32)
    const int64_t* a_rows;       // TA==0: gather of A rows by m (NULL = identity)
    const int64_t* k_rows;       // TB==1: gather of B rows by k (NULL = identity)
    int   M, N, K;
    int   lda, ldb, ldo;
    int   k_chunk;               // K range handled by one blockIdx.z
    long long slab_stride;       // floats between the outputs of consecutive blockIdx.z
    float alpha;
    const float* alpha_ptr;      // optional device scalar multiplied into alpha (logit scale)
    // TB==1 only: reduction rows k >= k_switch come from a second table (text rows
    // follow image rows in dZ^T); rows outside [0,k_valid) of either part are zeros.
    const float*   B2;
    const int64_t* k_rows2;
    int   ldb2;
    int   k_switch;
    int   k_valid1, k_valid2;
    // > 0: modality-aligned split-K -- slabs [0, nsplit1) cover k in [0, k_switch) and the rest cover
    // [k_switch, K), k_chunk rows each (image and text gradients stay in separate slabs); 0: uniform
    int   nsplit1;
    int   slab_count;            // dw_f32 only: slabs of the launch (its grid is 1-D, split index fastest)
    Epilogue epi;                // applied by the kernel when the launch has ONE slab (a split-K launch leaves it to the reduce)
    int   x3;                    // gemm_f32 (128 x 128 tiles): products on the bf16 matrix pipe from three-way splits (umlh_f32_x3)
};

// ---- bf16-mode argument blocks (kernels in umlh_kernels_bf16.hip, filled by umlh_api.cpp) ----
typedef unsigned short u16;
struct SegDescB {
    const u16*     feats;        // [*, ld] bf16
    const int64_t* feat_index;
    const int64_t* labels;
    const int64_t* label_index;
    const float*   scale_ptr;
    int   rows, ld, col0, blk0;
    float w_over_rows;
};

struct FwdArgsB {
    SegDescB seg[2];
    const u16* W;                // bf16 fragment-major shadow of the head weight (w_shadow_kernel)
    int   C, K;
    u16*  dzt;                   // bf16 dZ^T, column-chunk-major [cols/64][crows][64]; NULL = eval
    int   crows;                 // class rows per column chunk (C rounded up to 128)
    float* partials;
    int   plain;                 // 1 = plain dZ^T stores (default 0: write-through)
    int   dbg;                   // 9 = cycle stamps; other values: timing-only ablations (analysis build -DUMLH_ABLATIONS only)
    int   learn;                 // learnable_temp: also reduce sum_c p_c * raw_c (d loss / d scale)
    unsigned long long* stamps;  // diagnostic build only (UMLH_DBG_FWD=9): [grid][8] s_memtime stamps of wave 0
    float* row_stats;            // optional per-row {CE, top-1 correct} of segment 0 then segment 1 (whole-table evaluation)
    unsigned long long* xch;     // fwd_ce_bf16_q: [blocks][4 fields][128 rows] epoch-tagged exchange granules
    unsigned epoch;              // fwd_ce_bf16_q: tag of this launch (never 0)
    int   ntiles;                // fwd_ce_bf16_q: row tiles of this launch (the grid is rounded up to 8 of them)
    int   wtiles;                // fwd_ce_bf16_q: 32-class tiles per k-step of the W shadow (cpad / 32)
};

// bf16 GEMM out[m][n] = sum_k A[m][k] * B[k][n] (kernel dw_bf16<AM, OM>).  Written for dW = dZ^T F; the 2-layer
// head's projection forward / dH^T / dW_proj reuse it through the A-source and output modes.
//   AM 0: A column-chunk-major [K/64][lda][64] (k contiguous inside a chunk)     AM 1: A row-major, row m at A + a_rows[m]*lda
//   OM 0: fp32 split-K slabs   OM 1: bf16 row-major out[m*ldo + n]   OM 2: bf16 chunk-major out[((n>>6)*ldo + m)*64 + (n&63)]
struct DwArgsB {
    const u16* A;                // AM 0: dZ^T-style chunk-major; AM 1: row-major rows gathered by a_rows
    const u16* B;  const int64_t* k_rows;  int ldb;     // image-side feature rows
    const u16* B2; const int64_t* k_rows2; int ldb2;    // text-side feature rows (k >= k_switch)
    float* out;                  // fp32 slabs [splits][M][ldo]
    const u16* zeros;            // >= 16 B of zeros (source of masked loads)
    int   plain;                 // 1 = plain slab stores (default 0: write-through)
    int   dbg;                   // timing-only ablations, analysis build -DUMLH_ABLATIONS only: bit0 = no A traffic, bit1 = no F traffic, bit2 = no main loop
    int   M, N, K, lda, ldo, k_chunk, k_switch, k_valid1, k_valid2, nsplit;
    int   nsplit1;               // slabs [0, nsplit1) cover k in [0, k_switch), the rest [k_switch, K): k_chunk rows each
    const int64_t* a_rows;       // AM 1: row ids of A (never NULL: identity table for dense operands)
    int   bcs;                   // B column-chunk stride in elements: element (k, n) of either B source at
                                 // rid*ld + (n>>6)*bcs + (n&63); 64 = plain row-major rows
    void* out16;                 // OM 1/2: bf16 output
    long long slab_stride;
    unsigned long long* stamps;  // diagnostics (UMLH_DBG_DW=16+bits): [blocks][8 waves][8] cycle stamps, else NULL
};

// one launch sums the split-K slabs / row-chunk partials of every gradient of an encoder layer (fixed order s = 0 .. ns-1)
constexpr int UMLH_MULTI_REDUCE_MAX = 16;
struct ReduceDesc { const float* src; float* dst; long long stride; long long n; int ns; int blk0; };
struct MultiReduceArgs {
    ReduceDesc d[UMLH_MULTI_REDUCE_MAX]; int count;
    const float* s_num; const float* s_den; float s_mul;   // s_num != NULL: every sum is scaled by s_mul * s_num[0] / s_den[0]
};

struct OptArgs {
    int   kind;                  // UMLH_OPT_*
    float lr, decay;             // decay = 1 - lr*wd (AdamW)
    float neg_step_size;         // -(lr / (1 - beta1^t))
    float bc2_sqrt;              // sqrt(1 - beta2^t)
    float beta1, one_m_beta1, beta2, one_m_beta2, eps, momentum, wd;
    int   plain;                 // 1 = plain stores of the updated state (default 0: write-through, see store_wt_f32)
    int   x3;                    // the fp32 fragment-major shadow the update kernel refreshes is the three-plane bf16 split (umlh_f32_x3)
};

struct FinalizeArgs {
    const float* partials;       // fwd partials, or NULL when the tail is already final
    int   nb0, nb1;              // blocks of segment 0 / 1
    float inv_rows0, inv_rows1;  // 1 / global rows
    float w0, w1;                // loss weights (img_alpha, alpha)
    float* tail;                 // [2 + UMLH_N_SCALARS]: g_scales, scalars
    float* scalars_out;          // optional copy of the scalars
    float* scales; float* m_scales; float* v_scales;   // [2]
    int   update_mask;           // bit0: update img_scale, bit1: txt_scale
    OptArgs opt;
};

// gradient diagnostics of one step, by-products of the slab reduction (head_step_kernel)
struct DiagArgs {
    float* dst;                  // {dot, |g_img|^2, |g_txt|^2, sign agreements} of the step; NULL = off
    int   n_slabs_img;           // slabs [0, n_slabs_img) hold the image rows' partial sums
    float inv_w0, inv_w1;        // 1 / loss weight of each modality (0 if the weight is 0)
    float* part;                 // [blocks][4] per-workgroup partial sums; the workgroup that takes the LAST ticket adds them
    unsigned* ticket;            // in a fixed order (lane l: blocks l, l+64, ...; then a fixed lane tree) -> dst: the values do
                                 // not depend on the order the workgroups finished in.  ticket is 0 between launches.
    int   cols = 0;              // > 0: only columns [0, cols) of every class row count (heads with bias: the packed row is
};                               // [weight | bias | 0...] and the reference's diagnostics are over head.weight only, finetune.py:190-191)

// torch.optim single-tensor update of one element (see oracle/uml_oracle.py
// optimizer_step for the restated recurrence and its reference citations).
__device__ __forceinline__ void opt_update(const OptArgs& o, float g, float& p, float& m, float& v) {
    if (o.kind == UMLH_OPT_SGD) {
        if (o.wd != 0.f) g = g + o.wd * p;
        m = o.momentum * m + g;
        p = p - o.lr * m;
    } else {
        if (o.kind == UMLH_OPT_ADAMW) p = p * o.decay;
        else if (o.wd != 0.f) g = g + o.wd * p;
        m = m + (g - m) * o.one_m_beta1;
        v = v * o.beta2 + (o.one_m_beta2 * g) * g;
        float denom = sqrtf(v) / o.bc2_sqrt + o.eps;
        p = p + (o.neg_step_size * m) / denom;
    }
}

// Step scalars from the forward's per-block partials + the learnable logit scales' update (head_step_kernel's last block,
// finalize_kernel, and the finalize block of the single-launch step).  Works in a 256- or a 512-thread block: only threads
// < 256 touch data, every thread reaches every barrier; the summation order is the same in all callers.  COH: the partials
// were written through by forward blocks of the SAME launch and are read with device-coherent loads.
template <bool COH>
__device__ __forceinline__ void finalize_body(const FinalizeArgs& f, float (*sh)[256]) {
    const int tid = threadIdx.x;
    const bool on = tid < 256;
    if (f.partials != nullptr) {
        float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (on)
            for (int b = tid; b < f.nb0 + f.nb1; b += 256) {
                const float* q = f.partials + (size_t)b * 4;
                int o = b < f.nb0 ? 0 : 3;
                if (COH) {
                    s[o + 0] += __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s[o + 1] += __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s[o + 2] += __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    s[o + 0] += q[0]; s[o + 1] += q[1]; s[o + 2] += q[2];
                }
            }
        if (on) {
#pragma unroll
            for (int j = 0; j < 6; ++j) sh[j][tid] = s[j];
        }
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off)
#pragma unroll
                for (int j = 0; j < 6; ++j) sh[j][tid] += sh[j][tid + off];
            __syncthreads();
        }
        if (tid == 0) {
            float* t = f.tail;
            t[0] = sh[2][0] * f.w0 * f.inv_rows0;                    // d loss / d img_scale
            t[1] = sh[5][0] * f.w1 * f.inv_rows1;                    // d loss / d txt_scale
            float* sc = t + 2;
            sc[UMLH_S_LOSS_IMG] = sh[0][0] * f.inv_rows0;
            sc[UMLH_S_LOSS_TXT] = sh[3][0] * f.inv_rows1;
            sc[UMLH_S_ACC_IMG] = sh[1][0] * f.inv_rows0;
            sc[UMLH_S_ACC_TXT] = sh[4][0] * f.inv_rows1;
            sc[UMLH_S_GSCALE_IMG] = t[0];
            sc[UMLH_S_GSCALE_TXT] = t[1];
            sc[UMLH_S_CORRECT] = sh[1][0] + sh[4][0];
            sc[UMLH_S_LOSS_SUM] = sh[0][0] + sh[3][0];
        }
        __syncthreads();
    }
    if (tid < UMLH_N_CORE_SCALARS && f.scalars_out) f.scalars_out[tid] = f.tail[2 + tid];
    if (tid < 2 && ((f.update_mask >> tid) & 1)) {
        float p = f.scales[tid], m = f.m_scales[tid], v = f.v_scales[tid];
        opt_update(f.opt, f.tail[tid], p, m, v);
        f.scales[tid] = p; f.m_scales[tid] = m; f.v_scales[tid] = v;
    }
}

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2));
}
// x3 split of two fp32 values into packed bf16 pairs {a, b}: hi and mid by truncation (v_perm_b32 of the raw bits), lo by RNE
struct Split3 { unsigned hi, mid, lo; };
__device__ __forceinline__ Split3 split3_pair(float a, float b) {
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    Split3 r;
    r.hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                    // {a[31:16], b[31:16]}
    const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u), rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
    const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
    r.mid = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    r.lo = pack_bf16x2(ra - __builtin_bit_cast(float, va & 0xffff0000u), rb - __builtin_bit_cast(float, vb & 0xffff0000u));
    return r;
}

// --------------------------------------------------------------------------------------------------------------- //
// Claimed tasks of a one-launch step (step_bf16, step_f32).  The launch is a grid of persistent workgroups; the work
// is a list of TASKS (forward tiles, dW tiles, update slices, finalize) with dependencies that only point at earlier
// phases.  Nothing here depends on dispatch order, workgroup -> XCD placement or on the whole grid being resident:
//   * a task is run by whoever TAKES it: claim[t] = epoch via atomicMax, exactly one taker sees an older value;
//   * a workgroup walks its home tasks (b, b + G, ...) phase by phase, so it only ever holds tasks of phases >= the
//     one it is running;
//   * before a workgroup WAITS for task t it looks at claim[t]: a task nobody has taken (its home workgroup is not
//     resident yet: a second process / queue holds the CUs) is taken and run by the waiter itself, then the wait resumes.
//     A wait is therefore always on a task held by a RESIDENT workgroup that is running it or a lower phase: by
//     induction over the phases every wait ends (round 2 waited on lower block ids instead, which is only safe when
//     this launch has the device to itself: gpurun_out/n2_*.log);
//   * finished = done[t] granule {epoch, 1}, stored by one lane after every wave's write-through stores have drained.
// All words are epoch-tagged (epoch = launch/step tag, strictly increasing per handle): nothing is reset between
// launches.  Waits are bounded (50 ms); a wait that gives up sets status[0] (first code wins), the task and everything
// behind it is skipped (no update is applied from a gradient that timed out) and the host reads the word
// (umlh_step_status): umlh_train_step(s) does not return OK-with-NaN-weights any more.
// --------------------------------------------------------------------------------------------------------------- //
struct StepCtl {
    unsigned* claim;             // [ntask]
    unsigned long long* done;    // [ntask]
    unsigned* status;            // [4]: code (0 = ok), task waited on, epoch, phase
    unsigned epoch;              // never 0
};
constexpr int TW_OK = -1, TW_ABORT = -2;
constexpr unsigned long long UMLH_SPIN_TICKS = 5000000ull;     // 50 ms of s_memrealtime (100 MHz)

#if defined(__HIPCC__)
__device__ __forceinline__ unsigned ctl_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Tags are compared as OLDER THAN, never as "different" (a word that is newer than the epoch asked about belongs to a task that
// is long taken and done: the multi-step experiment of round 3 -- several steps' epochs alive in one launch -- ran update tasks
// twice under contention with "!=").  The host restarts the tags long before they could wrap.
__device__ __forceinline__ bool tag_older(unsigned tag, unsigned epoch) { return (int)(tag - epoch) < 0; }
// one lane takes task t; true = it is ours
__device__ __forceinline__ bool task_take(const StepCtl& c, int t) {
    return tag_older(__hip_atomic_fetch_max(c.claim + t, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c.epoch);
}
// Called by ONE whole wave: waits until tasks [first, first + count) are done.  Returns TW_OK, TW_ABORT (status set: by this
// wait's time-out or by anybody else) or the id of a task of the range that nobody had taken and that this wave has now
// TAKEN: the caller must run it (and may then wait again).
__device__ __forceinline__ int tasks_wait(const StepCtl& c, int first, int count, int lane, unsigned phase) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spin = 0;; ++spin) {
        const bool look = (spin & 7u) == 7u;
        bool ok = true;
        int untaken = 0x7fffffff;
        for (int i = lane; i < count; i += 64) {
            const unsigned long long v = __hip_atomic_load(c.done + first + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool d = !tag_older((unsigned)(v >> 32), c.epoch);
            ok = ok && d;
            if (look && !d && tag_older(ctl_load_u32(c.claim + first + i), c.epoch)) untaken = min(untaken, first + i);
        }
        if (__all(ok)) return TW_OK;
        if (look) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) untaken = min(untaken, __shfl_xor(untaken, off));
            if (untaken != 0x7fffffff) {
                unsigned old = c.epoch;
                if (lane == 0) old = __hip_atomic_fetch_max(c.claim + untaken, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tag_older((unsigned)__builtin_amdgcn_readfirstlane((int)old), c.epoch)) return untaken;
            }
            if (ctl_load_u32(c.status) != 0u) return TW_ABORT;
            if (__builtin_amdgcn_s_memrealtime() - t0 > UMLH_SPIN_TICKS) {
                if (lane == 0) {
                    unsigned expect = 0u;
                    if (__hip_atomic_compare_exchange_strong(c.status, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(c.status + 1, (unsigned)first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(c.status + 2, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(c.status + 3, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                return TW_ABORT;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// every thread of the workgroup calls it after the task's last store: drain, barrier, one granule
__device__ __forceinline__ void task_publish(const StepCtl& c, int t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(c.done + t, ((unsigned long long)c.epoch << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// everything the head blocks of the single-launch step need (step_bf16 in umlh_kernels_bf16.hip)
struct HeadFuse {
    const float* slabs; int n_slabs, n_slabs_img; long long slab_stride; int C, K;
    float* p; float* m; float* v; unsigned short* shadow; int cpad;
    float* grad_out;             // data-parallel split step: the summed gradient goes here and nothing is updated (p, m, v, shadow unused)
    int n_sub;                   // 256-thread sub-blocks of the update (two per workgroup)
    int dw_per_row;              // dW blocks per 128-class tile row (column tiles x K splits): the granules a sub-block waits for
    OptArgs o; FinalizeArgs f;
};
