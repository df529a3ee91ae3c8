// Single-launch multi-step "micro step" of the UML linear head for the reference's own operating point
// (batch 8 / 32 / 64 rows per modality, 12 800 iterations: engine/optimizer/default.py:3-45) and the grouped
// multi-head farm (finetune.py:406-448: sweep() over HYPER_DICT x alpha x seeds), gfx950.
//
// One persistent launch runs n_steps whole training steps (finetune.py:162-195) of G independent heads.  A head's
// class axis is cut into slices of 16 classes; workgroup (head, slice) OWNS that slice of W / m / v for the whole
// launch: the W slice lives in LDS, the optimizer moments in registers, and nothing but the gathered feature rows is
// read from memory per step.  Per step and workgroup:
//   1. forward   Z[slice, rows] = W_slice X^T on the fp32 MFMA (v_mfma_f32_16x16x4_f32, the exact fp32 fma chain of
//      the parity mode), MFMA issued "swapped" so a lane owns one sample column and 4 class logits;
//   2. slice-local softmax statistics (max, sum of exp, label logit, first arg-max, sum exp*raw) per sample, published
//      to the head's other slices as write-through (sc1) records + one flag per workgroup; every workgroup gathers
//      all records with sc1 loads (MI355X guide, Guideline 16 / "Valid forms", row 1) and merges them with the
//      online-softmax rule -- the only cross-workgroup exchange of a step;
//   3. dZ^T -> LDS, dW_slice = dZ^T X on the MFMA, the optimizer update of the slice straight from the accumulators.
// The feature rows of a step are gathered through registers (16-byte loads with per-lane row addresses, written to a
// double-buffered LDS image one K-chunk later: the loads of chunk q+2 are in flight while chunk q feeds the MFMAs),
// across phases and steps; the image is XOR-swizzled (slot ^= row & 15) so that the forward's ds_read_b128 and the
// backward's ds_read_b32 are conflict-free.  (An LDS-DMA ring was measured first: each global_load_lds costs the
// issuing wave ~100 cycles, eight per chunk as much as the chunk's whole MFMA time with one wave per SIMD.)
//
// All workgroups of a launch must be co-resident (spin waits): the host launches at most one workgroup per CU and
// chains micro launches of one process on one device; every spin is bounded and reports through a status word.
#include "umlh_common.h"
#include <atomic>
#include "umlh_micro.h"
#include <type_traits>

typedef float f32x4m __attribute__((ext_vector_type(4)));

namespace {

constexpr int MROWS = 64;          // row slots per step (4 sample tiles of 16)
constexpr int CS = UMLH_MICRO_CS;  // classes per workgroup
constexpr int LDZ = 66;            // dZ^T row stride (floats): banks 2*class + g distinct for ds_read_b32

__device__ __forceinline__ void wg_barrier() {
    // LDS writes of every wave visible before any wave passes; does NOT drain the LDS-DMA ring (no vmcnt wait)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void store_granule(unsigned long long* p, unsigned epoch, float v) {   // ONE aligned 8-B sc1 store
    __hip_atomic_store(p, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// torch.optim single-tensor update of one element with the hardware reciprocal / square root (1 ulp each) in place of
// the IEEE division and sqrtf of opt_update(): one wave per SIMD runs this on the critical path of every step
// (the correctly rounded forms cost ~100 VALU issues per element, 3x the MFMA time of the whole backward).  The update
// differs from opt_update's by ~1e-7 relative, i.e. <= 1e-7 * lr per step; parity tolerances are written in the tests.
template <int KIND>
__device__ __forceinline__ void opt_update_fast(const OptArgs& o, float inv_bc2_sqrt, float g, float& p, float& m, float& v) {
    if (KIND == UMLH_OPT_SGD) {
        if (o.wd != 0.f) g = g + o.wd * p;
        m = o.momentum * m + g;
        p = p - o.lr * m;
    } else {
        if (KIND == UMLH_OPT_ADAMW) p = p * o.decay;
        else if (o.wd != 0.f) g = g + o.wd * p;
        m = m + (g - m) * o.one_m_beta1;
        v = v * o.beta2 + (o.one_m_beta2 * g) * g;
        const float denom = __builtin_amdgcn_sqrtf(v) * inv_bc2_sqrt + o.eps;
        p = p + (o.neg_step_size * m) * __builtin_amdgcn_rcpf(denom);
    }
}

// Reductions over the four 16-lane rows of a wave (the class groups g of one sample) as two VALU row swaps
// (v_permlane16_swap, v_permlane32_swap) instead of ds_bpermute round trips: with both operands equal, the swap returns
// {the even rows' values in both rows of a pair, the odd rows' values in both rows}, so op(r[0], r[1]) is the pair's reduction.
__device__ __forceinline__ float rows_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// sum over the 16 lanes of a row, result in every lane: four DPP adds (xor 1, xor 2, half mirror, mirror), no LDS
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}
__device__ __forceinline__ void rows_argmax(float& m, int& am) {      // max, ties -> lowest class index
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    auto ai = __builtin_amdgcn_permlane16_swap((unsigned)am, (unsigned)am, false, false);
    float m0 = __uint_as_float(a[0]), m1 = __uint_as_float(a[1]);
    int i0 = (int)ai[0], i1 = (int)ai[1];
    bool t1 = m1 > m0 || (m1 == m0 && i1 < i0);
    m = t1 ? m1 : m0; am = t1 ? i1 : i0;
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    auto bi = __builtin_amdgcn_permlane32_swap((unsigned)am, (unsigned)am, false, false);
    m0 = __uint_as_float(b[0]); m1 = __uint_as_float(b[1]); i0 = (int)bi[0]; i1 = (int)bi[1];
    t1 = m1 > m0 || (m1 == m0 && i1 < i0);
    m = t1 ? m1 : m0; am = t1 ? i1 : i0;
}

// bf16 operand mode (BASELINE config "linear head bf16"; the precision mode of the general bf16 path: operands rounded
// to bf16 at use, fp32 accumulation, fp32 master weights and optimizer): the SAME fragments the fp32 MFMAs consume one float
// at a time -- lane (s16, g) holds 4 consecutive k of its row -- are exactly one v_mfma_f32_16x16x16_bf16 operand after two
// v_cvt_pk_bf16_f32, so a group of four fp32 MFMAs (128 matrix-pipe cycles) becomes one bf16 MFMA (16-32 cycles).
typedef short s16x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4m pack_bf16x4(float a, float b, float c, float d) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 w = {__builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2)),
                  __builtin_bit_cast(unsigned, __builtin_convertvector(f2{c, d}, bf2))};
    return __builtin_bit_cast(s16x4m, w);
}

// RES (bf16 operand mode, 128-wide chunks, d <= 640): the step's feature rows stay in LDS as bf16 -- NCH + 1 chunk regions of
// 64 rows x 128 columns (16 KB each) in a ring -- so the backward re-reads them instead of staging every chunk a second
// time: half the global loads, LDS stores and staging instructions of a step, and no barriers between forward chunks.
// Region of chunk c of step k: (base_k + c) mod (NCH + 1), base_{k+1} = base_k + NCH: the next step's chunk 0 goes to the one
// region this step does not use, its chunk c >= 1 to the region this step's chunk c-1 leaves after its backward position.
template <int NCH, int CW, bool RES = false>
struct MicroCfg {
    static constexpr int D = NCH * CW;
    static constexpr int LDW = D + 8;                       // W slice row stride: b128 reads of 16 rows conflict-free (LDW % 64 == 8)
    static constexpr int SLOTS = CW / 4;                    // 16-B slots per row of a chunk
    static constexpr int SM = SLOTS - 1 < 15 ? SLOTS - 1 : 15;   // swizzle mask
    static constexpr int DPW = CW / 16;                     // 16-byte pieces per lane per chunk (64 rows x CW floats over 256 lanes)
    static constexpr int XBYTES = MROWS * CW * 4;
    static constexpr int XTOTAL = RES ? (NCH + 1) * MROWS * CW * 2 : 2 * XBYTES;
    static constexpr int WBYTES = CS * LDW * 4;
    static constexpr int MISC = CS * LDZ * 4 + 2 * MROWS * 8 + 2 * MROWS * 4 + 4 * MROWS * 5 * 4 + 64 * 4;
    static constexpr int SMEM = WBYTES + XTOTAL + MISC;
    static constexpr int U = CW >= 64 ? CW / 64 : 1;        // 64-column blocks per chunk in the dW phase
    static constexpr int NP = 2 * NCH;                      // chunk positions of a step: NCH forward, NCH backward
};

template <int NCH, int CW, bool BF>
__global__ __launch_bounds__(256) void micro_steps_kernel(const UmlhMicroHead* __restrict__ heads, int n_heads, int n_steps) {
    constexpr bool RES = BF && CW == 128 && NCH <= 5;
    using K = MicroCfg<NCH, CW, RES>;
    constexpr int D = K::D, LDW = K::LDW, SM = K::SM, DPW = K::DPW, U = K::U, NP = K::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Wl = reinterpret_cast<float*>(smem);                                    // [CS][LDW]
    float* Xb = reinterpret_cast<float*>(smem + K::WBYTES);                        // [2][64][CW]        (RES: unused)
    unsigned short* Xh = reinterpret_cast<unsigned short*>(smem + K::WBYTES);      // RES: [NCH + 1][64][CW] bf16
    constexpr int RGN = MROWS * CW;                                                // elements of a chunk region
    float* dzT = reinterpret_cast<float*>(smem + K::WBYTES + K::XTOTAL);           // [CS][LDZ]
    unsigned long long* rowbase = reinterpret_cast<unsigned long long*>(dzT + CS * LDZ);   // [2][64] byte address of the row
    int* labs = reinterpret_cast<int*>(rowbase + 2 * MROWS);                        // [2][64]
    float* red = reinterpret_cast<float*>(labs + 2 * MROWS);                        // [4 parts][64][5]
    float* misc = red + 4 * MROWS * 5;                                              // [64]: per-tile scalar partials, abort flag

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4;

    // ---- which head / class slice ----
    int hi = 0;
    for (int i = 1; i < n_heads; ++i) hi = (int)blockIdx.x >= heads[i].wg0 ? i : hi;
    // by VALUE: the descriptor's fields live in registers for the whole launch (a reference would be re-read from
    // memory after every barrier / inline-asm wait: a dozen dependent L2 round trips per step)
    const UmlhMicroHead H = heads[hi];
    const int slice = (int)blockIdx.x - H.wg0;
    const int c0 = slice * CS, C = H.C, nwg = H.nwg;
    const bool learn = H.learnable != 0;
    unsigned* status = H.status;

    // ---- resident state: W slice -> LDS (rows >= C zero), m / v -> registers in the dW accumulator layout ----
    for (int i = tid; i < CS * (D / 4); i += 256) {
        const int r = i / (D / 4), q = i % (D / 4);
        f32x4m v = {0.f, 0.f, 0.f, 0.f};
        if (c0 + r < C) v = *reinterpret_cast<const f32x4m*>(H.w + (size_t)(c0 + r) * D + 4 * q);
        *reinterpret_cast<f32x4m*>(Wl + r * LDW + 4 * q) = v;
    }
    // element (c, u, r) of this thread: class 4g + r, column c*CW + 64u + 16*wave + s16
    // At 6 and 8 chunks (d = 768, 1024) the two moment arrays alone are 192 / 256 registers and the kernel spilled (d = 768 ran at
    // 18.8 us per step against 11.9 at d = 640): the SECOND moment then stays in memory -- each step reads and writes the
    // thread's 8 values of a chunk around that chunk's update (requested before the chunk's MFMA chains, L2-resident).
    // At 8 chunks (d = 1024) the first moment follows it.
    constexpr bool VMEM = NCH >= 6, MMEM = NCH >= 8;
    float mreg[MMEM ? 1 : NCH][U][4], vreg[VMEM ? 1 : NCH][U][4];
    float* vbase[4];                                        // VMEM: row pointers of this thread's 4 classes (clamped to valid rows)
    float* mbase[4];
    const int colw = 16 * wave + s16;                       // column inside a 64-wide block
    const bool colok = colw < CW;                           // CW < 64: only the first CW columns exist
    const bool adam = H.opt_kind != UMLH_OPT_SGD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        vbase[r] = H.v + (size_t)min(c0 + 4 * g + r, C - 1) * D + (colok ? colw : 0);
        mbase[r] = H.m + (size_t)min(c0 + 4 * g + r, C - 1) * D + (colok ? colw : 0);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cls = c0 + 4 * g + r, col = c * CW + 64 * u + colw;
                const bool ok = cls < C && colok;
                if (!MMEM) mreg[c][u][r] = ok ? H.m[(size_t)cls * D + col] : 0.f;
                if (!VMEM) vreg[c][u][r] = (ok && adam) ? H.v[(size_t)cls * D + col] : 0.f;
            }
    float scale[2] = {H.scales[0], H.scales[1]};
    float msc[2] = {0.f, 0.f}, vsc[2] = {0.f, 0.f};
    if (learn) { msc[0] = H.m_scales[0]; msc[1] = H.m_scales[1]; vsc[0] = H.v_scales[0]; vsc[1] = H.v_scales[1]; }

    // ---- row tables: threads 0..63 own one row slot each.  slot -> (modality, local row): image tiles first, then text
    // tiles; padding slots re-read a valid row (masked later).  The table of step k+2 is built during step k (row ids
    // requested at the top of the step, labels after the forward), so a step starts with its own and the next step's
    // tables in LDS and the feature loads can run across the step boundary.
    // Step offsets: a rolling window off[mod][j] = offs[mod][k + j], j = 0..3, advanced once per step (one load per
    // modality per step, a step ahead).  `rel` is a literal at every call site: the window stays in registers.
    int off[2][4];
#pragma unroll
    for (int md = 0; md < 2; ++md)
#pragma unroll
        for (int j = 0; j < 4; ++j) off[md][j] = H.offs[md] ? H.offs[md][j < n_steps ? j : n_steps] : 0;
    auto rows_of = [&](int rel, int mod) -> int { return off[mod][rel + 1] - off[mod][rel]; };
    long long rid_next = 0;
    auto fetch_rid = [&](int rel) {                         // issue the row-id load of step k + rel for this thread's slot
        const int ri = rows_of(rel, 0), rt = rows_of(rel, 1), ti = (ri + 15) >> 4;
        int mod = (tid >> 4) < ti ? 0 : 1;
        if (rt == 0) mod = 0;
        if (ri == 0) mod = 1;
        int local = mod == 0 ? tid : tid - 16 * ti;
        const int n = mod == 0 ? ri : rt;
        local = local < 0 ? 0 : (local >= n ? n - 1 : local);
        rid_next = H.index[mod][(size_t)(mod == 0 ? off[0][rel] : off[1][rel]) + local];
    };
    int lab_next = -1;
    auto publish_rowbase = [&](int rel, int k) {            // rid_next (of step k = current + rel) -> LDS row table [k & 1];
        const int ri = rows_of(rel, 0), rt = rows_of(rel, 1), ti = (ri + 15) >> 4;   // issues the (dependent) label load
        int mod = (tid >> 4) < ti ? 0 : 1;
        if (rt == 0) mod = 0;
        if (ri == 0) mod = 1;
        const int local = mod == 0 ? tid : tid - 16 * ti;
        const bool valid = local >= 0 && local < (mod == 0 ? ri : rt) && (tid >> 4) < ti + ((rt + 15) >> 4);
        rowbase[(k & 1) * MROWS + tid] = reinterpret_cast<unsigned long long>(H.feats[mod] + (size_t)rid_next * D);
        lab_next = valid ? (int)H.labels[mod][rid_next] : -1;
    };
    auto publish_labels = [&](int k) { labs[(k & 1) * MROWS + tid] = lab_next; };
    if (tid < MROWS) {
        fetch_rid(0); publish_rowbase(0, 0); publish_labels(0);
        if (n_steps > 1) { fetch_rid(1); publish_rowbase(1, 1); publish_labels(1); }
    }
    wg_barrier();

    // ---- feature rows: global -> registers -> LDS, one chunk (64 rows x CW floats) per position ----
    // piece i of this lane: 16-byte slot o16 = (wave*DPW + i)*64 + lane of the chunk image = (row, slot p); it is READ from
    // slot p ^ (row & SM) of the source row (the swizzle) and written linearly
    const float* rowp_cur[DPW];
    const float* rowp_nxt[DPW];
    int soff[DPW], doff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int o16 = (wave * DPW + i) * 64 + lane;
        const int row = o16 / K::SLOTS, p = o16 % K::SLOTS;
        soff[i] = 4 * (p ^ (row & SM));
        doff[i] = 4 * o16;
        rowp_cur[i] = reinterpret_cast<const float*>(rowbase[row]);
        rowp_nxt[i] = rowp_cur[i];
    }
    f32x4m xr[DPW];
    auto xload = [&](const float* const (&rowp)[DPW], int c) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) xr[i] = *reinterpret_cast<const f32x4m*>(rowp[i] + c * CW + soff[i]);
    };
    auto xstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < DPW; ++i) *reinterpret_cast<f32x4m*>(Xb + buf * (MROWS * CW) + doff[i]) = xr[i];
    };
    auto xstore_res = [&](int region) {                    // the staged chunk, rounded to bf16, into a region of the ring
#pragma unroll
        for (int i = 0; i < DPW; ++i)
            *reinterpret_cast<s16x4m*>(Xh + region * RGN + doff[i]) = pack_bf16x4(xr[i][0], xr[i][1], xr[i][2], xr[i][3]);
    };
    int rbase = 0;                                          // RES: region of this step's chunk 0
    auto region_of = [&](int base, int c) -> int { int r = base + c; return r > NCH ? r - (NCH + 1) : r; };
    if (RES) {
        // prologue: every chunk of step 0 into its region
#pragma unroll
        for (int c = 0; c < NCH; ++c) { xload(rowp_cur, c); xstore_res(c); }
    } else {
        // prologue: position 0 of step 0 into buffer 0, position 1 into the registers
        xload(rowp_cur, 0);
        xstore(0);
        xload(rowp_cur, 1 % NCH);
    }
    wg_barrier();

    constexpr int NOPT = (int)(sizeof(OptArgs) / 4);
    float opt_pre[NOPT];
    {
        const float* op = reinterpret_cast<const float*>(H.opt);
#pragma unroll
        for (int i = 0; i < NOPT; ++i) opt_pre[i] = op[i];
    }
    const unsigned epoch0 = H.epoch0;
    bool aborted = false;
    // phase stamps (diagnostic runs only; the values go to a buffer nothing else reads)
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool stamping = H.stamps != nullptr;
#define MSTAMP(i) do { if (stamping) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; } } while (0)
    if (stamping) tprev = __builtin_readcyclecounter();
    for (int k = 0; k < n_steps && !aborted; ++k) {
        const bool more = k + 1 < n_steps;
        const int ri = rows_of(0, 0), rt = rows_of(0, 1);
        const int ti = (ri + 15) >> 4, tt = (rt + 15) >> 4;
        const int mod = wave < ti ? 0 : 1;                               // modality of this wave's sample tile
        const bool tile_live = wave < ti + tt;
        const int local = mod == 0 ? 16 * wave + s16 : 16 * (wave - ti) + s16;
        const bool valid = tile_live && local < (mod == 0 ? ri : rt);
        const int lab = labs[(k & 1) * MROWS + 16 * wave + s16];
        if (tid < MROWS && k + 2 < n_steps) fetch_rid(2);
        if (tid == 0) misc[63] = 0.f;                       // abort flag of this step (read after the gather's barrier)
        if (more) {                                         // the next step's row pointers (its table was finished a step ago)
#pragma unroll
            for (int i = 0; i < DPW; ++i) rowp_nxt[i] = reinterpret_cast<const float*>(rowbase[((k + 1) & 1) * MROWS + ((wave * DPW + i) * 64 + lane) / K::SLOTS]);
        }
        // the step's optimizer scalars (requested one step ahead), made provably wave-uniform: scalar registers, scalar branches
        OptArgs o;
        {
            float tmp[NOPT];
#pragma unroll
            for (int i = 0; i < NOPT; ++i) tmp[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(opt_pre[i])));
            __builtin_memcpy(&o, tmp, sizeof(OptArgs));
            const float* op = reinterpret_cast<const float*>(H.opt + (more ? k + 1 : k));
#pragma unroll
            for (int i = 0; i < NOPT; ++i) opt_pre[i] = op[i];
        }
        const int okind = __builtin_amdgcn_readfirstlane(H.opt_kind);
        int off_next[2];                                    // offs[k + 4]: enters the window at the end of this step
#pragma unroll
        for (int md = 0; md < 2; ++md) off_next[md] = H.offs[md] ? H.offs[md][k + 4 < n_steps ? k + 4 : n_steps] : 0;
        MSTAMP(0);

        // Position p of the step (chunk p is complete in LDS buffer p & 1): write chunk p+1 (in the registers) to the other
        // buffer -- its last readers passed the barrier that ended position p-1 -- and request chunk p+2.
        auto stage = [&](int p) {
            if (RES) {
                // only the NEXT step's chunks move: chunk c is requested at position NCH + c - 1 and written, one position
                // later, to the region this step's chunk c - 1 has just left (chunk 0: to the spare region)
                if (more) {
                    const int nb = region_of(rbase, NCH);   // base of the next step
                    if (p >= NCH) xstore_res(region_of(nb, p - NCH));
                    if (p >= NCH - 1 && p + 1 < NP) xload(rowp_nxt, p + 1 - NCH);
                }
            } else {
                if (p + 1 < NP || more) xstore((p + 1) & 1);
                if (p + 2 < NP) xload(rowp_cur, (p + 2) % NCH);
                else if (more) xload(rowp_nxt, (p + 2 - NP) % NCH);
            }
            __builtin_amdgcn_sched_barrier(0);              // the requests stay ahead of the position's MFMAs
        };

        // ================= forward: raw[class 4g+r][sample s16 of tile `wave`] =================
        f32x4m acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            stage(c);
            MSTAMP(7);
            constexpr int T = CW / 16;
            if (RES) {
                if (tile_live) {
                    const unsigned short* xrow = Xh + region_of(rbase, c) * RGN + (16 * wave + s16) * CW;
                    const float* wrow = Wl + s16 * LDW + c * CW + 4 * g;
                    constexpr int T2 = CW / 16;
                    f32x4m a4[3];
                    s16x4m b4[3];
                    auto ld = [&](int t, int sl) {
                        a4[sl] = *reinterpret_cast<const f32x4m*>(wrow + 16 * t);
                        b4[sl] = *reinterpret_cast<const s16x4m*>(xrow + 4 * ((4 * t + g) ^ (s16 & SM)));
                    };
                    ld(0, 0);
                    ld(1, 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < T2; ++t) {
                        if (t + 2 < T2) ld(t + 2, (t + 2) % 3);
                        __builtin_amdgcn_sched_barrier(0);
                        const int sl = t % 3;
                        const s16x4m pa = pack_bf16x4(a4[sl][0], a4[sl][1], a4[sl][2], a4[sl][3]);
                        if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, b4[sl], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, b4[sl], acc0, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                MSTAMP(1);
                if (c == NCH - 1) wg_barrier();             // (one barrier for the whole forward: the chunks are resident)
                MSTAMP(8);
                continue;
            }
            if (tile_live) {
                const float* xrow = Xb + (c & 1) * (MROWS * CW) + (16 * wave + s16) * CW;
                const float* wrow = Wl + s16 * LDW + c * CW + 4 * g;
                // one wave per SIMD: nothing but this wave's own earlier loads can hide the LDS latency, so the operand
                // reads run two k-steps ahead of their MFMAs through a ring of three register sets
                f32x4m a4[3], b4[3];
                auto ld = [&](int t, int sl) {
                    a4[sl] = *reinterpret_cast<const f32x4m*>(wrow + 16 * t);
                    b4[sl] = *reinterpret_cast<const f32x4m*>(xrow + 4 * ((4 * t + g) ^ (s16 & SM)));
                };
                ld(0, 0);
                if (T > 1) ld(1, 1);
                __builtin_amdgcn_sched_barrier(0);          // pin the order: left alone, the scheduler sinks every read to its MFMAs
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if (t + 2 < T) ld(t + 2, (t + 2) % 3);
                    __builtin_amdgcn_sched_barrier(0);
                    const int sl = t % 3;
                    if (BF) {                               // k = 16t + 4g + e for element e of lane group g, in both operands
                        const s16x4m pa = pack_bf16x4(a4[sl][0], a4[sl][1], a4[sl][2], a4[sl][3]);
                        const s16x4m pb = pack_bf16x4(b4[sl][0], b4[sl][1], b4[sl][2], b4[sl][3]);
                        if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pb, acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pb, acc0, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sl][0], b4[sl][0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sl][1], b4[sl][1], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sl][2], b4[sl][2], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sl][3], b4[sl][3], acc1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            MSTAMP(1);
            wg_barrier();
            MSTAMP(8);
        }
        asm volatile("s_nop 7" ::: "memory");              // see the note at the dW chain: result latency of the last MFMA
        // next-but-one step's row table: the row ids were requested at the top of the step; the label load they feed is
        // issued here and consumed after the gather.  (Parity k & 1: this step's own table is no longer read.)
        if (tid < MROWS && k + 2 < n_steps) publish_rowbase(2, k + 2);
        // ---- slice-local softmax statistics of this lane's sample ----
        const float sc = scale[mod];
        const float NEG_INF = -__builtin_huge_valf();
        float raw[4], z[4];
        float mloc = NEG_INF;
        int amax = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            raw[r] = acc0[r] + acc1[r];
            const int cls = c0 + 4 * g + r;
            z[r] = cls < C ? raw[r] * sc : NEG_INF;
            if (z[r] > mloc) { mloc = z[r]; amax = cls; }
        }
        rows_argmax(mloc, amax);
        float e[4], sloc = 0.f, serw = 0.f, rawy = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cls = c0 + 4 * g + r;
            e[r] = cls < C ? expf(z[r] - mloc) : 0.f;
            sloc += e[r];
            serw += e[r] * raw[r];
            if (cls == lab) rawy = raw[r];
        }
        sloc = rows_sum(sloc);
        serw = rows_sum(serw);
        rawy = rows_sum(rawy);
        // ---- publish this slice's records: 8-byte {epoch, value} granules, one write-through store each (the data
        // IS the flag: MI355X guide, Guideline 16 recipe R2), into the parity buffer of this step ----
        const unsigned epoch = epoch0 + (unsigned)k + 1u;
        const int par = (int)(epoch & 1u);
        const int nfld = learn ? 5 : 4;
        if (g == 0 && tile_live) {
            unsigned long long* rec = H.xchg + (size_t)(par * nwg + slice) * (5 * MROWS) + 16 * wave + s16;
            store_granule(rec, epoch, mloc);
            store_granule(rec + MROWS, epoch, sloc);
            store_granule(rec + 2 * MROWS, epoch, rawy);
            store_granule(rec + 3 * MROWS, epoch, __int_as_float(amax));
            if (learn) store_granule(rec + 4 * MROWS, epoch, serw);
        }
        MSTAMP(2);
        // ---- gather: thread (sample = tid & 63, part = tid >> 6) sweeps the granules of slices part, part + 4, ...
        // until their tags carry this step's epoch (bounded), merging with the online-softmax rule ----
        {
            const int smp = tid & 63, part = tid >> 6;
            const bool slot_live = (smp >> 4) < ti + tt;                  // slots of absent tiles are never published
            float M = NEG_INF, S = 0.f, RY = 0.f, SR = 0.f;
            int AM = 0x7fffffff;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int bad = 0;
            for (int w = part; w < nwg && !bad; w += 4) {
                const unsigned long long* rec = H.xchg + (size_t)(par * nwg + w) * (5 * MROWS) + smp;
                unsigned long long gq[5];
                for (unsigned spin = 0;; ++spin) {
                    bool ok = true;
#pragma unroll
                    for (int f = 0; f < 5; ++f) {
                        if (f < nfld) {
                            gq[f] = __hip_atomic_load(rec + f * MROWS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ok && (unsigned)(gq[f] >> 32) == epoch;
                        }
                    }
                    if (__all(ok || !slot_live)) break;
                    if ((spin & 63u) == 63u) {
                        const unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (st != 0u || __builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { bad = 1; break; }   // 50 ms at 100 MHz (umlh_micro_status reports it)
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                const float m_ = slot_live ? __uint_as_float((unsigned)gq[0]) : NEG_INF, s_ = __uint_as_float((unsigned)gq[1]);
                const float ry_ = slot_live ? __uint_as_float((unsigned)gq[2]) : 0.f;
                const int am_ = (int)(unsigned)gq[3];
                const float sr_ = learn ? __uint_as_float((unsigned)gq[4]) : 0.f;
                const float Mn = fmaxf(M, m_);
                const float fo = M == NEG_INF ? 0.f : expf(M - Mn), fn = m_ == NEG_INF ? 0.f : expf(m_ - Mn);
                S = S * fo + s_ * fn;
                SR = SR * fo + sr_ * fn;
                RY += ry_;
                if (m_ > M || (m_ == M && am_ < AM)) AM = am_;
                M = Mn;
            }
            if (bad) {
                if (lane == 0) {
                    __hip_atomic_store(status, 1u + (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    misc[63] = 1.f;
                }
            }
            float* rp = red + (part * MROWS + smp) * 5;
            rp[0] = M; rp[1] = S; rp[2] = RY; rp[3] = __int_as_float(AM); rp[4] = SR;
        }
        if (tid < MROWS && k + 2 < n_steps) publish_labels(k + 2);
        wg_barrier();
        if (misc[63] != 0.f) { aborted = true; break; }
        MSTAMP(3);
        float M = NEG_INF, S = 0.f, RY = 0.f, SR = 0.f;
        int AM = 0x7fffffff;
#pragma unroll
        for (int part = 0; part < 4; ++part) {
            const float* rp = red + (part * MROWS + 16 * wave + s16) * 5;
            const float m_ = rp[0], s_ = rp[1], sr_ = rp[4];
            const int am_ = __float_as_int(rp[3]);
            const float Mn = fmaxf(M, m_);
            const float fo = M == NEG_INF ? 0.f : expf(M - Mn), fn = m_ == NEG_INF ? 0.f : expf(m_ - Mn);
            S = S * fo + s_ * fn;
            SR = SR * fo + sr_ * fn;
            RY += rp[2];
            if (m_ > M || (m_ == M && am_ < AM)) AM = am_;
            M = Mn;
        }
        MSTAMP(4);
        // ---- dZ^T of this slice -> LDS ----
        const float wmod = mod == 0 ? H.w_img : H.w_txt;
        const float w_over_rows = wmod / (float)(mod == 0 ? ri : rt);
        const float coef = valid ? w_over_rows * sc : 0.f;
        {
            const float f = tile_live ? expf(mloc - M) * __builtin_amdgcn_rcpf(S) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cls = c0 + 4 * g + r;
                const float p = e[r] * f;
                dzT[(4 * g + r) * LDZ + 16 * wave + s16] = (valid && cls < C) ? (p - (cls == lab ? 1.f : 0.f)) * coef : 0.f;
            }
        }
        // per-sample scalars (identical in every slice; slice 0 writes the step's row): loss, top-1, d loss / d scale
        float vl = 0.f, vc = 0.f, vg = 0.f;
        if (valid && g == 0) {
            vl = logf(S) + M - RY * sc;
            vc = AM == lab ? 1.f : 0.f;
            vg = SR / S - RY;
        }
        vl = row16_sum(vl); vc = row16_sum(vc); vg = row16_sum(vg);
        if (lane == 0) { misc[4 * wave + 0] = vl; misc[4 * wave + 1] = vc; misc[4 * wave + 2] = vg; }
        wg_barrier();
        // scalar sums per modality in tile order (every thread computes the same values)
        float sl[2] = {0.f, 0.f}, sa[2] = {0.f, 0.f}, sg[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ti + tt) {
                const int md = t < ti ? 0 : 1;
                sl[md] += misc[4 * t + 0]; sa[md] += misc[4 * t + 1]; sg[md] += misc[4 * t + 2];
            }
        }
        const float inv0 = ri > 0 ? 1.f / (float)ri : 0.f, inv1 = rt > 0 ? 1.f / (float)rt : 0.f;
        const float gs0 = sg[0] * H.w_img * inv0, gs1 = sg[1] * H.w_txt * inv1;
        if (slice == 0 && tid == 0 && H.scalars_out != nullptr) {
            float* so = H.scalars_out + (size_t)k * UMLH_N_SCALARS;
            so[UMLH_S_LOSS_IMG] = sl[0] * inv0; so[UMLH_S_LOSS_TXT] = sl[1] * inv1;
            so[UMLH_S_ACC_IMG] = sa[0] * inv0;  so[UMLH_S_ACC_TXT] = sa[1] * inv1;
            so[UMLH_S_GSCALE_IMG] = gs0;        so[UMLH_S_GSCALE_TXT] = gs1;
            so[UMLH_S_CORRECT] = sa[0] + sa[1]; so[UMLH_S_LOSS_SUM] = sl[0] + sl[1];
        }
        const float inv_bc2 = 1.f / o.bc2_sqrt;
        // A operand of the dW product: dzT[class s16][row 4m + g]
        float af[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) af[m] = dzT[s16 * LDZ + 4 * m + g];
        MSTAMP(5);

        // ================= backward + update: dW[class 4g+r][column], straight into the optimizer =================
        // (Issuing the optimizer arithmetic of chunk c-1 between the MFMAs of chunk c was measured and bought nothing: the
        // f32-input MFMA runs at the f32 vector rate, the two compete for the same issue slots.)
        auto wptr = [&](int c) -> float* { return Wl + (4 * g) * LDW + c * CW + (colok ? colw : 0); };
        auto upd_math = [&](auto kind, int c, const f32x4m (&dacc)[U], float (&pw)[U][4], float (&vv)[U][4], float (&mv)[U][4]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    opt_update_fast<decltype(kind)::value>(o, inv_bc2, dacc[u][r], pw[u][r], MMEM ? mv[u][r] : mreg[c][u][r],
                                                           VMEM ? vv[u][r] : vreg[c][u][r]);
        };
        // VMEM: this thread's second moments of chunk c (class 4g + r, column c*CW + 64u + colw); rows >= C / missing columns
        // read a valid element and are not written back
        auto v_ptr = [&](int c, int u, int r) -> float* { return vbase[r] + (c * CW + 64 * u); };   // (constant offset: an immediate)
        auto v_read = [&](int c, float (&vv)[U][4]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) vv[u][r] = adam ? *v_ptr(c, u, r) : 0.f;
        };
        auto v_write = [&](int c, const float (&vv)[U][4]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (adam && c0 + 4 * g + r < C && colok) *v_ptr(c, u, r) = vv[u][r];
        };
        auto m_read = [&](int c, float (&mv)[U][4]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) mv[u][r] = *(mbase[r] + (c * CW + 64 * u));
        };
        auto m_write = [&](int c, const float (&mv)[U][4]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c0 + 4 * g + r < C && colok) *(mbase[r] + (c * CW + 64 * u)) = mv[u][r];
        };
        auto w_read = [&](int c, float (&pw)[U][4]) {
            const float* wp0 = wptr(c);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) pw[u][r] = wp0[r * LDW + 64 * u];
        };
        auto w_write = [&](int c, const float (&pw)[U][4]) {       // rows >= C and (CW < 64) lanes without a column keep the padding:
            float* wp0 = wptr(c);                                   // their stores go to a dump word (branch-free)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* dst = (c0 + 4 * g + r < C && colok) ? wp0 + r * LDW + 64 * u : misc + 32;
                    *dst = pw[u][r];
                }
        };
        // the optimizer kind is decided ONCE around the whole phase (straight-line code inside)
        auto dw_phase = [&](auto kind) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            stage(NCH + c);
            MSTAMP(10);
            const float* xb = Xb + ((NCH + c) & 1) * (MROWS * CW);
            const unsigned short* xh = Xh + (RES ? region_of(rbase, c) : 0) * RGN;
            f32x4m dacc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dacc[u] = f32x4m{0.f, 0.f, 0.f, 0.f};
            // all B operands of the chunk first (16 U independent LDS reads in flight), then the MFMA chains.  Lane s16 of
            // wave w owns column 64u + 16w + s16: consecutive lanes read consecutive dwords of a row (the slot swizzle
            // keeps the four rows of a k-step on different banks).  RES: the same elements as bf16 halves of the resident rows.
            float bq[16][U];
            unsigned short bh[16][U];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int row = 4 * m + g;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int col = colok ? 64 * u + colw : CW - 1;
                    const int at = row * CW + 4 * ((col >> 2) ^ (row & SM)) + (col & 3);
                    if (RES) bh[m][u] = xh[at]; else bq[m][u] = xb[at];
                }
            }
            float pw[U][4];
            w_read(c, pw);
            float vv[U][4], mv[U][4];
            if (VMEM) v_read(c, vv);
            if (MMEM) m_read(c, mv);
            __builtin_amdgcn_sched_barrier(0);              // reads stay ahead of the MFMA chains
            if (BF) {                                       // rows 16mm + 4e + g for element e of lane group g, in both operands
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const s16x4m pa = pack_bf16x4(af[4 * mm], af[4 * mm + 1], af[4 * mm + 2], af[4 * mm + 3]);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const s16x4m pb = RES ? s16x4m{(short)bh[4 * mm][u], (short)bh[4 * mm + 1][u], (short)bh[4 * mm + 2][u], (short)bh[4 * mm + 3][u]}
                                              : pack_bf16x4(bq[4 * mm][u], bq[4 * mm + 1][u], bq[4 * mm + 2][u], bq[4 * mm + 3][u]);
                        dacc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pb, dacc[u], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int m = 0; m < 16; ++m)
#pragma unroll
                    for (int u = 0; u < U; ++u) dacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bq[m][u], dacc[u], 0, 0, 0);
            }
            // The update reads the accumulators right behind the last MFMA of the chain.  hipcc (ROCm 7.2) pads that read
            // with `s_nop 8`; on gfx950 v_mfma_f32_16x16x4_f32 has a 40-cycle result latency and the LAST accumulator
            // register came back without the final k-step's contribution (rows 60..63 of the batch missing from every
            // class 4g + 3: found by the oracle comparison at batch 64).  Eight more wait states, pinned in place.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MSTAMP(9);
            upd_math(kind, c, dacc, pw, vv, mv);
            w_write(c, pw);
            if (VMEM) v_write(c, vv);
            if (MMEM) m_write(c, mv);
            MSTAMP(6);
            wg_barrier();                                   // chunk's buffer free; after the last chunk: the whole W slice updated
            MSTAMP(11);
        }
        };
        if (okind == UMLH_OPT_ADAMW) dw_phase(std::integral_constant<int, UMLH_OPT_ADAMW>{});
        else if (okind == UMLH_OPT_ADAM) dw_phase(std::integral_constant<int, UMLH_OPT_ADAM>{});
        else dw_phase(std::integral_constant<int, UMLH_OPT_SGD>{});
        // learnable logit scales (head.py:69-70): every slice applies the same update to its private copy
        if (learn) {
            auto upds = [&](auto kind) {
                if (ri > 0) opt_update_fast<decltype(kind)::value>(o, inv_bc2, gs0, scale[0], msc[0], vsc[0]);
                if (rt > 0) opt_update_fast<decltype(kind)::value>(o, inv_bc2, gs1, scale[1], msc[1], vsc[1]);
            };
            if (okind == UMLH_OPT_ADAMW) upds(std::integral_constant<int, UMLH_OPT_ADAMW>{});
            else if (okind == UMLH_OPT_ADAM) upds(std::integral_constant<int, UMLH_OPT_ADAM>{});
            else upds(std::integral_constant<int, UMLH_OPT_SGD>{});
        }
        // advance: the next step's row pointers become current, the offsets window moves on
        if (RES) rbase = region_of(rbase, NCH);
#pragma unroll
        for (int i = 0; i < DPW; ++i) rowp_cur[i] = rowp_nxt[i];
#pragma unroll
        for (int md = 0; md < 2; ++md) { off[md][0] = off[md][1]; off[md][1] = off[md][2]; off[md][2] = off[md][3]; off[md][3] = off_next[md]; }
    }

    if (stamping && tid == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) H.stamps[(size_t)slice * 12 + i] = tacc[i];
    }
#undef MSTAMP
    // ---- write the resident state back ----
    wg_barrier();
    for (int i = tid; i < CS * (D / 4); i += 256) {
        const int r = i / (D / 4), q = i % (D / 4);
        if (c0 + r < C) *reinterpret_cast<f32x4m*>(H.w + (size_t)(c0 + r) * D + 4 * q) = *reinterpret_cast<const f32x4m*>(Wl + r * LDW + 4 * q);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cls = c0 + 4 * g + r, col = c * CW + 64 * u + colw;
                if (cls < C && colok) {
                    if (!MMEM) H.m[(size_t)cls * D + col] = mreg[c][u][r];
                    if (adam && !VMEM) H.v[(size_t)cls * D + col] = vreg[c][u][r];
                }
            }
    if (learn && slice == 0 && tid == 0) {
        H.scales[0] = scale[0]; H.scales[1] = scale[1];
        H.m_scales[0] = msc[0]; H.m_scales[1] = msc[1];
        H.v_scales[0] = vsc[0]; H.v_scales[1] = vsc[1];
    }
}

template <int NCH, int CW, bool BF>
int launch_one(const UmlhMicroHead* heads, int n_heads, int n_steps, int grid, hipStream_t st) {
    using K = MicroCfg<NCH, CW, BF && CW == 128 && NCH <= 5>;
    static std::atomic<unsigned long long> attr_done{0};   // bit d: done on device d
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ULL)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&micro_steps_kernel<NCH, CW, BF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, K::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_done.fetch_or(1ULL << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((micro_steps_kernel<NCH, CW, BF>), dim3(grid), dim3(256), K::SMEM, st, heads, n_heads, n_steps);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// chunking of a feature width: widest power-of-two chunk <= 128 dividing d; 0 = unsupported
int umlh_micro_chunking(int d, int* nch, int* cw) {
    static const int table[][2] = {{1, 16}, {1, 32}, {3, 16}, {1, 64}, {5, 16}, {3, 32}, {1, 128}, {2, 128}, {3, 128},
                                   {4, 128}, {5, 128}, {6, 128}, {8, 128}};
    for (const auto& t : table)
        if (t[0] * t[1] == d) { *nch = t[0]; *cw = t[1]; return 1; }
    return 0;
}

// bf16 operand mode: widths that are multiples of 128 (the general bf16 path's own requirement)
int umlh_micro_bf16_supported(int nch, int cw) { return cw == 128 && nch >= 1 && nch <= 8 && nch != 7; }

#define MICRO_CASE(N_, W_) if (nch == N_ && cw == W_) return launch_one<N_, W_, false>(heads, n_heads, n_steps, grid, st);
#define MICRO_CASE_BF(N_) if (nch == N_ && cw == 128) return launch_one<N_, 128, true>(heads, n_heads, n_steps, grid, st);
int umlh_micro_launch(int nch, int cw, int bf16, const UmlhMicroHead* heads, int n_heads, int n_steps, int grid, hipStream_t st) {
    if (bf16) {
        MICRO_CASE_BF(1) MICRO_CASE_BF(2) MICRO_CASE_BF(3) MICRO_CASE_BF(4) MICRO_CASE_BF(5) MICRO_CASE_BF(6) MICRO_CASE_BF(8)
        return (int)hipErrorInvalidValue;
    }
    MICRO_CASE(1, 16) MICRO_CASE(1, 32) MICRO_CASE(3, 16) MICRO_CASE(1, 64) MICRO_CASE(5, 16) MICRO_CASE(3, 32)
    MICRO_CASE(1, 128) MICRO_CASE(2, 128) MICRO_CASE(3, 128) MICRO_CASE(4, 128) MICRO_CASE(5, 128) MICRO_CASE(6, 128)
    MICRO_CASE(8, 128)
    return (int)hipErrorInvalidValue;
}

}  // extern "C"
