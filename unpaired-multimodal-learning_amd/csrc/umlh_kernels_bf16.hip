// bf16-operand (throughput-mode) kernels of the UML head step for gfx950.
//
// Operands (feature rows, head weight, dZ) are bf16 in HBM/LDS, every accumulation is
// fp32 in the MFMA (v_mfma_f32_32x32x16_bf16), the softmax/CE epilogue and the optimizer
// are fp32 on the fp32 master weights.  Validated on accuracy (+-0.1 pp) and loss, not on
// 1e-4 logits (bf16 rounding of operands scaled by 100 cannot meet that).
//
//   to_bf16        fp32 -> bf16 shadow copies (head weight each step, feature tables once)
//   fwd_ce_bf16    fused  X W^T * scale -> softmax-CE -> dZ^T (bf16), loss / top-1 / dscale
//   dw_bf16        dW = dZ^T F with the k-strided operand read through ds_read_b64_tr_b16
#include "umlh_common.h"
#include <type_traits>
#include <atomic>
#include <cstring>

// Timing-only ablations (skip the main loop / the epilogue / an operand's traffic) exist for kernel analysis and are
// compiled in only with -DUMLH_ABLATIONS: the shipped library has no path that skips work.  The cycle stamps
// (UMLH_DBG_FWD=9, UMLH_DBG_DW=16) do not change what is computed and stay available.
#ifdef UMLH_ABLATIONS
#define UMLH_ABL(cond) (cond)
#else
#define UMLH_ABL(cond) (false)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int KTB = 32;     // bf16 k elements staged per LDS chunk (two MFMA k-steps of 16)
constexpr int RSB = 40;     // LDS row stride in shorts for k-contiguous tiles: 64 B data + 16 B pad
                            // -> ds_read_b128 of 16 different rows hits 16 different 16-B bank groups

// write-through (sc1) stores: the data is on its way to memory when the instruction retires instead of sitting dirty in
// the XCD's L2 until the end-of-kernel write-back that the next (dependent) launch has to wait for
__device__ __forceinline__ void store_wt_b128(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ u16 f2bf(float f) {
    __bf16 b = (__bf16)f;                       // v_cvt_pk_bf16_f32, round-to-nearest-even
    return __builtin_bit_cast(u16, b);
}

__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, long long n) {
    long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 <= n) {
        f32x4v a = *reinterpret_cast<const f32x4v*>(src + i);
        f32x4v b = *reinterpret_cast<const f32x4v*>(src + i + 4);
        u32x4 o;
        o[0] = f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
        o[1] = f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
        o[2] = f2bf(b[0]) | ((unsigned)f2bf(b[1]) << 16);
        o[3] = f2bf(b[2]) | ((unsigned)f2bf(b[3]) << 16);
        *reinterpret_cast<u32x4*>(dst + i) = o;
    } else {
        for (long long e = i; e < n; ++e) dst[e] = f2bf(src[e]);
    }
}

__global__ __launch_bounds__(256) void iota_kernel(long long* dst, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = i;
}

// Head-weight shadow for the forward kernel: bf16, MFMA-FRAGMENT-MAJOR
//   [K/16 k-step][CPAD/32 class tile][64 lanes][8 bf16]
// lane l of tile t, k-step s holds W[32 t + (l & 31)][16 s + 8 (l >> 5) .. +8): exactly the A
// operand of v_mfma_f32_32x32x16_bf16, so a wave fetches one operand with ONE fully coalesced
// 1-KiB global_load_dwordx4 straight into registers (no LDS round trip for the streamed
// operand).  Class rows >= C are zero.
__global__ __launch_bounds__(256) void w_shadow_kernel(const float* __restrict__ w, u16* __restrict__ dst, int C, int K,
                                                       int cpad) {
    long long piece = (long long)blockIdx.x * 256 + threadIdx.x;          // one lane-fragment (16 B)
    const int tiles = cpad / 32;
    long long total = (long long)(K / 16) * tiles * 64;
    if (piece >= total) return;
    int lane = (int)(piece & 63);
    int tile = (int)((piece >> 6) % tiles);
    int ks = (int)((piece >> 6) / tiles);
    int cls = tile * 32 + (lane & 31);
    u32x4 o = {0u, 0u, 0u, 0u};
    if (cls < C) {
        const float* src = w + (size_t)cls * K + ks * 16 + 8 * (lane >> 5);
        f32x4v a = *reinterpret_cast<const f32x4v*>(src);
        f32x4v b = *reinterpret_cast<const f32x4v*>(src + 4);
        o[0] = f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
        o[1] = f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
        o[2] = f2bf(b[0]) | ((unsigned)f2bf(b[1]) << 16);
        o[3] = f2bf(b[2]) | ((unsigned)f2bf(b[3]) << 16);
    }
    *reinterpret_cast<u32x4*>(dst + piece * 8) = o;
}

// --------------------------------------------------------------------------- //
// transposed bf16 shadows of fp32 weights (2-layer head): dst(c, r) = bf16(src[r][c]) for a row-major
// [R][Cc] source, 64x64 tiles through LDS so that both sides move whole 128/256-B segments.
//   MODE 0: dst row-major [Cc][ldd]                      (W_proj^T: the B operand of H = X W_proj^T)
//   MODE 1: dst chunk-major over r: [(R+63)/64][ldd][64]  (W_head^T by class chunks: the A operand of dH^T);
//           r >= R inside the last chunk is written as 0
// --------------------------------------------------------------------------- //
template <int MODE>
__global__ __launch_bounds__(256) void transpose_shadow_kernel(const float* __restrict__ src, int R, int Cc, int ldd,
                                                               u16* __restrict__ dst) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int rr = (t >> 6) + 4 * j, cc = t & 63;          // 64 consecutive columns per row: 256-B reads
        const int r = r0 + rr, c = c0 + cc;
        tile[rr][cc] = (r < R && c < Cc) ? src[(size_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    const int cc = t >> 2, seg = (t & 3) * 16;                  // 16 consecutive r of one c: 32 B
    const int c = c0 + cc;
    if (c >= Cc) return;
    u32x4 o[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned v = f2bf(tile[seg + 2 * j][cc]) | ((unsigned)f2bf(tile[seg + 2 * j + 1][cc]) << 16);
        o[j >> 2][j & 3] = v;
    }
    if (MODE == 0) {
        if (r0 + seg + 16 <= R) {                               // ldd % 8 == 0 and r0, seg multiples of 16: aligned
            u32x4* d = reinterpret_cast<u32x4*>(dst + (size_t)c * ldd + r0 + seg);
            d[0] = o[0]; d[1] = o[1];
        } else {
            for (int j = 0; j < 16 && r0 + seg + j < R; ++j) dst[(size_t)c * ldd + r0 + seg + j] = f2bf(tile[seg + j][cc]);
        }
    } else {
        u32x4* d = reinterpret_cast<u32x4*>(dst + ((size_t)blockIdx.y * ldd + c) * 64 + seg);
        d[0] = o[0]; d[1] = o[1];
    }
}

// --------------------------------------------------------------------------- //
// fused forward + cross entropy, bf16 operands
// --------------------------------------------------------------------------- //
// Streamed operand W: global -> registers (fragment-major shadow, PD k-steps deep ring per wave,
// 16 KiB in flight per wave).  Shared operand X: the block's TS sample rows stay resident in LDS
// for a K-block of XK and every wave reads its B fragment with one ds_read_b128 per k-step.
// TakeHook (the one-launch step): the workgroup's takes of its home tasks were ISSUED at kernel start (returning atomics of
// threads 0..3, result in `old`) but not waited for; the forward body starts on its loads right away and settles the takes --
// verdicts into verdict[0..3] -- in front of its first barrier, where wave 0 waits for its own X loads anyway.  Nothing has
// been stored by then: a tile that turns out to be somebody else's (this workgroup arrived late and a waiter took it) returns false.
struct TakeHook { unsigned old; unsigned epoch; int* verdict; };
template <int CTW, int WC, int STW>
__device__ __forceinline__ bool fwd_ce_bf16_body(const FwdArgsB& a, const int bid, unsigned char* smem_raw, const TakeHook* hook = nullptr) {
    constexpr int WS = 8 / WC;
    constexpr int CPAD = 32 * CTW * WC;
    constexpr int TS = 32 * STW * WS;
    constexpr int XK = WS <= 2 ? 512 : (WS == 4 ? 256 : 128);   // K-block resident in LDS
    constexpr int XRS = XK + 8;                                  // row stride (shorts): odd multiple of 16 B
    constexpr int PD = STW == 1 ? 8 : 4;                         // k-steps of W fragments in flight (ring depth)
    constexpr int NPX = (TS * (XK / 8)) / 512;                   // 16-B pieces of the X block per thread
    u16* Xt = reinterpret_cast<u16*>(smem_raw);                  // [TS][XRS]
    constexpr int STAGE_BYTES = 8 * CTW * 32 * 80;               // dZ staging: 8 waves x [CTW*32 rows][80 B (64 data + 16 pad)]
    constexpr int XT_BYTES = TS * XRS * 2;
    constexpr int UNION_BYTES = XT_BYTES > STAGE_BYTES ? XT_BYTES : STAGE_BYTES;
    unsigned* dzstage = reinterpret_cast<unsigned*>(smem_raw);   // aliases Xt: used only after the last barrier of pass 2
    float* red = reinterpret_cast<float*>(smem_raw + UNION_BYTES);   // [WS][WC][32][4]
    float* red2 = red + 8 * 32 * 8;                              // [WS][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave % WC, ws = wave / WC;
    const int h = lane >> 5, l31 = lane & 31;
    const int sidx = bid >= a.seg[1].blk0 ? 1 : 0;
    const SegDescB& sg = a.seg[sidx];
#define STAMP(i) do { if (a.stamps && lane == 0) a.stamps[((size_t)bid * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    STAMP(0);
    const int row0 = (bid - sg.blk0) * TS;
    const int C = a.C, K = a.K;

    // X pieces of this thread: piece p -> row p / (XK/8), 16-B column p % (XK/8).  Rows past the
    // segment are clamped to its last row (finite garbage the epilogue discards).
    const u16* xsrc[NPX];
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
        int p = tid + 512 * q;
        int r = min(row0 + p / (XK / 8), sg.rows - 1);
        int64_t rid = sg.feat_index ? sg.feat_index[r] : (int64_t)r;
        xsrc[q] = sg.feats + (size_t)rid * sg.ld + 8 * (p % (XK / 8));
    }

    // labels of this lane's samples: two dependent global loads (row id, then label) issued NOW so
    // their latency hides behind the whole main loop instead of stalling the epilogue
    int labs[STW];
#pragma unroll
    for (int st = 0; st < STW; ++st) {
        int r = min(row0 + ws * 32 * STW + st * 32 + l31, sg.rows - 1);
        labs[st] = (int)sg.labels[sg.label_index ? sg.label_index[r] : (int64_t)r];
    }

    f32x16 acc[CTW][STW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
        for (int st = 0; st < STW; ++st)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][st][i] = 0.f;

    const int nks = K / 16;                                      // total k-steps (multiple of PD)
    // fragment (k-step ks, class tile t) = 1 KiB at W + ((ks * CPAD/32 + t) * 64 + lane) * 8
    const u16* wlane = a.W + ((size_t)(wc * CTW) * 64 + lane) * 8;
    auto wfrag = [&](int ks, int ct) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(wlane + ((size_t)ks * (CPAD / 32) + ct) * 512);
    };
    // a wave whose sample rows all lie past the segment (batch 32 in a 64-row block) or whose classes are all
    // padding neither streams W nor issues MFMAs (wave-uniform); its accumulators stay 0 and are masked below
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool live = row0 + (wave_u / WC) * 32 * STW < sg.rows && (wave_u % WC) * CTW * 32 < C;
    bf16x8 ring[PD][CTW];
    if (live) {
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) ring[d][ct] = wfrag(min(d, nks - 1), ct);
    }

    STAMP(1);
    for (int kb0 = 0; kb0 < (UMLH_ABL(a.dbg == 2 || a.dbg == 5) ? 0 : K); kb0 += XK) {
        const int kbw = min(XK, K - kb0);                        // multiple of 128
        // ---- stage the X block: global -> registers -> LDS ----
        u32x4 xr[NPX];
#pragma unroll
        for (int q = 0; q < NPX; ++q) {
            int col = 8 * ((tid + 512 * q) % (XK / 8));
            xr[q] = *reinterpret_cast<const u32x4*>(xsrc[q] + kb0 + min(col, kbw - 8) - col);
        }
        if (hook != nullptr && kb0 == 0 && tid < 4) hook->verdict[tid] = tag_older(hook->old, hook->epoch) ? 1 : 0;
        __syncthreads();                                         // previous block fully consumed
        if (hook != nullptr && kb0 == 0 && !hook->verdict[0]) return false;   // (uniform: the forward tile's verdict)
#pragma unroll
        for (int q = 0; q < NPX; ++q) {
            int p = tid + 512 * q;
            *reinterpret_cast<u32x4*>(Xt + (p / (XK / 8)) * XRS + 8 * (p % (XK / 8))) = xr[q];
        }
        __syncthreads();
        // ---- k-steps of this block ----
        const int ks0 = kb0 / 16, nst = kbw / 16;                // nst is a multiple of PD
        const u16* xrow = Xt + (ws * 32 * STW + l31) * XRS + h * 8;
        // one group = PD k-steps; REFILL = the slot just consumed is reloaded PD k-steps ahead.  The
        // very last group of the kernel has nothing left to fetch: issuing clamped refills there cost
        // PD/nks of extra L2->CU traffic on the stream that bounds this kernel, and made the epilogue
        // wait on them before it could reuse the ring registers.
        auto kgroup = [&](int s, auto refill) {
            constexpr bool REFILL = decltype(refill)::value;
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                bf16x8 b[STW];
#pragma unroll
                for (int st = 0; st < STW; ++st)
                    b[st] = *reinterpret_cast<const bf16x8*>(xrow + st * 32 * XRS + (s + d) * 16);
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                    for (int st = 0; st < STW; ++st)
                        acc[ct][st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[d][ct], b[st], acc[ct][st], 0, 0, 0);
                if constexpr (REFILL) {
                    const int nxt = ks0 + s + d + PD;
#pragma unroll
                    for (int ct = 0; ct < CTW; ++ct) ring[d][ct] = wfrag(nxt, ct);
                }
                // pin the emitted order per k-step: B read, its MFMAs, then the slot's refill loads
                // (left alone, hipcc sinks all refills to the end of the group, which leaves the
                // next group's first fragment zero time to arrive)
                __builtin_amdgcn_sched_group_barrier(0x100, STW, 0);        // B reads
                __builtin_amdgcn_sched_group_barrier(0x008, CTW * STW, 0);  // MFMAs
                if constexpr (REFILL) __builtin_amdgcn_sched_group_barrier(0x020, CTW, 0);    // CTW VMEM reads
            }
        };
        const bool last_block = kb0 + XK >= K;
        const int nfull = last_block ? nst - PD : nst;
        if (live) {
            for (int s = 0; s < nfull; s += PD) kgroup(s, std::true_type{});
            if (last_block) kgroup(nst - PD, std::false_type{});
        }
    }

    STAMP(2);
    if (UMLH_ABL(a.dbg == 1 || a.dbg == 5)) {          // ablation: keep the accumulators live, skip the epilogue
        float t = 0.f;
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int st = 0; st < STW; ++st)
#pragma unroll
                for (int i = 0; i < 16; ++i) t += acc[ct][st][i];
        if (t == 123.456f) a.partials[0] = t;
        return true;
    }
    // ---------------- epilogue (per 32-sample tile of this wave) ----------------
    // VALU-bound (64 logits per lane per tile), so every pass is kept to very few instructions per
    // element and free of per-element class masks / sign selects:
    //   * the scale's sign and the padded classes of a partial tile are folded into the accumulators
    //     once, in wave-uniform branches that are almost never taken;
    //   * max = v_max3 chains; first arg-max = (value == max) overwrite in descending order with the
    //     register number as an inline constant; label logit = select tree on the bits of the label's
    //     register number (63 selects instead of 64 compares + 64 selects);
    //   * exp / sums / dZ scaling work on register PAIRS (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32);
    //   * dZ is packed by v_cvt_pk_bf16_f32 and staged in LDS; the one-hot term is patched into the
    //     staged tile by the one lane that owns the label element.
    static_assert(CTW == 1 || CTW == 2 || CTW == 4, "label select tree needs a power-of-two CTW");
    constexpr int NREG = CTW * 16;                       // logits per lane per sample tile
    constexpr int ZRS = 20;                              // staged dZ row stride (dwords): 64 B of data + 16 B pad
    const float scale = *sg.scale_ptr;
    const float sgn = scale < 0.f ? -1.f : 1.f;
    const float LOG2E = 1.4426950408889634f;
    const float ascale = __builtin_fabsf(scale);
    // exponent slope; the floor only matters for scale == 0, where it keeps padded classes at e = 0
    const float asl2 = __builtin_fmaxf(ascale * LOG2E, 1e-20f);
    const float MASKED = -1e30f;
    const bool learn = a.learn != 0;
    if (scale < 0.f) {                                   // argmax/softmax of scale*raw == those of |scale| * (-raw)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int st = 0; st < STW; ++st)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][st][i] = -acc[ct][st][i];
    }
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
        const int cbase = (wc * CTW + ct) * 32;
        if (cbase + 32 > C) {                            // wave-uniform: only the tile that straddles C
#pragma unroll
            for (int st = 0; st < STW; ++st)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[ct][st][i] = cbase + acc_row(i, h) < C ? acc[ct][st][i] : MASKED;
        }
    }
    const int wave_c0 = wc * CTW * 32;                   // first class of this wave
    float bl = 0.f, bc = 0.f, bg = 0.f;                  // block sums: loss, correct, dscale
#pragma unroll
    for (int st = 0; st < STW; ++st) {
        const int smp = ws * 32 * STW + st * 32 + l31;
        const int r = row0 + smp;
        const bool valid = r < sg.rows;
        const int lab = valid ? labs[st] : -1;
        // ---- pass 1: max over this lane's logits (independent chain per class tile) ----
        float mkc[CTW];
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            mkc[ct] = acc[ct][st][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mkc[ct] = __builtin_fmaxf(mkc[ct], acc[ct][st][i]);
        }
        float mk = mkc[0];
#pragma unroll
        for (int ct = 1; ct < CTW; ++ct) mk = __builtin_fmaxf(mk, mkc[ct]);
        mk = __builtin_fmaxf(mk, __shfl_xor(mk, 32));    // max over the wave's CTW*32 classes of this sample
        // ---- label logit: register number = (tile, i) with i&3 = row&3, i>>2 = row>>3, h = (row>>2)&1 ----
        const int rel = lab - wave_c0;                   // label relative to the wave's classes
        const bool mine = rel >= 0 && rel < CTW * 32 && ((rel >> 2) & 1) == h;
        float rawy;
        {
            const int reg = (rel & 3) | ((rel >> 3) & 3) << 2 | (rel >> 5) << 4;
            float t[NREG];
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) t[ct * 16 + i] = acc[ct][st][i];
#pragma unroll
            for (int bit = 0, n = NREG; n > 1; ++bit, n >>= 1) {
                const bool up = (reg >> bit) & 1;
#pragma unroll
                for (int j = 0; j < n / 2; ++j) t[j] = up ? t[2 * j + 1] : t[2 * j];
            }
            rawy = mine ? t[0] : 0.f;
        }
        rawy += __shfl_xor(rawy, 32);                    // both lanes of the sample: label logit or 0
        STAMP(6);
        if (UMLH_ABL(a.dbg == 4)) { bl += mk + rawy; continue; }   // ablation: max + label passes only
        // ---- pass 2: first arg-max, e = exp(z - WAVE-LOCAL max), sums.  The waves' partial results are
        // merged afterwards with the online-softmax rule (one LDS exchange instead of max-then-sum) ----
        // wave-local max scaled logit, in log2 units.  A wave whose classes are ALL padding has
        // max = MASKED: keep its exponent base at 0 so every e underflows to 0 (with the base at
        // MASKED*asl2 the fma's rounding residual, ~1e24, would overflow exp2 instead).
        const float mwl = wave_c0 >= C ? 0.f : mk * asl2;
        int first = NREG;                                // lowest register number holding the max
        f32x2v se2 = {0.f, 0.f}, serw2 = {0.f, 0.f};
        const f32x2v sl2v = {asl2, asl2}, nmwl = {-mwl, -mwl};
#pragma unroll
        for (int ct = CTW - 1; ct >= 0; --ct) {
#pragma unroll
            for (int i = 14; i >= 0; i -= 2) {
                const f32x2v raw = {acc[ct][st][i], acc[ct][st][i + 1]};
                first = raw[1] == mk ? ct * 16 + i + 1 : first;
                first = raw[0] == mk ? ct * 16 + i : first;
                const f32x2v t = __builtin_elementwise_fma(raw, sl2v, nmwl);
                const f32x2v e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                se2 += e;
                serw2 = __builtin_elementwise_fma(e, raw, serw2);
                acc[ct][st][i] = e[0];
                acc[ct][st][i + 1] = e[1];
            }
        }
        // class of register number n at this lane: tile n>>4, row (n&3) + 8*((n>>2)&3) + 4h
        int mi = first < NREG ? wave_c0 + (first >> 4) * 32 + (first & 3) + 8 * ((first >> 2) & 3) + 4 * h : 0x7fffffff;
        mi = min(mi, __shfl_xor(mi, 32));
        float se = se2[0] + se2[1], serw = serw2[0] + serw2[1];
        se += __shfl_xor(se, 32);
        serw += __shfl_xor(serw, 32);
        const float rawy_wave = rawy;
        STAMP(7);
        float fown = 1.f;                                // exp(own wave max - global max)
        bool hit = mi == lab;                            // the wave-local first arg-max is the label
        if (WC > 1) {
            // field-major exchange buffer red[ws][field][wave][32]: lanes read consecutive floats (no bank
            // conflicts; the h = 1 half reads the same addresses as h = 0).  Fields: max, sum of exp
            // (sign bit = "local arg-max is the label"), sum of exp*raw, label logit.
            float* rb = red + ws * (4 * WC * 32) + l31;
            if (st > 0) __syncthreads();                 // previous tile's records fully consumed
            if (h == 0) {
                rb[(0 * WC + wc) * 32] = mk;
                rb[(1 * WC + wc) * 32] = hit ? -se : se;
                rb[(2 * WC + wc) * 32] = serw;
                rb[(3 * WC + wc) * 32] = rawy;
            }
            __syncthreads();
            const float mown = mk;
            float om[WC], os[WC];
#pragma unroll
            for (int w = 0; w < WC; ++w) { om[w] = rb[(0 * WC + w) * 32]; os[w] = rb[(1 * WC + w) * 32]; }
            // global max; on ties the lowest wave (= lowest classes) wins, and its flag is the verdict
            mk = om[0];
            float win = os[0];
#pragma unroll
            for (int w = 1; w < WC; ++w) { win = om[w] > mk ? os[w] : win; mk = __builtin_fmaxf(mk, om[w]); }
            hit = win < 0.f;
            se = 0.f;
            float fw[WC];
#pragma unroll
            for (int w = 0; w < WC; ++w) {
                // a wave whose classes are all padding has max MASKED and zero sums: its factor underflows to 0
                fw[w] = __builtin_amdgcn_exp2f((om[w] - mk) * asl2);
                se = __builtin_fmaf(__builtin_fabsf(os[w]), fw[w], se);
            }
            serw = 0.f;
            if (learn) {
#pragma unroll
                for (int w = 0; w < WC; ++w) serw = __builtin_fmaf(rb[(2 * WC + w) * 32], fw[w], serw);
            }
            // the label logit sits in exactly one wave's record
            const int wown = min(max(lab, 0) / (CTW * 32), WC - 1);
            rawy = rb[(3 * WC + wown) * 32];
            fown = __builtin_amdgcn_exp2f((mown - mk) * asl2);
        }
        const float mx = mk * ascale;                    // max scaled logit
        STAMP(3);
        const float zy = rawy * ascale;                  // (sgn*raw_y) * |scale| = scale * raw_y
        // ---- pass 3: dZ^T (bf16) ----
        if (a.dzt != nullptr && !UMLH_ABL(a.dbg == 3 || a.dbg == 4)) {
            const float coef = valid ? sg.w_over_rows * scale : 0.f;
            const float ic = coef * fown * __builtin_amdgcn_rcpf(se);   // dZ = e_local * exp(m_wave - m) / S * coef - onehot*coef
            if (WC == 1 && st == 0) __syncthreads();       // staging aliases the X tile: every wave must have left the main loop
            unsigned* dzs = dzstage + wave * (CTW * 32 * ZRS);                 // [CTW*32 class rows][ZRS dwords]
            u16* dzs16 = reinterpret_cast<u16*>(dzs);
            const f32x2v icv = {ic, ic};
            // v_cvt_pk gives (row lr, row lr+1) of ONE column; the staged tile wants two COLUMNS of one
            // row per dword.  Lanes l, l^1 swap their packed pairs (DPP quad_perm [1,0,3,2]) and a byte
            // permute picks: even lane -> row lr [own.lo, nbr.lo], odd lane -> row lr+1 [nbr.hi, own.hi].
            const bool odd = lane & 1;
            const unsigned psel = odd ? 0x03020706u : 0x05040100u;          // bytes 0-3 = own (S1), 4-7 = neighbour (S0)
            unsigned* dzl = dzs + (4 * h + (odd ? 1 : 0)) * ZRS + (l31 >> 1);
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2v m = f32x2v{acc[ct][st][i], acc[ct][st][i + 1]} * icv;
                    const unsigned own = __builtin_bit_cast(unsigned, __builtin_convertvector(m, bf16x2v));
                    const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, true);
                    dzl[(ct * 32 + (i & 3) + 8 * (i >> 2)) * ZRS] = __builtin_amdgcn_perm(nbr, own, psel);
                }
            }
            // one-hot term: the label element of this sample lives in exactly one wave; its h = 0 lane
            // recomputes e from the label logit (same fma + exp2 as pass 2) and overwrites the staged value
            if (h == 0 && rel >= 0 && rel < CTW * 32) {
                const float ey = __builtin_amdgcn_exp2f(__builtin_fmaf(rawy_wave, asl2, -mwl));
                const f32x2v v = {__builtin_fmaf(ey, ic, -coef), 0.f};
                const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
                dzs16[rel * (2 * ZRS) + l31] = (u16)(pk & 0xffffu);
            }
            // wave-local transpose done (same wave wrote and reads: program order + lgkmcnt suffice).
            // Each lane now stores 16 B = 8 columns of one class row: 16 rows x 64 B per instruction
            // instead of 4-B scattered stores (the store tail was issue-bound, not bandwidth-bound).
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int colbase = sg.col0 + row0 + ws * 32 * STW + st * 32;      // first column of this 32-sample tile
            u16* gbase = a.dzt + ((size_t)(colbase >> 6) * a.crows) * 64 + (colbase & 63);
#pragma unroll
            for (int it = 0; it < CTW * 2; ++it) {
                const int lr = it * 16 + (lane >> 2);                           // row within the wave's class range
                const int cls = wave_c0 + lr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(dzs + lr * ZRS + (lane & 3) * 4);
                if (cls < C) {
                    if (!a.plain) store_wt_b128(gbase + (size_t)cls * 64 + (lane & 3) * 8, v);
                    else *reinterpret_cast<u32x4*>(gbase + (size_t)cls * 64 + (lane & 3) * 8) = v;
                }
            }
            if (STW > 1) __builtin_amdgcn_wave_barrier();                      // next tile reuses the staging slice
        }
        STAMP(4);
        if (wc == 0 && h == 0 && valid) {
            const float ce = __logf(se) + mx - zy;
            bl += ce;
            bc += hit ? 1.f : 0.f;
            bg += sgn * (serw / se - rawy);
            if (a.row_stats != nullptr) {
                float* rs = a.row_stats + 2 * ((size_t)(sidx ? a.seg[0].rows : 0) + r);
                rs[0] = ce;
                rs[1] = hit ? 1.f : 0.f;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bl += __shfl_xor(bl, off);
        bc += __shfl_xor(bc, off);
        bg += __shfl_xor(bg, off);
    }
    if (wc == 0 && lane == 0) { red2[ws * 4 + 0] = bl; red2[ws * 4 + 1] = bc; red2[ws * 4 + 2] = bg; }
    __syncthreads();
    if (tid == 0) {
        float l = 0.f, c = 0.f, g = 0.f;
#pragma unroll
        for (int w = 0; w < WS; ++w) { l += red2[w * 4 + 0]; c += red2[w * 4 + 1]; g += red2[w * 4 + 2]; }
        float* o = a.partials + (size_t)bid * 4;
        store_out_f32(o + 0, l, a.plain); store_out_f32(o + 1, c, a.plain); store_out_f32(o + 2, g, a.plain); store_out_f32(o + 3, 0.f, a.plain);
    }
    STAMP(5);
    return true;
#undef STAMP
}

template <int CTW, int WC, int STW>
__global__ __launch_bounds__(512) void fwd_ce_bf16(FwdArgsB a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    (void)fwd_ce_bf16_body<CTW, WC, STW>(a, (int)blockIdx.x, smem_dyn);
}

// --------------------------------------------------------------------------- //
// fused forward + cross entropy, bf16 operands, 2-D decomposition (class groups x row tiles)
// --------------------------------------------------------------------------- //
// fwd_ce_bf16 gives every workgroup 32 sample rows and ALL classes, so each CU streams the whole W shadow
// (1 MB at cfg2) from L2 at the ~46 B/clk one CU sustains: 25k of its 35k cycles.  Here a workgroup owns 128
// sample rows (resident in LDS for the whole K) and a GROUP of 256 classes (one 32-class tile per wave, streamed
// through a 16-deep register ring): 128 KB of X + 256 KB of W per CU = 0.37x the stream for the same MFMA work.
// The NQ workgroups of a row tile (adjacent block ids = consecutive dispatch slots) exchange their per-row
// (max, sum of exp, sum of exp*raw, label logit) through 8-byte epoch-tagged granules in global memory -- the
// online-softmax merge of fwd_ce_bf16's wave records, one level up -- and then write their own slice of dZ^T.
// Spins are bounded; a timed-out exchange poisons the step's loss with NaN instead of hanging.
// MEASURED (cfg2, cycle stamps scripts/fwd_stamps.py): correct, but not faster than fwd_ce_bf16 -- 24.7 vs 21.8 us by the
// raw event interval.  The main loop shrinks as designed (25k -> 12k cycles) yet the workgroup lives 37k cycles against 34.6k:
// ~9k before the first MFMA (row-id round trip, then the gather from HBM), and an epilogue of 16k (6.2k records + 6k merge
// through device-coherent memory + 3.7k dZ) that the 1-D kernel half hides behind the stream of its slower wave per SIMD
// (its two waves end the stream 8k cycles apart; here four barriers keep all eight in step).  Neither removing the W refills nor
// the B reads (analysis build, dbg 21 / 22) moves the loop by more than 2k cycles.  Opt-in only: UMLH_BF16_FWD2D=1.
__device__ __forceinline__ void fq_store_granule(unsigned long long* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int FQ_TS = 128;                       // sample rows per workgroup
constexpr int FQ_XRS = 520;                      // LDS row stride (shorts): 1040 B = odd multiple of 16 B
constexpr int FQ_XT_BYTES = FQ_TS * FQ_XRS * 2;
constexpr int FQ_SMEM = FQ_XT_BYTES + sizeof(float) * (4 * 4 * 8 * 32 + 4 * 128 + 16);

template <int NQ, int NKS>
__global__ __launch_bounds__(512) void fwd_ce_bf16_q(FwdArgsB a) {
    constexpr int STW = 4, XRS = FQ_XRS, PD = 16, ZRS = 20, NG = NKS / PD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u16* Xt = reinterpret_cast<u16*>(smem_raw);                          // [128][XRS]: the tile's rows, whole K
    unsigned* dzstage = reinterpret_cast<unsigned*>(smem_raw);           // aliases Xt after the main loop
    float* rec = reinterpret_cast<float*>(smem_raw + FQ_XT_BYTES);       // [st][field][wave][32]
    float* fin = rec + 4 * 4 * 8 * 32;                                   // [field][128]: max, coef / S, coef
    float* red2 = fin + 4 * 128;                                         // [2][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    // XCD-aware decode: workgroups b and b + 8 share an XCD (round-robin dispatch).  The NQ class groups of a row tile take
    // CONSECUTIVE slots of ONE XCD: the tile's gathered rows come from HBM once (the partners hit them in their L2), the
    // exchange stays inside that L2's neighbourhood, and partners are dispatched back to back (the oldest unfinished row tile
    // of an XCD is always fully resident, so the bounded spins below cannot wait on an undispatched partner).
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int q = slot % NQ, rtile = (slot / NQ) * 8 + xcd;
    if (rtile >= a.ntiles) return;                                      // grid is rounded up to 8 tiles (whole partner sets leave)
    const int sidx = rtile >= a.seg[1].blk0 ? 1 : 0;
    const SegDescB& sg = a.seg[sidx];
#define QSTAMP(i) do { if (a.stamps && lane == 0) a.stamps[((size_t)(rtile * NQ + q) * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    QSTAMP(0);
    const int row0 = (rtile - sg.blk0) * FQ_TS;
    const int C = a.C;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wave_c0 = (q * 8 + wave_u) * 32;                          // first class of this wave's tile (a tile past C
                                                                         // multiplies the shadow's zero padding and is masked)
    // X pieces: K block kb (128 wide) = 2048 pieces of 16 B; piece tid + 512 j -> row (tid >> 4) + 32 j, column tid & 15
    const u16* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int r = min(row0 + (tid >> 4) + 32 * j, sg.rows - 1);
        int64_t rid = sg.feat_index ? sg.feat_index[r] : (int64_t)r;
        xsrc[j] = sg.feats + (size_t)rid * sg.ld + 8 * (tid & 15);
    }
    int labs[STW];
#pragma unroll
    for (int st = 0; st < STW; ++st) {
        int r = min(row0 + st * 32 + l31, sg.rows - 1);
        labs[st] = (int)sg.labels[sg.label_index ? sg.label_index[r] : (int64_t)r];
    }
    f32x16 acc[STW];
#pragma unroll
    for (int st = 0; st < STW; ++st)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[st][i] = 0.f;

    // fragment (k-step ks, class tile t) = 1 KiB at W + ((ks * wtiles + t) * 64 + lane) * 8
    const u16* wlane = a.W + ((size_t)(q * 8 + wave_u) * 64 + lane) * 8;
    const size_t wstep = (size_t)a.wtiles * 512;
    bf16x8 ring[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) ring[d] = *reinterpret_cast<const bf16x8*>(wlane + d * wstep);
    u32x4 xr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const u32x4*>(xsrc[j]);
    QSTAMP(1);
    const u16* xrow = Xt + l31 * XRS + h * 8;
    // Straight-line main loop (NKS is a template parameter: a runtime group loop made hipcc shuffle the ring between
    // its peeled variants and spill).  Per 128-wide K block: publish the block (its rows were loaded one block ago), start
    // the loads of the next one, ONE barrier (the tile is never overwritten), then 8 k-steps of {4 B reads, 4 MFMAs, refill}.
    // B fragments are read one k-step ahead of their MFMAs (two register sets)
    bf16x8 b[2][STW];
#pragma unroll
    for (int kb = 0; kb < 2 * NG; ++kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<u32x4*>(Xt + ((tid >> 4) + 32 * j) * XRS + kb * 128 + 8 * (tid & 15)) = xr[j];
        if (kb + 1 < 2 * NG) {                          // (requesting every block up front instead measured slower: 21.6k vs 20.5k cycles)
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const u32x4*>(xsrc[j] + (kb + 1) * 128);
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < STW; ++st) b[0][st] = *reinterpret_cast<const bf16x8*>(xrow + st * 32 * XRS + kb * 128);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int ks = kb * 8 + d, slot = ks % PD, cur = d & 1;
            if (d + 1 < 8) {
#pragma unroll
                for (int st = 0; st < STW; ++st) b[cur ^ 1][st] = UMLH_ABL(a.dbg == 22) ? b[cur][st] : *reinterpret_cast<const bf16x8*>(xrow + st * 32 * XRS + (ks + 1) * 16);
            }
#pragma unroll
            for (int st = 0; st < STW; ++st)
                acc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[slot], b[cur][st], acc[st], 0, 0, 0);
            if (ks + PD < NKS && !UMLH_ABL(a.dbg == 21)) ring[slot] = *reinterpret_cast<const bf16x8*>(wlane + (ks + PD) * wstep);
            if (d + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, STW, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, STW, 0);
            if (ks + PD < NKS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    }
    QSTAMP(2);

    // ---------------- epilogue, phase A: wave-local records of the 4 sample tiles ----------------
    const float scale = *sg.scale_ptr;
    const float sgn = scale < 0.f ? -1.f : 1.f;
    const float LOG2E = 1.4426950408889634f;
    const float ascale = __builtin_fabsf(scale);
    const float asl2 = __builtin_fmaxf(ascale * LOG2E, 1e-20f);
    const float MASKED = -1e30f;
    const bool learn = a.learn != 0;
    if (scale < 0.f) {
#pragma unroll
        for (int st = 0; st < STW; ++st)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[st][i] = -acc[st][i];
    }
    if (wave_c0 + 32 > C) {                              // wave-uniform: the tile that straddles C, and all-padding tiles
#pragma unroll
        for (int st = 0; st < STW; ++st)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[st][i] = wave_c0 + acc_row(i, h) < C ? acc[st][i] : MASKED;
    }
    float mown[STW], rawy_w[STW];
#pragma unroll
    for (int st = 0; st < STW; ++st) {
        const bool valid = row0 + st * 32 + l31 < sg.rows;
        const int lab = valid ? labs[st] : -1;
        float mk = acc[st][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mk = __builtin_fmaxf(mk, acc[st][i]);
        mk = __builtin_fmaxf(mk, __shfl_xor(mk, 32));
        const int rel = lab - wave_c0;
        const bool mine = rel >= 0 && rel < 32 && ((rel >> 2) & 1) == h;
        float rawy;
        {
            const int reg = (rel & 3) | ((rel >> 3) & 3) << 2;
            float t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = acc[st][i];
#pragma unroll
            for (int bit = 0, n = 16; n > 1; ++bit, n >>= 1) {
                const bool up = (reg >> bit) & 1;
#pragma unroll
                for (int j = 0; j < n / 2; ++j) t[j] = up ? t[2 * j + 1] : t[2 * j];
            }
            rawy = mine ? t[0] : 0.f;
        }
        rawy += __shfl_xor(rawy, 32);
        const float mwl = wave_c0 >= C ? 0.f : mk * asl2;
        int first = 16;
        f32x2v se2 = {0.f, 0.f}, serw2 = {0.f, 0.f};
        const f32x2v sl2v = {asl2, asl2}, nmwl = {-mwl, -mwl};
#pragma unroll
        for (int i = 14; i >= 0; i -= 2) {
            const f32x2v raw = {acc[st][i], acc[st][i + 1]};
            first = raw[1] == mk ? i + 1 : first;
            first = raw[0] == mk ? i : first;
            const f32x2v t = __builtin_elementwise_fma(raw, sl2v, nmwl);
            const f32x2v e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            se2 += e;
            serw2 = __builtin_elementwise_fma(e, raw, serw2);
            acc[st][i] = e[0];
            acc[st][i + 1] = e[1];
        }
        int mi = first < 16 ? wave_c0 + (first & 3) + 8 * ((first >> 2) & 3) + 4 * h : 0x7fffffff;
        mi = min(mi, __shfl_xor(mi, 32));
        float se = se2[0] + se2[1], serw = serw2[0] + serw2[1];
        se += __shfl_xor(se, 32);
        serw += __shfl_xor(serw, 32);
        const bool hit = mi == lab;
        if (h == 0) {
            float* rb = rec + (st * 4 * 8 + wave) * 32 + l31;
            rb[0 * 8 * 32] = mk;
            rb[1 * 8 * 32] = hit ? -se : se;
            rb[2 * 8 * 32] = serw;
            rb[3 * 8 * 32] = rawy;
        }
        mown[st] = mk;
        rawy_w[st] = rawy;
    }
    __syncthreads();
    QSTAMP(3);
    // ---------------- phase B: thread t < 128 merges row t over the 8 waves, then over the NQ class groups ----------------
    float bl = 0.f, bc = 0.f, bg = 0.f;
    if (tid < 128) {
        const float* rb = rec + ((tid >> 5) * 4 * 8) * 32 + (tid & 31);
        float mk = rb[0], win = rb[8 * 32];
        float om[8], os[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) { om[w] = rb[w * 32]; os[w] = rb[(8 + w) * 32]; }
#pragma unroll
        for (int w = 1; w < 8; ++w) { win = om[w] > mk ? os[w] : win; mk = __builtin_fmaxf(mk, om[w]); }
        float se = 0.f, serw = 0.f, rawy = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float fw = __builtin_amdgcn_exp2f((om[w] - mk) * asl2);
            se = __builtin_fmaf(__builtin_fabsf(os[w]), fw, se);
            serw = __builtin_fmaf(rb[(16 + w) * 32], fw, serw);
            rawy += rb[(24 + w) * 32];
        }
        bool hit = win < 0.f;
        bool bad = false;
        if (NQ > 1) {
            unsigned long long* mine_rec = a.xch + ((size_t)(rtile * NQ + q) * 4) * 128 + tid;
            fq_store_granule(mine_rec, a.epoch, mk);
            fq_store_granule(mine_rec + 128, a.epoch, hit ? -se : se);
            fq_store_granule(mine_rec + 256, a.epoch, serw);
            fq_store_granule(mine_rec + 384, a.epoch, rawy);
            float gm[NQ], gs[NQ], gw[NQ], gy[NQ];
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
            for (int p = 0; p < NQ; ++p) {
                if (p == q) { gm[p] = mk; gs[p] = hit ? -se : se; gw[p] = serw; gy[p] = rawy; continue; }
                const unsigned long long* pr = a.xch + ((size_t)(rtile * NQ + p) * 4) * 128 + tid;
                unsigned long long g4[4];
                for (unsigned spin = 0; !bad; ++spin) {
                    bool ok = true;
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        g4[f] = __hip_atomic_load(pr + f * 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned)(g4[f] >> 32) == a.epoch;
                    }
                    if (__all(ok)) break;
                    if ((spin & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) bad = true;   // 50 ms at 100 MHz
                    __builtin_amdgcn_s_sleep(1);
                }
                gm[p] = __uint_as_float((unsigned)g4[0]); gs[p] = __uint_as_float((unsigned)g4[1]);
                gw[p] = __uint_as_float((unsigned)g4[2]); gy[p] = __uint_as_float((unsigned)g4[3]);
            }
            mk = gm[0]; win = gs[0];
#pragma unroll
            for (int p = 1; p < NQ; ++p) { win = gm[p] > mk ? gs[p] : win; mk = __builtin_fmaxf(mk, gm[p]); }
            hit = win < 0.f;
            se = 0.f; serw = 0.f; rawy = 0.f;
#pragma unroll
            for (int p = 0; p < NQ; ++p) {
                const float fw = __builtin_amdgcn_exp2f((gm[p] - mk) * asl2);
                se = __builtin_fmaf(__builtin_fabsf(gs[p]), fw, se);
                serw = __builtin_fmaf(gw[p], fw, serw);
                rawy += gy[p];
            }
        }
        const int r = row0 + tid;
        const bool valid = r < sg.rows;
        const float coef = valid ? sg.w_over_rows * scale : 0.f;
        fin[tid] = mk;
        fin[128 + tid] = coef * __builtin_amdgcn_rcpf(se);
        fin[256 + tid] = coef;
        if (q == 0 && valid) {
            const float ce = __logf(se) + mk * ascale - rawy * ascale;
            bl = bad ? __builtin_nanf("") : ce;
            bc = hit ? 1.f : 0.f;
            bg = learn ? sgn * (serw / se - rawy) : 0.f;
            if (a.row_stats != nullptr) {
                float* rs = a.row_stats + 2 * ((size_t)(sidx ? a.seg[0].rows : 0) + r);
                rs[0] = ce;
                rs[1] = hit ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            bl += __shfl_xor(bl, off);
            bc += __shfl_xor(bc, off);
            bg += __shfl_xor(bg, off);
        }
        if (lane == 0) { red2[wave * 4 + 0] = bl; red2[wave * 4 + 1] = bc; red2[wave * 4 + 2] = bg; }
    }
    __syncthreads();
    QSTAMP(4);
    if (q == 0 && tid == 0) {
        float* o = a.partials + (size_t)rtile * 4;
        o[0] = red2[0] + red2[4]; o[1] = red2[1] + red2[5]; o[2] = red2[2] + red2[6]; o[3] = 0.f;
    }
    // ---------------- phase C: this wave's 32 class rows of dZ^T, tile by tile ----------------
    if (a.dzt != nullptr) {
        unsigned* dzs = dzstage + wave * (32 * ZRS);                      // [32 class rows][ZRS dwords]
        u16* dzs16 = reinterpret_cast<u16*>(dzs);
        const bool odd = lane & 1;
        const unsigned psel = odd ? 0x03020706u : 0x05040100u;
        unsigned* dzl = dzs + (4 * h + (odd ? 1 : 0)) * ZRS + (l31 >> 1);
#pragma unroll
        for (int st = 0; st < STW; ++st) {
            const int smp = st * 32 + l31;
            const float mg = fin[smp], icb = fin[128 + smp], coef = fin[256 + smp];
            const float fown = __builtin_amdgcn_exp2f((mown[st] - mg) * asl2);
            const float ic = icb * fown;
            const f32x2v icv = {ic, ic};
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2v m = f32x2v{acc[st][i], acc[st][i + 1]} * icv;
                const unsigned own = __builtin_bit_cast(unsigned, __builtin_convertvector(m, bf16x2v));
                const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, true);
                dzl[((i & 3) + 8 * (i >> 2)) * ZRS] = __builtin_amdgcn_perm(nbr, own, psel);
            }
            const int lab = row0 + smp < sg.rows ? labs[st] : -1;
            const int rel = lab - wave_c0;
            if (h == 0 && rel >= 0 && rel < 32) {
                const float mwl = mown[st] * asl2;                        // a tile that holds a label is never all padding
                const float ey = __builtin_amdgcn_exp2f(__builtin_fmaf(rawy_w[st], asl2, -mwl));
                const f32x2v v = {__builtin_fmaf(ey, ic, -coef), 0.f};
                const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v));
                dzs16[rel * (2 * ZRS) + l31] = (u16)(pk & 0xffffu);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int colbase = sg.col0 + row0 + st * 32;
            u16* gbase = a.dzt + ((size_t)(colbase >> 6) * a.crows) * 64 + (colbase & 63);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int lr = it * 16 + (lane >> 2);
                const int cls = wave_c0 + lr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(dzs + lr * ZRS + (lane & 3) * 4);
                if (cls < C) {
                    if (!a.plain) store_wt_b128(gbase + (size_t)cls * 64 + (lane & 3) * 8, v);
                    else *reinterpret_cast<u32x4*>(gbase + (size_t)cls * 64 + (lane & 3) * 8) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    QSTAMP(5);
#undef QSTAMP
}

// --------------------------------------------------------------------------- //
// dW[m][n] = sum_r dZ^T[m][r] * F[r][n]   (bf16 operands, fp32 split-K slabs)
// 128x128 tile, 4 waves (2x2) of 64x64.  A rows are k-contiguous (ds_read_b128);
// the feature rows F are k-major in memory, so the B fragment (8 consecutive k for one
// column) comes from the hardware transposing read ds_read_b64_tr_b16.
// --------------------------------------------------------------------------- //
constexpr int DBM = 128, DBN = 128;
constexpr int DKT = 64;     // reduction rows (dZ^T columns) per chunk = one column chunk of dZ^T
constexpr int RSA = 72;     // shorts per LDS row of the dZ^T tile: 128 B data + 16 B pad (9 x 16 B: conflict-free b128)
constexpr int RSF = 160;    // shorts per LDS row of the F tile: 256 B data + 64 B pad (4 k-rows -> 4 bank quarters)

constexpr int DNS = 4;      // register stages: 3 chunks (96 KiB per CU) in flight while one is consumed
constexpr int DIDS = 4096;  // max reduction rows per workgroup (row ids staged in LDS)
constexpr int DMASK = (int)0x80000000;

// 512 threads = 8 waves (2 along M x 4 along N, 64x32 outputs each): two waves per SIMD, so one
// wave's address arithmetic / LDS traffic overlaps the other's MFMAs (with 4 waves per CU every
// phase of a chunk was serialised: 2.9k VALU instructions per wave and 22 us measured).
// GATED (the single-launch forward + dW below): the dZ^T loads wait until the forward blocks that write this split's columns
// have published their granules; everything that does not depend on dZ^T (row ids, the first feature chunks) is issued first.
// device-coherent (sc1) 16-byte load through a raw buffer descriptor: reads what another XCD's workgroup wrote through to memory
// earlier in the SAME launch without invalidating this XCD's whole L2 (an acquire fence per gated workgroup did that 32 times
// per XCD and cost +12 us per step); out-of-range offsets return zeros (the branch-free masking of the plain path's zero page)
__device__ __forceinline__ u32x4 load_coherent_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 1 << 4 /* sc1 */);
}

struct DwGate { StepCtl ctl; int fwd_base, self_base;   // task ids of forward block 0 / dW block 0 of this launch
                int ts, total_cols;
                unsigned long long* timeline; };          // UMLH_DBG_STEP=1: [task][4] s_memrealtime stamps (100 MHz): start, gate open, end
constexpr int DW_LDS_BYTES = 2 * DBM * RSA * 2 + 2 * DKT * RSF * 2 + DIDS * 4;   // two A tiles, two F tiles, the split's row ids

// Returns TW_OK when the tile is written; GATED only: TW_ABORT (a wait gave up: nothing stored) or the id of a forward task
// this workgroup has TAKEN while waiting (nobody held it): the caller runs it and calls again (nothing was stored yet).
template <int AM, int OM, bool GATED>
__device__ __forceinline__ int dw_bf16_body(const DwArgsB& g, const int vbid, const DwGate& gate, unsigned char* lds, int* sh_rc) {
    // two LDS buffers: chunk c+1 is written while chunk c is consumed -> ONE barrier per chunk
    u16 (*At)[DBM * RSA] = reinterpret_cast<u16 (*)[DBM * RSA]>(lds);
    u16 (*Ft)[DKT * RSF] = reinterpret_cast<u16 (*)[DKT * RSF]>(lds + 2 * DBM * RSA * 2);
    int* ids = reinterpret_cast<int*>(lds + 2 * DBM * RSA * 2 + 2 * DKT * RSF * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int h = lane >> 5, l31 = lane & 31;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    // XCD-aware decode of the 1-D grid: workgroups b and b+8 share an XCD (round-robin dispatch),
    // so split z = b % nsplit keeps every tile of one K-split -- which all re-read the same dZ^T
    // columns and feature rows -- on ONE XCD's L2 (speed only; any placement is correct).
    const int nx = (g.N + DBN - 1) / DBN;
    const int bid = vbid;
    const int z = bid % g.nsplit, t = bid / g.nsplit;
    const int m0 = (t / nx) * DBM, n0 = (t % nx) * DBN;
    // modality-aligned split-K: slabs [0, nsplit1) cover the image rows [0, k_switch), the rest the text
    // rows [k_switch, K) -- their sums stay separable (per-modality gradient diagnostics)
    const int kb = z < g.nsplit1 ? z * g.k_chunk : g.k_switch + (z - g.nsplit1) * g.k_chunk;    // multiple of DKT
    const int ke = min(z < g.nsplit1 ? g.k_switch : g.K, kb + g.k_chunk);
    const int nchunks = g.k_chunk / DKT;                // multiple of DNS

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#define DSTAMP(i) do { if (g.stamps && lane == 0) g.stamps[((size_t)vbid * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    DSTAMP(0);
    if (GATED && gate.timeline && tid == 0) gate.timeline[(size_t)(gate.self_base + vbid) * 4 + 0] = __builtin_amdgcn_s_memrealtime();

    struct Stage { u32x4 a[2]; u32x4 f[2]; };
    Stage st[DNS];
    // per-thread invariants of the two A pieces and two F pieces it stages per chunk
    //   A piece p (0..1023): tile row p>>3, 16-B column p&7;   F piece p: chunk row p>>4, 16-B column p&15
    const int mclamp = g.M - 1 - m0;                    // rows >= M: garbage that is never stored
    const u16* a_thr[2];
    int a_col[2], f_row[2], f_col[2];
    bool f_colok[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int p = tid + 512 * q;
        a_col[q] = 8 * (p & 7);
        if (AM == 0) a_thr[q] = g.A + ((size_t)m0 + min(p >> 3, mclamp)) * 64 + a_col[q];
        else a_thr[q] = g.A + (size_t)g.a_rows[m0 + min(p >> 3, mclamp)] * g.lda + a_col[q];
        f_row[q] = p >> 4;
        const int col = min(n0 + 8 * (p & 15), g.N - 8);
        f_col[q] = (col >> 6) * g.bcs + (col & 63);     // row-major rows: bcs = 64 -> col
        f_colok[q] = n0 + 8 * (p & 15) < g.N;
    }
    // GATED: dZ^T comes through a buffer descriptor with device-coherent loads (AM == 0 only)
    __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(g.A), 0, GATED ? (int)(((size_t)(g.K + 63) / 64) * g.lda * 128) : 0, 0x00020000);
    auto a_load = [&](int q, int k0, size_t achunk) -> u32x4 {
        const bool on = k0 + a_col[q] < ke && !UMLH_ABL(g.dbg & 1);
        if (GATED) {
            const unsigned off = (unsigned)((a_thr[q] - g.A + achunk) * 2);
            return load_coherent_b128(a_rsrc, on ? off : 0xfffffff0u);
        }
        const u16* ap = on ? a_thr[q] + achunk : g.zeros;
        return *reinterpret_cast<const u32x4*>(ap);
    };
    // Branch-free loads: masked pieces read a zero page (a select on the loaded value would make the
    // compiler wait for the load right here and serialise the pipeline).
    auto gloadA = [&](Stage& sg, int c) {
        const int k0 = kb + c * DKT;
        const size_t achunk = AM == 0 ? (size_t)(k0 >> 6) * g.lda * 64 : (size_t)k0;
#pragma unroll
        for (int q = 0; q < 2; ++q) sg.a[q] = a_load(q, k0, achunk);
    };
    auto gloadF = [&](Stage& sg, int c) {
        const int k0 = kb + c * DKT;                    // whole chunk lies in one modality (k_switch % 64 == 0)
        const bool seg2 = k0 >= g.k_switch;
        const u16* fb = seg2 ? g.B2 : g.B;
        const int ld = seg2 ? g.ldb2 : g.ldb;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rid = ids[c * DKT + f_row[q]];
            const u16* fp = (rid != DMASK && f_colok[q] && !UMLH_ABL(g.dbg & 2)) ? fb + (size_t)rid * ld + f_col[q] : g.zeros;
            sg.f[q] = *reinterpret_cast<const u32x4*>(fp);
        }
    };
    auto gload = [&](Stage& sg, int c) { gloadA(sg, c); gloadF(sg, c); };
    auto lstore = [&](const Stage& sg, int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int p = tid + 512 * q;
            *reinterpret_cast<u32x4*>(At[buf] + (p >> 3) * RSA + 8 * (p & 7)) = sg.a[q];
            *reinterpret_cast<u32x4*>(Ft[buf] + (p >> 4) * RSF + 8 * (p & 15)) = sg.f[q];
        }
    };
    const int a_off = (wm * 64 + l31) * RSA + h * 8;
    const int f_off = (8 * h + q4) * RSF + wn * 32 + (g16 & 1) * 16 + 4 * p4;
    auto compute = [&](int buf) {
        const u16* a_frag = At[buf] + a_off;
        const u16* f_frag = Ft[buf] + f_off;
#pragma unroll
        for (int s = 0; s < DKT / 16; ++s) {
            bf16x8 av[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8*>(a_frag + i * 32 * RSA + s * 16);
            // 16-lane group g16: column block (g16&1) of the 32-wide tile, k half h = g16>>1;
            // lane 4q+p addresses row q, columns 4p..4p+3; it receives column (lane&15), rows 0..3.
            const u16* base = f_frag + s * 16 * RSF;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * RSF));
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const bf16x8 bv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv, acc[i], 0, 0, 0);
        }
    };

    // row ids of this split -> LDS once (no dependent global loads inside the pipeline); DMASK marks a
    // masked (padding / out-of-range) row.  The id-independent dZ^T loads of the first DNS chunks are
    // issued first so their latency overlaps this round trip.
    const int lastc = nchunks - 1;
    if (!GATED) {
#pragma unroll
        for (int d = 0; d < DNS; ++d) gloadA(st[d], min(d, lastc));
    }
    for (int i = tid; i < g.k_chunk; i += 512) {
        int k = kb + i;
        bool seg2 = k >= g.k_switch;
        int kl = seg2 ? k - g.k_switch : k;
        int lim = seg2 ? g.k_valid2 : g.k_valid1;
        const int64_t* ip = seg2 ? g.k_rows2 : g.k_rows;
        bool valid = k < ke && kl < lim;
        ids[i] = valid ? (int)ip[kl] : DMASK;
    }
    __syncthreads();
    DSTAMP(1);
#pragma unroll
    for (int d = 0; d < DNS; ++d) gloadF(st[d], min(d, lastc));
    if (GATED) {
        // forward block b writes columns [b * ts, (b + 1) * ts) of dZ^T (image blocks first, text columns start at a multiple
        // of ts); this split reads [kb, ke).  One sweep = all granules of the range in flight (tasks_wait, umlh_common.h).
        if (wave == 0) {
            const int b0 = kb / gate.ts, nb = (min(ke, gate.total_cols) - kb + gate.ts - 1) / gate.ts;
            const int rc = nb > 0 ? tasks_wait(gate.ctl, gate.fwd_base + b0, nb, lane, 1u) : TW_OK;
            if (lane == 0) *sh_rc = rc;
        }
        __syncthreads();
        const int rc = *sh_rc;
        __syncthreads();                                // (the word is rewritten by the next wait)
        if (rc != TW_OK) return rc;
        if (gate.timeline && tid == 0) gate.timeline[(size_t)(gate.self_base + vbid) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int d = 0; d < DNS; ++d) gloadA(st[d], min(d, lastc));
    }
    lstore(st[0], 0);                                   // chunk 0 -> buffer 0
    gload(st[0], min(DNS, lastc));
    __syncthreads();
    DSTAMP(2);
    // iteration for chunk c (buffer c&1): stage chunk c+1 into the other buffer, refill its register
    // stage with chunk c+1+DNS, run the MFMAs of chunk c, one barrier.  DNS is even, so (c+1)&1 and
    // the stage index (c+1)%DNS are compile-time constants inside the unrolled body.
    // Staging is interleaved with the MFMAs, one piece per k-step: {fragment reads, MFMA, ds_write of
    // piece s of chunk c+1, MFMA, global load of piece s of chunk c+1+DNS}.  Issued as separate phases
    // (all stores, all loads, then all MFMAs) the eight waves move in lockstep between barriers and the
    // LDS-store transfer (13 cycles per ds_write_b128 per wave) is never hidden behind matrix work.
    static_assert(DKT / 16 == 4, "one staging piece (a0, a1, f0, f1) per k-step");
    auto store_piece = [&](const Stage& sg, int buf, int pc) {
        const int q = pc & 1, pidx = tid + 512 * q;
        if (pc < 2) *reinterpret_cast<u32x4*>(At[buf] + (pidx >> 3) * RSA + 8 * (pidx & 7)) = sg.a[q];
        else        *reinterpret_cast<u32x4*>(Ft[buf] + (pidx >> 4) * RSF + 8 * (pidx & 15)) = sg.f[q];
    };
    auto load_piece = [&](Stage& sg, int c, int pc, const int (&rids)[2]) {
        const int q = pc & 1;
        const int k0 = kb + c * DKT;
        if (pc < 2) {
            const size_t achunk = AM == 0 ? (size_t)(k0 >> 6) * g.lda * 64 : (size_t)k0;
            sg.a[q] = a_load(q, k0, achunk);
        } else {
            const bool seg2 = k0 >= g.k_switch;
            const u16* fb = seg2 ? g.B2 : g.B;
            const int ld = seg2 ? g.ldb2 : g.ldb;
            const int rid = rids[q];
            const u16* fp = (rid != DMASK && f_colok[q] && !UMLH_ABL(g.dbg & 2)) ? fb + (size_t)rid * ld + f_col[q] : g.zeros;
            sg.f[q] = *reinterpret_cast<const u32x4*>(fp);
        }
    };
    for (int c = 0; c < (UMLH_ABL(g.dbg & 4) ? 0 : nchunks); c += DNS) {      // dbg bit2: skip the main loop (fixed-cost probe)
#pragma unroll
        for (int d = 0; d < DNS; ++d) {
            const int nd = (d + 1) % DNS;
            const int buf = d & 1, nbuf = (d + 1) & 1;
            const int cnext = min(c + d + 1 + DNS, lastc);            // (past the end: a harmless re-store of the last chunk)
            const u16* a_frag = At[buf] + a_off;
            const u16* f_frag = Ft[buf] + f_off;
            const int rids[2] = {ids[cnext * DKT + f_row[0]], ids[cnext * DKT + f_row[1]]};
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // row ids of the chunk to fetch
            // fragments are read one k-step ahead of their MFMAs (two register sets): only the first k-step of
            // a chunk -- right after the barrier that publishes its buffer -- waits on LDS latency
            bf16x8 av[2][2], bvv[2];
            auto load_frag = [&](int ks, int slot) {
#pragma unroll
                for (int i = 0; i < 2; ++i) av[slot][i] = *reinterpret_cast<const bf16x8*>(a_frag + i * 32 * RSA + ks * 16);
                const u16* base = f_frag + ks * 16 * RSF;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * RSF));
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bvv[slot] = __builtin_bit_cast(bf16x8, v);
            };
            load_frag(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int ks = 0; ks < DKT / 16; ++ks) {
                const int cur = ks & 1;
                if (ks + 1 < DKT / 16) load_frag(ks + 1, cur ^ 1);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cur][0], bvv[cur], acc[0], 0, 0, 0);
                store_piece(st[nd], nbuf, ks);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cur][1], bvv[cur], acc[1], 0, 0, 0);
                load_piece(st[nd], cnext, ks, rids);
                if (ks + 1 < DKT / 16) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // next k-step's fragment reads
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // ds_write of the staged piece
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // global load of the next piece
            }
            __syncthreads();
        }
        if (c < 3 * DNS) DSTAMP(3 + c / DNS);
    }
    DSTAMP(6);

    const int n = n0 + wn * 32 + l31;
    if (n < g.N) {
        float* out = g.out + (size_t)z * g.slab_stride;
        u16* o16 = static_cast<u16*>(g.out16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int m = m0 + wm * 64 + i * 32 + acc_row(e, h);
                if (m >= g.M) continue;
                if (OM == 0) store_out_f32(out + (size_t)m * g.ldo + n, acc[i][e], g.plain);
                else {
                    const u16 v = f2bf(acc[i][e]);
                    if (OM == 1) o16[(size_t)m * g.ldo + n] = v;
                    else o16[((size_t)(n >> 6) * g.ldo + m) * 64 + (n & 63)] = v;
                }
            }
    }
    DSTAMP(7);
    if (GATED && gate.timeline && tid == 0) gate.timeline[(size_t)(gate.self_base + vbid) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    return TW_OK;
#undef DSTAMP
}

// --------------------------------------------------------------------------- //
// Round 3: the same tile (128 x 128, 8 waves of 64 x 32, 64 reduction rows per chunk) with BOTH operands staged by LDS-DMA
// (global_load_lds_dwordx4: no VGPR staging, no ds_write -- the ds_write_b128 stream, 13 cycles per wave-instruction, was
// 416 of the old loop's ~1270 cycles per chunk, on top of 384 cycles of fragment reads for 512 cycles of MFMA).
//   * three LDS stages of (16 KB dZ^T tile + 16 KB feature tile); a chunk's 32 one-KiB pieces are issued two chunks ahead,
//     four per wave; ONE raw s_barrier per chunk behind a counted s_waitcnt vmcnt (the DMA of the chunk after next stays in
//     flight across it); no ordinary global load inside the loop (row ids live in LDS), so hipcc inserts no vmcnt(0)
//   * the LDS-DMA destination is lane-linear (wave-uniform base + 16 B x lane), so the bank swizzles sit on the per-lane
//     SOURCE address and on the reads (guide rule 21):  dZ^T tile [128 rows][8 x 16 B]: 16-B column c of row r is stored at
//     column c ^ ((r >> 1) & 7) -- the b128 fragment reads of 16 rows hit 16 different 16-B slots;  feature tile [64 k-rows]
//     [16 x 16 B]: chunk ch of row r at ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) -- the transposing reads (ds_read_b64_tr_b16)
//     of 4 rows x 32 columns per half-wave are conflict-free (guide T10, image (b))
//   * masked pieces (rows past the split, padding rows, columns past N) read a zero page through their per-lane source
// GATED: as dw_bf16_body -- row ids and the feature tiles of the first two chunks are issued before the wait for the forward
// tiles, the dZ^T pieces after it, with device-coherent (sc1) DMA loads.
// --------------------------------------------------------------------------- //
constexpr int DST = 2;                                  // LDS stages
constexpr int DKB = 128;                                // reduction rows per chunk: the per-chunk chain (barrier, DMA issue, first fragment
                                                        // latency) costs ~800 cycles whatever the chunk holds -- with 64 rows (512 cycles of MFMA)
                                                        // it was 60 % of the loop (scripts/dw_stamps.py: 1355 cycles per chunk)
constexpr int DA_BYTES = DBM * 2 * DKB, DF_BYTES = DKB * 256;
constexpr int DSTAGE_BYTES = DA_BYTES + DF_BYTES;       // 64 KB
constexpr int DW_DMA_LDS_BYTES = DST * DSTAGE_BYTES + DIDS * 4;

template <int AM, int OM, bool GATED>
__device__ __forceinline__ int dw_bf16_body_dma(const DwArgsB& g, const int vbid, const DwGate& gate, unsigned char* lds, int* sh_rc) {
    int* ids = reinterpret_cast<int*>(lds + DST * DSTAGE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int h = lane >> 5, l31 = lane & 31;
    const int g16 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int nx = (g.N + DBN - 1) / DBN;
    const int z = vbid % g.nsplit, t = vbid / g.nsplit;
    const int m0 = (t / nx) * DBM, n0 = (t % nx) * DBN;
    const int kb = z < g.nsplit1 ? z * g.k_chunk : g.k_switch + (z - g.nsplit1) * g.k_chunk;    // multiple of 64
    const int ke = min(z < g.nsplit1 ? g.k_switch : g.K, kb + g.k_chunk);
    const int nchunks = g.k_chunk / DKB;                // (k_chunk is a multiple of 256)

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#define DSTAMP(i) do { if (g.stamps && lane == 0) g.stamps[((size_t)vbid * 8 + wave) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
    DSTAMP(0);
    if (GATED && gate.timeline && tid == 0) gate.timeline[(size_t)(gate.self_base + vbid) * 4 + 0] = __builtin_amdgcn_s_memrealtime();

    // ---- per-lane invariants of the pieces this wave issues: pieces wave + 8q (q = 0..3) of each tile ----
    // dZ^T piece p = rows 4p .. 4p+3 of [128 rows][16 x 16 B]; lane -> row 4p + (lane >> 4), physical column lane & 15, logical
    // column ^ (row & 15)  (row & 15 is the same for all four pieces of a wave)
    const int a_r0 = 4 * wave + (lane >> 4);
    const int a_c = (lane & 15) ^ (a_r0 & 15);                             // logical 16-B column = k offset 8 * a_c
    const int mclamp = g.M - 1 - m0;                                       // rows >= M: garbage that is never stored
    const u16* a_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = a_r0 + 32 * q;
        // AM 0: column-chunk-major [K/64][lda][64]: the 128-wide chunk spans two 64-column chunks
        if (AM == 0) a_src[q] = g.A + (size_t)(a_c >> 3) * g.lda * 64 + ((size_t)m0 + min(r, mclamp)) * 64 + 8 * (a_c & 7);
        else a_src[q] = g.A + (size_t)g.a_rows[m0 + min(r, mclamp)] * g.lda + 8 * a_c;
    }
    // feature piece p = k-rows 4p .. 4p+3 of [128 k-rows][16 x 16 B]; lane -> row 4p + (lane >> 4), physical chunk lane & 15,
    // logical chunk ^ swz(row), swz(r) = ((r & 3) << 2) | ((r >> 2) & 3)
    const int f_r0 = 4 * wave + (lane >> 4);
    const int f_ch = (lane & 15) ^ (((f_r0 & 3) << 2) | ((f_r0 >> 2) & 3));     // same for the four pieces of a wave
    const int f_colraw = n0 + 8 * f_ch;
    const bool f_colok = f_colraw < g.N;
    const int f_colc = min(f_colraw, g.N - 8);
    const int f_col = (f_colc >> 6) * g.bcs + (f_colc & 63);              // row-major rows: bcs = 64 -> col
    auto stage_ptr = [&](int st) -> unsigned char* { return lds + st * DSTAGE_BYTES; };
    auto issueA = [&](int c, int st) {
        const int k0 = kb + c * DKB;
        const size_t achunk = AM == 0 ? (size_t)(k0 >> 6) * g.lda * 64 : (size_t)k0;
        const bool on = k0 + 8 * a_c < ke;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u16* sp = on ? a_src[q] + achunk : g.zeros;
            // (the dZ^T pieces of a GATED launch were written through by forward tiles of the same launch: coherent loads)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(stage_ptr(st) + (wave + 8 * q) * 1024), 16, 0, GATED ? 16 : 0);
        }
    };
    auto issueF = [&](int c, int st) {
        const int k0 = kb + c * DKB;                    // the two 64-row halves of a chunk may lie in different modalities
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool seg2 = k0 + 32 * q >= g.k_switch;  // (k_switch is a multiple of 64: rows 32q .. 32q+31 lie on one side)
            const u16* fb = seg2 ? g.B2 : g.B;
            const int ld = seg2 ? g.ldb2 : g.ldb;
            const int rid = ids[c * DKB + f_r0 + 32 * q];
            const long long off = (long long)rid * ld + f_col;             // formed unconditionally: a select, not a branch, picks the source
            const u16* sp = (rid != DMASK && f_colok) ? fb + off : g.zeros;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(stage_ptr(st) + DA_BYTES + (wave + 8 * q) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addresses (LDS byte addresses in stage 0) ----
    // A: row r = wm*64 + i*32 + l31, k-step s (0..7): logical column 2s + h, stored at column ^ (r & 15) = ^ (l31 & 15)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned a_ad[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) a_ad[s8] = lds0 + (unsigned)((wm * 64 + l31) * 256 + (((2 * s8 + h) ^ (l31 & 15)) << 4));
    // F: k-row R = 16s + 8h + q4 (+ 4 for the upper half of the fragment); element column wn*32 + (g16 & 1)*16 + 4*p4, i.e.
    // 16-B chunk ch = wn*4 + (g16 & 1)*2 + (p4 >> 1), 8-byte half p4 & 1; swz(R) = (q4 << 2) | ((2h + hi) & 3) for every s
    const int f_chr = wn * 4 + (g16 & 1) * 2 + (p4 >> 1);
    const unsigned lds_lo = lds0 + (unsigned)(DA_BYTES + (8 * h + q4) * 256 + ((f_chr ^ ((q4 << 2) | ((2 * h) & 3))) << 4) + 8 * (p4 & 1));
    const unsigned lds_hi = lds0 + (unsigned)(DA_BYTES + (8 * h + q4 + 4) * 256 + ((f_chr ^ ((q4 << 2) | ((2 * h + 1) & 3))) << 4) + 8 * (p4 & 1));

    // row ids of this split -> LDS once; DMASK marks a masked (padding / out-of-range) row
    for (int i = tid; i < g.k_chunk; i += 512) {
        int k = kb + i;
        bool seg2 = k >= g.k_switch;
        int kl = seg2 ? k - g.k_switch : k;
        int lim = seg2 ? g.k_valid2 : g.k_valid1;
        const int64_t* ip = seg2 ? g.k_rows2 : g.k_rows;
        bool valid = k < ke && kl < lim;
        ids[i] = valid ? (int)ip[kl] : DMASK;
    }
    __syncthreads();                                    // (also drains the id loads: no ordinary vector load is pending from here on)
    DSTAMP(1);
    issueF(0, 0);
    if (GATED) {
        if (wave == 0) {
            const int b0 = kb / gate.ts, nb = (min(ke, gate.total_cols) - kb + gate.ts - 1) / gate.ts;
            const int rc = nb > 0 ? tasks_wait(gate.ctl, gate.fwd_base + b0, nb, lane, 1u) : TW_OK;
            if (lane == 0) *sh_rc = rc;
        }
        __syncthreads();                                // (emits vmcnt(0): the feature pieces above have landed -- harmless, they are needed next)
        const int rc = *sh_rc;
        __syncthreads();                                // (the word is rewritten by the next wait)
        if (rc != TW_OK) return rc;
        if (gate.timeline && tid == 0) gate.timeline[(size_t)(gate.self_base + vbid) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    issueA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DSTAMP(2);
    for (int c = 0; c < nchunks; ++c) {
        const int st = c & 1;
        // the other stage was last read in iteration c-1, which every wave has left (the barrier below)
        if (c + 1 < nchunks) { issueA(c + 1, st ^ 1); issueF(c + 1, st ^ 1); }
        // fragment reads run four k-steps ahead of their MFMAs (inline asm: behind the ds_read_tr16 builtin hipcc drains the
        // LDS-DMA queue with s_waitcnt vmcnt(0) in front of every chunk's first read, and its own waits do not count asm loads
        // anyway); every k-step waits for ITS four reads with a counted lgkmcnt (LDS operations complete in order; at most 16
        // are outstanding) that names their destinations (guide 5.7, form (ii))
        const unsigned so = (unsigned)(st * DSTAGE_BYTES);
        u32x4 fa[4][2];
        s16x4 flo[4], fhi[4];
#define DW_READ(ks_, slot_)                                                                                               \
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[slot_][0]) : "v"(a_ad[ks_] + so));                                   \
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fa[slot_][1]) : "v"(a_ad[ks_] + so));                       \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(flo[slot_]) : "v"(lds_lo + so), "i"((ks_) * 16 * 256)); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fhi[slot_]) : "v"(lds_hi + so), "i"((ks_) * 16 * 256));
        DW_READ(0, 0) DW_READ(1, 1) DW_READ(2, 2) DW_READ(3, 3)
#pragma unroll
        for (int ks = 0; ks < DKB / 16; ++ks) {
            const int sl = ks & 3;
            if (ks <= 4) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(fa[sl][0]), "+v"(fa[sl][1]), "+v"(flo[sl]), "+v"(fhi[sl]) :: "memory");
            if (ks == 5) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[sl][0]), "+v"(fa[sl][1]), "+v"(flo[sl]), "+v"(fhi[sl]) :: "memory");
            if (ks == 6) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[sl][0]), "+v"(fa[sl][1]), "+v"(flo[sl]), "+v"(fhi[sl]) :: "memory");
            if (ks == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[sl][0]), "+v"(fa[sl][1]), "+v"(flo[sl]), "+v"(fhi[sl]) :: "memory");
            s16x8 v = {flo[sl][0], flo[sl][1], flo[sl][2], flo[sl][3], fhi[sl][0], fhi[sl][1], fhi[sl][2], fhi[sl][3]};
            const bf16x8 bv = __builtin_bit_cast(bf16x8, v);
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa[sl][0]), a1 = __builtin_bit_cast(bf16x8, fa[sl][1]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // keep this k-step's MFMAs in front of the refill and the next wait (guide rule 18)
            if (ks == 0) { DW_READ(4, 0) }
            if (ks == 1) { DW_READ(5, 1) }
            if (ks == 2) { DW_READ(6, 2) }
            if (ks == 3) { DW_READ(7, 3) }
        }
#undef DW_READ
        // chunk c+1 must have landed before anybody reads it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c < 3) DSTAMP(3 + c);
    }
    DSTAMP(6);

    if (OM == 0) {
        // fp32 slab tile: through the (now free) LDS stage 0, wave-private [64 rows][32 columns], so that a lane stores 16 B of
        // one row (8 rows x 128 B per instruction) instead of 32 scattered 4-byte write-through stores (scalar sc1 stores are
        // one fabric write each; the same change took 3 us off the fp32 forward).  No padding needed: the ds_write_b32 of a
        // register is 32 consecutive floats per lane half, the b128 reads of 8 rows x 8 quads are conflict-free.
        float* stg = reinterpret_cast<float*>(lds) + wave * (64 * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) stg[(i * 32 + acc_row(e, h)) * 32 + l31] = acc[i][e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* out = g.out + (size_t)z * g.slab_stride;
        const int nq = n0 + wn * 32 + 4 * (lane & 7);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3);
            const int m = m0 + wm * 64 + row;
            const f32x4v v = *reinterpret_cast<const f32x4v*>(stg + row * 32 + 4 * (lane & 7));
            if (m < g.M && nq < g.N) store_out_f32x4(out + (size_t)m * g.ldo + nq, v, g.plain);
        }
    } else {
    const int n = n0 + wn * 32 + l31;
    if (n < g.N) {
        u16* o16 = static_cast<u16*>(g.out16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int m = m0 + wm * 64 + i * 32 + acc_row(e, h);
                if (m >= g.M) continue;
                const u16 v = f2bf(acc[i][e]);
                if (OM == 1) o16[(size_t)m * g.ldo + n] = v;
                else o16[((size_t)(n >> 6) * g.ldo + m) * 64 + (n & 63)] = v;
            }
    }
    }
    DSTAMP(7);
    if (GATED && gate.timeline && tid == 0) gate.timeline[(size_t)(gate.self_base + vbid) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    return TW_OK;
#undef DSTAMP
}

template <int AM, int OM>
__global__ __launch_bounds__(512) void dw_bf16_dma(DwArgsB g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    DwGate none;
    none.ctl = StepCtl{nullptr, nullptr, nullptr, 0u}; none.fwd_base = none.self_base = 0; none.ts = 32; none.total_cols = 0; none.timeline = nullptr;
    (void)dw_bf16_body_dma<AM, OM, false>(g, (int)blockIdx.x, none, smem_dyn, nullptr);
}

template <int AM, int OM>
__global__ __launch_bounds__(512) void dw_bf16(DwArgsB g) {
    __shared__ __attribute__((aligned(16))) unsigned char dw_lds[DW_LDS_BYTES];
    DwGate none;
    none.ctl = StepCtl{nullptr, nullptr, nullptr, 0u}; none.fwd_base = none.self_base = 0; none.ts = 32; none.total_cols = 0; none.timeline = nullptr;
    (void)dw_bf16_body<AM, OM, false>(g, (int)blockIdx.x, none, dw_lds, nullptr);
}

// --------------------------------------------------------------------------- //
// The whole step of a linear bf16 head as ONE launch of persistent workgroups over CLAIMED TASKS (umlh_common.h, StepCtl):
//   phase 0  forward tiles     [0, nfwd)                       fwd_ce_bf16_body, publishes after its write-through dZ^T stores
//   phase 1  dW tiles          [nfwd, nfwd + ndw)              dw_bf16_body<GATED>: waits for the forward tiles of its K range
//   phase 2  update slices     [nfwd + ndw, .. + nupd)         two 256-thread sub-blocks of head_step_kernel's work each (same
//            + finalize        the last task                   arithmetic, same order); wait for the dW tiles of their 128-class
//                                                              tile rows; finalize waits for every forward tile
// nupd = 0: forward + dW only (the update is a separate launch: gradient diagnostics on).  hf.grad_out: the update slices
// write the summed gradient (data-parallel split step) instead of stepping the weights.
// Workgroup b runs its home tasks b, b + G, ... of each phase in phase order.  A task whose home workgroup has not arrived
// is taken by the first workgroup that needs its result (tasks_wait), so no wait depends on dispatch order or residency.
// Handed-off data (dZ^T, slabs, partials) is stored write-through and read with device-coherent loads.
// --------------------------------------------------------------------------- //
// the dW body of the one-launch step: the LDS-DMA tile (round 3); -DUMLH_STEP_DW_OLD builds the register-staged one for A/B runs
#ifdef UMLH_STEP_DW_OLD
#define UMLH_STEP_DW_BODY dw_bf16_body
#define UMLH_STEP_DW_LDS DW_LDS_BYTES
#else
#define UMLH_STEP_DW_BODY dw_bf16_body_dma
#define UMLH_STEP_DW_LDS DW_DMA_LDS_BYTES
#endif
struct StepShape { int nfwd, ndw, nupd, nfin;
                   int lazy;      // test switch (UMLH_STEP_LAZY=1): every 4th workgroup leaves its forward and dW home tasks alone,
                                  // as if it had not been dispatched yet -- whoever needs them takes them (tasks_wait)
                   int fs, fper; };   // home forward tile of workgroup b = (b % fs) * fper + b / fs  (fs = 0: tile b).  With fs = the
                                  // number of K splits and fper tiles per split, the forward tiles of K range z run on the workgroups
                                  // with b % fs == z -- under round-robin placement the XCD whose dW tiles (split z = b % nsplit) read
                                  // those feature rows and that slice of dZ^T a few microseconds later: their loads find the lines in
                                  // that XCD's L2.  Placement is a speed matter only (every load involved stays what it was).

template <int CTW, int WC>
__device__ __forceinline__ int step_update_task(const HeadFuse& hf, const DwGate& gate, const StepShape& sh, int u, int* sh_rc) {
    const int t_self = sh.nfwd + sh.ndw + u;
    unsigned long long* tl = gate.timeline ? gate.timeline + (size_t)t_self * 4 : nullptr;
    if (tl && threadIdx.x == 0) tl[0] = __builtin_amdgcn_s_memrealtime();
    // ---- update: sub-block `sub`, thread t of 256 ----
    const int sub = 2 * u + ((int)threadIdx.x >> 8), t = (int)threadIdx.x & 255;
    const long long n4 = (long long)hf.C * hf.K / 4;
    const long long g4 = (long long)sub * 256 + t;
    const bool live = sub < hf.n_sub && g4 < n4;
    const long long i = (live ? g4 : 0) * 4;
    f32x4v gi = {0.f, 0.f, 0.f, 0.f}, gt = gi, p0 = gi, m0 = gi, v0 = gi;
    const bool upd = hf.grad_out == nullptr;                 // (grid-uniform) else: the data-parallel split step's gradient only
    if (live && upd) {
        p0 = *reinterpret_cast<f32x4v*>(hf.p + i);
        m0 = *reinterpret_cast<f32x4v*>(hf.m + i);
        if (hf.o.kind != UMLH_OPT_SGD) v0 = *reinterpret_cast<f32x4v*>(hf.v + i);
    }
    if ((int)threadIdx.x < 64) {                             // wave 0: the dW tiles of the workgroup's tile row(s)
        const long long e0 = (long long)(2 * u) * 1024, e1 = min(e0 + 2047, (long long)hf.C * hf.K - 1);
        const int m_lo = (int)(e0 / hf.K) / 128, m_hi = (int)(e1 / hf.K) / 128;
        const int rc = tasks_wait(gate.ctl, sh.nfwd + m_lo * hf.dw_per_row, (m_hi - m_lo + 1) * hf.dw_per_row, (int)threadIdx.x, 2u);
        if (threadIdx.x == 0) *sh_rc = rc;
    }
    __syncthreads();
    const int rc = *sh_rc;
    __syncthreads();
    if (rc != TW_OK) return rc;
    if (tl && threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
    if (live) {
        {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hf.slabs), 0,
                                                                          (int)((size_t)hf.n_slabs * hf.slab_stride * 4), 0x00020000);
            for (int sI = 0; sI < hf.n_slabs_img; ++sI)
                gi += __builtin_bit_cast(f32x4v, load_coherent_b128(rs, (unsigned)(((size_t)sI * hf.slab_stride + i) * 4)));
            for (int sI = hf.n_slabs_img; sI < hf.n_slabs; ++sI)
                gt += __builtin_bit_cast(f32x4v, load_coherent_b128(rs, (unsigned)(((size_t)sI * hf.slab_stride + i) * 4)));
        }
        const f32x4v g0 = gi + gt;
        if (!upd) {
            store_out_f32x4(hf.grad_out + i, g0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pa = p0[j], mb = m0[j], vc = v0[j];
                opt_update(hf.o, g0[j], pa, mb, vc);
                p0[j] = pa; m0[j] = mb; v0[j] = vc;
            }
            store_out_f32x4(hf.p + i, p0, 0);
            store_out_f32x4(hf.m + i, m0, 0);
            if (hf.o.kind != UMLH_OPT_SGD) store_out_f32x4(hf.v + i, v0, 0);
            if (hf.shadow != nullptr) {
                const int cls = (int)(i / hf.K), k = (int)(i % hf.K);
                const long long piece = ((long long)(k >> 4) * (hf.cpad / 32) + (cls >> 5)) * 64 + (cls & 31) + 32 * ((k >> 3) & 1);
                typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
                const u32x2s w = {pack_bf16x2(p0[0], p0[1]), pack_bf16x2(p0[2], p0[3])};
                asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(hf.shadow + piece * 8 + (k & 7)), "v"(w) : "memory");
            }
        }
    }
    if (tl && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl[2] = __builtin_amdgcn_s_memrealtime(); }
    return TW_OK;
}

struct StepArgs { FwdArgsB a; DwArgsB g; DwGate gate; StepShape sh; HeadFuse hf; };   // THE kernel argument (one struct: the cold
                                                                                       // path below re-reads it from the kernarg segment)
template <int CTW, int WC>
__device__ __forceinline__ void step_fwd_task(const StepArgs& p, int t, unsigned char* smem, const TakeHook* hook = nullptr) {
    unsigned long long* tl = p.gate.timeline ? p.gate.timeline + (size_t)t * 4 : nullptr;
    if (tl && threadIdx.x == 0) tl[0] = __builtin_amdgcn_s_memrealtime();
    if (!fwd_ce_bf16_body<CTW, WC, 1>(p.a, t, smem, hook)) return;   // (taken by somebody else: nothing was stored)
    if (tl && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl[2] = __builtin_amdgcn_s_memrealtime(); }
    task_publish(p.gate.ctl, t);
}

template <int CTW, int WC>
__device__ __forceinline__ int step_fin_task(const StepArgs& p, int t, unsigned char* smem, int* sh_ctl) {
    unsigned long long* tl = p.gate.timeline ? p.gate.timeline + (size_t)t * 4 : nullptr;
    if (tl && threadIdx.x == 0) tl[0] = __builtin_amdgcn_s_memrealtime();
    if ((int)threadIdx.x < 64) {
        const int w = tasks_wait(p.gate.ctl, 0, p.sh.nfwd, (int)threadIdx.x, 3u);
        if (threadIdx.x == 0) sh_ctl[0] = w;
    }
    __syncthreads();
    const int rc = sh_ctl[0];
    __syncthreads();
    if (rc != TW_OK) return rc;
    if (tl && threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
    finalize_body<true>(p.hf.f, reinterpret_cast<float (*)[256]>(smem));
    if (tl && threadIdx.x == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
    return TW_OK;
}

// COLD path, not inlined (its copies of the bodies and its loops must not take part in the register allocation of the hot,
// straight-line kernel below; it re-reads the kernel argument from the kernarg segment).  Runs `first` (a task this workgroup
// has TAKEN: one that a wait handed back because nobody held it, or -- take = 1 -- a home task beyond the first of a phase,
// which is taken here) and then `then` (the home task whose wait was interrupted; -1 = none), each to completion; a wait in
// either can hand back a lower-phase task again, which runs first (depth: update -> dW -> forward).
template <int CTW, int WC>
__device__ __attribute__((noinline)) void step_cold(const StepArgs* kp, int first, int then, int take, unsigned char* smem, int* sh_ctl) {
    const StepArgs& p = *kp;
    const StepCtl& ctl = p.gate.ctl;
    const int b_dw = p.sh.nfwd, b_upd = p.sh.nfwd + p.sh.ndw, b_fin = p.sh.nfwd + p.sh.ndw + p.sh.nupd;
    if (take) {
        if (threadIdx.x == 0) sh_ctl[1] = task_take(ctl, first) ? 1 : 0;
        __syncthreads();
        const int mine = sh_ctl[1];
        __syncthreads();
        if (!mine) return;
    }
    // pending tasks, innermost last: s0 (= `then`, or `first`), s1, s2 (registers, no private array)
    int s0 = then >= 0 ? then : first, s1 = then >= 0 ? first : -1, s2 = -1;
    int sp = then >= 0 ? 2 : 1;
    while (sp > 0) {
        const int t = sp == 1 ? s0 : (sp == 2 ? s1 : s2);
        int rc = TW_OK;
        if (t < b_dw) { step_fwd_task<CTW, WC>(p, t, smem); --sp; continue; }
        else if (t < b_upd) rc = UMLH_STEP_DW_BODY<0, 0, true>(p.g, t - b_dw, p.gate, smem, sh_ctl);
        else if (t < b_fin) rc = step_update_task<CTW, WC>(p.hf, p.gate, p.sh, t - b_upd, sh_ctl);
        else rc = step_fin_task<CTW, WC>(p, t, smem, sh_ctl);
        if (rc == TW_OK) { task_publish(ctl, t); --sp; }
        else if (rc == TW_ABORT) { __syncthreads(); sp = 0; }    // status is set: these tasks and what depends on them are skipped
        else { if (sp == 1) s1 = rc; else s2 = rc; ++sp; }       // (depth <= 3: update -> dW -> forward)
    }
}

// HOT path: straight-line -- the first home task of each phase inline (a grid as wide as the phases, the normal case, has no
// other), in execution order forward, dW, finalize (needs the forward tiles only; its home is the LAST workgroup, which has
// no update slice when the grid is wider than the update), update.
template <int CTW, int WC>
__global__ __launch_bounds__(512) void step_bf16(StepArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    __shared__ int sh_ctl[8];                                // [0] wait verdict, [1] take verdict (cold path), [4..7] this workgroup's first home
                                                             // task of each phase: taken or not (32 bytes of static LDS: the dynamic base stays 16-byte aligned)
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const StepCtl& ctl = p.gate.ctl;
    const StepArgs* kp = (const StepArgs*)__builtin_amdgcn_kernarg_segment_ptr();   // for the cold path: it reads the argument from there
    const int b_dw = p.sh.nfwd, b_upd = p.sh.nfwd + p.sh.ndw, b_fin = p.sh.nfwd + p.sh.ndw + p.sh.nupd;   // task ids: forward, dW, update, finalize
    const bool lazy = p.sh.lazy && (b & 3) == 1;
    const int fb = (p.sh.fs > 0 && G >= p.sh.nfwd && b < p.sh.nfwd) ? (b % p.sh.fs) * p.sh.fper + b / p.sh.fs : b;   // home forward tile (StepShape)
    // all first home tasks are taken NOW, by four lanes of one instruction: the takes of the later phases cost nothing when
    // their turn comes, and nobody who waits for one of them finds it untaken while this workgroup is still in an earlier phase
    TakeHook hook = {ctl.epoch, ctl.epoch, sh_ctl + 4};       // (old == epoch reads as "not mine")
    if (threadIdx.x < 4) {
        const int k = (int)threadIdx.x;
        int t = -1;
        if (k == 0 && !lazy && b < p.sh.nfwd) t = fb;
        if (k == 1 && !lazy && b < p.sh.ndw) t = b_dw + b;
        if (k == 2 && p.sh.nfin > 0 && b == G - 1) t = b_fin;
        if (k == 3 && b < p.sh.nupd) t = b_upd + b;
        if (t >= 0) hook.old = __hip_atomic_fetch_max(ctl.claim + t, ctl.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // issued, not waited for
    }
    if (!lazy && b < p.sh.nfwd) {
        // the forward tile starts on its loads NOW and settles the four takes in front of its first barrier (TakeHook): the
        // round trip of the atomics (0.5-1 us when 256 workgroups start together) is off the step's critical path
        step_fwd_task<CTW, WC>(p, fb, smem_dyn, &hook);
    } else {
        if (threadIdx.x < 4) sh_ctl[4 + threadIdx.x] = tag_older(hook.old, ctl.epoch) ? 1 : 0;
        __syncthreads();
    }
#pragma unroll 1
    for (int i = b + G; i < p.sh.nfwd && !lazy; i += G) step_cold<CTW, WC>(kp, i, -1, 1, smem_dyn, sh_ctl);
    if (sh_ctl[5]) {
        const int rc = UMLH_STEP_DW_BODY<0, 0, true>(p.g, b, p.gate, smem_dyn, sh_ctl);   // the same dynamic LDS, laid out for dW
        if (rc == TW_OK) task_publish(ctl, b_dw + b);
        else if (rc == TW_ABORT) __syncthreads();            // status is set: this task and what depends on it is skipped
        else step_cold<CTW, WC>(kp, rc, b_dw + b, 0, smem_dyn, sh_ctl);   // a forward tile nobody had taken: run it, then this tile from the start (nothing was stored)
    }
#pragma unroll 1
    for (int i = b + G; i < p.sh.ndw && !lazy; i += G) step_cold<CTW, WC>(kp, b_dw + i, -1, 1, smem_dyn, sh_ctl);
    if (sh_ctl[6]) {
        const int rc = step_fin_task<CTW, WC>(p, b_fin, smem_dyn, sh_ctl);
        if (rc == TW_OK) task_publish(ctl, b_fin);
        else if (rc >= 0) step_cold<CTW, WC>(kp, rc, b_fin, 0, smem_dyn, sh_ctl);
    }
    if (sh_ctl[7]) {
        const int rc = step_update_task<CTW, WC>(p.hf, p.gate, p.sh, b, sh_ctl);
        if (rc == TW_OK) task_publish(ctl, b_upd + b);
        else if (rc >= 0) step_cold<CTW, WC>(kp, rc, b_upd + b, 0, smem_dyn, sh_ctl);   // a dW tile nobody had taken
    }
#pragma unroll 1
    for (int i = b + G; i < p.sh.nupd; i += G) step_cold<CTW, WC>(kp, b_upd + i, -1, 1, smem_dyn, sh_ctl);
}

// --------------------------------------------------------------------------- //
extern "C" {

int umlh_launch_to_bf16(const float* src, void* dst, long long n, hipStream_t stream) {
    if (n <= 0) return 0;
    int blocks = (int)((n + 2047) / 2048);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(blocks), dim3(256), 0, stream, src, (u16*)dst, n);
    return (int)hipGetLastError();
}

int umlh_launch_iota(long long* dst, long long n, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, n);
    return (int)hipGetLastError();
}

int umlh_launch_w_shadow(const float* w, void* dst, int C, int K, int cpad, hipStream_t stream) {
    long long total = (long long)(K / 16) * (cpad / 32) * 64;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(w_shadow_kernel, dim3(blocks), dim3(256), 0, stream, w, (u16*)dst, C, K, cpad);
    return (int)hipGetLastError();
}

// samples per block for a class-tile configuration (ctw, wc) and stw
int umlh_bf16_fwd_ts(int wc, int stw) { return 32 * stw * (8 / wc); }

static size_t fwd_smem_bytes_b(int ctw, int wc, int stw) {
    int ws = 8 / wc, ts = 32 * stw * ws;
    int xk = ws <= 2 ? 512 : (ws == 4 ? 256 : 128);
    size_t xt = (size_t)ts * (xk + 8) * 2, stage = (size_t)8 * ctw * 32 * 80;
    return (xt > stage ? xt : stage) + sizeof(float) * (size_t)(8 * 32 * 8 + ws * 4 + 16);
}

#define FWDB_CASE(CT, W, S)                                                                          \
    if (ctw == CT && wc == W && stw == S) {                                                          \
        size_t sm = fwd_smem_bytes_b(CT, W, S);                                                      \
        static std::atomic<unsigned long long> attr_done{0};  /* bit d: done on device d (the attribute is per device) */ \
        int dev_ = 0;                                                                                \
        (void)hipGetDevice(&dev_);                                                                   \
        if (!((attr_done.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ULL)) {                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_ce_bf16<CT, W, S>),\
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
            if (e != hipSuccess) return (int)e;                                                      \
            attr_done.fetch_or(1ULL << (dev_ & 63), std::memory_order_release);                      \
        }                                                                                            \
        FwdArgsB c_ = *a; c_.plain = umlh_plain_stores();                                            \
        hipLaunchKernelGGL((fwd_ce_bf16<CT, W, S>), dim3(grid), dim3(512), sm, stream, c_);          \
        return (int)hipGetLastError();                                                               \
    }

// 2-D forward: grid = row tiles (128 rows; seg[].blk0 in tile units) x nq class groups of 256
#define FWDQ_CASE(NQ_, NKS_)                                                                                       \
    if (nq == NQ_ && a->K == 16 * NKS_) {                                                                          \
        static std::atomic<unsigned long long> attr_done{0};                                                       \
        if (!((attr_done.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ULL)) {                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_ce_bf16_q<NQ_, NKS_>),           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, FQ_SMEM);               \
            if (e != hipSuccess) return (int)e;                                                                    \
            attr_done.fetch_or(1ULL << (dev_ & 63), std::memory_order_release);                                    \
        }                                                                                                          \
        FwdArgsB c_ = *a; c_.plain = umlh_plain_stores();                                                          \
        hipLaunchKernelGGL((fwd_ce_bf16_q<NQ_, NKS_>), dim3((tiles + 7) / 8 * 8 * NQ_), dim3(512), FQ_SMEM, stream, c_);        \
        return (int)hipGetLastError();                                                                             \
    }

int umlh_bf16_launch_fwd_q(const FwdArgsB* a, int nq, int tiles, hipStream_t stream) {
    if (tiles <= 0) return 0;
    if (!a->xch || a->epoch == 0 || a->C > 256 * nq || a->wtiles < 8 * nq || a->ntiles != tiles) return (int)hipErrorInvalidValue;
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    FWDQ_CASE(2, 32) FWDQ_CASE(3, 32) FWDQ_CASE(4, 32) FWDQ_CASE(2, 16) FWDQ_CASE(3, 16) FWDQ_CASE(4, 16)
    return (int)hipErrorInvalidValue;
}

int umlh_bf16_launch_fwd(const FwdArgsB* a, int ctw, int wc, int stw, int grid, hipStream_t stream) {
    if (grid <= 0) return 0;
    FWDB_CASE(1, 1, 1) FWDB_CASE(1, 2, 1) FWDB_CASE(1, 4, 1) FWDB_CASE(1, 8, 1) FWDB_CASE(2, 8, 1) FWDB_CASE(4, 8, 1)
    FWDB_CASE(2, 8, 2) FWDB_CASE(4, 8, 2)
    return (int)hipErrorInvalidValue;
}

// mode 0: dst[c*ldd + r]; mode 1: dst[((r>>6)*ldd + c)*64 + (r&63)]  (see transpose_shadow_kernel)
int umlh_bf16_launch_transpose_shadow(const float* src, int R, int Cc, int ldd, void* dst, int mode, hipStream_t stream) {
    if (R <= 0 || Cc <= 0) return 0;
    if (ldd % 8 != 0) return (int)hipErrorInvalidValue;
    dim3 grid((Cc + 63) / 64, (R + 63) / 64);
    if (mode == 0) hipLaunchKernelGGL((transpose_shadow_kernel<0>), grid, dim3(256), 0, stream, src, R, Cc, ldd, (u16*)dst);
    else hipLaunchKernelGGL((transpose_shadow_kernel<1>), grid, dim3(256), 0, stream, src, R, Cc, ldd, (u16*)dst);
    return (int)hipGetLastError();
}

#define STEP_CASE(CT, W)                                                                                           \
    if (ctw == CT && wc == W) {                                                                                    \
        size_t sm = fwd_smem_bytes_b(CT, W, 1);                                                                    \
        if (sm < (size_t)UMLH_STEP_DW_LDS) sm = UMLH_STEP_DW_LDS;                                                  \
        static std::atomic<unsigned long long> attr_done{0};  /* bit d: done on device d (the attribute is per device) */ \
        if (!((attr_done.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ULL)) {                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_bf16<CT, W>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);               \
            if (e != hipSuccess) return (int)e;                                                                    \
            attr_done.fetch_or(1ULL << (dev_ & 63), std::memory_order_release);                                    \
        }                                                                                                          \
        hipLaunchKernelGGL((step_bf16<CT, W>), dim3(grid), dim3(512), sm, stream, sa);           \
        return (int)hipGetLastError();                                                                             \
    }

// forward, dW and (hf != NULL) the update + finalize in one launch of `grid` persistent workgroups over claimed tasks.
// claim: [ntask] u32, done: [ntask] u64, status: [4] u32 (device words, epoch-tagged: never reset); *ntask_out = tasks of the launch.
int umlh_bf16_step_tasks(int nfwd, int M, int N, int splits, long long n_head, int with_head) {
    const int ndw = ((N + DBN - 1) / DBN) * ((M + DBM - 1) / DBM) * splits;
    const int n_sub = (int)((n_head / 4 + 255) / 256);
    return nfwd + ndw + (with_head ? (n_sub + 1) / 2 + 1 : 0);
}

int umlh_bf16_launch_step(const FwdArgsB* a, int ctw, int wc, int nfwd, const DwArgsB* g, int splits, unsigned* claim,
                          unsigned long long* done, unsigned* status, unsigned epoch, int ts, int total_cols, const HeadFuse* hf,
                          unsigned long long* timeline, int cus, int lazy, hipStream_t stream) {
    if (nfwd <= 0 || g->M <= 0 || g->N <= 0 || !claim || !done || !status || epoch == 0 || !a->dzt) return (int)hipErrorInvalidValue;
    if (g->k_chunk > DIDS || g->k_chunk % (DKT * DNS) != 0 || g->nsplit != splits) return (int)hipErrorInvalidValue;
    if (g->k_switch % DKT != 0 || g->bcs < 64 || g->N % 8 != 0 || umlh_plain_stores()) return (int)hipErrorInvalidValue;
    if (hf && (hf->K % 8 != 0 || hf->C != g->M || hf->K != g->N || hf->slab_stride % 4 != 0)) return (int)hipErrorInvalidValue;
    const int nx = (g->N + DBN - 1) / DBN;
    const int ndw = nx * ((g->M + DBM - 1) / DBM) * splits;
    FwdArgsB fa = *a; fa.plain = 0;
    DwArgsB ga = *g; ga.plain = 0;
    HeadFuse h;
    memset(&h, 0, sizeof(h));
    StepShape shp = {nfwd, ndw, 0, 0, lazy, 0, 0};
    {   // forward tiles of a K range on the workgroups of "its" XCD (StepShape::fs; UMLH_STEP_XCD=0: tile b on workgroup b)
        static const bool xcd_map = [] { const char* e = getenv("UMLH_STEP_XCD"); return !(e && atoi(e) == 0); }();
        const int per = ts > 0 && g->k_chunk % ts == 0 ? g->k_chunk / ts : 0;
        if (xcd_map && splits > 1 && per > 0 && nfwd == per * splits && g->k_switch == g->nsplit1 * g->k_chunk) { shp.fs = splits; shp.fper = per; }
    }
    if (hf) {
        h = *hf;
        const long long n4 = (long long)h.C * h.K / 4;
        h.n_sub = (int)((n4 + 255) / 256);
        h.dw_per_row = nx * splits;
        shp.nupd = (h.n_sub + 1) / 2;
        shp.nfin = 1;
    }
    DwGate gate;
    gate.ctl = StepCtl{claim, done, status, epoch};
    gate.fwd_base = 0; gate.self_base = nfwd; gate.ts = ts; gate.total_cols = total_cols; gate.timeline = timeline;
    int grid = nfwd > ndw ? nfwd : ndw;
    if (shp.nupd + 1 > grid) grid = shp.nupd + 1;
    if (cus > 0 && grid > cus) grid = cus;
    StepArgs sa;
    sa.a = fa; sa.g = ga; sa.gate = gate; sa.sh = shp; sa.hf = h;
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    STEP_CASE(1, 1) STEP_CASE(1, 2) STEP_CASE(1, 4) STEP_CASE(1, 8) STEP_CASE(2, 8) STEP_CASE(4, 8)
    return (int)hipErrorInvalidValue;
}

#define DW_DMA_CASE(A_, O_)                                                                                        \
    if (am == A_ && om == O_) {                                                                                    \
        static std::atomic<unsigned long long> attr_done{0};                                                       \
        if (!((attr_done.load(std::memory_order_acquire) >> (dev_ & 63)) & 1ULL)) {                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_bf16_dma<A_, O_>),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, DW_DMA_LDS_BYTES);      \
            if (e != hipSuccess) return (int)e;                                                                    \
            attr_done.fetch_or(1ULL << (dev_ & 63), std::memory_order_release);                                    \
        }                                                                                                          \
        hipLaunchKernelGGL((dw_bf16_dma<A_, O_>), grid, dim3(512), DW_DMA_LDS_BYTES, stream, c);                   \
        return (int)hipGetLastError();                                                                             \
    }

int umlh_bf16_launch_dw(const DwArgsB* g, int splits, int am, int om, hipStream_t stream) {
    if (g->M <= 0 || g->N <= 0) return 0;
    if (g->k_chunk > DIDS || g->k_chunk % (DKT * DNS) != 0 || g->nsplit != splits) return (int)hipErrorInvalidValue;
    if (g->k_switch % DKT != 0 || g->bcs < 64 || g->N % 8 != 0) return (int)hipErrorInvalidValue;
    if ((am == 1 && !g->a_rows) || (om != 0 && (!g->out16 || splits != 1))) return (int)hipErrorInvalidValue;
    dim3 grid(((g->N + DBN - 1) / DBN) * ((g->M + DBM - 1) / DBM) * splits);
    DwArgsB c = *g;
    c.plain = umlh_plain_stores();
    // UMLH_BF16_DW=0: the register-staged tile of rounds 1-2 (A/B timing); default: the LDS-DMA tile
    static const bool old_tile = [] { const char* e = getenv("UMLH_BF16_DW"); return e && atoi(e) == 0; }();
    if (!old_tile) {
        int dev_ = 0;
        (void)hipGetDevice(&dev_);
        DW_DMA_CASE(0, 0) DW_DMA_CASE(1, 1) DW_DMA_CASE(0, 2)
        return (int)hipErrorInvalidValue;
    }
    if (am == 0 && om == 0) hipLaunchKernelGGL((dw_bf16<0, 0>), grid, dim3(512), 0, stream, c);
    else if (am == 1 && om == 1) hipLaunchKernelGGL((dw_bf16<1, 1>), grid, dim3(512), 0, stream, c);
    else if (am == 0 && om == 2) hipLaunchKernelGGL((dw_bf16<0, 2>), grid, dim3(512), 0, stream, c);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

}  // extern "C"
