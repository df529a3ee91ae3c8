// Host side of the C ABI declared in include/umlh.h: argument checks, workspace
// partitioning, per-step launch sequences.  No allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>

#include "umlh_common.h"
#include "umlh_enc.h"
#include "umlh_micro.h"
#include <mutex>

extern "C" {
int umlh_micro_chunking(int d, int* nch, int* cw);
int umlh_micro_launch(int nch, int cw, int bf16, const UmlhMicroHead* heads, int n_heads, int n_steps, int grid, hipStream_t st);
int umlh_micro_bf16_supported(int nch, int cw);
int umlh_f32_fwd_config(int C, int* ctw, int* wc);
int umlh_f32_launch_fwd(const FwdArgs* a, int ctw, int wc, int grid, hipStream_t stream);
int umlh_f32_launch_gemm(const GemmArgs* g, int ta, int tb, int splits, hipStream_t stream);
int umlh_f32_launch_gemm_enc(const GemmArgs* g, int ta, int tb, int splits, hipStream_t stream);
int umlh_launch_reduce_update(int mode, const float* slabs, int n_slabs, long long slab_stride, long long n,
                              float* grad_out, float* p, float* m, float* v, const OptArgs* o, long long frozen_lo,
                              long long frozen_hi, hipStream_t stream);
int umlh_launch_finalize(const FinalizeArgs* f, hipStream_t stream);
int umlh_launch_w_shadow32(const float* w, float* dst, int C, int K, int cpad, hipStream_t stream);
int umlh_launch_head_step(const float* slabs, int n_slabs, long long slab_stride, int C, int K, float* p, float* m,
                          float* v, const OptArgs* o, void* shadow, int cpad, const FinalizeArgs* f, float* grad_out,
                          const DiagArgs* dg, float* shadow32, hipStream_t stream);
int umlh_launch_zero_shot(const float* feats, const int64_t* labels, long long n, int d, int C, float* w,
                          hipStream_t stream);
int umlh_launch_to_bf16(const float* src, void* dst, long long n, hipStream_t stream);
int umlh_bf16_fwd_ts(int wc, int stw);
int umlh_launch_w_shadow(const float* w, void* dst, int C, int K, int cpad, hipStream_t stream);
int umlh_launch_iota(long long* dst, long long n, hipStream_t stream);
}

extern "C" {
int umlh_bf16_launch_fwd(const FwdArgsB* a, int ctw, int wc, int stw, int grid, hipStream_t stream);
int umlh_bf16_launch_fwd_q(const FwdArgsB* a, int nq, int tiles, hipStream_t stream);
int umlh_bf16_step_tasks(int nfwd, int M, int N, int splits, long long n_head, int with_head);
int umlh_bf16_launch_step(const FwdArgsB* a, int ctw, int wc, int nfwd, const DwArgsB* g, int splits, unsigned* claim,
                          unsigned long long* done, unsigned* status, unsigned epoch, int ts, int total_cols, const HeadFuse* hf,
                          unsigned long long* timeline, int cus, int lazy, hipStream_t stream);
int umlh_bf16_launch_dw(const DwArgsB* g, int splits, int am, int om, hipStream_t stream);
int umlh_p2p_launch(void* const* regions, int n_ranks, int rank, float* msg, long long n, long long n_max, unsigned epoch, hipStream_t st);
int umlh_p2p_status_offset(long long n_max, int n_ranks, unsigned long long* off);
int umlh_enc_launch_bias_act(float* y, const float* b, long long M, int N, int relu, hipStream_t st);
int umlh_enc_launch_relu_bwd(const float* y, float* dy, long long n, hipStream_t st);
int umlh_enc_launch_dropout(float* x, long long n, float p, unsigned long long seed, hipStream_t st);
int umlh_enc_launch_colsum(const float* x, int M, int N, float* out, hipStream_t st);
int umlh_enc_launch_add_inplace(float* y, const float* x, long long n, hipStream_t st);
int umlh_enc_launch_add_layernorm(const float* x, const float* r, const float* gamma, const float* beta, int M, int N, float eps,
                                  float* s_out, float* y, float* mean, float* rstd, hipStream_t st);
int umlh_enc_launch_layernorm_bwd(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                                  int M, int N, float* ds, float* dgamma, float* dbeta, hipStream_t st);
int umlh_enc_launch_add_pos(float* x, const float* pos, int T, int B, int Z, hipStream_t st);
int umlh_enc_launch_pos_grad(const float* dx, int T, int B, int Z, float* dpos, hipStream_t st);
int umlh_enc_launch_gather_rows(const float* x, const int64_t* idx, int n, int Z, float* out, int scatter, hipStream_t st);
int umlh_enc_launch_attention_fwd(const float* qkv, const int64_t* lengths, int T, int B, int Z, int H, float p,
                                  unsigned long long seed, const unsigned long long* seed_ptr, float* ctx, float* lse, hipStream_t st);
int umlh_enc_launch_attention_bwd(const float* qkv, const int64_t* lengths, const float* lse, const float* dctx, int T, int B,
                                  int Z, int H, float p, unsigned long long seed, const unsigned long long* seed_ptr, float* dqkv, hipStream_t st);
int umlh_bf16_launch_transpose_shadow(const float* src, int R, int Cc, int ldd, void* dst, int mode, hipStream_t stream);
}

static thread_local char g_err[512] = "";
static int device_cus(int dev);

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static inline long long round_up(long long x, long long m) { return (x + m - 1) / m * m; }

struct Layout {                 // workspace partition, in floats from the base
    long long dzt, h, dht, slabs_head, slabs_proj, partials, diag_part, grads, w16, iota, zeros, dbg, wpt16, wht16, xch, fuse_flags, w32s, total;
    long long ctl_tasks;        // one-launch step: task capacity of the control region at fuse_flags: [cap] u64 done, [cap] u32 claim, [16] u32 status
    int fwd_nq;                 // bf16 2-D forward (fwd_ce_bf16_q): class groups per row tile, 0 = the 1-D kernel
    long long mc_flags, mc_xchg, mc_ext, mc_tab, mc_desc;   // micro-step region (umlh_kernels_micro.hip); mc_flags = 0: unsupported shape
    int mc_nwg, mc_nch, mc_cw;
    long long n_iota;
    int rcap_img, rcap_txt, ldz;     // padded row capacities
    int scap_head, scap_proj;        // split-K slab capacities
    long long n_head, n_proj;        // parameter counts
    int max_blocks;
};

static int split_cap(int M, int N) {
    long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
    long long want = 256 / tiles;
    if (want < 2) want = 2;           // image rows and text rows never share a slab (plan_splits)
    if (want > 64) want = 64;
    return (int)want;
}

// Modality-aligned split-K plan for dW_head: `r0` reduction rows of the image segment followed by `r1`
// of the text segment, at most `want` slabs of `chunk` rows (a multiple of `quantum`) each, every slab
// inside one modality.  Keeping the two partial sums in separate slabs makes the reference's
// per-modality gradients (finetune.py:190-191) by-products of the slab reduction.
struct SplitPlan { int chunk, n_img, n_txt; };
static SplitPlan plan_splits(int r0, int r1, int want, int quantum, int min_chunk) {
    SplitPlan sp = {min_chunk, 0, 0};
    const long long tot = (long long)r0 + r1;
    if (tot <= 0) return sp;
    int s0 = r0 > 0 ? (int)((want * (long long)r0 + tot / 2) / tot) : 0;
    if (r0 > 0 && s0 < 1) s0 = 1;
    if (r1 > 0 && s0 > want - 1) s0 = want - 1;
    int s1 = r1 > 0 ? want - s0 : 0;
    if (r1 > 0 && s1 < 1) s1 = 1;
    long long c0 = s0 > 0 ? (r0 + s0 - 1) / s0 : 0, c1 = s1 > 0 ? (r1 + s1 - 1) / s1 : 0;
    long long chunk = round_up(c0 > c1 ? c0 : c1, quantum);
    if (chunk < min_chunk) chunk = min_chunk;
    sp.chunk = (int)chunk;
    sp.n_img = (int)((r0 + chunk - 1) / chunk);
    sp.n_txt = (int)((r1 + chunk - 1) / chunk);
    return sp;
}

// per-launch tables of one head: int offs_img[MAXS + 1], int offs_txt[MAXS + 1], OptArgs opt[MAXS]
static long long micro_table_floats() {
    return 2LL * (UMLH_MICRO_MAX_STEPS + 1) + 2 + (long long)UMLH_MICRO_MAX_STEPS * (sizeof(OptArgs) / sizeof(float));
}

static bool make_layout(const umlh_config_t& c, Layout& L) {
    if (c.d_img < 1 || c.d_shared < 1 || c.num_classes < 1 || c.num_classes > 1024) return false;
    if (!c.has_proj && c.d_img != c.d_shared) return false;
    if (c.max_rows_img < 0 || c.max_rows_txt < 0 || c.max_rows_img + c.max_rows_txt < 1) return false;
    if (c.optimizer < UMLH_OPT_SGD || c.optimizer > UMLH_OPT_ADAMW) return false;
    if (c.precision != UMLH_PREC_FP32 && c.precision != UMLH_PREC_BF16) return false;
    // bf16 mode: the shared dim is the K of the fused forward (X tile staged in 128-wide K blocks); with img_proj the
    // image width is the K of H = X W_proj^T, whose row-major A operand is read in whole 64-wide chunks
    if (c.precision == UMLH_PREC_BF16 && (c.d_shared % 128 != 0 || (c.has_proj && c.d_img % 64 != 0))) return false;
    L.rcap_img = (int)round_up(c.max_rows_img, 256);
    L.rcap_txt = (int)round_up(c.max_rows_txt, 256);
    L.ldz = L.rcap_img + L.rcap_txt;
    L.n_head = (long long)c.num_classes * c.d_shared;
    L.n_proj = c.has_proj ? (long long)c.d_shared * c.d_img : 0;
    L.scap_head = split_cap(c.num_classes, c.d_shared);
    if (c.precision == UMLH_PREC_BF16 && L.scap_head < (L.ldz + 4095) / 4096 + 1) L.scap_head = (L.ldz + 4095) / 4096 + 1;
    L.scap_proj = c.has_proj ? split_cap(c.d_shared, c.d_img) : 0;
    if (c.has_proj && c.precision == UMLH_PREC_BF16 && L.scap_proj < (L.rcap_img + 4095) / 4096) L.scap_proj = (L.rcap_img + 4095) / 4096;
    L.max_blocks = L.ldz / 32 + 2;
    long long off = 0;
    auto take = [&](long long n) { long long o = off; off += round_up(n, 64); return o; };
    // fp32: dZ^T [C][ldz] floats.  bf16: [ldz/64][crows][64] shorts, crows = C rounded up to 128
    L.dzt = take(c.precision == UMLH_PREC_BF16 ? (round_up(c.num_classes, 128) * (long long)L.ldz + 1) / 2
                                               : (long long)c.num_classes * L.ldz);
    L.h = take(c.has_proj ? (long long)L.rcap_img * c.d_shared : 0);
    L.dht = take(c.has_proj ? (long long)c.d_shared * L.rcap_img : 0);
    L.slabs_head = take((long long)L.scap_head * L.n_head);
    L.slabs_proj = take((long long)L.scap_proj * L.n_proj);
    L.partials = take((long long)L.max_blocks * 4);
    L.diag_part = take(4 * ((L.n_head + 1023) / 1024 + 2) + 64);     // gradient diagnostics: [head_step blocks][4] partials, then the ticket
    L.grads = take(2 * L.n_head + L.n_proj + 2 + UMLH_N_SCALARS);    // 2 x n_head: the data-parallel message carries the image and
                                                                     // the text gradient separately when diagnostics are on
    L.w16 = take(c.precision == UMLH_PREC_BF16 ? 1024LL * c.d_shared / 2 : 0);   // bf16 chunk-major shadow of w_head (<= 1024 class rows)
    L.n_iota = L.rcap_img > L.rcap_txt ? L.rcap_img : L.rcap_txt;     // identity row ids: batch rows, classes, image-feature columns
    if (L.n_iota < 1024) L.n_iota = 1024;
    if (L.n_iota < c.d_img) L.n_iota = c.d_img;
    L.iota = take(c.precision == UMLH_PREC_BF16 ? 2LL * L.n_iota : 0);   // int64 0..n_iota-1
    L.zeros = take(64);
    L.dbg = take((long long)L.max_blocks * 128);         // diagnostic stamps: [blocks][8 waves][8] u64
    const bool bfp = c.precision == UMLH_PREC_BF16 && c.has_proj;
    L.wpt16 = take(bfp ? (L.n_proj + 1) / 2 : 0);         // bf16 W_proj^T [d_img][d_shared]
    L.wht16 = take(bfp ? 1024LL * round_up(c.d_shared, 128) / 2 : 0);   // bf16 W_head^T by class chunks [16][d_shared^128][64]
    // 2-D forward (128-row tiles x groups of 256 classes, cross-workgroup softmax merge; K resident in LDS): measured
    // slower than the 1-D kernel at cfg2 (umlh_kernels_bf16.hip) -> opt-in, UMLH_BF16_FWD2D=1, whenever the shape allows.
    L.fwd_nq = 0; L.xch = 0;
    if (c.precision == UMLH_PREC_BF16 && c.num_classes > 256 && c.d_shared % 256 == 0 && c.d_shared <= 512) {
        const char* e = getenv("UMLH_BF16_FWD2D");
        const int mode = e ? atoi(e) : -1;
        const int nq = (c.num_classes + 255) / 256;
        if (mode == 1) {
            L.fwd_nq = nq;
            L.xch = take(2LL * (L.ldz / 128 + 2) * nq * 4 * 128);      // 8-byte granules
        }
    }
    // fp32 mode: fragment-major fp32 shadow of w_head for the streamed forward (fwd_ce_f32 MODE 2): [K/16][cpad/32][64][8] floats
    L.w32s = 0;
    if (c.precision == UMLH_PREC_FP32 && c.d_shared % 32 == 0) {
        int ctw = 0, wc = 0;
        if (umlh_f32_fwd_config(c.num_classes, &ctw, &wc) > 0) L.w32s = take((long long)c.d_shared * 32 * ctw * wc * 3 / 2);   // (x3: three bf16 planes = 1.5x the fp32 shadow)
    }
    // one-launch step (step_bf16): per task one done granule (u64) and one claim word (u32), then the status words
    L.ctl_tasks = L.max_blocks + 4096 + L.n_head / 2048 + 8;
    L.fuse_flags = c.precision == UMLH_PREC_BF16 && !c.has_proj ? take(3LL * L.ctl_tasks + 16) : 0;
    // micro-step path: linear head whose width has a supported chunking (bf16 operand mode: widths that are multiples of 128)
    L.mc_flags = L.mc_xchg = L.mc_ext = L.mc_tab = L.mc_desc = 0;
    L.mc_nwg = (c.num_classes + UMLH_MICRO_CS - 1) / UMLH_MICRO_CS;
    L.mc_nch = L.mc_cw = 0;
    if (!c.has_proj && umlh_micro_chunking(c.d_shared, &L.mc_nch, &L.mc_cw) &&
        (c.precision == UMLH_PREC_FP32 || umlh_micro_bf16_supported(L.mc_nch, L.mc_cw))) {
        L.mc_flags = take(64 + 64);                                              // [nwg <= 64] epoch flags, then the status word
        L.mc_xchg = take(2LL * L.mc_nwg * 5 * UMLH_MICRO_MAX_ROWS * 2);       // 8-byte granules
        L.mc_ext = 0;
        L.mc_tab = take(micro_table_floats());
        L.mc_desc = take((long long)UMLH_MICRO_MAX_HEADS * sizeof(UmlhMicroHead) / sizeof(float) + 16);
    }
    L.total = off;
    return true;
}

struct umlh_handle_s {
    umlh_config_t cfg;
    umlh_buffers_t buf;
    Layout L;
    bool bound;
    int ctw, wc, ts;            // fwd_ce tile configuration
    int stw;                    // bf16: 32-sample tiles per wave
    unsigned fwd_epoch;         // bf16 2-D forward: launch tag of the exchange granules
    unsigned fuse_epoch;        // single-launch forward + dW: launch tag of the forward blocks' granules
    int fuse;                   // forward and dW of a linear bf16 head as one launch (default; UMLH_BF16_FUSE=0: two launches)
    int dbg_step;               // UMLH_DBG_STEP=1: step_bf16 writes a per-task timeline into the debug buffer
    int step_lazy;              // UMLH_STEP_LAZY=1 (tests): see StepShape::lazy
    int step_grid;              // persistent workgroups of the one-launch step (CUs of the device; UMLH_STEP_GRID overrides)
    long long step_launches;    // one-launch steps taken (tests assert the path that ran)
    const HeadFuse* pending_head;   // set by train_step_impl: the update may ride in the same launch (step_bf16)
    bool head_fused_done;       // forward_backward took it
    // state carried from umlh_grad_step to umlh_apply_update
    int last_rows_img, last_rows_txt;
    bool iota_ready;            // bf16: identity row-id table in the workspace initialised
    bool shadow_fresh;          // bf16: the W shadow was written by the previous step's update kernel
    float* diag_dst;            // where this step's 4 gradient diagnostics go (written by the head-step launch)
    int n_slabs_img;            // dW_head slabs that hold image rows (the rest hold text rows)
    bool diagnostics;           // umlh_enable_diagnostics
    int  diag_cols = 0;         // umlh_set_diagnostic_columns (0 = every column of w_head)
    float* row_stats;           // per-row {CE, correct} output of the next forward (umlh_eval_rows), else NULL
    int dbg_fwd, dbg_dw;        // timing-only ablation / cycle-stamp switches (UMLH_DBG_FWD / UMLH_DBG_DW), read once at create
    hipEvent_t ev[UMLH_N_PHASES + 1];   // phase boundaries, valid when profiling
    bool profiling;
    // micro-step path
    unsigned micro_epoch;       // last epoch published by this handle's workgroups (flags are zeroed at bind)
    unsigned char* stage;       // pinned host staging [2][stage_bytes] for the per-launch tables / descriptors (lazy)
    size_t stage_bytes;
    hipEvent_t stage_ev[2];     // copy-out of staging buffer i has completed
    int stage_next;
    int micro_off;              // UMLH_MICRO=0: never take the micro path
    long long micro_launches;   // persistent launches this handle took part in (tests assert the path that ran)
    // data-parallel stepping
    int frozen_proj_row;        // umlh_freeze_proj_row: row of w_proj the optimizer leaves alone (-1 = none)
    bool dp_diag;               // layout of the gradient message: [g_img | g_txt | g_proj | g_scales | scalars] instead of [g_head | ...]
    int n_ranks;                // > 1 (or dp_force): umlh_train_steps runs grad -> all-reduce -> update per step
    int dp_force;               // UMLH_FORCE_DP=1 / umlh_set_allreduce with one rank: take the split path also alone (pricing, tests)
    umlh_allreduce_fn ar_fn;    // custom transport (tests: gloo through a host callback), else RCCL through `comm`
    void* ar_ctx;
    void* p2p_region[8];        // umlh_p2p_attach: every rank's exchange region as mapped here (p2p_region[rank] = this rank's own)
    int p2p_on, p2p_rank;       // direct peer-to-peer all-reduce instead of RCCL / the callback
    unsigned p2p_epoch;
    void* comm;                 // ncclComm_t
    bool comm_owned;
    hipStream_t comm_stream;    // second stream: the head-gradient all-reduce of a 2-layer head runs beside the img_proj backward GEMMs
    hipEvent_t ev_head_ready, ev_head_done;
    bool overlap_pending;       // forward_backward calls dp_after_head() behind the dW_head GEMM
    const umlh_hyper_t* overlap_hy;
    const umlh_batch_t* overlap_img; const umlh_batch_t* overlap_txt;
    int device;                 // HIP device the handle was created on: every launching entry point runs there
    int global_rows_img, global_rows_txt;   // global row counts of the last umlh_grad_step (gate the update on every rank alike)
};

// Every entry point that launches kernels for a handle makes the handle's device current for its duration
// (a process may drive several GPUs; the caller's current device is restored on return).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int want) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != want && hipSetDevice(want) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static void dp_release(umlh_handle_t h);
static int grad_step_impl(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt, const umlh_hyper_t* hy, hipStream_t st,
                          bool comm);
static int apply_update_impl(umlh_handle_t h, const umlh_hyper_t* hy, float* scalars_out, hipStream_t st);

static inline void mark(umlh_handle_t h, int i, hipStream_t st) {
    if (h->profiling) (void)hipEventRecord(h->ev[i], st);
}

const char* umlh_last_error(void) { return g_err; }
int umlh_version(void) { return 4; }   // 3: round 2 (grouped / micro / data-parallel / encoder-plan / InfoNCE entry points, umlh_enc_layer_t.seed_device, umlh_seq_mse_backward scratch); 4: round 3 (umlh_step_status / _launches, umlh_p2p_*)

int umlh_freeze_proj_row(umlh_handle_t h, int32_t row) {
    if (!h) return fail(UMLH_E_INVALID, "umlh_freeze_proj_row: null handle");
    if (row >= 0 && (!h->cfg.has_proj || row >= h->cfg.d_shared || h->cfg.d_img % 4 != 0))
        return fail(UMLH_E_INVALID, "umlh_freeze_proj_row: row %d of a [%d, %d] img_proj (needs img_proj and d_img %% 4 == 0)", row, h->cfg.d_shared, h->cfg.d_img);
    h->frozen_proj_row = row < 0 ? -1 : row;
    return UMLH_OK;
}

int umlh_set_diagnostic_columns(umlh_handle_t h, int32_t cols) {
    if (!h) return fail(UMLH_E_INVALID, "umlh_set_diagnostic_columns: null handle");
    if (cols < 0 || cols > h->cfg.d_shared) return fail(UMLH_E_INVALID, "umlh_set_diagnostic_columns: cols must be in [0, d_shared]");
    h->diag_cols = cols == h->cfg.d_shared ? 0 : cols;
    return UMLH_OK;
}

int umlh_enable_diagnostics(umlh_handle_t h, int32_t on) {
    if (!h) return fail(UMLH_E_INVALID, "umlh_enable_diagnostics: null handle");
    h->diagnostics = on != 0;
    return UMLH_OK;
}

uint64_t umlh_workspace_bytes(const umlh_config_t* cfg) {
    Layout L;
    if (!cfg || !make_layout(*cfg, L)) return 0;
    return (uint64_t)L.total * sizeof(float);
}

int umlh_create(const umlh_config_t* cfg, umlh_handle_t* out) {
    if (!cfg || !out) return fail(UMLH_E_INVALID, "umlh_create: null argument");
    Layout L;
    if (!make_layout(*cfg, L))
        return fail(UMLH_E_INVALID,
                    "umlh_create: unsupported config (d_img=%d d_shared=%d C=%d has_proj=%d opt=%d prec=%d rows=%d/%d)",
                    cfg->d_img, cfg->d_shared, cfg->num_classes, cfg->has_proj, cfg->optimizer, cfg->precision,
                    cfg->max_rows_img, cfg->max_rows_txt);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(UMLH_E_NOGPU, "umlh_create: no HIP device visible");
    umlh_handle_s* h = new (std::nothrow) umlh_handle_s();
    if (!h) return fail(UMLH_E_INVALID, "umlh_create: out of host memory");
    h->cfg = *cfg;
    h->L = L;
    h->bound = false;
    h->ts = umlh_f32_fwd_config(cfg->num_classes, &h->ctw, &h->wc);
    h->stw = 1;
    { const char* d = getenv("UMLH_DBG_FWD"); h->dbg_fwd = d ? atoi(d) : 0; }
    { const char* d = getenv("UMLH_DBG_DW"); h->dbg_dw = d ? atoi(d) : 0; }
    if (cfg->precision == UMLH_PREC_BF16) {
        // two 32-sample tiles per wave would halve the L2->CU stream of the head weight, but measured
        // slower on MI355X (51 vs 26 us at cfg2: half the CUs idle, VGPR-limited ring) -> opt-in only
        const char* e = getenv("UMLH_BF16_STW");
        int want = e ? atoi(e) : 1;
        if (h->wc == 8 && h->ctw >= 2 && want == 2) h->stw = 2;
        h->ts = umlh_bf16_fwd_ts(h->wc, h->stw);
        if (L.fwd_nq) h->ts = 128;
    }
    h->fwd_epoch = 0;
    h->fuse_epoch = 0;
    h->pending_head = nullptr; h->head_fused_done = false;
    { const char* e = getenv("UMLH_DBG_STEP"); h->dbg_step = (e && atoi(e) == 1) ? 1 : 0; }
    h->step_grid = 0; h->step_launches = 0;
    { const char* e = getenv("UMLH_STEP_LAZY"); h->step_lazy = (e && atoi(e) == 1) ? 1 : 0; }
    { const char* e = getenv("UMLH_BF16_FUSE"); h->fuse = e ? atoi(e) : 2; }   // 2 (default): the whole step as one launch; 1: forward + dW as one; 0: separate launches
    h->last_rows_img = h->last_rows_txt = 0;
    h->global_rows_img = h->global_rows_txt = 0;
    h->profiling = false;
    h->micro_epoch = 0;
    h->micro_launches = 0;
    h->frozen_proj_row = -1;
    h->dp_diag = false; h->n_ranks = 1; h->ar_fn = nullptr; h->ar_ctx = nullptr; h->comm = nullptr; h->comm_owned = false;
    h->comm_stream = nullptr; h->overlap_pending = false;
    h->p2p_on = 0; h->p2p_rank = 0; h->p2p_epoch = 0; memset(h->p2p_region, 0, sizeof(h->p2p_region));
    { const char* e = getenv("UMLH_FORCE_DP"); h->dp_force = (e && atoi(e) == 1) ? 1 : 0; }
    h->stage = nullptr; h->stage_bytes = 0; h->stage_next = 0;
    { const char* e = getenv("UMLH_MICRO"); h->micro_off = (e && atoi(e) == 0) ? 1 : 0; }
    h->device = 0;
    (void)hipGetDevice(&h->device);
    h->step_grid = device_cus(h->device);
    { const char* e = getenv("UMLH_STEP_GRID"); if (e && atoi(e) > 0) h->step_grid = atoi(e); }   // fewer workgroups than tasks is always correct (tests: partial residency)
    memset(&h->buf, 0, sizeof(h->buf));
    *out = h;
    return UMLH_OK;
}

int umlh_profile_enable(umlh_handle_t h, int enable) {
    if (!h) return fail(UMLH_E_INVALID, "umlh_profile_enable: null handle");
    DeviceGuard dg_(h->device);
    if (enable && !h->profiling) {
        for (int i = 0; i <= UMLH_N_PHASES; ++i)
            if (hipEventCreate(&h->ev[i]) != hipSuccess) return fail(UMLH_E_HIP, "umlh_profile_enable: hipEventCreate failed");
        h->profiling = true;
    } else if (!enable && h->profiling) {
        for (int i = 0; i <= UMLH_N_PHASES; ++i) (void)hipEventDestroy(h->ev[i]);
        h->profiling = false;
    }
    return UMLH_OK;
}

int umlh_profile_read(umlh_handle_t h, float* ms_out) {
    if (!h || !ms_out) return fail(UMLH_E_INVALID, "umlh_profile_read: null argument");
    if (!h->profiling) return fail(UMLH_E_INVALID, "umlh_profile_read: profiling not enabled");
    DeviceGuard dg_(h->device);
    if (hipEventSynchronize(h->ev[UMLH_N_PHASES]) != hipSuccess) return fail(UMLH_E_HIP, "umlh_profile_read: sync failed");
    for (int i = 0; i < UMLH_N_PHASES; ++i)
        if (hipEventElapsedTime(&ms_out[i], h->ev[i], h->ev[i + 1]) != hipSuccess)
            return fail(UMLH_E_HIP, "umlh_profile_read: elapsed failed (run a step first)");
    return UMLH_OK;
}

int umlh_destroy(umlh_handle_t h) {
    if (h && h->profiling) umlh_profile_enable(h, 0);
    if (h) {
        DeviceGuard dg_(h->device);
        dp_release(h);
        if (h->comm_stream) {
            (void)hipStreamSynchronize(h->comm_stream);
            (void)hipEventDestroy(h->ev_head_ready); (void)hipEventDestroy(h->ev_head_done);
            (void)hipStreamDestroy(h->comm_stream);
        }
    }
    if (h && h->stage) {
        DeviceGuard dg_(h->device);
        (void)hipEventSynchronize(h->stage_ev[0]); (void)hipEventSynchronize(h->stage_ev[1]);
        (void)hipEventDestroy(h->stage_ev[0]); (void)hipEventDestroy(h->stage_ev[1]);
        (void)hipHostFree(h->stage);
    }
    delete h;
    return UMLH_OK;
}

static inline float* ws(umlh_handle_t h, long long off) { return static_cast<float*>(h->buf.workspace) + off; }

// gradient message layout (see umlh_grad_step): head part, img_proj part, then g_scales(2) + scalars
static inline long long frozen_lo(const umlh_handle_s* h) { return h->frozen_proj_row < 0 ? 0 : (long long)h->frozen_proj_row * h->cfg.d_img; }
static inline long long frozen_hi(const umlh_handle_s* h) { return h->frozen_proj_row < 0 ? 0 : (long long)(h->frozen_proj_row + 1) * h->cfg.d_img; }
static inline long long msg_head_len(const umlh_handle_s* h) { return h->dp_diag ? 2 * h->L.n_head : h->L.n_head; }
static inline long long msg_tail_off(const umlh_handle_s* h) { return msg_head_len(h) + h->L.n_proj; }
static inline long long msg_len(const umlh_handle_s* h) { return msg_tail_off(h) + 2 + UMLH_N_SCALARS; }

int umlh_bind(umlh_handle_t h, const umlh_buffers_t* b) {
    if (!h || !b) return fail(UMLH_E_INVALID, "umlh_bind: null argument");
    if (!b->w_head || !b->m_head || !b->workspace) return fail(UMLH_E_INVALID, "umlh_bind: w_head/m_head/workspace required");
    if (h->cfg.optimizer != UMLH_OPT_SGD && !b->v_head) return fail(UMLH_E_INVALID, "umlh_bind: v_head required for adam/adamw");
    if (h->cfg.has_proj && (!b->w_proj || !b->m_proj || (h->cfg.optimizer != UMLH_OPT_SGD && !b->v_proj)))
        return fail(UMLH_E_INVALID, "umlh_bind: img_proj buffers required when has_proj");
    if (!b->scales) return fail(UMLH_E_INVALID, "umlh_bind: scales[2] required");
    if (h->cfg.learnable_temp && (!b->m_scales || !b->v_scales))
        return fail(UMLH_E_INVALID, "umlh_bind: m_scales/v_scales required when learnable_temp");
    if (b->workspace_bytes < (uint64_t)h->L.total * sizeof(float))
        return fail(UMLH_E_UNBOUND, "umlh_bind: workspace too small (%llu < %llu bytes)",
                    (unsigned long long)b->workspace_bytes, (unsigned long long)h->L.total * sizeof(float));
    if ((reinterpret_cast<uintptr_t>(b->workspace) & 255) != 0) return fail(UMLH_E_INVALID, "umlh_bind: workspace must be 256-B aligned");
    h->buf = *b;
    h->bound = true;
    h->iota_ready = false;
    h->shadow_fresh = false;
    {                             // ticket of the gradient-diagnostics reduction (head_step_kernel leaves it at 0 after every launch)
        DeviceGuard dg_(h->device);
        const long long np = 4 * ((h->L.n_head + 1023) / 1024 + 2);
        if (hipMemset(ws(h, h->L.diag_part) + np, 0, 64 * sizeof(float)) != hipSuccess) return fail(UMLH_E_HIP, "umlh_bind: clearing the diagnostics ticket failed");
    }
    if (h->L.fuse_flags) {        // done granules / claim words / status of the one-launch step: tag 0 = never written
        DeviceGuard dg_(h->device);
        if (hipMemset(ws(h, h->L.fuse_flags), 0, sizeof(float) * (size_t)(3 * h->L.ctl_tasks + 16)) != hipSuccess)
            return fail(UMLH_E_HIP, "umlh_bind: clearing the step control words failed");
        h->fuse_epoch = 0;
    }
    if (h->L.fwd_nq) {            // exchange granules of the 2-D forward: tag 0 = never written
        DeviceGuard dg_(h->device);
        if (hipMemset(ws(h, h->L.xch), 0, sizeof(float) * 2 * (size_t)(h->L.ldz / 128 + 2) * h->L.fwd_nq * 4 * 128) != hipSuccess)
            return fail(UMLH_E_HIP, "umlh_bind: clearing the forward exchange region failed");
        h->fwd_epoch = 0;
    }
    if (h->L.mc_flags) {          // epoch flags, status word and exchange records start from zero (bind time only)
        DeviceGuard dg_(h->device);
        const size_t n = (size_t)(h->L.mc_tab - h->L.mc_flags) * sizeof(float);
        if (hipMemset(ws(h, h->L.mc_flags), 0, n) != hipSuccess) return fail(UMLH_E_HIP, "umlh_bind: clearing the micro-step region failed");
        h->micro_epoch = 0;
    }
    return UMLH_OK;
}

// forward blocks of the image segment: in bf16 mode the image columns of dZ^T are padded to whole
// 64-column chunks (every dW chunk then lies in one modality); the padding tile(s) run with all rows
// masked and write zeros
static inline int fwd_blocks_img(const umlh_handle_s* h, int rows) {
    int nb = (rows + h->ts - 1) / h->ts;
    if (h->cfg.precision == UMLH_PREC_BF16 && h->ts < 64 && nb > 0) {
        int per = 64 / h->ts;
        nb = (nb + per - 1) / per * per;
    }
    return nb;
}

static int check_batch(umlh_handle_t h, const umlh_batch_t* b, int cap, const char* who) {
    if (!b) return UMLH_OK;
    if (b->rows < 0 || b->rows > cap) return fail(UMLH_E_INVALID, "%s: rows=%d outside [0,%d]", who, b->rows, cap);
    if (b->rows > 0 && (!b->feats || !b->labels)) return fail(UMLH_E_INVALID, "%s: feats/labels null", who);
    if (b->rows > 0 && b->global_rows < b->rows) return fail(UMLH_E_INVALID, "%s: global_rows=%d < rows=%d", who, b->global_rows, b->rows);
    return UMLH_OK;
}

static OptArgs make_opt(const umlh_config_t& c, const umlh_hyper_t& hy) {
    OptArgs o;
    memset(&o, 0, sizeof(o));
    o.plain = umlh_plain_stores();
    o.kind = c.optimizer;
    o.lr = (float)hy.lr;
    o.decay = (float)(1.0 - hy.lr * c.weight_decay);
    double t = (double)(hy.step < 1 ? 1 : hy.step);
    double bc1 = 1.0 - std::pow(c.beta1, t);
    double bc2 = 1.0 - std::pow(c.beta2, t);
    o.neg_step_size = (float)(-(hy.lr / bc1));
    o.bc2_sqrt = (float)std::sqrt(bc2);
    o.beta1 = (float)c.beta1;
    o.one_m_beta1 = (float)(1.0 - c.beta1);
    o.beta2 = (float)c.beta2;
    o.one_m_beta2 = (float)(1.0 - c.beta2);
    o.eps = (float)c.eps;
    o.momentum = (float)c.momentum;
    o.wd = (float)c.weight_decay;
    return o;
}

#define HIPCHK(expr, what)                                                                  \
    do {                                                                                    \
        int _e = (expr);                                                                    \
        if (_e != 0) return fail(UMLH_E_HIP, "%s: HIP error %d (%s)", what, _e, hipGetErrorString((hipError_t)_e)); \
    } while (0)

int umlh_zero_shot_init(umlh_handle_t h, const float* text_feats, const int64_t* text_labels, int64_t n_text,
                        void* stream) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_zero_shot_init: handle not bound");
    if (!text_feats || !text_labels || n_text < 0) return fail(UMLH_E_INVALID, "umlh_zero_shot_init: bad arguments");
    DeviceGuard dg_(h->device);
    HIPCHK(umlh_launch_zero_shot(text_feats, text_labels, n_text, h->cfg.d_shared, h->cfg.num_classes,
                                 h->buf.w_head, (hipStream_t)stream), "zero_shot");
    return UMLH_OK;
}

// H = X_img[index] W_proj^T into the workspace (head.py:79)
static int launch_proj_forward(umlh_handle_t h, const umlh_batch_t* img, float* H, hipStream_t st) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = img->feats; g.a_rows = img->index; g.lda = h->cfg.d_img;
    g.B = h->buf.w_proj; g.ldb = h->cfg.d_img;
    g.out = H; g.ldo = h->cfg.d_shared;
    g.M = img->rows; g.N = h->cfg.d_shared; g.K = h->cfg.d_img;
    g.k_chunk = g.K; g.slab_stride = 0; g.alpha = 1.f;
    g.k_switch = INT_MAX; g.k_valid1 = g.K; g.k_valid2 = 0;
    return umlh_f32_launch_gemm(&g, 0, 0, 1, st);
}

int umlh_logits(umlh_handle_t h, const umlh_batch_t* b, int modality, float* out, void* stream) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_logits: handle not bound");
    if (!b || !out || (modality != 0 && modality != 1)) return fail(UMLH_E_INVALID, "umlh_logits: bad arguments");
    DeviceGuard dg_(h->device);
    int rc = check_batch(h, b, modality == 0 ? h->L.rcap_img : h->L.rcap_txt, "umlh_logits");
    if (rc) return rc;
    if (b->rows == 0) return UMLH_OK;
    hipStream_t st = (hipStream_t)stream;
    const float* F = b->feats;
    const int64_t* idx = b->index;
    int ld = h->cfg.d_shared;
    if (modality == 0 && h->cfg.has_proj) {
        HIPCHK(launch_proj_forward(h, b, ws(h, h->L.h), st), "proj forward");
        F = ws(h, h->L.h);
        idx = nullptr;
    }
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = F; g.a_rows = idx; g.lda = ld;
    g.B = h->buf.w_head; g.ldb = h->cfg.d_shared;
    g.out = out; g.ldo = h->cfg.num_classes;
    g.M = b->rows; g.N = h->cfg.num_classes; g.K = h->cfg.d_shared;
    g.k_chunk = g.K; g.alpha = 1.f;
    g.k_switch = INT_MAX; g.k_valid1 = g.K;
    g.alpha_ptr = h->buf.scales + modality;             // device-resident logit scale
    HIPCHK(umlh_f32_launch_gemm(&g, 0, 0, 1, st), "logits gemm");
    return UMLH_OK;
}

int umlh_project(umlh_handle_t h, const umlh_batch_t* b, float* out, void* stream) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_project: handle not bound");
    if (!b || !out) return fail(UMLH_E_INVALID, "umlh_project: null argument");
    if (!h->cfg.has_proj) return fail(UMLH_E_INVALID, "umlh_project: head has no img_proj");
    DeviceGuard dg_(h->device);
    int rc = check_batch(h, b, h->L.rcap_img, "umlh_project");
    if (rc) return rc;
    if (b->rows == 0) return UMLH_OK;
    HIPCHK(launch_proj_forward(h, b, out, (hipStream_t)stream), "proj forward");
    return UMLH_OK;
}

int umlh_optimizer_step(int32_t optimizer, float* param, const float* grad, float* m, float* v, int64_t n, double lr,
                        int64_t step, double beta1, double beta2, double eps, double momentum, double weight_decay,
                        void* stream) {
    if (optimizer < UMLH_OPT_SGD || optimizer > UMLH_OPT_ADAMW)
        return fail(UMLH_E_INVALID, "umlh_optimizer_step: unknown optimizer %d", optimizer);
    if (!param || !grad || !m || (optimizer != UMLH_OPT_SGD && !v) || n < 0)
        return fail(UMLH_E_INVALID, "umlh_optimizer_step: null buffer");
    umlh_config_t c;
    memset(&c, 0, sizeof(c));
    c.optimizer = optimizer; c.beta1 = beta1; c.beta2 = beta2; c.eps = eps; c.momentum = momentum;
    c.weight_decay = weight_decay;
    umlh_hyper_t hy;
    memset(&hy, 0, sizeof(hy));
    hy.lr = lr; hy.step = step;
    OptArgs o = make_opt(c, hy);
    HIPCHK(umlh_launch_reduce_update(1, grad, 1, n, n, nullptr, param, m, v, &o, 0, 0, (hipStream_t)stream),
           "optimizer step");
    return UMLH_OK;
}

extern "C" int umlh_launch_multi_opt(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const long long* cnt,
                                     const OptArgs* o, hipStream_t stream);

int umlh_optimizer_step_multi(int32_t optimizer, int32_t n_tensors, float* const* params, const float* const* grads, float* const* m,
                              float* const* v, const int64_t* n, double lr, int64_t step, double beta1, double beta2, double eps,
                              double momentum, double weight_decay, void* stream) {
    if (optimizer < UMLH_OPT_SGD || optimizer > UMLH_OPT_ADAMW) return fail(UMLH_E_INVALID, "umlh_optimizer_step_multi: unknown optimizer %d", optimizer);
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !m || !n || (optimizer != UMLH_OPT_SGD && !v))))
        return fail(UMLH_E_INVALID, "umlh_optimizer_step_multi: null argument");
    umlh_config_t c;
    memset(&c, 0, sizeof(c));
    c.optimizer = optimizer; c.beta1 = beta1; c.beta2 = beta2; c.eps = eps; c.momentum = momentum; c.weight_decay = weight_decay;
    umlh_hyper_t hy;
    memset(&hy, 0, sizeof(hy));
    hy.lr = lr; hy.step = step;
    OptArgs o = make_opt(c, hy);
    for (int t0 = 0; t0 < n_tensors; t0 += UMLH_MULTI_OPT_MAX) {
        const int k = n_tensors - t0 < UMLH_MULTI_OPT_MAX ? n_tensors - t0 : UMLH_MULTI_OPT_MAX;
        long long cnt[UMLH_MULTI_OPT_MAX];
        for (int i = 0; i < k; ++i) {
            if (!params[t0 + i] || !grads[t0 + i] || !m[t0 + i] || n[t0 + i] < 0 || (optimizer != UMLH_OPT_SGD && !v[t0 + i]))
                return fail(UMLH_E_INVALID, "umlh_optimizer_step_multi: tensor %d has a null buffer", t0 + i);
            cnt[i] = n[t0 + i];
        }
        HIPCHK(umlh_launch_multi_opt(k, params + t0, grads + t0, m + t0, v ? v + t0 : nullptr, cnt, &o, (hipStream_t)stream), "optimizer step (multi)");
    }
    return UMLH_OK;
}

extern "C" {
int umlh_seq_launch_fwd(const float* z, const float* w, const float* b, const float* x, const int64_t* lengths, int B, int T,
                        int Z, int D, float* recon, float* dres, float* row_partial, float* loss_cnt, hipStream_t st);
int umlh_seq_launch_bwd(const float* z, const float* w, const float* dres, const float* loss_cnt, const float* grad_out, int B,
                        int T, int Z, int D, float* dz, float* dw, float* db, int with_params, hipStream_t st);
int umlh_seq_launch_l2norm(const float* x, int n, int D, float* y, float* norm, hipStream_t st);
int umlh_seq_launch_nce_rows(float* dots, int n, float inv_temp, float* row_loss, float* loss, hipStream_t st);
int umlh_seq_launch_l2norm_bwd(const float* dy, const float* y, const float* norm, const float* grad_out, float scale, int n, int D,
                               float* dx, hipStream_t st);
}

int umlh_seq_mse_forward(const float* z, const float* w, const float* bias, const float* x, const int64_t* lengths, int32_t B,
                         int32_t T, int32_t Z, int32_t D, float* recon, float* dres, float* row_partial, float* loss_cnt,
                         void* stream) {
    if (!z || !w || !bias || !x || !dres || !row_partial || !loss_cnt) return fail(UMLH_E_INVALID, "umlh_seq_mse_forward: null buffer");
    if (B < 1 || T < 1 || Z < 1 || D < 1 || Z > 8192) return fail(UMLH_E_INVALID, "umlh_seq_mse_forward: bad shape B=%d T=%d Z=%d D=%d", B, T, Z, D);
    HIPCHK(umlh_seq_launch_fwd(z, w, bias, x, lengths, B, T, Z, D, recon, dres, row_partial, loss_cnt, (hipStream_t)stream), "seq fwd");
    return UMLH_OK;
}

// split-K factor / row chunking of the decoder gradient: dW [D,Z] over B*T rows (same rule as the encoder layers)
static int seq_splits(int D, int Z, int R) {
    const long long tiles = (long long)((D + 63) / 64) * ((Z + 63) / 64);
    long long s = (768 + tiles - 1) / tiles;
    if (s > R / 64) s = R / 64;
    if (s > 32) s = 32;
    return (int)(s < 1 ? 1 : s);
}
static int seq_row_chunk(int R) { int c = (R + 63) / 64; return c < 64 ? 64 : c; }

uint64_t umlh_seq_mse_backward_scratch_floats(int32_t B, int32_t T, int32_t Z, int32_t D) {
    if (B < 1 || T < 1 || Z < 1 || D < 1) return 0;
    const int R = B * T, chunk = seq_row_chunk(R);
    return (uint64_t)seq_splits(D, Z, R) * D * Z + (uint64_t)((R + chunk - 1) / chunk) * D + 64;
}

int umlh_seq_mse_backward(const float* z, const float* w, const float* dres, const float* loss_cnt, const float* grad_out,
                          int32_t B, int32_t T, int32_t Z, int32_t D, float* dz, float* dw, float* db, float* scratch, void* stream) {
    if (!z || !w || !dres || !loss_cnt || !grad_out || !dz || !dw || !db) return fail(UMLH_E_INVALID, "umlh_seq_mse_backward: null buffer");
    if (B < 1 || T < 1 || Z < 1 || D < 1 || D > 8192) return fail(UMLH_E_INVALID, "umlh_seq_mse_backward: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int R = B * T, sp = seq_splits(D, Z, R);
    if (!scratch || sp == 1) {
        HIPCHK(umlh_seq_launch_bwd(z, w, dres, loss_cnt, grad_out, B, T, Z, D, dz, dw, db, 1, st), "seq bwd");
        return UMLH_OK;
    }
    // dz as before; dW[d][k] = s * sum_r dres[r][d] z[r][k] as split-K slabs, db[d] = s * sum_r dres[r][d] as row-chunk partials,
    // both summed and scaled (s = 2 * grad_out / denominator, device-side) by one multi-reduce
    HIPCHK(umlh_seq_launch_bwd(z, w, dres, loss_cnt, grad_out, B, T, Z, D, dz, dw, db, 0, st), "seq bwd (dz)");
    const int chunk = seq_row_chunk(R), nr = (R + chunk - 1) / chunk;
    float* slabs = scratch;
    float* part = scratch + (size_t)sp * D * Z;
    int ns = 1;
    int rc = umlh_gemm_f32_epi(dres, z, nullptr, D, Z, R, D, Z, 1, 1, nullptr, sp, slabs, 1, &ns, st);
    if (rc) return rc;
    HIPCHK(umlh_enc_launch_colsum_partial(dres, R, D, chunk, part, st), "seq bwd (db partials)");
    MultiReduceArgs red;
    memset(&red, 0, sizeof(red));
    red.d[0] = ReduceDesc{slabs, dw, (long long)D * Z, (long long)D * Z, ns, 0};
    red.d[1] = ReduceDesc{part, db, (long long)D, (long long)D, nr, 0};
    red.count = 2;
    red.s_num = grad_out; red.s_den = loss_cnt + 1; red.s_mul = 2.f;
    HIPCHK(umlh_enc_launch_multi_reduce(&red, st), "seq bwd (reduce)");
    return UMLH_OK;
}

// SequenceInfoNCELoss (MultiBench/models.py:145-175) over n valid rows
int umlh_infonce_forward(const float* pred, const float* target, int32_t n, int32_t D, float temperature, float* pred_hat,
                         float* target_hat, float* pred_norm, float* probs, float* row_loss, float* loss, void* stream) {
    if (!pred || !target || !pred_hat || !target_hat || !pred_norm || !probs || !row_loss || !loss)
        return fail(UMLH_E_INVALID, "umlh_infonce_forward: null buffer");
    if (n < 1 || D < 1 || !(temperature > 0.f)) return fail(UMLH_E_INVALID, "umlh_infonce_forward: bad shape (n=%d D=%d)", n, D);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(umlh_seq_launch_l2norm(pred, n, D, pred_hat, pred_norm, st), "infonce l2norm(pred)");
    HIPCHK(umlh_seq_launch_l2norm(target, n, D, target_hat, row_loss, st), "infonce l2norm(target)");   // (row_loss: scratch for the unused norms)
    int ns = 1;
    const int rc = umlh_gemm_f32_epi(pred_hat, target_hat, probs, n, n, D, D, D, 0, 0, nullptr, 1, nullptr, 0, &ns, st);
    if (rc) return rc;
    HIPCHK(umlh_seq_launch_nce_rows(probs, n, 1.f / temperature, row_loss, loss, st), "infonce rows");
    return UMLH_OK;
}

int umlh_infonce_backward(const float* pred_hat, const float* target_hat, const float* pred_norm, const float* probs,
                          const float* grad_out, int32_t n, int32_t D, float temperature, float* dhat, float* dpred, void* stream) {
    if (!pred_hat || !target_hat || !pred_norm || !probs || !grad_out || !dhat || !dpred)
        return fail(UMLH_E_INVALID, "umlh_infonce_backward: null buffer");
    if (n < 1 || D < 1 || !(temperature > 0.f)) return fail(UMLH_E_INVALID, "umlh_infonce_backward: bad shape");
    hipStream_t st = (hipStream_t)stream;
    int ns = 1;
    const int rc = umlh_gemm_f32_epi(probs, target_hat, dhat, n, D, n, n, D, 0, 1, nullptr, 1, nullptr, 0, &ns, st);   // dhat = (softmax - I) target_hat
    if (rc) return rc;
    HIPCHK(umlh_seq_launch_l2norm_bwd(dhat, pred_hat, pred_norm, grad_out, 1.f / ((float)n * temperature), n, D, dpred, st), "infonce l2norm bwd");
    return UMLH_OK;
}

extern "C" int umlh_launch_feistel_perm(long long n, unsigned long long seed, long long* out, hipStream_t stream);

int umlh_random_permutation(int64_t n, uint64_t seed, int64_t* out, void* stream) {
    if (n < 0 || (n > 0 && !out)) return fail(UMLH_E_INVALID, "umlh_random_permutation: bad arguments");
    HIPCHK(umlh_launch_feistel_perm(n, seed, reinterpret_cast<long long*>(out), (hipStream_t)stream), "feistel perm");
    return UMLH_OK;
}

int umlh_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n < 0) return fail(UMLH_E_INVALID, "umlh_to_bf16: bad arguments");
    HIPCHK(umlh_launch_to_bf16(src, dst, n, (hipStream_t)stream), "to_bf16");
    return UMLH_OK;
}

// ---- MultiBench shared encoder ops (kernels: umlh_kernels_enc.hip, GEMMs: gemm_f32) ----
int umlh_gemm_f32(const float* A, const float* B, float* out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                  int32_t ldo, int32_t ta, int32_t tb, const int64_t* a_rows, const int64_t* k_rows, float alpha,
                  int32_t splits, float* slabs, void* stream) {
    if (!A || !B || !out || M < 0 || N < 0 || K < 1 || (ta && a_rows) || (!tb && k_rows) || splits < 1 || (splits > 1 && !slabs))
        return fail(UMLH_E_INVALID, "umlh_gemm_f32: bad arguments");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.a_rows = a_rows; g.k_rows = k_rows;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldo = ldo;
    g.alpha = alpha; g.k_switch = K; g.k_valid1 = K;
    // dense operands and a short K range per workgroup: the latency-oriented kernel (see gemm_enc)
    const bool dense = !a_rows && !k_rows && !(ta == 1 && tb == 0) && (K + splits - 1) / splits <= 512;
    auto launch = dense ? umlh_f32_launch_gemm_enc : umlh_f32_launch_gemm;
    if (splits == 1) {
        g.out = out; g.k_chunk = K; g.slab_stride = 0;
        HIPCHK(launch(&g, ta, tb, 1, (hipStream_t)stream), "gemm_f32");
        return UMLH_OK;
    }
    const int chunk = (int)round_up((K + splits - 1) / splits, KT);
    const int ns = (K + chunk - 1) / chunk;
    g.out = slabs; g.k_chunk = chunk; g.slab_stride = (long long)M * ldo;
    HIPCHK(launch(&g, ta, tb, ns, (hipStream_t)stream), "gemm_f32 (split-K)");
    const long long n = (long long)M * ldo;
    Epilogue none;
    memset(&none, 0, sizeof(none));
    HIPCHK(umlh_enc_launch_reduce_epilogue(slabs, ns, n, n, N, &none, out, (hipStream_t)stream), "split-K reduce");
    return UMLH_OK;
}

int umlh_gemm_f32_epi(const float* A, const float* B, float* out, int M, int N, int K, int lda, int ldb, int ta, int tb,
                      const Epilogue* epi, int splits, float* slabs, int defer, int* ns_out, hipStream_t stream) {
    if (!A || !B || M < 1 || N < 1 || K < 1 || splits < 1 || ((defer || splits > 1) && !slabs) || (!defer && !out))
        return fail(UMLH_E_INVALID, "gemm_f32 (epilogue): bad arguments");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldo = N;
    g.alpha = 1.f; g.k_switch = K; g.k_valid1 = K;
    const int chunk = splits == 1 ? K : (int)round_up((K + splits - 1) / splits, KT);
    const int ns = (K + chunk - 1) / chunk;
    g.k_chunk = chunk;
    g.slab_stride = (long long)M * N;
    if (ns_out) *ns_out = ns;
    static const bool legacy = [] { const char* e = getenv("UMLH_ENC_GEMM"); return e && atoi(e) == 0; }();   // timing comparisons
    auto launch = legacy ? umlh_f32_launch_gemm : umlh_f32_launch_gemm_enc;
    if (!defer && ns == 1) {
        g.out = out;
        if (epi) g.epi = *epi;
        HIPCHK(launch(&g, ta, tb, 1, stream), "gemm_enc (epilogue)");
        return UMLH_OK;
    }
    g.out = slabs;
    HIPCHK(launch(&g, ta, tb, ns, stream), "gemm_enc (slabs)");
    if (defer) return UMLH_OK;
    Epilogue none;
    memset(&none, 0, sizeof(none));
    HIPCHK(umlh_enc_launch_reduce_epilogue(slabs, ns, g.slab_stride, g.slab_stride, N, epi ? epi : &none, out, stream), "split-K reduce (epilogue)");
    return UMLH_OK;
}

int umlh_bias_act(float* y, const float* bias, int64_t M, int32_t N, int32_t relu, void* stream) {
    if (!y || M < 0 || N < 1) return fail(UMLH_E_INVALID, "umlh_bias_act: bad arguments");
    HIPCHK(umlh_enc_launch_bias_act(y, bias, M, N, relu, (hipStream_t)stream), "bias_act");
    return UMLH_OK;
}

int umlh_relu_backward(const float* y, float* dy, int64_t n, void* stream) {
    if (!y || !dy || n < 0) return fail(UMLH_E_INVALID, "umlh_relu_backward: bad arguments");
    HIPCHK(umlh_enc_launch_relu_bwd(y, dy, n, (hipStream_t)stream), "relu_bwd");
    return UMLH_OK;
}

int umlh_dropout(float* x, int64_t n, float p, uint64_t seed, void* stream) {
    if (!x || n < 0 || !(p >= 0.f && p < 1.f)) return fail(UMLH_E_INVALID, "umlh_dropout: bad arguments");
    HIPCHK(umlh_enc_launch_dropout(x, n, p, seed, (hipStream_t)stream), "dropout");
    return UMLH_OK;
}

int umlh_add_inplace(float* y, const float* x, int64_t n, void* stream) {
    if (!y || !x || n < 0) return fail(UMLH_E_INVALID, "umlh_add_inplace: bad arguments");
    HIPCHK(umlh_enc_launch_add_inplace(y, x, n, (hipStream_t)stream), "add_inplace");
    return UMLH_OK;
}

int umlh_colsum(const float* x, int32_t M, int32_t N, float* out, void* stream) {
    if (!x || !out || M < 0 || N < 1) return fail(UMLH_E_INVALID, "umlh_colsum: bad arguments");
    HIPCHK(umlh_enc_launch_colsum(x, M, N, out, (hipStream_t)stream), "colsum");
    return UMLH_OK;
}

int umlh_add_layernorm_forward(const float* x, const float* r, const float* gamma, const float* beta, int32_t M, int32_t N,
                               float eps, float* s, float* y, float* mean, float* rstd, void* stream) {
    if (!x || !gamma || !beta || !s || !y || !mean || !rstd || M < 0 || N < 1)
        return fail(UMLH_E_INVALID, "umlh_add_layernorm_forward: bad arguments");
    HIPCHK(umlh_enc_launch_add_layernorm(x, r, gamma, beta, M, N, eps, s, y, mean, rstd, (hipStream_t)stream), "add_layernorm");
    return UMLH_OK;
}

int umlh_layernorm_backward(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                            int32_t M, int32_t N, float* ds, float* dgamma, float* dbeta, void* stream) {
    if (!dy || !s || !gamma || !mean || !rstd || !ds || !dgamma || !dbeta || M < 0 || N < 1)
        return fail(UMLH_E_INVALID, "umlh_layernorm_backward: bad arguments");
    HIPCHK(umlh_enc_launch_layernorm_bwd(dy, s, gamma, mean, rstd, M, N, ds, dgamma, dbeta, (hipStream_t)stream), "layernorm_bwd");
    return UMLH_OK;
}

int umlh_add_positions(float* x, const float* pos, int32_t T, int32_t B, int32_t Z, void* stream) {
    if (!x || !pos || T < 1 || B < 1 || Z < 1) return fail(UMLH_E_INVALID, "umlh_add_positions: bad arguments");
    HIPCHK(umlh_enc_launch_add_pos(x, pos, T, B, Z, (hipStream_t)stream), "add_pos");
    return UMLH_OK;
}

int umlh_positions_backward(const float* dx, int32_t T, int32_t B, int32_t Z, float* dpos, void* stream) {
    if (!dx || !dpos || T < 1 || B < 1 || Z < 1) return fail(UMLH_E_INVALID, "umlh_positions_backward: bad arguments");
    HIPCHK(umlh_enc_launch_pos_grad(dx, T, B, Z, dpos, (hipStream_t)stream), "pos_grad");
    return UMLH_OK;
}

int umlh_gather_rows(const float* x, const int64_t* idx, int32_t n, int32_t Z, float* out, int32_t scatter, void* stream) {
    if (!x || !idx || !out || n < 0 || Z < 1) return fail(UMLH_E_INVALID, "umlh_gather_rows: bad arguments");
    HIPCHK(umlh_enc_launch_gather_rows(x, idx, n, Z, out, scatter, (hipStream_t)stream), "gather_rows");
    return UMLH_OK;
}

int umlh_attention_forward(const float* qkv, const int64_t* lengths, int32_t T, int32_t B, int32_t Z, int32_t H, float p,
                           uint64_t seed, float* ctx, float* lse, void* stream) {
    if (!qkv || !ctx || !lse || B < 1 || !(p >= 0.f && p < 1.f)) return fail(UMLH_E_INVALID, "umlh_attention_forward: bad arguments");
    if (T < 1 || T > 128 || H < 1 || Z % H != 0 || Z / H > 64)
        return fail(UMLH_E_INVALID, "umlh_attention_forward: T=%d Z=%d H=%d outside the kernel's envelope (T <= 128, Z/H <= 64)", T, Z, H);
    HIPCHK(umlh_enc_launch_attention_fwd(qkv, lengths, T, B, Z, H, p, seed, nullptr, ctx, lse, (hipStream_t)stream), "attention_fwd");
    return UMLH_OK;
}

int umlh_attention_backward(const float* qkv, const int64_t* lengths, const float* lse, const float* dctx, int32_t T, int32_t B,
                            int32_t Z, int32_t H, float p, uint64_t seed, float* dqkv, void* stream) {
    if (!qkv || !lse || !dctx || !dqkv || B < 1 || !(p >= 0.f && p < 1.f))
        return fail(UMLH_E_INVALID, "umlh_attention_backward: bad arguments");
    if (T < 1 || T > 128 || H < 1 || Z % H != 0 || Z / H > 64)
        return fail(UMLH_E_INVALID, "umlh_attention_backward: T=%d Z=%d H=%d outside the kernel's envelope", T, Z, H);
    HIPCHK(umlh_enc_launch_attention_bwd(qkv, lengths, lse, dctx, T, B, Z, H, p, seed, nullptr, dqkv, (hipStream_t)stream), "attention_bwd");
    return UMLH_OK;
}

static int dp_after_head(umlh_handle_t h, int n_slabs_head, hipStream_t st);

// Everything of a step up to (not including) the parameter update.
static int forward_backward(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt,
                            const umlh_hyper_t* hy, bool want_grad, hipStream_t st, int* n_slabs_head,
                            int* n_slabs_proj) {
    const umlh_config_t& c = h->cfg;
    const Layout& L = h->L;
    const int ri = img ? img->rows : 0, rt = txt ? txt->rows : 0;
    const int TS = h->ts;
    const int nb0 = fwd_blocks_img(h, ri), nb1 = (rt + TS - 1) / TS;
    const int r0p = nb0 * TS, r1p = nb1 * TS;
    float* H = ws(h, L.h);
    float* dzt = ws(h, L.dzt);

    if (c.precision == UMLH_PREC_BF16) {
        if ((ri > 0 && !img->feats_bf16) || (rt > 0 && !txt->feats_bf16))
            return fail(UMLH_E_INVALID, "bf16 mode: batch.feats_bf16 is required (umlh_to_bf16 of the feature table)");
        u16* w16 = reinterpret_cast<u16*>(ws(h, L.w16));
        u16* dz16 = reinterpret_cast<u16*>(dzt);
        u16* h16 = reinterpret_cast<u16*>(H);                                  // bf16 H = X W_proj^T, row-major [r][d_shared]
        u16* dht16 = reinterpret_cast<u16*>(ws(h, L.dht));                     // bf16 dH^T, chunk-major [r/64][d_shared^128][64]
        const u16* zeros16 = reinterpret_cast<const u16*>(ws(h, L.zeros));
        const int dsp = (int)round_up(c.d_shared, 128);
        const bool proj = c.has_proj && ri > 0;
        mark(h, 0, st);
        // the GEMM loaders are branch-free: every pointer must be dereferenceable, also for an absent
        // modality or a dense (index-less) batch -> identity row ids + a zero page from the workspace
        int64_t* iota = reinterpret_cast<int64_t*>(ws(h, L.iota));
        if (!h->iota_ready) {
            HIPCHK(umlh_launch_iota(reinterpret_cast<long long*>(iota), L.n_iota, st), "iota");
            HIPCHK((int)hipMemsetAsync(ws(h, L.zeros), 0, 64 * sizeof(float), st), "zero page");
            h->iota_ready = true;
        }
        auto base_args = [&]() {
            DwArgsB g;
            memset(&g, 0, sizeof(g));
            g.zeros = zeros16; g.bcs = 64; g.a_rows = iota; g.nsplit = g.nsplit1 = 1;
            g.dbg = h->dbg_dw;
            if (g.dbg >= 16) { g.dbg -= 16; g.stamps = reinterpret_cast<unsigned long long*>(ws(h, L.dbg)); }   // +16: cycle stamps
            return g;
        };
        if (proj) {
            // H = X_img W_proj^T  (head.py:79): rows gathered by the batch index, bf16 out.  W_proj^T is a
            // per-step bf16 shadow of the fp32 master (6.5 MB at cfg3, against ~180 GFLOP of GEMMs per step).
            u16* wpt16 = reinterpret_cast<u16*>(ws(h, L.wpt16));
            HIPCHK(umlh_bf16_launch_transpose_shadow(h->buf.w_proj, c.d_shared, c.d_img, c.d_shared, wpt16, 0, st), "W_proj^T shadow");
            DwArgsB g = base_args();
            g.A = static_cast<const u16*>(img->feats_bf16); g.lda = c.d_img; g.a_rows = img->index ? img->index : iota;
            g.B = g.B2 = wpt16; g.k_rows = g.k_rows2 = iota; g.ldb = g.ldb2 = c.d_shared;
            g.M = ri; g.N = c.d_shared; g.K = c.d_img;
            g.k_chunk = (int)round_up(c.d_img, 256); g.k_switch = c.d_img; g.k_valid1 = c.d_img; g.k_valid2 = 0;
            g.out16 = h16; g.ldo = c.d_shared;
            if (g.k_chunk > 4096) return fail(UMLH_E_INVALID, "bf16 img_proj: d_img %d > 4096 unsupported", c.d_img);
            HIPCHK(umlh_bf16_launch_dw(&g, 1, 1, 1, st), "proj forward (bf16)");
        }
        // bf16 shadow of the fp32 master weight, refreshed every call (the caller may have
        // rewritten w_head: load_state_dict, zero-shot init)
        const int cpad = 32 * h->ctw * h->wc;
        const int crows = (int)round_up(c.num_classes, 128);
        // refreshed here unless the previous step of the same umlh_train_steps call just wrote it
        // from its update kernel (between calls the caller may have rewritten w_head)
        if (!h->shadow_fresh)
            HIPCHK(umlh_launch_w_shadow(h->buf.w_head, w16, c.num_classes, c.d_shared, cpad, st), "w_shadow");
        h->shadow_fresh = false;
        mark(h, 1, st);
        FwdArgsB fb;
        memset(&fb, 0, sizeof(fb));
        SegDescB& b0 = fb.seg[0];
        SegDescB& b1 = fb.seg[1];
        if (ri > 0) {
            b0.feats = proj ? h16 : static_cast<const u16*>(img->feats_bf16);
            b0.feat_index = proj ? nullptr : img->index;
            b0.labels = img->labels; b0.label_index = img->index; b0.ld = c.d_shared; b0.rows = ri;
            b0.w_over_rows = hy->img_alpha / (float)img->global_rows;
        }
        b0.scale_ptr = h->buf.scales; b0.col0 = 0; b0.blk0 = 0;
        if (rt > 0) {
            b1.feats = static_cast<const u16*>(txt->feats_bf16); b1.feat_index = txt->index;
            b1.labels = txt->labels; b1.label_index = txt->index; b1.ld = c.d_shared; b1.rows = rt;
            b1.w_over_rows = hy->alpha / (float)txt->global_rows;
        }
        b1.scale_ptr = h->buf.scales + 1; b1.col0 = r0p; b1.blk0 = nb0;
        fb.W = w16; fb.C = c.num_classes; fb.K = c.d_shared;
        fb.dzt = want_grad ? dz16 : nullptr; fb.crows = crows;
        fb.partials = ws(h, L.partials);
        fb.dbg = h->dbg_fwd;
        fb.learn = c.learnable_temp;
        fb.row_stats = h->row_stats;
        fb.stamps = (fb.dbg == 9 || fb.dbg >= 20) ? reinterpret_cast<unsigned long long*>(ws(h, L.dbg)) : nullptr;
        if (L.fwd_nq) {
            fb.xch = reinterpret_cast<unsigned long long*>(ws(h, L.xch));
            if (++h->fwd_epoch == 0) h->fwd_epoch = 1;
            fb.epoch = h->fwd_epoch;
            fb.wtiles = cpad / 32;
            fb.ntiles = nb0 + nb1;
            HIPCHK(umlh_bf16_launch_fwd_q(&fb, L.fwd_nq, nb0 + nb1, st), "fwd_ce_bf16_q");
        }
        // forward + dW as one launch (linear head, 1-D forward, write-through stores).  In profiling mode the interval
        // mark 1 -> 2 is then empty and mark 2 -> 3 holds the one launch.
        // (not while the stream is being captured into a HIP graph: the granules' epoch tag is a launch argument, a replay
        // would find the previous replay's tags and pass its gates early)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool capturing = h->fuse && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
        // (the coherent loads address dZ^T and the slabs through 32-bit buffer offsets)
        const bool fits32 = (unsigned long long)((r0p + r1p + 63) / 64) * crows * 128ull < (1ull << 31) &&
                            (unsigned long long)L.scap_head * L.n_head * 4ull < (1ull << 31);
        const bool fused = h->fuse && !capturing && fits32 && want_grad && !proj && !L.fwd_nq && h->stw == 1 && L.fuse_flags &&
                           !umlh_plain_stores() && fb.dbg == 0 && h->dbg_dw == 0;
        if (!L.fwd_nq && !fused)
        HIPCHK(umlh_bf16_launch_fwd(&fb, h->ctw, h->wc, h->stw, nb0 + nb1, st), "fwd_ce_bf16");
        mark(h, 2, st);
        *n_slabs_head = 0; *n_slabs_proj = 0;
        if (!want_grad) return UMLH_OK;
        // ---- dW_head = dZ^T [H or X_img ; X_txt] ----
        {
            // chunk: multiples of 4 x 64 columns (4-stage pipeline), at most 4096 (row ids of a split live in LDS)
            SplitPlan sp = plan_splits(r0p, r1p, L.scap_head, 256, 256);
            if (sp.chunk > 4096) { sp.chunk = 4096; sp.n_img = (r0p + 4095) / 4096; sp.n_txt = (r1p + 4095) / 4096; }
            const int splits = sp.n_img + sp.n_txt;
            if (splits > L.scap_head)
                return fail(UMLH_E_INVALID, "bf16 dW: %d + %d reduction rows need more than %d split-K slabs", r0p, r1p, L.scap_head);
            const u16* any16 = ri > 0 ? static_cast<const u16*>(img->feats_bf16) : static_cast<const u16*>(txt->feats_bf16);
            DwArgsB g = base_args();
            g.A = dz16; g.lda = crows;
            g.B = proj ? h16 : (ri > 0 ? static_cast<const u16*>(img->feats_bf16) : any16);
            g.k_rows = (ri > 0 && img->index && !proj) ? img->index : iota; g.ldb = c.d_shared;
            g.B2 = rt > 0 ? static_cast<const u16*>(txt->feats_bf16) : any16;
            g.k_rows2 = (rt > 0 && txt->index) ? txt->index : iota; g.ldb2 = c.d_shared;
            g.out = ws(h, L.slabs_head); g.ldo = c.d_shared;
            g.M = c.num_classes; g.N = c.d_shared; g.K = r0p + r1p;
            g.k_chunk = sp.chunk; g.k_switch = r0p; g.k_valid1 = ri; g.k_valid2 = rt; g.slab_stride = L.n_head;
            g.nsplit = splits; g.nsplit1 = sp.n_img; h->n_slabs_img = sp.n_img;
            const int ndw_blocks = ((c.d_shared + 127) / 128) * ((c.num_classes + 127) / 128) * splits;
            const bool with_head = fused && h->pending_head && h->fuse >= 2;
            if (fused && umlh_bf16_step_tasks(nb0 + nb1, g.M, g.N, splits, L.n_head, with_head) <= L.ctl_tasks) {
                // forward + dW (+ update + finalize) as ONE launch of persistent workgroups over claimed tasks (StepCtl)
                if (h->fuse_epoch > 0xFFFFFF00u) {            // tag wrap (2^32 launches): start the epoch-tagged words over
                    HIPCHK((int)hipMemsetAsync(ws(h, L.fuse_flags), 0, sizeof(float) * (size_t)(3 * L.ctl_tasks), st), "step control words");
                    h->fuse_epoch = 0;
                }
                ++h->fuse_epoch;
                HeadFuse hf;
                if (with_head) { hf = *h->pending_head; hf.n_slabs = splits; hf.n_slabs_img = sp.n_img; }
                unsigned long long* done = reinterpret_cast<unsigned long long*>(ws(h, L.fuse_flags));
                unsigned* claim = reinterpret_cast<unsigned*>(done + L.ctl_tasks);
                unsigned* status = claim + L.ctl_tasks;
                const int ntask = umlh_bf16_step_tasks(nb0 + nb1, g.M, g.N, splits, L.n_head, with_head);
                HIPCHK(umlh_bf16_launch_step(&fb, h->ctw, h->wc, nb0 + nb1, &g, splits, claim, done, status, h->fuse_epoch, TS,
                                             (nb0 + nb1) * TS, with_head ? &hf : nullptr,
                                             (h->dbg_step && 4LL * ntask <= 64LL * L.max_blocks)
                                                 ? reinterpret_cast<unsigned long long*>(ws(h, L.dbg)) : nullptr, h->step_grid, h->step_lazy, st), "step_bf16");
                h->head_fused_done = with_head;
                h->step_launches++;
            } else {
                if (fused) HIPCHK(umlh_bf16_launch_fwd(&fb, h->ctw, h->wc, h->stw, nb0 + nb1, st), "fwd_ce_bf16");
                HIPCHK(umlh_bf16_launch_dw(&g, splits, 0, 0, st), "dw_bf16");
            }
            *n_slabs_head = splits;
        }
        mark(h, 3, st);
        if (h->overlap_pending) { int rc2 = dp_after_head(h, *n_slabs_head, st); if (rc2) return rc2; }
        if (proj) {
            // dH^T[n][r] = sum_c W_head[c][n] dZ^T[c][r]  (image columns), bf16 chunk-major out
            u16* wht16 = reinterpret_cast<u16*>(ws(h, L.wht16));
            HIPCHK(umlh_bf16_launch_transpose_shadow(h->buf.w_head, c.num_classes, c.d_shared, dsp, wht16, 1, st), "W_head^T shadow");
            const int kc = (int)round_up(c.num_classes, 64);
            DwArgsB g = base_args();
            g.A = wht16; g.lda = dsp;
            g.B = g.B2 = dz16; g.k_rows = g.k_rows2 = iota; g.ldb = g.ldb2 = 64; g.bcs = crows * 64;
            g.M = c.d_shared; g.N = r0p; g.K = kc;
            g.k_chunk = (int)round_up(kc, 256); g.k_switch = kc; g.k_valid1 = c.num_classes; g.k_valid2 = 0;
            g.out16 = dht16; g.ldo = dsp;
            HIPCHK(umlh_bf16_launch_dw(&g, 1, 0, 2, st), "dH^T (bf16)");
            // dW_proj[n][k] = sum_r dH^T[n][r] X_img[r][k]
            SplitPlan sp = plan_splits(r0p, 0, L.scap_proj, 256, 256);
            if (sp.chunk > 4096) { sp.chunk = 4096; sp.n_img = (r0p + 4095) / 4096; }
            if (sp.n_img > L.scap_proj)
                return fail(UMLH_E_INVALID, "bf16 dW_proj: %d reduction rows need more than %d split-K slabs", r0p, L.scap_proj);
            DwArgsB p = base_args();
            p.A = dht16; p.lda = dsp;
            p.B = p.B2 = static_cast<const u16*>(img->feats_bf16); p.k_rows = p.k_rows2 = img->index ? img->index : iota;
            p.ldb = p.ldb2 = c.d_img;
            p.M = c.d_shared; p.N = c.d_img; p.K = r0p;
            p.k_chunk = sp.chunk; p.k_switch = r0p; p.k_valid1 = ri; p.k_valid2 = 0;
            p.out = ws(h, L.slabs_proj); p.ldo = c.d_img; p.slab_stride = L.n_proj;
            p.nsplit = p.nsplit1 = sp.n_img;
            HIPCHK(umlh_bf16_launch_dw(&p, sp.n_img, 0, 0, st), "dW_proj (bf16)");
            *n_slabs_proj = sp.n_img;
        }
        mark(h, 4, st);
        return UMLH_OK;
    }

    mark(h, 0, st);
    if (ri > 0 && c.has_proj) HIPCHK(launch_proj_forward(h, img, H, st), "proj forward");
    mark(h, 1, st);

    FwdArgs fa;
    memset(&fa, 0, sizeof(fa));
    SegDesc& s0 = fa.seg[0];
    SegDesc& s1 = fa.seg[1];
    if (ri > 0) {
        s0.feats = c.has_proj ? H : img->feats;
        s0.feat_index = c.has_proj ? nullptr : img->index;
        s0.labels = img->labels; s0.label_index = img->index;
        s0.ld = c.d_shared; s0.rows = ri;
        s0.w_over_rows = hy->img_alpha / (float)img->global_rows;
    }
    s0.scale_ptr = h->buf.scales; s0.col0 = 0; s0.blk0 = 0;
    if (rt > 0) {
        s1.feats = txt->feats; s1.feat_index = txt->index;
        s1.labels = txt->labels; s1.label_index = txt->index;
        s1.ld = c.d_shared; s1.rows = rt;
        s1.w_over_rows = hy->alpha / (float)txt->global_rows;
    }
    s1.scale_ptr = h->buf.scales + 1; s1.col0 = r0p; s1.blk0 = nb0;
    fa.W = h->buf.w_head; fa.C = c.num_classes; fa.K = c.d_shared;
    fa.dzt = want_grad ? dzt : nullptr; fa.ldz = L.ldz;
    fa.partials = ws(h, L.partials);
    fa.row_stats = h->row_stats;
    fa.stamps = (h->dbg_fwd == 9 || h->dbg_fwd >= 20) ? reinterpret_cast<unsigned long long*>(ws(h, L.dbg)) : nullptr;
    fa.dbg = h->dbg_fwd;
    fa.learn = c.learnable_temp;
    if (L.w32s) {
        // refreshed here unless the previous step of the same umlh_train_steps call (or the data-parallel update) just wrote
        // it from its update kernel (between calls the caller may have rewritten w_head)
        if (!h->shadow_fresh)
            HIPCHK(umlh_launch_w_shadow32(h->buf.w_head, ws(h, L.w32s), c.num_classes, c.d_shared, 32 * h->ctw * h->wc, st), "w_shadow32");
        h->shadow_fresh = false;
        fa.Ws = ws(h, L.w32s);
        fa.x3 = umlh_f32_x3();
    }
    HIPCHK(umlh_f32_launch_fwd(&fa, h->ctw, h->wc, nb0 + nb1, st), "fwd_ce");
    mark(h, 2, st);
    if (!want_grad) return UMLH_OK;

    // dW_head[c][k] = sum_r dZ^T[c][r] F[r][k]  over image rows then text rows
    const int rcols = r0p + r1p;
    {
        const SplitPlan sp = plan_splits(r0p, r1p, L.scap_head, KT, 64);
        const int chunk = sp.chunk, splits = sp.n_img + sp.n_txt;
        if (splits > L.scap_head)
            return fail(UMLH_E_INVALID, "dW: %d + %d reduction rows need more than %d split-K slabs", r0p, r1p, L.scap_head);
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = dzt; g.lda = L.ldz;
        g.M = c.num_classes; g.N = c.d_shared; g.K = rcols;
        g.B = c.has_proj ? H : (img ? img->feats : nullptr);
        g.k_rows = c.has_proj ? nullptr : (img ? img->index : nullptr);
        g.ldb = c.d_shared;
        g.B2 = txt ? txt->feats : nullptr; g.k_rows2 = txt ? txt->index : nullptr; g.ldb2 = c.d_shared;
        g.k_switch = r0p; g.k_valid1 = ri; g.k_valid2 = rt;
        g.out = ws(h, L.slabs_head); g.ldo = c.d_shared;
        g.k_chunk = chunk; g.slab_stride = L.n_head; g.alpha = 1.f; g.nsplit1 = sp.n_img; h->n_slabs_img = sp.n_img;
        HIPCHK(umlh_f32_launch_gemm(&g, 0, 1, splits, st), "dW_head gemm");
        *n_slabs_head = splits;
    }
    mark(h, 3, st);
    if (h->overlap_pending) { int rc2 = dp_after_head(h, *n_slabs_head, st); if (rc2) return rc2; }
    *n_slabs_proj = 0;
    if (c.has_proj && ri > 0) {
        // dH^T[n][r] = sum_c W_head[c][n] dZ^T[c][r]   (image columns only)
        float* dht = ws(h, L.dht);
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = h->buf.w_head; g.lda = c.d_shared;            // A(m=n, k=c) = W[c][n]  (k-major)
        g.B = dzt; g.ldb = L.ldz;                           // B(n=r, k=c) = dZ^T[c][r]
        g.M = c.d_shared; g.N = ri; g.K = c.num_classes;
        g.out = dht; g.ldo = L.rcap_img;
        g.k_chunk = g.K; g.alpha = 1.f;
        g.k_switch = INT_MAX; g.k_valid1 = g.K;
        HIPCHK(umlh_f32_launch_gemm(&g, 1, 1, 1, st), "dH gemm");
        // dW_proj[n][k] = sum_r dH^T[n][r] X_img[index[r]][k]
        int want = L.scap_proj;
        int chunk = (int)round_up((ri + want - 1) / want, KT);
        if (chunk < 64) chunk = 64;
        int splits = (ri + chunk - 1) / chunk;
        GemmArgs p;
        memset(&p, 0, sizeof(p));
        p.A = dht; p.lda = L.rcap_img;
        p.B = img->feats; p.k_rows = img->index; p.ldb = c.d_img;
        p.M = c.d_shared; p.N = c.d_img; p.K = ri;
        p.out = ws(h, L.slabs_proj); p.ldo = c.d_img;
        p.k_chunk = chunk; p.slab_stride = L.n_proj; p.alpha = 1.f;
        p.k_switch = INT_MAX; p.k_valid1 = ri;
        HIPCHK(umlh_f32_launch_gemm(&p, 0, 1, splits, st), "dW_proj gemm");
        *n_slabs_proj = splits;
    }
    mark(h, 4, st);
    return UMLH_OK;
}

static FinalizeArgs make_finalize(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt,
                                  const umlh_hyper_t* hy, bool from_partials, float* scalars_out, bool update) {
    const int ri = img ? img->rows : h->last_rows_img, rt = txt ? txt->rows : h->last_rows_txt;
    FinalizeArgs f;
    memset(&f, 0, sizeof(f));
    f.partials = from_partials ? ws(h, h->L.partials) : nullptr;
    f.nb0 = fwd_blocks_img(h, ri);
    f.nb1 = (rt + h->ts - 1) / h->ts;
    f.inv_rows0 = (img && img->rows > 0) ? 1.f / (float)img->global_rows : 0.f;
    f.inv_rows1 = (txt && txt->rows > 0) ? 1.f / (float)txt->global_rows : 0.f;
    f.w0 = hy ? hy->img_alpha : 1.f;
    f.w1 = hy ? hy->alpha : 1.f;
    f.tail = ws(h, h->L.grads) + msg_tail_off(h);
    f.scalars_out = scalars_out;
    f.scales = h->buf.scales; f.m_scales = h->buf.m_scales; f.v_scales = h->buf.v_scales;
    f.update_mask = 0;
    // a parameter is stepped iff its modality has rows on ANY rank (torch skips parameters without a gradient):
    // the data-parallel update must take the same decision on every rank, so it looks at the GLOBAL row counts
    const int gi = img ? img->global_rows : h->global_rows_img, gt = txt ? txt->global_rows : h->global_rows_txt;
    if (update && h->cfg.learnable_temp) f.update_mask = ((img ? ri : gi) > 0 ? 1 : 0) | ((txt ? rt : gt) > 0 ? 2 : 0);
    if (hy) f.opt = make_opt(h->cfg, *hy);
    return f;
}

static int check_step(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt, const umlh_hyper_t* hy,
                      const char* who, bool allow_empty_local = false) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "%s: handle not bound", who);
    if (!hy) return fail(UMLH_E_INVALID, "%s: hyper is null", who);
    int rc = check_batch(h, img, h->cfg.max_rows_img, who);
    if (rc) return rc;
    rc = check_batch(h, txt, h->cfg.max_rows_txt, who);
    if (rc) return rc;
    // finetune.py:123 "At least one of the loaders should be provided"
    if ((img ? img->rows : 0) + (txt ? txt->rows : 0) == 0) {
        // data-parallel split: a rank may hold no row of a (small or unevenly sharded) global batch
        if (allow_empty_local && (img ? img->global_rows : 0) + (txt ? txt->global_rows : 0) > 0) return UMLH_OK;
        return fail(UMLH_E_INVALID, "%s: both modalities empty", who);
    }
    return UMLH_OK;
}

static int train_step_impl(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt, const umlh_hyper_t* hy,
                           float* scalars_out, hipStream_t st, bool keep_shadow) {
    int sh = 0, sp = 0;
    // gradient diagnostics: written to the caller's scalar row (or the workspace tail) by the head-step launch
    h->dp_diag = false;
    float* tail = ws(h, h->L.grads) + msg_tail_off(h);
    h->diag_dst = h->diagnostics ? (scalars_out ? scalars_out : tail + 2) + UMLH_N_CORE_SCALARS : nullptr;
    const umlh_config_t& c = h->cfg;
    OptArgs o = make_opt(h->cfg, *hy);
    FinalizeArgs f = make_finalize(h, img, txt, hy, true, scalars_out, true);
    // the update (and the step scalars) may ride in the forward + dW launch: linear bf16 head, no gradient diagnostics
    // (in profiling mode the interval mark 2 -> 3 then holds the whole step and the others are empty)
    HeadFuse hfuse;
    memset(&hfuse, 0, sizeof(hfuse));
    h->head_fused_done = false;
    h->pending_head = nullptr;
    if (c.precision == UMLH_PREC_BF16 && !c.has_proj && c.d_shared % 8 == 0 && !h->diag_dst) {
        hfuse.slabs = ws(h, h->L.slabs_head); hfuse.slab_stride = h->L.n_head; hfuse.C = c.num_classes; hfuse.K = c.d_shared;
        hfuse.p = h->buf.w_head; hfuse.m = h->buf.m_head; hfuse.v = h->buf.v_head;
        hfuse.shadow = reinterpret_cast<unsigned short*>(ws(h, h->L.w16)); hfuse.cpad = 32 * h->ctw * h->wc;
        hfuse.o = o; hfuse.f = f;
        h->pending_head = &hfuse;
    }
    int rc = forward_backward(h, img, txt, hy, true, st, &sh, &sp);
    h->pending_head = nullptr;
    if (rc) return rc;
    if (h->head_fused_done) {
        h->head_fused_done = false;
        h->shadow_fresh = keep_shadow;
        mark(h, 5, st);
        return UMLH_OK;
    }
    DiagArgs dg;
    dg.dst = h->diag_dst; dg.n_slabs_img = h->n_slabs_img;
    dg.part = ws(h, h->L.diag_part);
    dg.ticket = reinterpret_cast<unsigned*>(dg.part + 4 * ((h->L.n_head + 1023) / 1024 + 2));
    dg.inv_w0 = hy->img_alpha != 0.f ? 1.f / hy->img_alpha : 0.f;
    dg.inv_w1 = hy->alpha != 0.f ? 1.f / hy->alpha : 0.f;
    dg.cols = h->diag_cols;
    if (c.d_shared % 8 == 0) {
        // one launch: slab sum + optimizer + (bf16) next step's W shadow + scalars / logit scales
        const bool bf = c.precision == UMLH_PREC_BF16;
        HIPCHK(umlh_launch_head_step(ws(h, h->L.slabs_head), sh, h->L.n_head, c.num_classes, c.d_shared, h->buf.w_head,
                                     h->buf.m_head, h->buf.v_head, &o, bf ? ws(h, h->L.w16) : nullptr,
                                     32 * h->ctw * h->wc, &f, nullptr, &dg, h->L.w32s ? ws(h, h->L.w32s) : nullptr, st), "head step");
        h->shadow_fresh = (bf || h->L.w32s) && keep_shadow;
    } else {
        HIPCHK(umlh_launch_finalize(&f, st), "finalize");
        HIPCHK(umlh_launch_reduce_update(1, ws(h, h->L.slabs_head), sh, h->L.n_head, h->L.n_head, nullptr,
                                         h->buf.w_head, h->buf.m_head, h->buf.v_head, &o, 0, 0, st), "update head");
    }
    if (sp > 0)
        HIPCHK(umlh_launch_reduce_update(1, ws(h, h->L.slabs_proj), sp, h->L.n_proj, h->L.n_proj, nullptr,
                                         h->buf.w_proj, h->buf.m_proj, h->buf.v_proj, &o, frozen_lo(h), frozen_hi(h), st), "update proj");
    mark(h, 5, st);
    return UMLH_OK;
}


// --------------------------------------------------------------------------------------------------------------- //
// micro-step path (umlh_kernels_micro.hip): whole evaluation intervals of one or MANY heads in one persistent launch
// --------------------------------------------------------------------------------------------------------------- //
struct MicroItem {
    umlh_handle_t h;
    const umlh_stream_t* img; const umlh_stream_t* txt;
    const double* lr; int64_t first_step; float alpha, img_alpha; float* scalars_out;
};

static int device_cus(int dev) {
    static int cus[64];
    static std::once_flag once[64];
    const int d = dev & 63;
    std::call_once(once[d], [&] {
        hipDeviceProp_t p;
        cus[d] = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 0;
    });
    return cus[d];
}

// Shape / row-count test of the micro path for `n_steps` steps of one head (no side effects).
static bool micro_eligible(umlh_handle_t h, const umlh_stream_t* img, const umlh_stream_t* txt, int n_steps) {
    if (!h->L.mc_flags || h->micro_off || h->diagnostics || h->profiling || n_steps < 1) return false;
    if (h->L.mc_nwg > device_cus(h->device) || h->L.mc_nwg > 64) return false;
    for (int k = 0; k < n_steps; ++k) {
        const int ri = img ? img->offsets[k + 1] - img->offsets[k] : 0, rt = txt ? txt->offsets[k + 1] - txt->offsets[k] : 0;
        if (ri < 0 || rt < 0 || ri + rt == 0) return false;
        if (ri > h->cfg.max_rows_img || rt > h->cfg.max_rows_txt) return false;
        if ((ri + 15) / 16 + (rt + 15) / 16 > UMLH_MICRO_MAX_ROWS / 16) return false;
    }
    return true;
}

static int micro_stage(umlh_handle_t h, unsigned char** out) {
    const size_t need = (size_t)micro_table_floats() * sizeof(float) + (size_t)UMLH_MICRO_MAX_HEADS * sizeof(UmlhMicroHead) + 64;
    if (!h->stage) {
        if (hipHostMalloc(reinterpret_cast<void**>(&h->stage), 2 * need, hipHostMallocDefault) != hipSuccess)
            return fail(UMLH_E_HIP, "micro step: pinned staging allocation failed");
        h->stage_bytes = need;
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming) != hipSuccess) return fail(UMLH_E_HIP, "micro step: event");
        (void)hipEventRecord(h->stage_ev[0], nullptr); (void)hipEventRecord(h->stage_ev[1], nullptr);
    }
    const int i = h->stage_next;
    h->stage_next ^= 1;
    if (hipEventSynchronize(h->stage_ev[i]) != hipSuccess) return fail(UMLH_E_HIP, "micro step: staging event");   // copy-out of its last use finished
    *out = h->stage + (size_t)i * h->stage_bytes;
    return i;
}

// Micro launches of one process on one device run one after the other, also when they are enqueued on different
// streams: two persistent grids that are each resident alone could otherwise strand each other's workgroups.
static std::mutex g_micro_mu;
static hipEvent_t g_micro_ev[64];
static bool g_micro_ev_ok[64];

// One launch: `n` heads x `n_steps` steps starting at step offset `k0` of every item's streams.
static int micro_launch_group(const MicroItem* it, int n, int k0, int n_steps, hipStream_t st) {
    umlh_handle_t h0 = it[0].h;
    unsigned char* stage0 = nullptr;
    const int sidx0 = micro_stage(h0, &stage0);
    if (sidx0 < 0) return sidx0;
    const size_t tab_bytes = (size_t)micro_table_floats() * sizeof(float);
    UmlhMicroHead* descs = reinterpret_cast<UmlhMicroHead*>(stage0 + tab_bytes);
    int grid = 0;
    for (int i = 0; i < n; ++i) {
        umlh_handle_t h = it[i].h;
        unsigned char* stg = stage0;
        int sidx = sidx0;
        if (i > 0) { sidx = micro_stage(h, &stg); if (sidx < 0) return sidx; }
        int* offs_i = reinterpret_cast<int*>(stg);
        int* offs_t = offs_i + UMLH_MICRO_MAX_STEPS + 1;
        OptArgs* opt = reinterpret_cast<OptArgs*>(offs_t + UMLH_MICRO_MAX_STEPS + 1 + 2);
        for (int k = 0; k <= n_steps; ++k) {
            offs_i[k] = it[i].img ? it[i].img->offsets[k0 + k] : 0;
            offs_t[k] = it[i].txt ? it[i].txt->offsets[k0 + k] : 0;
        }
        for (int k = 0; k < n_steps; ++k) {
            umlh_hyper_t hy;
            memset(&hy, 0, sizeof(hy));
            hy.lr = it[i].lr[k0 + k]; hy.step = it[i].first_step + k0 + k;
            opt[k] = make_opt(h->cfg, hy);
        }
        float* dtab = ws(h, h->L.mc_tab);
        HIPCHK((int)hipMemcpyAsync(dtab, stg, tab_bytes, hipMemcpyHostToDevice, st), "micro step: table upload");
        if (i > 0) HIPCHK((int)hipEventRecord(h->stage_ev[sidx], st), "micro step: staging event");
        UmlhMicroHead& d = descs[i];
        memset(&d, 0, sizeof(d));
        const umlh_stream_t* s2[2] = {it[i].img, it[i].txt};
        int* doffs[2] = {reinterpret_cast<int*>(dtab), reinterpret_cast<int*>(dtab) + UMLH_MICRO_MAX_STEPS + 1};
        for (int m = 0; m < 2; ++m) {
            if (!s2[m]) continue;
            d.feats[m] = s2[m]->feats; d.labels[m] = s2[m]->labels; d.index[m] = s2[m]->index; d.offs[m] = doffs[m];
        }
        d.w = h->buf.w_head; d.m = h->buf.m_head; d.v = h->buf.v_head;
        d.scales = h->buf.scales; d.m_scales = h->buf.m_scales; d.v_scales = h->buf.v_scales;
        d.opt = reinterpret_cast<const OptArgs*>(reinterpret_cast<int*>(dtab) + 2 * (UMLH_MICRO_MAX_STEPS + 1) + 2);
        d.scalars_out = it[i].scalars_out ? it[i].scalars_out + (size_t)k0 * UMLH_N_SCALARS : nullptr;
        d.xchg = reinterpret_cast<unsigned long long*>(ws(h, h->L.mc_xchg));
        d.status = reinterpret_cast<unsigned*>(ws(h, h->L.mc_flags)) + 64;
        static const bool dbg_micro = [] { const char* e = getenv("UMLH_DBG_MICRO"); return e && atoi(e) == 1; }();
        d.stamps = (dbg_micro && (long long)h->L.max_blocks * 128 * sizeof(float) >= (size_t)h->L.mc_nwg * 96)
                       ? reinterpret_cast<unsigned long long*>(ws(h, h->L.dbg)) : nullptr;
        d.epoch0 = h->micro_epoch;
        h->micro_epoch += (unsigned)n_steps;
        h->micro_launches += 1;
        d.C = h->cfg.num_classes; d.d = h->cfg.d_shared; d.nwg = h->L.mc_nwg; d.wg0 = grid;
        d.learnable = h->cfg.learnable_temp; d.opt_kind = h->cfg.optimizer;
        d.w_img = it[i].img_alpha; d.w_txt = it[i].alpha;
        grid += d.nwg;
        h->shadow_fresh = false;
    }
    UmlhMicroHead* ddesc = reinterpret_cast<UmlhMicroHead*>(ws(h0, h0->L.mc_desc));
    HIPCHK((int)hipMemcpyAsync(ddesc, descs, sizeof(UmlhMicroHead) * (size_t)n, hipMemcpyHostToDevice, st), "micro step: descriptor upload");
    HIPCHK((int)hipEventRecord(h0->stage_ev[sidx0], st), "micro step: staging event");
    {
        std::lock_guard<std::mutex> lk(g_micro_mu);
        const int dv = h0->device & 63;
        if (!g_micro_ev_ok[dv]) {
            HIPCHK((int)hipEventCreateWithFlags(&g_micro_ev[dv], hipEventDisableTiming), "micro step: chain event");
            g_micro_ev_ok[dv] = true;
        } else {
            HIPCHK((int)hipStreamWaitEvent(st, g_micro_ev[dv], 0), "micro step: chain wait");
        }
        HIPCHK(umlh_micro_launch(h0->L.mc_nch, h0->L.mc_cw, h0->cfg.precision == UMLH_PREC_BF16, ddesc, n, n_steps, grid, st), "micro_steps_kernel");
        HIPCHK((int)hipEventRecord(g_micro_ev[dv], st), "micro step: chain record");
    }
    return UMLH_OK;
}

// All items: same number of steps; heads of one launch share the chunking (feature width) and fit the chip together.
static int micro_run(const MicroItem* it, int n_items, int n_steps, hipStream_t st) {
    const int cus = device_cus(it[0].h->device);
    for (int k0 = 0; k0 < n_steps; k0 += UMLH_MICRO_MAX_STEPS) {
        const int ns = n_steps - k0 < UMLH_MICRO_MAX_STEPS ? n_steps - k0 : UMLH_MICRO_MAX_STEPS;
        int a = 0;
        while (a < n_items) {
            int b = a, wgs = 0;
            while (b < n_items && b - a < UMLH_MICRO_MAX_HEADS && wgs + it[b].h->L.mc_nwg <= cus &&
                   it[b].h->cfg.d_shared == it[a].h->cfg.d_shared && it[b].h->cfg.precision == it[a].h->cfg.precision) {
                wgs += it[b].h->L.mc_nwg;
                ++b;
            }
            if (b == a) return fail(UMLH_E_INVALID, "micro step: head %d does not fit the device", a);
            int rc = micro_launch_group(it + a, b - a, k0, ns, st);
            if (rc) return rc;
            a = b;
        }
    }
    return UMLH_OK;
}

int umlh_micro_status(umlh_handle_t h, int32_t* status_out) {
    if (!h || !h->bound || !status_out) return fail(UMLH_E_INVALID, "umlh_micro_status: bad arguments");
    *status_out = 0;
    if (!h->L.mc_flags) return UMLH_OK;
    DeviceGuard dg_(h->device);
    unsigned v = 0;
    HIPCHK((int)hipMemcpy(&v, reinterpret_cast<unsigned*>(ws(h, h->L.mc_flags)) + 64, sizeof(v), hipMemcpyDeviceToHost), "umlh_micro_status");
    *status_out = (int32_t)v;
    return UMLH_OK;
}

// status of the one-launch step's bounded waits: 0 = every wait of every step so far ended; else {code, first task of the range
// waited on, epoch, phase} of the FIRST wait that gave up (the step that hit it and all later ones applied no update)
int umlh_step_status(umlh_handle_t h, int32_t* status_out) {
    if (!h || !h->bound || !status_out) return fail(UMLH_E_INVALID, "umlh_step_status: bad arguments");
    status_out[0] = status_out[1] = status_out[2] = status_out[3] = 0;
    DeviceGuard dg_(h->device);
    if (h->L.fuse_flags) {
        const unsigned* st = reinterpret_cast<const unsigned*>(reinterpret_cast<unsigned long long*>(ws(h, h->L.fuse_flags)) + h->L.ctl_tasks) + h->L.ctl_tasks;
        HIPCHK((int)hipMemcpy(status_out, st, 4 * sizeof(int32_t), hipMemcpyDeviceToHost), "umlh_step_status");
    }
    if (status_out[0] == 0 && h->p2p_on) {       // the direct all-reduce's waits (30 s bound: a peer that never arrives)
        unsigned long long off = 0;
        const long long n_max = 2 * h->L.n_head + h->L.n_proj + 2 + UMLH_N_SCALARS;
        if (umlh_p2p_status_offset(n_max, h->n_ranks, &off) == 0) {
            unsigned v = 0;
            HIPCHK((int)hipMemcpy(&v, static_cast<unsigned char*>(h->p2p_region[h->p2p_rank]) + off, sizeof(v), hipMemcpyDeviceToHost), "umlh_step_status");
            if (v) { status_out[0] = 2; status_out[1] = (int32_t)(v >> 8); status_out[2] = (int32_t)h->p2p_epoch; status_out[3] = (int32_t)(v & 255u); }
        }
    }
    return UMLH_OK;
}

int umlh_step_launches(umlh_handle_t h, int64_t* out) {
    if (!h || !out) return fail(UMLH_E_INVALID, "umlh_step_launches: bad arguments");
    *out = h->step_launches;
    return UMLH_OK;
}

int umlh_micro_launches(umlh_handle_t h, int64_t* out) {
    if (!h || !out) return fail(UMLH_E_INVALID, "umlh_micro_launches: bad arguments");
    *out = h->micro_launches;
    return UMLH_OK;
}

static int check_streams(umlh_handle_t h, const umlh_stream_t* img, const umlh_stream_t* txt, int32_t n_steps, const double* lr,
                         const char* who) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "%s: handle not bound", who);
    if (n_steps < 0 || !lr || (!img && !txt)) return fail(UMLH_E_INVALID, "%s: bad arguments", who);
    if ((img && (!img->offsets || !img->index)) || (txt && (!txt->offsets || !txt->index)))
        return fail(UMLH_E_INVALID, "%s: index/offsets required", who);
    return UMLH_OK;
}

int umlh_train_steps_grouped(const umlh_group_item_t* items, int32_t n_items, int32_t n_steps, void* stream) {
    if (!items || n_items < 1 || n_steps < 0) return fail(UMLH_E_INVALID, "umlh_train_steps_grouped: bad arguments");
    MicroItem* mi = new (std::nothrow) MicroItem[n_items];
    if (!mi) return fail(UMLH_E_INVALID, "umlh_train_steps_grouped: out of host memory");
    bool all_micro = true;
    int rc = UMLH_OK;
    for (int i = 0; i < n_items && !rc; ++i) {
        const umlh_group_item_t& g = items[i];
        rc = check_streams(g.handle, g.img, g.txt, n_steps, g.lr, "umlh_train_steps_grouped");
        if (rc) break;
        if (g.handle->device != items[0].handle->device) { rc = fail(UMLH_E_INVALID, "umlh_train_steps_grouped: heads on different devices"); break; }
        for (int j = 0; j < i; ++j)
            if (items[j].handle == g.handle) rc = fail(UMLH_E_INVALID, "umlh_train_steps_grouped: a handle appears twice");
        mi[i] = MicroItem{g.handle, g.img, g.txt, g.lr, g.first_step, g.alpha, g.img_alpha, g.scalars_out};
        all_micro = all_micro && micro_eligible(g.handle, g.img, g.txt, n_steps);
    }
    if (!rc && n_steps > 0) {
        DeviceGuard dg_(items[0].handle->device);
        if (all_micro) {
            rc = micro_run(mi, n_items, n_steps, (hipStream_t)stream);
        } else {          // shapes outside the micro kernel's envelope: the heads step one after the other on the general path
            for (int i = 0; i < n_items && !rc; ++i)
                rc = umlh_train_steps(items[i].handle, items[i].img, items[i].txt, n_steps, items[i].lr, items[i].first_step,
                                      items[i].alpha, items[i].img_alpha, items[i].scalars_out, stream);
        }
    }
    delete[] mi;
    return rc;
}

int umlh_train_step(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt, const umlh_hyper_t* hy,
                    float* scalars_out, void* stream) {
    int rc = check_step(h, img, txt, hy, "umlh_train_step");
    if (rc) return rc;
    DeviceGuard dg_(h->device);
    h->shadow_fresh = false;          // the caller may have rewritten w_head since the last call
    return train_step_impl(h, img, txt, hy, scalars_out, (hipStream_t)stream, false);
}

int umlh_train_steps(umlh_handle_t h, const umlh_stream_t* img, const umlh_stream_t* txt, int32_t n_steps,
                     const double* lr, int64_t first_step, float alpha, float img_alpha, float* scalars_out,
                     void* stream) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_train_steps: handle not bound");
    if (n_steps < 0 || !lr || (!img && !txt)) return fail(UMLH_E_INVALID, "umlh_train_steps: bad arguments");
    if ((img && (!img->offsets || !img->index)) || (txt && (!txt->offsets || !txt->index)))
        return fail(UMLH_E_INVALID, "umlh_train_steps: index/offsets required");
    DeviceGuard dg_(h->device);
    const bool dp = h->n_ranks > 1 || h->dp_force;
    if (dp && h->n_ranks > 1 && !h->comm && !h->ar_fn && !h->p2p_on)
        return fail(UMLH_E_UNBOUND, "umlh_train_steps: %d ranks but no communicator", h->n_ranks);
    if (!dp && micro_eligible(h, img, txt, n_steps)) {       // batch <= 64 linear head: one persistent launch for all n_steps
        MicroItem it{h, img, txt, lr, first_step, alpha, img_alpha, scalars_out};
        return micro_run(&it, 1, n_steps, (hipStream_t)stream);
    }
    for (int k = 0; k < n_steps; ++k) {
        umlh_batch_t bi, bt;
        memset(&bi, 0, sizeof(bi));
        memset(&bt, 0, sizeof(bt));
        if (img) {
            bi.feats = img->feats; bi.feats_bf16 = img->feats_bf16; bi.labels = img->labels;
            bi.index = img->index + img->offsets[k];
            bi.rows = bi.global_rows = img->offsets[k + 1] - img->offsets[k];
        }
        if (txt) {
            bt.feats = txt->feats; bt.feats_bf16 = txt->feats_bf16; bt.labels = txt->labels;
            bt.index = txt->index + txt->offsets[k];
            bt.rows = bt.global_rows = txt->offsets[k + 1] - txt->offsets[k];
        }
        umlh_hyper_t hy;
        hy.lr = lr[k]; hy.step = first_step + k; hy.alpha = alpha; hy.img_alpha = img_alpha; hy.flags = 0; hy.reserved = 0;
        int rc = check_step(h, img ? &bi : nullptr, txt ? &bt : nullptr, &hy, "umlh_train_steps");
        if (rc) return rc;
        if (k == 0) h->shadow_fresh = false;
        float* so = scalars_out ? scalars_out + (size_t)k * UMLH_N_SCALARS : nullptr;
        if (dp) {
            // data parallel, equal shards: every rank brings the same row counts, the CE means divide by rows x ranks;
            // gradients -> SUM all-reduce (RCCL, enqueued on the same stream) -> identical update on every rank
            bi.global_rows = bi.rows * h->n_ranks;
            bt.global_rows = bt.rows * h->n_ranks;
            if (k > 0) hy.flags |= UMLH_F_WEIGHTS_UNCHANGED;      // the shadow of the weights this loop wrote is current
            rc = grad_step_impl(h, img ? &bi : nullptr, txt ? &bt : nullptr, &hy, (hipStream_t)stream, true);
            if (!rc) rc = apply_update_impl(h, &hy, so, (hipStream_t)stream);
        } else {
            rc = train_step_impl(h, img ? &bi : nullptr, txt ? &bt : nullptr, &hy, so, (hipStream_t)stream, k + 1 < n_steps);
        }
        if (rc) return rc;
    }
    h->shadow_fresh = false;
    return UMLH_OK;
}

// ---- RCCL, loaded at run time (the library has no link-time dependency on it; a process that imported torch already
// holds librccl.so.1 and gets that copy) ----
#include <dlfcn.h>
#include <rccl/rccl.h>
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};
static RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(lib, "ncclAllReduce"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(lib, "ncclGetErrorString"));
        if (api.get_unique_id && api.comm_init_rank && api.all_reduce && api.comm_destroy) api.lib = lib;
    });
    return api.lib ? &api : nullptr;
}

int umlh_comm_unique_id(void* id_out) {
    if (!id_out) return fail(UMLH_E_INVALID, "umlh_comm_unique_id: null argument");
    RcclApi* r = rccl_api();
    if (!r) return fail(UMLH_E_HIP, "umlh_comm_unique_id: librccl.so.1 could not be loaded");
    ncclUniqueId id;
    ncclResult_t e = r->get_unique_id(&id);
    if (e != ncclSuccess) return fail(UMLH_E_HIP, "ncclGetUniqueId: %s", r->error_string ? r->error_string(e) : "error");
    memcpy(id_out, &id, UMLH_COMM_ID_BYTES);
    return UMLH_OK;
}

static int dp_streams(umlh_handle_t h) {
    if (h->comm_stream) return UMLH_OK;
    HIPCHK((int)hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking), "data parallel: comm stream");
    HIPCHK((int)hipEventCreateWithFlags(&h->ev_head_ready, hipEventDisableTiming), "data parallel: event");
    HIPCHK((int)hipEventCreateWithFlags(&h->ev_head_done, hipEventDisableTiming), "data parallel: event");
    return UMLH_OK;
}

static void dp_release(umlh_handle_t h) {
    h->p2p_on = 0;
    if (h->comm && h->comm_owned) { RcclApi* r = rccl_api(); if (r) (void)r->comm_destroy(static_cast<ncclComm_t>(h->comm)); }
    h->comm = nullptr; h->comm_owned = false; h->ar_fn = nullptr; h->ar_ctx = nullptr; h->n_ranks = 1;
}

int umlh_comm_init_rank(umlh_handle_t h, const void* id, int32_t n_ranks, int32_t rank) {
    if (!h || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(UMLH_E_INVALID, "umlh_comm_init_rank: bad arguments");
    RcclApi* r = rccl_api();
    if (!r) return fail(UMLH_E_HIP, "umlh_comm_init_rank: librccl.so.1 could not be loaded");
    DeviceGuard dg_(h->device);
    dp_release(h);
    ncclUniqueId uid;
    memcpy(&uid, id, UMLH_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t e = r->comm_init_rank(&comm, n_ranks, uid, rank);
    if (e != ncclSuccess) return fail(UMLH_E_HIP, "ncclCommInitRank: %s", r->error_string ? r->error_string(e) : "error");
    h->comm = comm; h->comm_owned = true; h->n_ranks = n_ranks;
    return dp_streams(h);
}

int umlh_set_comm(umlh_handle_t h, void* nccl_comm, int32_t n_ranks) {
    if (!h || n_ranks < 1 || (!nccl_comm && n_ranks > 1)) return fail(UMLH_E_INVALID, "umlh_set_comm: bad arguments");
    if (nccl_comm && !rccl_api()) return fail(UMLH_E_HIP, "umlh_set_comm: librccl.so.1 could not be loaded");
    DeviceGuard dg_(h->device);
    dp_release(h);
    h->comm = nccl_comm; h->comm_owned = false; h->n_ranks = n_ranks;
    return nccl_comm ? dp_streams(h) : UMLH_OK;
}

// regions[q]: rank q's exchange region (umlh_p2p_region_bytes(umlh_grad_buffer floats, n_ranks) bytes from umlh_p2p_alloc on rank
// q, zero-filled) as mapped into THIS process: regions[rank] the local allocation, the others through umlh_p2p_open of the
// handles their owners exported.  The caller keeps the mappings alive until the handle is destroyed or another transport is set.
int umlh_p2p_attach(umlh_handle_t h, void* const* regions, int32_t n_ranks, int32_t rank) {
    if (!h || !regions || n_ranks < 1 || n_ranks > 8 || rank < 0 || rank >= n_ranks) return fail(UMLH_E_INVALID, "umlh_p2p_attach: bad arguments");
    if (h->cfg.has_proj) return fail(UMLH_E_INVALID, "umlh_p2p_attach: the 2-layer head overlaps two all-reduces per step on two streams; use RCCL there");
    for (int q = 0; q < n_ranks; ++q) if (!regions[q]) return fail(UMLH_E_INVALID, "umlh_p2p_attach: region %d is null", q);
    DeviceGuard dg_(h->device);
    dp_release(h);
    for (int q = 0; q < n_ranks; ++q) h->p2p_region[q] = regions[q];
    h->p2p_on = 1; h->p2p_rank = rank; h->p2p_epoch = 0; h->n_ranks = n_ranks;
    if (n_ranks == 1) h->dp_force = 1;
    return dp_streams(h);
}

int umlh_set_allreduce(umlh_handle_t h, umlh_allreduce_fn fn, void* ctx, int32_t n_ranks) {
    if (!h || n_ranks < 1 || (!fn && n_ranks > 1)) return fail(UMLH_E_INVALID, "umlh_set_allreduce: bad arguments");
    DeviceGuard dg_(h->device);
    dp_release(h);
    h->ar_fn = fn; h->ar_ctx = ctx; h->n_ranks = n_ranks;
    if (fn && n_ranks == 1) h->dp_force = 1;
    return fn ? dp_streams(h) : UMLH_OK;
}

// SUM all-reduce of `n` floats in place, enqueued on `st`
static int dp_allreduce(umlh_handle_t h, float* buf, long long n, hipStream_t st) {
    if (n <= 0) return UMLH_OK;
    if (h->p2p_on) {                 // direct reduce-scatter + all-gather over the peers' mapped regions (umlh_p2p.hip)
        if (++h->p2p_epoch == 0) h->p2p_epoch = 1;
        const long long n_max = 2 * h->L.n_head + h->L.n_proj + 2 + UMLH_N_SCALARS;
        HIPCHK(umlh_p2p_launch(h->p2p_region, h->n_ranks, h->p2p_rank, buf, n, n_max, h->p2p_epoch, st), "p2p all-reduce");
        return UMLH_OK;
    }
    if (h->ar_fn) {
        int rc = h->ar_fn(h->ar_ctx, buf, (uint64_t)n, st);
        return rc ? fail(UMLH_E_HIP, "data parallel: the all-reduce callback failed with code %d", rc) : UMLH_OK;
    }
    if (h->comm) {
        RcclApi* r = rccl_api();
        ncclResult_t e = r->all_reduce(buf, buf, (size_t)n, ncclFloat, ncclSum, static_cast<ncclComm_t>(h->comm), st);
        if (e != ncclSuccess) return fail(UMLH_E_HIP, "ncclAllReduce: %s", r->error_string ? r->error_string(e) : "error");
        return UMLH_OK;
    }
    return h->n_ranks > 1 ? fail(UMLH_E_UNBOUND, "data parallel: %d ranks but no communicator (umlh_comm_init_rank / umlh_set_comm / umlh_set_allreduce)", h->n_ranks)
                          : UMLH_OK;
}

// Sum the dW_head slabs into the gradient message: [g_head], or [g_img | g_txt] when the per-modality gradient
// diagnostics are on (finetune.py:190-191,203-206 need the GLOBAL per-modality gradients: dot products and norms are
// not linear in the ranks' partial sums, so the two sums travel separately and the update kernel forms the diagnostics
// from the all-reduced pair).
static int dp_reduce_head(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt, const umlh_hyper_t* hy, int sh,
                          hipStream_t st) {
    OptArgs o = make_opt(h->cfg, *hy);
    FinalizeArgs f = make_finalize(h, img, txt, hy, true, nullptr, false);
    float* grads = ws(h, h->L.grads);
    const long long nh = h->L.n_head;
    if (h->dp_diag) {
        const int si = h->n_slabs_img < sh ? h->n_slabs_img : sh;
        FinalizeArgs f2 = f;
        f2.partials = nullptr;                           // the step scalars are formed once (first launch)
        DiagArgs none; none.dst = nullptr; none.n_slabs_img = si; none.inv_w0 = none.inv_w1 = 0.f; none.part = nullptr; none.ticket = nullptr;
        if (si > 0) HIPCHK(umlh_launch_head_step(ws(h, h->L.slabs_head), si, nh, h->cfg.num_classes, h->cfg.d_shared, nullptr, nullptr,
                                                 nullptr, &o, nullptr, 32 * h->ctw * h->wc, &f, grads, &none, nullptr, st), "reduce head (image rows)");
        else HIPCHK((int)hipMemsetAsync(grads, 0, sizeof(float) * nh, st), "zero image gradient");
        none.n_slabs_img = 0;
        if (sh - si > 0) HIPCHK(umlh_launch_head_step(ws(h, h->L.slabs_head) + (size_t)si * nh, sh - si, nh, h->cfg.num_classes, h->cfg.d_shared,
                                                      nullptr, nullptr, nullptr, &o, nullptr, 32 * h->ctw * h->wc, si > 0 ? &f2 : &f, grads + nh,
                                                      &none, nullptr, st), "reduce head (text rows)");
        else HIPCHK((int)hipMemsetAsync(grads + nh, 0, sizeof(float) * nh, st), "zero text gradient");
        return UMLH_OK;
    }
    if (h->cfg.d_shared % 8 == 0) {
        // image slabs and text slabs are summed separately, then added: the order of the fused step and of the one-launch
        // gradient (step_update_task), so the message is bit-identical whichever launch form wrote it
        DiagArgs order; order.dst = nullptr; order.n_slabs_img = h->n_slabs_img < sh ? h->n_slabs_img : sh;
        order.inv_w0 = order.inv_w1 = 0.f; order.part = nullptr; order.ticket = nullptr;
        HIPCHK(umlh_launch_head_step(ws(h, h->L.slabs_head), sh, nh, h->cfg.num_classes, h->cfg.d_shared, nullptr, nullptr, nullptr, &o,
                                     nullptr, 32 * h->ctw * h->wc, &f, grads, &order, nullptr, st), "reduce head");
    } else {
        HIPCHK(umlh_launch_finalize(&f, st), "finalize");
        HIPCHK(umlh_launch_reduce_update(0, ws(h, h->L.slabs_head), sh, nh, nh, grads, nullptr, nullptr, nullptr, &o, 0, 0, st), "reduce head");
    }
    return UMLH_OK;
}

// 2-layer head, data parallel: the head gradient is complete behind the dW_head GEMM; its all-reduce (12.8 MB at cfg3)
// runs on the second stream beside the img_proj backward GEMMs (dH^T, dW_proj) of the step's stream.
static int dp_after_head(umlh_handle_t h, int n_slabs_head, hipStream_t st) {
    h->overlap_pending = false;
    int rc = dp_reduce_head(h, h->overlap_img, h->overlap_txt, h->overlap_hy, n_slabs_head, st);
    if (rc) return rc;
    HIPCHK((int)hipEventRecord(h->ev_head_ready, st), "data parallel: event record");
    HIPCHK((int)hipStreamWaitEvent(h->comm_stream, h->ev_head_ready, 0), "data parallel: stream wait");
    rc = dp_allreduce(h, ws(h, h->L.grads), msg_head_len(h), h->comm_stream);
    if (rc) return rc;
    HIPCHK((int)hipEventRecord(h->ev_head_done, h->comm_stream), "data parallel: event record");
    return UMLH_OK;
}

// Gradients of one step into the message buffer.  `comm`: also all-reduce it (the data-parallel multi-step loop); the
// head part then overlaps the img_proj backward when the head has one.
static int grad_step_impl(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt, const umlh_hyper_t* hy, hipStream_t st,
                          bool comm) {
    int sh = 0, sp = 0;
    h->last_rows_img = img ? img->rows : 0;
    h->last_rows_txt = txt ? txt->rows : 0;
    h->global_rows_img = img ? img->global_rows : 0;
    h->global_rows_txt = txt ? txt->global_rows : 0;
    h->dp_diag = h->diagnostics && h->cfg.d_shared % 8 == 0;
    float* grads = ws(h, h->L.grads);
    if (h->last_rows_img + h->last_rows_txt == 0) {       // no local row: this rank contributes zeros to the all-reduce
        HIPCHK((int)hipMemsetAsync(grads, 0, sizeof(float) * msg_len(h), st), "zero gradient buffer");
        return comm ? dp_allreduce(h, grads, msg_len(h), st) : UMLH_OK;
    }
    if (!(hy->flags & UMLH_F_WEIGHTS_UNCHANGED)) h->shadow_fresh = false;
    h->diag_dst = nullptr;                                // the diagnostics of a split step are formed by umlh_apply_update
    const bool overlap = comm && h->cfg.has_proj && h->comm_stream != nullptr && (img ? img->rows : 0) > 0;
    h->overlap_pending = overlap;
    h->overlap_hy = hy; h->overlap_img = img; h->overlap_txt = txt;
    // linear bf16 head: the slab sum into the message (and the step scalars) rides in the forward + dW launch
    HeadFuse hfuse;
    memset(&hfuse, 0, sizeof(hfuse));
    h->head_fused_done = false;
    h->pending_head = nullptr;
    if (h->cfg.precision == UMLH_PREC_BF16 && !h->cfg.has_proj && h->cfg.d_shared % 8 == 0 && !h->dp_diag && !overlap) {
        hfuse.slabs = ws(h, h->L.slabs_head); hfuse.slab_stride = h->L.n_head; hfuse.C = h->cfg.num_classes; hfuse.K = h->cfg.d_shared;
        hfuse.grad_out = grads; hfuse.cpad = 32 * h->ctw * h->wc;
        hfuse.o = make_opt(h->cfg, *hy); hfuse.f = make_finalize(h, img, txt, hy, true, nullptr, false);
        h->pending_head = &hfuse;
    }
    int rc = forward_backward(h, img, txt, hy, true, st, &sh, &sp);
    h->pending_head = nullptr;
    h->overlap_pending = false;
    if (rc) return rc;
    const bool head_done = h->head_fused_done;
    h->head_fused_done = false;
    if (!overlap && !head_done) { rc = dp_reduce_head(h, img, txt, hy, sh, st); if (rc) return rc; }
    OptArgs o = make_opt(h->cfg, *hy);
    if (h->cfg.has_proj) {
        float* gp = grads + msg_head_len(h);
        if (sp > 0)
            HIPCHK(umlh_launch_reduce_update(0, ws(h, h->L.slabs_proj), sp, h->L.n_proj, h->L.n_proj, gp, nullptr, nullptr, nullptr, &o, 0, 0, st),
                   "reduce proj");
        else
            HIPCHK((int)hipMemsetAsync(gp, 0, sizeof(float) * h->L.n_proj, st), "zero proj grad");
    }
    mark(h, 5, st);
    if (comm) {
        if (overlap) {                                    // head part is in flight on the second stream: the rest here, then join
            rc = dp_allreduce(h, grads + msg_head_len(h), msg_len(h) - msg_head_len(h), st);
            if (rc) return rc;
            HIPCHK((int)hipStreamWaitEvent(st, h->ev_head_done, 0), "data parallel: stream wait");
        } else {
            rc = dp_allreduce(h, grads, msg_len(h), st);
            if (rc) return rc;
        }
    }
    return UMLH_OK;
}

int umlh_grad_step(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt, const umlh_hyper_t* hy,
                   void* stream) {
    int rc = check_step(h, img, txt, hy, "umlh_grad_step", true);
    if (rc) return rc;
    DeviceGuard dg_(h->device);
    return grad_step_impl(h, img, txt, hy, (hipStream_t)stream, false);
}

int umlh_debug_buffer(umlh_handle_t h, void** device_ptr, uint64_t* n_bytes) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_debug_buffer: handle not bound");
    if (device_ptr) *device_ptr = ws(h, h->L.dbg);
    if (n_bytes) *n_bytes = (uint64_t)h->L.max_blocks * 128 * sizeof(float);
    return UMLH_OK;
}

int umlh_grad_buffer(umlh_handle_t h, float** device_ptr, uint64_t* n_floats) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_grad_buffer: handle not bound");
    if (device_ptr) *device_ptr = ws(h, h->L.grads);
    const bool diag = h->diagnostics && h->cfg.d_shared % 8 == 0;        // the layout umlh_grad_step will use
    if (n_floats) *n_floats = (uint64_t)((diag ? 2 : 1) * h->L.n_head + h->L.n_proj + 2 + UMLH_N_SCALARS);
    return UMLH_OK;
}

static int apply_update_impl(umlh_handle_t h, const umlh_hyper_t* hy, float* scalars_out, hipStream_t st) {
    OptArgs o = make_opt(h->cfg, *hy);
    float* grads = ws(h, h->L.grads);
    FinalizeArgs f = make_finalize(h, nullptr, nullptr, hy, false, scalars_out, true);
    if (h->cfg.d_shared % 8 == 0) {
        const bool bf = h->cfg.precision == UMLH_PREC_BF16;
        DiagArgs dg;
        dg.dst = nullptr; dg.n_slabs_img = 1; dg.inv_w0 = dg.inv_w1 = 0.f;
        dg.part = ws(h, h->L.diag_part);
        dg.ticket = reinterpret_cast<unsigned*>(dg.part + 4 * ((h->L.n_head + 1023) / 1024 + 2));
        dg.cols = h->diag_cols;
        if (h->dp_diag) {
            // the two all-reduced per-modality gradients are the two "slabs" of the update kernel: it sums them, steps the
            // weights and accumulates dot / norms / sign agreement of the GLOBAL gradients (finetune.py:203-206)
            float* dst = (scalars_out ? scalars_out : f.tail + 2) + UMLH_N_CORE_SCALARS;
            dg.dst = dst;
            dg.inv_w0 = hy->img_alpha != 0.f ? 1.f / hy->img_alpha : 0.f;
            dg.inv_w1 = hy->alpha != 0.f ? 1.f / hy->alpha : 0.f;
        }
        HIPCHK(umlh_launch_head_step(grads, h->dp_diag ? 2 : 1, h->L.n_head, h->cfg.num_classes, h->cfg.d_shared, h->buf.w_head,
                                     h->buf.m_head, h->buf.v_head, &o, bf ? ws(h, h->L.w16) : nullptr, 32 * h->ctw * h->wc, &f, nullptr,
                                     h->dp_diag ? &dg : nullptr, h->L.w32s ? ws(h, h->L.w32s) : nullptr, st), "update head");
        h->shadow_fresh = bf || h->L.w32s;              // the next umlh_grad_step may trust it (see umlh_grad_step)
    } else {
        HIPCHK(umlh_launch_reduce_update(1, grads, 1, h->L.n_head, h->L.n_head, nullptr, h->buf.w_head, h->buf.m_head,
                                         h->buf.v_head, &o, 0, 0, st), "update head");
        HIPCHK(umlh_launch_finalize(&f, st), "finalize");
    }
    if (h->cfg.has_proj && h->global_rows_img > 0)
        HIPCHK(umlh_launch_reduce_update(1, grads + msg_head_len(h), 1, h->L.n_proj, h->L.n_proj, nullptr, h->buf.w_proj,
                                         h->buf.m_proj, h->buf.v_proj, &o, frozen_lo(h), frozen_hi(h), st), "update proj");
    return UMLH_OK;
}

int umlh_apply_update(umlh_handle_t h, const umlh_hyper_t* hy, float* scalars_out, void* stream) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_apply_update: handle not bound");
    if (!hy) return fail(UMLH_E_INVALID, "umlh_apply_update: hyper is null");
    DeviceGuard dg_(h->device);
    return apply_update_impl(h, hy, scalars_out, (hipStream_t)stream);
}

int umlh_eval_batch(umlh_handle_t h, const umlh_batch_t* b, float* scalars_out, void* stream) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_eval_batch: handle not bound");
    if (!b || !scalars_out) return fail(UMLH_E_INVALID, "umlh_eval_batch: null argument");
    DeviceGuard dg_(h->device);
    int rc = check_batch(h, b, h->cfg.max_rows_img, "umlh_eval_batch");
    if (rc) return rc;
    if (b->rows == 0) return fail(UMLH_E_INVALID, "umlh_eval_batch: empty batch");
    hipStream_t st = (hipStream_t)stream;
    umlh_hyper_t hy;
    memset(&hy, 0, sizeof(hy));
    hy.lr = 0; hy.step = 1; hy.alpha = 1.f; hy.img_alpha = 1.f;
    int sh = 0, sp = 0;
    h->shadow_fresh = false;
    h->diag_dst = nullptr;
    rc = forward_backward(h, b, nullptr, &hy, false, st, &sh, &sp);
    if (rc) return rc;
    FinalizeArgs f = make_finalize(h, b, nullptr, &hy, true, scalars_out, false);
    HIPCHK(umlh_launch_finalize(&f, st), "finalize");
    return UMLH_OK;
}

int umlh_eval_rows(umlh_handle_t h, const umlh_batch_t* b, float* row_stats, void* stream) {
    if (!h || !h->bound) return fail(UMLH_E_UNBOUND, "umlh_eval_rows: handle not bound");
    if (!b || !row_stats) return fail(UMLH_E_INVALID, "umlh_eval_rows: null argument");
    DeviceGuard dg_(h->device);
    int rc = check_batch(h, b, h->cfg.max_rows_img, "umlh_eval_rows");
    if (rc) return rc;
    if (b->rows == 0) return fail(UMLH_E_INVALID, "umlh_eval_rows: empty batch");
    umlh_hyper_t hy;
    memset(&hy, 0, sizeof(hy));
    hy.lr = 0; hy.step = 1; hy.alpha = 1.f; hy.img_alpha = 1.f;
    int sh = 0, sp = 0;
    h->shadow_fresh = false;
    h->diag_dst = nullptr;
    h->row_stats = row_stats;
    rc = forward_backward(h, b, nullptr, &hy, false, (hipStream_t)stream, &sh, &sp);
    h->row_stats = nullptr;
    return rc;
}
