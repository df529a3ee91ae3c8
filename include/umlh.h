/*
 * umlh.h -- C ABI of the MI355X-native UML head fine-tune hot path.
 *
 * The reference (OEmiliatanO/Unpaired-Multimodal-Learning) has no FFI / plugin
 * boundary: the hot path sits behind a Python duck-typed contract between
 * vision_language/finetune.py:train() and the model / optimizer / scheduler /
 * loader objects it is handed.  This header is the
 * boundary a maintainer binds underneath that contract (ctypes stub in
 * INTEGRATION.md).  Each entry point cites the reference lines it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - plain C types only; every pointer marked "device" is HIP device memory
 *     owned by the caller (torch allocates it); the library never allocates,
 *     frees or synchronises, so every call is graph-capturable (on a capturing stream the bf16 step takes its
 *     launch-per-kernel form: the one-launch step tags its in-launch hand-offs with a per-launch argument; the
 *     single-launch micro path and the opt-in 2-D forward use such tags too and are not meant for capture).
 *   - every function returns 0 on success or a negative UMLH_E_* code;
 *     umlh_last_error() returns a thread-local message for the last failure.
 *   - all work is enqueued on the hipStream_t passed as `void* stream`
 *     (asynchronous w.r.t. the host); a handle is not thread-safe.
 *   - features are row-major fp32 [rows, dim]; labels are int64 (torch.long);
 *     optional `index` (int64) gathers rows from a device-resident table, which
 *     replaces the DataLoader collate of finetune.py:33-39,165-172.
 *
 * Environment switches read by the library (tuning / analysis; defaults are the measured-fastest settings):
 *   UMLH_WT=0            plain instead of write-through (sc1) stores for what a kernel hands to the next launch
 *   UMLH_BF16_FUSE=2|1|0 bf16 mode, linear head: the whole step as one launch (step_bf16, default) | forward + dW as one
 *                        launch (fwd_dw_bf16) + the update kernel | three launches
 *   UMLH_BF16_FWD2D=1    bf16 mode: the 2-D forward (128-row tiles x 256-class groups, in-launch softmax merge); slower at cfg2
 *   UMLH_BF16_STW=2      bf16 mode: two sample tiles per wave of the 1-D forward
 *   UMLH_MICRO=0         never take the single-launch micro step
 *   UMLH_F32_DW=0        fp32 dW through the generic GEMM instead of dw_f32
 *   UMLH_ENC_GRAPH=0 / UMLH_ENC_FORK=1   encoder plans: no HIP-graph replay / weight-gradient work on a forked branch
 *   UMLH_FORCE_DP=1      take the data-parallel split step (grad -> all-reduce -> update) with one rank
 *   UMLH_DBG_FWD / UMLH_DBG_DW / UMLH_DBG_MICRO / UMLH_DBG_STEP   in-kernel stamps (scripts/fwd_stamps.py, dw_stamps.py,
 *                        micro_stamps.py, step_timeline.py)
 */
#ifndef UMLH_H
#define UMLH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UMLH_OK              0
#define UMLH_E_INVALID      -1   /* bad argument / unsupported shape            */
#define UMLH_E_UNBOUND      -2   /* umlh_bind() not called / workspace too small */
#define UMLH_E_HIP          -3   /* a HIP runtime call failed                   */
#define UMLH_E_NOGPU        -4   /* no gfx950 device visible                    */

/* engine/optimizer/optim.py:6,15-31 AVAI_OPTIMS */
#define UMLH_OPT_SGD   0   /* torch.optim.SGD(momentum=0.9, nesterov=False)  optim.py:34-48 */
#define UMLH_OPT_ADAM  1   /* torch.optim.Adam(betas=(0.9,0.999))            optim.py:50-59 */
#define UMLH_OPT_ADAMW 2   /* torch.optim.AdamW(betas=(0.9,0.999))           optim.py:61-71 */

#define UMLH_PREC_FP32 0   /* fp32 operands, fp32 results: the parity mode (logits / loss 1e-4).  The fused step forms its products
                            * from three-way bf16 splits of both operands on the bf16 MFMA (six exact piece products per product, fp32
                            * accumulation; what is dropped is 2^-24 of a product rms, 2^-20 worst case: as accurate against float64 as the fp32 MFMA chain) -- or, with UMLH_F32_X3=0 in the environment, on
                            * the f32-input MFMA as in ABI <= 3.  umlh_logits / umlh_gemm_f32 always use the f32-input MFMA. */
#define UMLH_PREC_BF16 1   /* bf16 operands, fp32 accumulate: the throughput mode         */

typedef struct umlh_handle_s* umlh_handle_t;

/* Shape + optimizer of one head.  Mirrors the constructor arguments of
 * engine/models/head.py:39-70 (UML) / :101-125 (UMLClip) and of
 * engine/optimizer/optim.py:15 build_optimizer. */
typedef struct {
    int32_t d_img;          /* image feature dim (vision_model.num_features)                 */
    int32_t d_shared;       /* head input dim (text_indim if has_proj else d_img)            */
    int32_t num_classes;    /* C, 1..1024                                                    */
    int32_t has_proj;       /* 1: img_proj Linear(d_img -> d_shared, bias=False) head.py:64-66 */
    int32_t learnable_temp; /* 1: img_scale / txt_scale are parameters        head.py:69-70 */
    int32_t optimizer;      /* UMLH_OPT_*                                                    */
    int32_t precision;      /* UMLH_PREC_*                                                   */
    int32_t max_rows_img;   /* per-step row capacity, image modality                         */
    int32_t max_rows_txt;   /* per-step row capacity, text modality                          */
    double  beta1, beta2, eps, momentum, weight_decay;  /* python floats of optim.py:9,12 */
} umlh_config_t;

/* Caller-owned device state (torch tensors): parameters in reference
 * parameters() order img_proj.weight, head.weight, img_scale, txt_scale, and the
 * optimizer state torch.optim keeps for each (exp_avg / momentum_buffer, exp_avg_sq). */
typedef struct {
    float* w_head;  float* m_head;  float* v_head;     /* [C, d_shared]                    */
    float* w_proj;  float* m_proj;  float* v_proj;     /* [d_shared, d_img] or NULL        */
    float* scales;  float* m_scales; float* v_scales;  /* [2] = {img_scale, txt_scale}     */
    void*    workspace;                                /* >= umlh_workspace_bytes(cfg)     */
    uint64_t workspace_bytes;
} umlh_buffers_t;

/* One modality's rows for one step: what fetch_next() + .to(device) deliver in
 * finetune.py:164-174, as (table, index) instead of a collated copy. */
typedef struct {
    const float*   feats;       /* device [table_rows, dim] fp32                           */
    const int64_t* labels;      /* device [table_rows]                                     */
    const int64_t* index;       /* device [rows] row ids into feats/labels, NULL = 0..rows-1 */
    int32_t        rows;        /* rows processed by THIS rank in this step (0 = absent)   */
    int32_t        global_rows; /* denominator of the CE mean: rows summed over all ranks  */
    const void*    feats_bf16;  /* device [table_rows, dim] bf16 copy of feats (umlh_to_bf16);
                                   required by train/grad/eval calls in UMLH_PREC_BF16 mode  */
} umlh_batch_t;

/* Per-step scalars: lr = scheduler value used by this optimizer.step()
 * (engine/optimizer/scheduler.py:58-81), step = 1-based count of optimizer steps
 * (Adam bias correction), alpha = text loss weight (finetune.py:188). */
typedef struct {
    double  lr;
    int64_t step;
    float   alpha;      /* weight of the text CE   (args.alpha)      finetune.py:188 */
    float   img_alpha;  /* weight of the image CE  (img_alpha = 1.0) finetune.py:160 */
    int32_t flags;      /* UMLH_F_* */
    int32_t reserved;
} umlh_hyper_t;
/* The caller guarantees that w_head has not been written by anyone but this handle since its
 * last umlh_apply_update / umlh_train_step: the bf16 weight shadow the update kernel produced
 * may be reused instead of rebuilt (bf16 mode only; default is to rebuild on every call). */
#define UMLH_F_WEIGHTS_UNCHANGED 1

/* Device float[UMLH_N_SCALARS] written by train/grad/eval calls. */
#define UMLH_S_LOSS_IMG   0   /* mean CE over image rows   finetune.py:186 */
#define UMLH_S_LOSS_TXT   1   /* mean CE over text rows    finetune.py:187 */
#define UMLH_S_ACC_IMG    2   /* top-1 accuracy, image     finetune.py:197 */
#define UMLH_S_ACC_TXT    3   /* top-1 accuracy, text      finetune.py:198 */
#define UMLH_S_GSCALE_IMG 4   /* d loss / d img_scale                      */
#define UMLH_S_GSCALE_TXT 5   /* d loss / d txt_scale                      */
#define UMLH_S_CORRECT    6   /* eval: number of correct rows (as float)   */
#define UMLH_S_LOSS_SUM   7   /* eval: sum of per-row CE                   */
/* Per-step gradient diagnostics (finetune.py:190-191,203-206,238): raw sums over all C*d elements
 * of the UNWEIGHTED per-modality head gradients g_img = dL_img/dW, g_txt = dL_txt/dW -- by-products
 * of the split-K slab reduction (image and text rows occupy separate slabs).  Written by
 * umlh_train_step(s) after umlh_enable_diagnostics(h, 1) (off by default: the accumulation costs
 * about 2.5 us per step at C*d = 512 000) when d_shared % 8 == 0; left untouched by umlh_eval_batch; in the
 * data-parallel split they are written by umlh_apply_update from the all-reduced per-modality gradients (the message
 * then carries g_img and g_txt separately); zero for a modality whose loss weight is 0.  The caller forms
 *   grad_direction_sim = DOT / sqrt(N2_IMG * N2_TXT),  img_grad_norm = sqrt(N2_IMG),
 *   txt_grad_norm = sqrt(N2_TXT),  grad_agreement_rate = AGREE / (C * d). */
#define UMLH_S_GRAD_DOT      8   /* sum g_img * g_txt                          */
#define UMLH_S_GRAD_N2_IMG   9   /* sum g_img^2                                */
#define UMLH_S_GRAD_N2_TXT   10  /* sum g_txt^2                                */
#define UMLH_S_GRAD_AGREE    11  /* #elements with sign(g_img) == sign(g_txt)  */
#define UMLH_N_CORE_SCALARS  8
#define UMLH_N_SCALARS    12

const char* umlh_last_error(void);
int  umlh_version(void);        /* ABI revision: 3 = round 2 (grouped / micro / data-parallel / encoder-plan / InfoNCE entry points); 4 = round 3 (umlh_step_status / umlh_step_launches, umlh_p2p_*) */

/* Bytes of workspace a handle with this config needs (0 on invalid config). */
uint64_t umlh_workspace_bytes(const umlh_config_t* cfg);

int  umlh_create(const umlh_config_t* cfg, umlh_handle_t* out);
int  umlh_destroy(umlh_handle_t h);
/* Attach parameter / optimizer-state / workspace buffers (state_dict round trips
 * stay on the torch side: finetune.py:249,274). */
int  umlh_bind(umlh_handle_t h, const umlh_buffers_t* bufs);

/* Per-step gradient diagnostics on/off (finetune.py:190-191,203-206: the reference computes them on
 * every step with two extra backward passes; here they ride on the slab reduction).  Off by default. */
int  umlh_enable_diagnostics(umlh_handle_t h, int32_t on);
/* Heads with bias (head.py:65,68 bias=True, run as packed rows [weight | bias | 0...]): the reference forms its
 * diagnostics over head.weight only (finetune.py:190-191), so only columns [0, cols) of every class row of w_head enter the
 * dot product, the norms and the sign-agreement count (cols = 0 or d_shared: all columns). */
int  umlh_set_diagnostic_columns(umlh_handle_t h, int32_t cols);

/* img_proj WITH bias (head.py:65 `bias=True`, run as a bias-free projection over rows [x | 1 | 0...]): row `row` of w_proj is the
 * constant row that copies the ones column into the projected rows (the head's own bias column needs it); the optimizer
 * leaves that row, and its moments, untouched.  row < 0 clears.  Needs d_img % 4 == 0. */
int  umlh_freeze_proj_row(umlh_handle_t h, int32_t row);

/* head.weight.data = get_zero_shot_weights(...)  head.py:22-37,96-98: per-class
 * mean of the text rows (rows of classes without text stay 0), rows L2-normalised. */
int  umlh_zero_shot_init(umlh_handle_t h, const float* text_feats, const int64_t* text_labels,
                         int64_t n_text, void* stream);

/* model(images, text_features) logits, unfused, for callers that want the
 * tensors (head.py:77-84 / :131-137).  modality 0 = image path (through img_proj
 * when has_proj, scaled by img_scale), 1 = text path (txt_scale).
 * logits_out: device [rows, C] fp32. */
int  umlh_logits(umlh_handle_t h, const umlh_batch_t* batch, int modality, float* logits_out, void* stream);

/* model.extract_features(images) for the projected head: out[rows, d_shared] =
 * img_proj(feats) (head.py:87-90).  Requires has_proj. */
int  umlh_project(umlh_handle_t h, const umlh_batch_t* batch, float* out, void* stream);

/* One whole training step (finetune.py:180-195 minus logging): fused
 * features x W^T * scale -> softmax-CE -> dZ, dW(+dW_proj, dscale), optimizer
 * update of every parameter.  Either batch may have rows == 0 (modality absent,
 * finetune.py:373-380).  scalars_out: device float[UMLH_N_SCALARS] or NULL. */
int  umlh_train_step(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt,
                     const umlh_hyper_t* hyper, float* scalars_out, void* stream);

/* Many consecutive training steps from device-resident tables with no host work in
 * between (the whole `for i in range(max_iters)` body of finetune.py:162-195 between two
 * evaluations).  Step k consumes index[offsets[k] .. offsets[k+1]) of each modality
 * (offsets: HOST int32[n_steps+1]; a modality with table == NULL is absent), uses
 * lr[k] (HOST double[n_steps]) and optimizer step number first_step + k, and writes its
 * scalars to scalars_out + k*UMLH_N_SCALARS (device, may be NULL). */
typedef struct {
    const float*   feats;       /* device [table_rows, dim] fp32                           */
    const void*    feats_bf16;  /* device bf16 shadow (UMLH_PREC_BF16) or NULL             */
    const int64_t* labels;      /* device [table_rows]                                     */
    const int64_t* index;       /* device int64: concatenated per-step row ids             */
    const int32_t* offsets;     /* HOST  int32 [n_steps + 1]                               */
} umlh_stream_t;
int  umlh_train_steps(umlh_handle_t h, const umlh_stream_t* img, const umlh_stream_t* txt, int32_t n_steps,
                      const double* lr, int64_t first_step, float alpha, float img_alpha,
                      float* scalars_out, void* stream);

/* Batch <= 64 linear heads (the reference's own operating point: batch 8 / 32 / 64, 12 800 iterations,
 * engine/optimizer/default.py:3-45): umlh_train_steps runs all n_steps inside ONE persistent launch when the head has
 * no img_proj, its width has a supported chunking (fp32: d in {16,32,48,64,80,96,128,256,384,512,640,768,1024}; bf16
 * operand mode: the multiples of 128 among them), diagnostics are off and every step has at most 4 sample tiles
 * (ceil(rows_img/16) + ceil(rows_txt/16) <= 4).  bf16 mode reads the fp32 tables and rounds operands at use (same
 * values as the bf16 shadows of the three-kernel path); for d <= 640 the step's rows stay in LDS as bf16.  The
 * class axis is cut into 16-class slices, one workgroup per slice owns its part of W / m / v for the whole call; the
 * slices of a head exchange their softmax statistics once per step (in-launch, bounded waits).  UMLH_MICRO=0 in the
 * environment disables the path.  A wait that gives up (another process starving the device of CUs) is reported
 * through umlh_micro_status (0 = ok); the head's state is then undefined.
 *
 * umlh_train_steps_grouped: the sweep of finetune.py:406-448 (HYPER_DICT grid x alpha x seeds: many independent heads over
 * the SAME feature tables) as grouped launches -- G heads [G, C, d] advance n_steps each inside the same persistent
 * launch(es), every head with its own index streams, lr table, optimizer state and scalar rows.  Results are bit-identical
 * to G separate umlh_train_steps calls.  Heads outside the micro envelope fall back to per-head stepping. */
typedef struct {
    umlh_handle_t handle;
    const umlh_stream_t* img;    /* NULL = modality absent */
    const umlh_stream_t* txt;
    const double* lr;            /* HOST double[n_steps] */
    int64_t first_step;
    float   alpha, img_alpha;
    float*  scalars_out;         /* device float[n_steps * UMLH_N_SCALARS] or NULL */
} umlh_group_item_t;
int  umlh_train_steps_grouped(const umlh_group_item_t* items, int32_t n_items, int32_t n_steps, void* stream);
int  umlh_micro_status(umlh_handle_t h, int32_t* status_out);
/* number of persistent micro-step launches this handle has taken part in (which path ran: tests, logging) */
int  umlh_micro_launches(umlh_handle_t h, int64_t* out);

/* The one-launch step of a linear bf16 head (forward, dW, update as claimed tasks of ONE launch: finetune.py:180-195 without a
 * kernel boundary).  Its in-launch waits are bounded (50 ms).  A wait that gives up (the device is being starved by another
 * tenant, or a fault) sets a device status word; the step that hit it and every later step of the handle apply NO update from
 * that point (a gradient that timed out is never stepped) and umlh_step_status returns, in status_out[0..3], {code (0 = ok),
 * first task of the range waited on, launch tag, phase}.  It synchronises with the device.  The host mirror raises UmlhError
 * whenever it reads the step scalars and finds the code set.  umlh_step_launches: one-launch steps taken by this handle. */
int  umlh_step_status(umlh_handle_t h, int32_t* status_out);

/* Direct peer-to-peer all-reduce of the gradient message (csrc/umlh_p2p.hip): reduce-scatter + all-gather over exchange regions
 * that every rank allocates in its own HBM (umlh_p2p_alloc of umlh_p2p_region_bytes(n_floats of umlh_grad_buffer, n_ranks)),
 * exports (umlh_p2p_export: a 64-byte hipIpcMemHandle_t to hand to the peers over the caller's store) and maps from its peers
 * (umlh_p2p_open).  umlh_p2p_attach makes it the handle's transport in place of RCCL / the callback (linear heads; the sums are
 * formed in rank order, bit-identical on every rank).  No reference call site exists (the reference is single-GPU); SURVEY 8(e)
 * asks for it because a ring is per-link bound on the xGMI mesh.  Its waits give up after 30 s (a peer that never arrives) and
 * report through umlh_step_status (code 2).  UNMEASURED on a multi-GPU node: RCCL remains the default transport. */
uint64_t umlh_p2p_region_bytes(int64_t n_max_floats, int32_t n_ranks);
int  umlh_p2p_alloc(uint64_t bytes, void** out);
int  umlh_p2p_free(void* p);
int  umlh_p2p_export(void* p, void* handle64);
int  umlh_p2p_open(const void* handle64, void** out);
int  umlh_p2p_close(void* p);
int  umlh_p2p_attach(umlh_handle_t h, void* const* regions, int32_t n_ranks, int32_t rank);
int  umlh_step_launches(umlh_handle_t h, int64_t* out);

/* Data-parallel split of the step: gradients only, laid out as ONE flat fp32
 * buffer [g_head | g_proj | g_scales(2) | scalars(UMLH_N_SCALARS)] inside the
 * workspace, already divided by batch->global_rows so a SUM all-reduce over ranks
 * yields the single-GPU gradient; then the update from that buffer.  With the gradient diagnostics enabled
 * (d_shared % 8 == 0) the head part is [g_img | g_txt]: the two per-modality gradients travel separately and
 * umlh_apply_update forms dot / norms / sign agreement from the all-reduced pair (pass the step's alpha / img_alpha in
 * its hyper), so the diagnostics of a data-parallel step are those of the global batch. */
/* Data-parallel transport.  With more than one rank attached, umlh_train_steps runs every step as
 *   gradients -> SUM all-reduce -> identical update,
 * all enqueued from C on the caller's stream (no host work per step): equal shards are assumed (every rank passes the same
 * row counts per step; the CE means divide by rows x n_ranks).  For a 2-layer head the head gradient's all-reduce runs on
 * a second stream beside the img_proj backward GEMMs.  The reference is single-process; there is no
 * reference call site.
 *   umlh_comm_unique_id / umlh_comm_init_rank: an RCCL communicator owned by the handle (librccl.so.1 is loaded at run
 *     time; rank 0 draws the 128-byte id and hands it to the others, e.g. by a torch.distributed broadcast);
 *   umlh_set_comm: attach an existing ncclComm_t instead;
 *   umlh_set_allreduce: a custom transport -- fn(ctx, device_buf, n_floats, stream) must SUM-all-reduce the buffer in
 *     place and be ordered with `stream` (tests plug gloo through a host callback; n_ranks == 1 makes umlh_train_steps take
 *     the split path alone, for pricing). */
#define UMLH_COMM_ID_BYTES 128
typedef int (*umlh_allreduce_fn)(void* ctx, float* device_buf, uint64_t n_floats, void* stream);
int  umlh_comm_unique_id(void* id_out /* UMLH_COMM_ID_BYTES */);
int  umlh_comm_init_rank(umlh_handle_t h, const void* id, int32_t n_ranks, int32_t rank);
int  umlh_set_comm(umlh_handle_t h, void* nccl_comm, int32_t n_ranks);
int  umlh_set_allreduce(umlh_handle_t h, umlh_allreduce_fn fn, void* ctx, int32_t n_ranks);

int  umlh_grad_step(umlh_handle_t h, const umlh_batch_t* img, const umlh_batch_t* txt,
                    const umlh_hyper_t* hyper, void* stream);
int  umlh_grad_buffer(umlh_handle_t h, float** device_ptr, uint64_t* n_floats);
/* Diagnostic builds only (UMLH_DBG_FWD=9): per-workgroup cycle stamps of the forward kernel,
 * a region of the workspace no other code reads. */
int  umlh_debug_buffer(umlh_handle_t h, void** device_ptr, uint64_t* n_bytes);
int  umlh_apply_update(umlh_handle_t h, const umlh_hyper_t* hyper, float* scalars_out, void* stream);

/* validate() inner loop for one batch (finetune.py:295-308): logits -> argmax ->
 * CE, nothing leaves the device.  Writes scalars_out[UMLH_S_CORRECT] (count) and
 * scalars_out[UMLH_S_LOSS_SUM] (sum of row losses); the caller forms the
 * per-batch means (finetune.py:311-312). */
int  umlh_eval_batch(umlh_handle_t h, const umlh_batch_t* batch, float* scalars_out, void* stream);

/* Whole-table evaluation: the same forward as umlh_eval_batch over up to max_rows_img rows, but the per-row
 * results are kept -- row_stats[2*r] = CE of row r, row_stats[2*r + 1] = 1 if its first arg-max is the label.
 * validate() (finetune.py:291-315) forms its per-batch mean losses from these for ANY batch size, so a
 * validation pass is one launch per 4096-row slab instead of one per batch. */
int  umlh_eval_rows(umlh_handle_t h, const umlh_batch_t* rows, float* row_stats, void* stream);

/* Per-phase device timing of umlh_train_step / umlh_grad_step with HIP events recorded
 * on the step's stream (bench.py's roofline leg; rocprofv3 --kernel-trace must agree).
 * Phases: 0 img_proj forward GEMM, 1 fused forward+CE, 2 dW_head GEMM, 3 img_proj
 * backward GEMMs, 4 slab reduce + optimizer + scalars.  enable creates the events,
 * read synchronises on the last one and returns milliseconds of the latest step. */
#define UMLH_N_PHASES 5
int  umlh_profile_enable(umlh_handle_t h, int enable);
int  umlh_profile_read(umlh_handle_t h, float* ms_out /* host float[UMLH_N_PHASES] */);

/* fp32 -> bf16 (round-to-nearest-even) copy of n elements: builds the bf16 shadow of a
 * device-resident feature table once, for UMLH_PREC_BF16 handles. */
int  umlh_to_bf16(const float* src, void* dst_bf16, int64_t n, void* stream);

/* MultiBench alternation step (MultiBench/train.py:354-399, models.py:194-243): per-modality
 * decoder Linear(z -> D) fused with the masked next-step MSE, forward and backward.
 *   recon[b,t,:] = W z[b,t,:] + bias;  loss = sum_{t<T-1, t+1<len_b} |recon[b,t]-x[b,t+1]|^2 / (D*#live + 1e-8)
 * (T == 1: plain reconstruction MSE, models.py:209-210; lengths == NULL: unmasked mean).
 * z [B,T,Z], w [D,Z], bias [D], x [B,T,D] fp32 device; lengths int64[B] device or NULL.
 * recon [B,T,D] optional; dres [B*T*D] and row_partial [B*T] scratch that the backward reads;
 * loss_cnt device float[2] = {loss, denominator}.  Backward: grad_out device scalar (d/d loss);
 * dz [B,T,Z], dw [D,Z], db [D]; scratch: umlh_seq_mse_backward_scratch_floats(B,T,Z,D) device floats for the split-K
 * slabs of dW and the row-chunk partials of db (NULL: one workgroup walks all B*T rows per output tile). */
int  umlh_seq_mse_forward(const float* z, const float* w, const float* bias, const float* x, const int64_t* lengths,
                          int32_t B, int32_t T, int32_t Z, int32_t D, float* recon, float* dres, float* row_partial,
                          float* loss_cnt, void* stream);
int  umlh_seq_mse_backward(const float* z, const float* w, const float* dres, const float* loss_cnt, const float* grad_out,
                           int32_t B, int32_t T, int32_t Z, int32_t D, float* dz, float* dw, float* db, float* scratch,
                           void* stream);
uint64_t umlh_seq_mse_backward_scratch_floats(int32_t B, int32_t T, int32_t Z, int32_t D);

/* SequenceInfoNCELoss (MultiBench/models.py:145-175), the contrastive alternative to the next-step MSE: over the n valid
 * (batch, time) rows, logits = normalize(pred) normalize(target)^T / temperature, labels = the diagonal, loss = mean CE.
 * pred / target [n, D] dense fp32 rows (the caller selects the valid rows, as the reference's boolean indexing does).
 * Forward leaves: pred_hat, target_hat [n, D] (unit rows), pred_norm [n], probs [n, n] = softmax(logits) - I, row_loss [n],
 * loss (device scalar).  Backward: grad_out device scalar; dhat [n, D] scratch; dpred [n, D] = d loss / d pred * grad_out.
 * (No gradient flows to the targets: they are the model's inputs.) */
int  umlh_infonce_forward(const float* pred, const float* target, int32_t n, int32_t D, float temperature, float* pred_hat,
                          float* target_hat, float* pred_norm, float* probs, float* row_loss, float* loss, void* stream);
int  umlh_infonce_backward(const float* pred_hat, const float* target_hat, const float* pred_norm, const float* probs,
                           const float* grad_out, int32_t n, int32_t D, float temperature, float* dhat, float* dpred, void* stream);

/* ---- MultiBench shared encoder (MultiBench/models.py:39-127: Conv1d k=1 -> positions -> 5 x post-norm
 * nn.TransformerEncoderLayer(z, nhead, dim_feedforward=2048, relu, dropout) under a causal + key-padding
 * mask), forward and backward, fp32.  Token rows are m = t*B + b of a [T,B,*] activation (torch's
 * sequence-first layout).  The host mirror (multibench/encoder.py) strings these together exactly as
 * torch.nn.TransformerEncoderLayer.forward does; all pointers are device fp32 unless noted. ---- */

/* out[m][n] = alpha * sum_k A(m,k) B(n,k).  ta = 0: A[m*lda + k], rows optionally gathered by a_rows[m];
 * ta = 1: A[k*lda + m].  tb = 0: B[n*ldb + k];  tb = 1: B[k*ldb + n], rows optionally gathered by k_rows[k].
 * (ta,tb) in {(0,0),(0,1),(1,1)}: y = x W^T, dx = dy W, dW = dy^T x.  fp32 MFMA (exact fp32 products).
 * splits > 1: split-K over `splits` slabs of M*ldo floats in `slabs` (caller scratch), summed into out in slab
 * order -- the encoder's GEMMs have few output tiles and a long K, so one tile per CU is latency-bound. */
int  umlh_gemm_f32(const float* A, const float* B, float* out, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                   int32_t ldo, int32_t ta, int32_t tb, const int64_t* a_rows, const int64_t* k_rows, float alpha,
                   int32_t splits, float* slabs, void* stream);
/* y[m][n] = act(y[m][n] + bias[n]) in place (bias may be NULL; relu != 0: max(.,0)) */
int  umlh_bias_act(float* y, const float* bias, int64_t M, int32_t N, int32_t relu, void* stream);
/* dy[i] = y[i] > 0 ? dy[i] : 0 in place */
int  umlh_relu_backward(const float* y, float* dy, int64_t n, void* stream);
/* x[i] = keep_i ? x[i]/(1-p) : 0 in place; keep_i is a pure function of (seed, i): the same call on the
 * gradient is the backward pass.  p == 0: no-op. */
int  umlh_dropout(float* x, int64_t n, float p, uint64_t seed, void* stream);
/* y[i] += x[i]  (gradient fan-in of a residual branch) */
int  umlh_add_inplace(float* y, const float* x, int64_t n, void* stream);
/* out[n] = sum_m x[m][n]  (bias gradients) */
int  umlh_colsum(const float* x, int32_t M, int32_t N, float* out, void* stream);
/* s = x + r (r may be NULL); y = LayerNorm(s; gamma, beta, eps); mean/rstd [M] saved for the backward */
int  umlh_add_layernorm_forward(const float* x, const float* r, const float* gamma, const float* beta, int32_t M, int32_t N,
                                float eps, float* s, float* y, float* mean, float* rstd, void* stream);
/* ds [M,N] (gradient of both summands of s), dgamma [N], dbeta [N] */
int  umlh_layernorm_backward(const float* dy, const float* s, const float* gamma, const float* mean, const float* rstd,
                             int32_t M, int32_t N, float* ds, float* dgamma, float* dbeta, void* stream);
/* x[t,b,:] += pos[t,:]  /  dpos[t,:] = sum_b dx[t,b,:] */
int  umlh_add_positions(float* x, const float* pos, int32_t T, int32_t B, int32_t Z, void* stream);
int  umlh_positions_backward(const float* dx, int32_t T, int32_t B, int32_t Z, float* dpos, void* stream);
/* scatter == 0: out[j,:] = x[idx[j],:];  scatter != 0: out[idx[j],:] = x[j,:] (out pre-zeroed, idx unique) */
int  umlh_gather_rows(const float* x, const int64_t* idx, int32_t n, int32_t Z, float* out, int32_t scatter, void* stream);
/* Causal multi-head self-attention with key padding over packed in-projections qkv [T,B,3Z] (torch
 * MultiheadAttention layout), H heads; lengths int64[B] or NULL.  ctx [T,B,Z] (heads concatenated, before
 * out_proj), lse [B,H,T] saved for the backward.  p/seed: attention-probability dropout.  T <= 128, Z/H <= 64. */
int  umlh_attention_forward(const float* qkv, const int64_t* lengths, int32_t T, int32_t B, int32_t Z, int32_t H, float p,
                            uint64_t seed, float* ctx, float* lse, void* stream);
int  umlh_attention_backward(const float* qkv, const int64_t* lengths, const float* lse, const float* dctx, int32_t T, int32_t B,
                             int32_t Z, int32_t H, float p, uint64_t seed, float* dqkv, void* stream);

/* One whole post-norm nn.TransformerEncoderLayer (MultiBench/models.py:57-60: d_model = Z, nhead = H, dim_feedforward = d_ff,
 * relu, dropout p, causal + key-padding mask), forward / backward, as ONE call each: the launch sequence the host mirror
 * used to drive op by op is enqueued from C (same kernels, same arithmetic, same dropout seeds: seed for the attention
 * probabilities, seed+1 / +2 / +3 for dropout1 / the FFN dropout / dropout2).
 *   params[12] = in_proj_weight [3Z,Z], in_proj_bias, out_proj.weight [Z,Z], out_proj.bias, linear1.weight [d_ff,Z],
 *                linear1.bias, linear2.weight [Z,d_ff], linear2.bias, norm1.weight, norm1.bias, norm2.weight, norm2.bias
 *   h_in / h_out [T*B, Z] token rows m = t*B + b;  saved: caller buffer of umlh_encoder_layer_saved_floats(cfg) floats that
 *   the backward of the same layer reads;  scratch: umlh_encoder_layer_scratch_floats(cfg) floats, reusable across layers.
 *   backward: dh_out = d loss / d h_out -> grads[12] (same order and shapes as params) and dh_in. */
typedef struct {
    int32_t T, B, Z, H, d_ff;
    float   p, eps;
    uint64_t seed;
    const uint64_t* seed_device;   /* optional device word added to `seed` by every kernel (NULL = none): lets a launch sequence
                                    * captured in a HIP graph draw fresh dropout masks at every replay */
} umlh_enc_layer_t;
uint64_t umlh_encoder_layer_saved_floats(const umlh_enc_layer_t* cfg);
uint64_t umlh_encoder_layer_scratch_floats(const umlh_enc_layer_t* cfg);
int  umlh_encoder_layer_forward(const umlh_enc_layer_t* cfg, const float* const* params, const float* h_in, const int64_t* lengths,
                                float* saved, float* scratch, float* h_out, void* stream);
int  umlh_encoder_layer_backward(const umlh_enc_layer_t* cfg, const float* const* params, const float* h_in, const int64_t* lengths,
                                 const float* saved, const float* dh_out, float* scratch, float* const* grads, float* dh_in,
                                 void* stream);

/* The whole stack of n_layers identical layers in one call each way.  P / G: 12 pointers per layer in the order above.
 * forward: layer li reads (li ? h + (li-1)*M*Z : h0) and writes h + li*M*Z (M = T*B) and saved + li*saved_floats; its dropout
 * streams start at cfg->seed + 7919*li.  backward: dh_out = gradient of the last layer's output, dh = two [M,Z] scratch
 * buffers, the gradient of h0 is left in dh0. */
int  umlh_encoder_stack_forward(const umlh_enc_layer_t* cfg, int32_t n_layers, const float* const* P, const float* h0,
                                const int64_t* lengths, float* saved, float* scratch, float* h, void* stream);
int  umlh_encoder_stack_backward(const umlh_enc_layer_t* cfg, int32_t n_layers, const float* const* P, const float* h0,
                                 const int64_t* lengths, const float* saved, const float* h, const float* dh_out, float* scratch,
                                 float* const* G, float* dh, float* dh0, void* stream);

/* Encoder plan: the layer stack bound to fixed buffers so that its launch sequences replay from HIP graphs (one graph launch per
 * forward / backward instead of 7 + 16 kernel launches per layer: at MOSEI sizes the alternation step is bound by the host's
 * launch rate).  The caller owns `workspace` (umlh_encoder_plan_floats(cfg, n_layers) device floats, alive as long as the plan)
 * and the parameter tensors P (12 per layer; their ADDRESSES are baked into the graphs -- make a new plan when they move).
 * Buffers inside the workspace (float offsets from umlh_encoder_plan_offsets):
 *   [0] h0 [T*B, Z]    input token rows, written by the caller before forward
 *   [1] lengths        int64[B] key-padding lengths (has_lengths != 0), written by the caller before forward
 *   [2] h_last [T*B,Z] output of the last layer
 *   [3] dh_out [T*B,Z] gradient of h_last, written by the caller before backward
 *   [4] dh0 [T*B, Z]   gradient of h0 after backward
 *   [5] grads          gradients of all parameters after backward, flat in P order
 * forward: seed = base of this pass's dropout streams (cfg->seed is ignored).  The first call of each direction runs the plain
 * launch sequence, the second captures it, later calls replay.  A plan serves ONE forward/backward pair at a time and one
 * stream at a time. */
typedef struct umlh_enc_plan_s* umlh_enc_plan_t;
uint64_t umlh_encoder_plan_floats(const umlh_enc_layer_t* cfg, int32_t n_layers);
int  umlh_encoder_plan_create(const umlh_enc_layer_t* cfg, int32_t n_layers, const float* const* P, int32_t has_lengths,
                              float* workspace, umlh_enc_plan_t* out);
int  umlh_encoder_plan_offsets(umlh_enc_plan_t plan, uint64_t offsets[6]);
int  umlh_encoder_plan_forward(umlh_enc_plan_t plan, uint64_t seed, void* stream);
int  umlh_encoder_plan_backward(umlh_enc_plan_t plan, void* stream);
void umlh_encoder_plan_destroy(umlh_enc_plan_t plan);

/* A pseudo-random permutation of 0..n-1 written as int64 (device), keyed by seed: 4-round Feistel
 * network + cycle walking, no sort.  Epoch shuffles for throughput runs; NOT the reference's
 * sampler order (that is reproduced host-side by the loader, finetune.py:370-371). */
int  umlh_random_permutation(int64_t n, uint64_t seed, int64_t* out, void* stream);

/* Standalone optimizer.step() for one parameter tensor from a caller-computed
 * gradient (engine/optimizer/optim.py:34-71; torch.optim single-tensor recurrences):
 * the same update kernel the fused step applies.  v may be NULL for SGD. */
int  umlh_optimizer_step(int32_t optimizer, float* param, const float* grad, float* m, float* v, int64_t n,
                         double lr, int64_t step, double beta1, double beta2, double eps, double momentum,
                         double weight_decay, void* stream);
/* The same update for MANY parameter tensors in one launch per 48 tensors (optimizer.step() over all 70 tensors of the
 * MultiBench model, MultiBench/main.py:122): host arrays of n_tensors device pointers / element counts; v may be NULL for SGD. */
int  umlh_optimizer_step_multi(int32_t optimizer, int32_t n_tensors, float* const* params, const float* const* grads,
                               float* const* m, float* const* v, const int64_t* n, double lr, int64_t step, double beta1,
                               double beta2, double eps, double momentum, double weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UMLH_H */
