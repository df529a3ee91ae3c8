// Descriptor of one head of a micro-step launch (umlh_kernels_micro.hip); filled by umlh_api.cpp, read by the kernel
// from a device array.
#pragma once
#include "umlh_common.h"

#define UMLH_MICRO_CS        16     /* classes per workgroup (one 16-row MFMA tile)                       */
#define UMLH_MICRO_MAX_ROWS  64     /* row slots per step: ceil(rows_img/16) + ceil(rows_txt/16) <= 4 tiles */
#define UMLH_MICRO_MAX_STEPS 512    /* steps per launch (per-step tables live in the workspace)           */
#define UMLH_MICRO_MAX_HEADS 64     /* heads per launch                                                   */

struct UmlhMicroHead {
    const float*   feats[2];        // device [table_rows, d] fp32: image / text table
    const int64_t* labels[2];
    const int64_t* index[2];        // concatenated per-step row ids
    const int*     offs[2];         // device int[n_steps + 1] (NULL = modality absent)
    float* w; float* m; float* v;   // [C, d] head weight and optimizer moments (v unused for SGD)
    float* scales; float* m_scales; float* v_scales;   // [2]
    const OptArgs* opt;             // device [n_steps]: the optimizer scalars of each step (lr schedule folded in)
    float* scalars_out;             // device [n_steps][UMLH_N_SCALARS] or NULL
    unsigned long long* xchg;       // [2 parities][nwg][5 fields][64 rows] granules {epoch << 32 | value bits}: max, sum exp,
                                    // raw label logit, first arg-max, sum exp * raw (learnable logit scales only)
    unsigned* status;               // 0 = ok; 1 + step: a bounded wait of that step gave up (all slices of the head abort)
    unsigned long long* stamps;     // diagnostics (UMLH_DBG_MICRO=1): [nwg][8] cycle sums per phase of wave 0, else NULL
    unsigned epoch0;                // epochs of this launch are epoch0 + 1 .. epoch0 + n_steps
    int   C, d, nwg, wg0;           // classes, width, workgroups (class slices) of this head, first block in the grid
    int   learnable, opt_kind;
    float w_img, w_txt;             // loss weights (img_alpha, alpha)
};
