"""ctypes binding of include/umlh.h + the in-tree hipcc build of libumlh.so."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")
_INCLUDE = os.path.join(_ROOT, "include")
_SO = os.path.join(_HERE, "libumlh.so")

OPT_IDS = {"sgd": 0, "adam": 1, "adamw": 2}          # engine/optimizer/optim.py:6 AVAI_OPTIMS
PREC_IDS = {"fp32": 0, "bf16": 1}
(S_LOSS_IMG, S_LOSS_TXT, S_ACC_IMG, S_ACC_TXT, S_GSCALE_IMG, S_GSCALE_TXT, S_CORRECT, S_LOSS_SUM,
 S_GRAD_DOT, S_GRAD_N2_IMG, S_GRAD_N2_TXT, S_GRAD_AGREE) = range(12)
N_CORE_SCALARS = 8
N_SCALARS = 12

SOURCES = ["umlh_p2p.hip", "umlh_kernels_f32.hip", "umlh_kernels_bf16.hip", "umlh_kernels_micro.hip", "umlh_kernels_seq.hip", "umlh_kernels_enc.hip", "umlh_api.cpp", "umlh_encoder.cpp"]
EXPORTS = ["umlh_last_error", "umlh_version", "umlh_enable_diagnostics", "umlh_set_diagnostic_columns", "umlh_freeze_proj_row", "umlh_workspace_bytes", "umlh_create", "umlh_destroy", "umlh_bind",
           "umlh_zero_shot_init", "umlh_logits", "umlh_train_step", "umlh_grad_step", "umlh_grad_buffer",
           "umlh_apply_update", "umlh_eval_batch", "umlh_eval_rows", "umlh_project", "umlh_optimizer_step",
           "umlh_profile_enable", "umlh_profile_read", "umlh_to_bf16",
           "umlh_train_steps", "umlh_train_steps_grouped", "umlh_micro_status", "umlh_micro_launches", "umlh_step_status", "umlh_step_launches", "umlh_p2p_region_bytes", "umlh_p2p_alloc", "umlh_p2p_free", "umlh_p2p_export", "umlh_p2p_open", "umlh_p2p_close", "umlh_p2p_attach", "umlh_comm_unique_id", "umlh_comm_init_rank",
           "umlh_set_comm", "umlh_set_allreduce", "umlh_seq_mse_forward", "umlh_seq_mse_backward", "umlh_seq_mse_backward_scratch_floats", "umlh_infonce_forward", "umlh_infonce_backward",
           "umlh_random_permutation", "umlh_debug_buffer",
           "umlh_gemm_f32", "umlh_add_inplace", "umlh_bias_act", "umlh_relu_backward", "umlh_dropout", "umlh_colsum",
           "umlh_add_layernorm_forward", "umlh_layernorm_backward", "umlh_add_positions", "umlh_positions_backward",
           "umlh_gather_rows", "umlh_attention_forward", "umlh_attention_backward", "umlh_optimizer_step_multi",
           "umlh_encoder_layer_saved_floats", "umlh_encoder_layer_scratch_floats", "umlh_encoder_layer_forward",
           "umlh_encoder_layer_backward", "umlh_encoder_stack_forward", "umlh_encoder_stack_backward",
           "umlh_encoder_plan_floats", "umlh_encoder_plan_create", "umlh_encoder_plan_offsets", "umlh_encoder_plan_forward",
           "umlh_encoder_plan_backward", "umlh_encoder_plan_destroy"]


class UmlhError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("d_img", C.c_int32), ("d_shared", C.c_int32), ("num_classes", C.c_int32),
                ("has_proj", C.c_int32), ("learnable_temp", C.c_int32), ("optimizer", C.c_int32),
                ("precision", C.c_int32), ("max_rows_img", C.c_int32), ("max_rows_txt", C.c_int32),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("momentum", C.c_double),
                ("weight_decay", C.c_double)]


class Buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_head", "m_head", "v_head", "w_proj", "m_proj", "v_proj",
                                          "scales", "m_scales", "v_scales", "workspace")] + \
               [("workspace_bytes", C.c_uint64)]


class Batch(C.Structure):
    _fields_ = [("feats", C.c_void_p), ("labels", C.c_void_p), ("index", C.c_void_p),
                ("rows", C.c_int32), ("global_rows", C.c_int32), ("feats_bf16", C.c_void_p)]


class Stream(C.Structure):
    _fields_ = [("feats", C.c_void_p), ("feats_bf16", C.c_void_p), ("labels", C.c_void_p), ("index", C.c_void_p),
                ("offsets", C.POINTER(C.c_int32))]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)   # umlh_allreduce_fn
COMM_ID_BYTES = 128


class GroupItem(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("img", C.POINTER(Stream)), ("txt", C.POINTER(Stream)),
                ("lr", C.POINTER(C.c_double)), ("first_step", C.c_int64), ("alpha", C.c_float), ("img_alpha", C.c_float),
                ("scalars_out", C.c_void_p)]


class EncLayer(C.Structure):
    _fields_ = [("T", C.c_int32), ("B", C.c_int32), ("Z", C.c_int32), ("H", C.c_int32), ("d_ff", C.c_int32),
                ("p", C.c_float), ("eps", C.c_float), ("seed", C.c_uint64), ("seed_device", C.c_void_p)]


class Hyper(C.Structure):
    _fields_ = [("lr", C.c_double), ("step", C.c_int64), ("alpha", C.c_float), ("img_alpha", C.c_float),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


def lib_path() -> str:
    return _SO


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Cross-compile the HIP sources for gfx950 into umlh/libumlh.so (in-tree, so the
    binary travels with the repo snapshot to the GPU box)."""
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(_CSRC, "umlh_common.h"), os.path.join(_CSRC, "umlh_micro.h"), os.path.join(_INCLUDE, "umlh.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return _SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", _INCLUDE, "-I", _CSRC, *srcs, "-o", _SO + ".tmp"]
    if os.environ.get("UMLH_BUILD_ABLATIONS") == "1":     # kernel-analysis build: timing-only work-skipping switches
        cmd.insert(1, "-DUMLH_ABLATIONS")
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise UmlhError(f"hipcc failed ({res.returncode}):\n{res.stderr[-4000:]}")
    os.replace(_SO + ".tmp", _SO)
    return _SO


_LIB = None


def load_library():
    """dlopen libumlh.so and declare every prototype of include/umlh.h.  Raises
    UmlhError if the extension has not been built: there is no fallback path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise UmlhError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). The HIP extension is mandatory; there is no CPU fallback.")
    lib = C.CDLL(_SO)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    lib.umlh_last_error.restype = C.c_char_p
    lib.umlh_last_error.argtypes = []
    lib.umlh_version.restype = C.c_int
    lib.umlh_workspace_bytes.restype = u64
    lib.umlh_workspace_bytes.argtypes = [C.POINTER(Config)]
    lib.umlh_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.umlh_destroy.argtypes = [vp]
    lib.umlh_bind.argtypes = [vp, C.POINTER(Buffers)]
    lib.umlh_enable_diagnostics.argtypes = [vp, C.c_int32]
    lib.umlh_set_diagnostic_columns.argtypes = [vp, C.c_int32]
    lib.umlh_freeze_proj_row.argtypes = [vp, C.c_int32]
    lib.umlh_eval_rows.argtypes = [vp, C.POINTER(Batch), vp, vp]
    i32, f32 = C.c_int32, C.c_float
    lib.umlh_gemm_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, f32, i32, vp, vp]
    lib.umlh_bias_act.argtypes = [vp, vp, i64, i32, i32, vp]
    lib.umlh_add_inplace.argtypes = [vp, vp, i64, vp]
    lib.umlh_relu_backward.argtypes = [vp, vp, i64, vp]
    lib.umlh_dropout.argtypes = [vp, i64, f32, u64, vp]
    lib.umlh_colsum.argtypes = [vp, i32, i32, vp, vp]
    lib.umlh_add_layernorm_forward.argtypes = [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp]
    lib.umlh_layernorm_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.umlh_add_positions.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.umlh_positions_backward.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.umlh_gather_rows.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    lib.umlh_attention_forward.argtypes = [vp, vp, i32, i32, i32, i32, f32, u64, vp, vp, vp]
    lib.umlh_attention_backward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp, vp]
    lib.umlh_zero_shot_init.argtypes = [vp, vp, vp, i64, vp]
    lib.umlh_logits.argtypes = [vp, C.POINTER(Batch), C.c_int, vp, vp]
    lib.umlh_train_step.argtypes = [vp, C.POINTER(Batch), C.POINTER(Batch), C.POINTER(Hyper), vp, vp]
    lib.umlh_grad_step.argtypes = [vp, C.POINTER(Batch), C.POINTER(Batch), C.POINTER(Hyper), vp]
    lib.umlh_grad_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.umlh_apply_update.argtypes = [vp, C.POINTER(Hyper), vp, vp]
    lib.umlh_eval_batch.argtypes = [vp, C.POINTER(Batch), vp, vp]
    lib.umlh_project.argtypes = [vp, C.POINTER(Batch), vp, vp]
    lib.umlh_optimizer_step.argtypes = [i32, vp, vp, vp, vp, i64, C.c_double, i64, C.c_double, C.c_double,
                                        C.c_double, C.c_double, C.c_double, vp]
    lib.umlh_to_bf16.argtypes = [vp, vp, i64, vp]
    pv = C.POINTER(vp)
    lib.umlh_optimizer_step_multi.argtypes = [i32, i32, pv, pv, pv, pv, C.POINTER(i64), C.c_double, i64, C.c_double, C.c_double,
                                              C.c_double, C.c_double, C.c_double, vp]
    lib.umlh_encoder_layer_saved_floats.restype = u64
    lib.umlh_encoder_layer_saved_floats.argtypes = [C.POINTER(EncLayer)]
    lib.umlh_encoder_layer_scratch_floats.restype = u64
    lib.umlh_encoder_layer_scratch_floats.argtypes = [C.POINTER(EncLayer)]
    lib.umlh_encoder_layer_forward.argtypes = [C.POINTER(EncLayer), pv, vp, vp, vp, vp, vp, vp]
    lib.umlh_encoder_layer_backward.argtypes = [C.POINTER(EncLayer), pv, vp, vp, vp, vp, vp, pv, vp, vp]
    lib.umlh_encoder_stack_forward.argtypes = [C.POINTER(EncLayer), i32, pv, vp, vp, vp, vp, vp, vp]
    lib.umlh_encoder_stack_backward.argtypes = [C.POINTER(EncLayer), i32, pv, vp, vp, vp, vp, vp, vp, pv, vp, vp, vp]
    lib.umlh_encoder_plan_floats.restype = u64
    lib.umlh_encoder_plan_floats.argtypes = [C.POINTER(EncLayer), i32]
    lib.umlh_encoder_plan_create.argtypes = [C.POINTER(EncLayer), i32, pv, i32, vp, C.POINTER(vp)]
    lib.umlh_encoder_plan_offsets.argtypes = [vp, C.POINTER(u64)]
    lib.umlh_encoder_plan_forward.argtypes = [vp, u64, vp]
    lib.umlh_encoder_plan_backward.argtypes = [vp, vp]
    lib.umlh_encoder_plan_destroy.restype = None
    lib.umlh_encoder_plan_destroy.argtypes = [vp]
    lib.umlh_train_steps.argtypes = [vp, C.POINTER(Stream), C.POINTER(Stream), i32, C.POINTER(C.c_double), i64,
                                     C.c_float, C.c_float, vp, vp]
    lib.umlh_train_steps_grouped.argtypes = [C.POINTER(GroupItem), i32, i32, vp]
    lib.umlh_micro_status.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.umlh_micro_launches.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.umlh_step_status.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.umlh_step_launches.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.umlh_p2p_region_bytes.restype = u64
    lib.umlh_p2p_region_bytes.argtypes = [i64, i32]
    lib.umlh_p2p_alloc.argtypes = [u64, C.POINTER(vp)]
    lib.umlh_p2p_free.argtypes = [vp]
    lib.umlh_p2p_export.argtypes = [vp, vp]
    lib.umlh_p2p_open.argtypes = [vp, C.POINTER(vp)]
    lib.umlh_p2p_close.argtypes = [vp]
    lib.umlh_p2p_attach.argtypes = [vp, C.POINTER(vp), i32, i32]
    lib.umlh_comm_unique_id.argtypes = [vp]
    lib.umlh_comm_init_rank.argtypes = [vp, vp, i32, i32]
    lib.umlh_set_comm.argtypes = [vp, vp, i32]
    lib.umlh_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, i32]
    lib.umlh_seq_mse_forward.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.umlh_seq_mse_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.umlh_infonce_forward.argtypes = [vp, vp, i32, i32, C.c_float, vp, vp, vp, vp, vp, vp, vp]
    lib.umlh_infonce_backward.argtypes = [vp, vp, vp, vp, vp, i32, i32, C.c_float, vp, vp, vp]
    lib.umlh_seq_mse_backward_scratch_floats.restype = u64
    lib.umlh_seq_mse_backward_scratch_floats.argtypes = [i32, i32, i32, i32]
    lib.umlh_random_permutation.argtypes = [i64, u64, vp, vp]
    lib.umlh_debug_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.umlh_profile_enable.argtypes = [vp, C.c_int]
    lib.umlh_profile_read.argtypes = [vp, C.POINTER(C.c_float)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("umlh_last_error", "umlh_workspace_bytes", "umlh_encoder_layer_saved_floats", "umlh_encoder_layer_scratch_floats"):
            fn.restype = C.c_int
    _LIB = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_library().umlh_last_error()
        raise UmlhError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
