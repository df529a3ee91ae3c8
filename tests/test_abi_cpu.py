"""CPU-side checks of the C-ABI shared library: it builds for gfx950, loads, and
exports every symbol include/umlh.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import umlh
    umlh.build_library()
    return umlh.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "umlh.h")).read()
    declared = set(re.findall(r"\b(umlh_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), name


def test_version_and_workspace_query(lib):
    from umlh._lib import Config
    assert lib.umlh_version() >= 1
    ok = Config(512, 512, 1000, 0, 0, 2, 0, 4096, 4096, 0.9, 0.999, 1e-8, 0.9, 0.01)
    assert lib.umlh_workspace_bytes(C.byref(ok)) > 1000 * 8192 * 4
    bad = Config(512, 256, 1000, 0, 0, 2, 0, 4096, 4096, 0.9, 0.999, 1e-8, 0.9, 0.01)   # dims differ w/o proj
    assert lib.umlh_workspace_bytes(C.byref(bad)) == 0
    too_many = Config(64, 64, 2000, 0, 0, 2, 0, 32, 32, 0.9, 0.999, 1e-8, 0.9, 0.0)
    assert lib.umlh_workspace_bytes(C.byref(too_many)) == 0


def test_null_and_unbound_calls_fail_with_message(lib):
    assert lib.umlh_create(None, None) < 0
    assert b"umlh_create" in lib.umlh_last_error()
    assert lib.umlh_train_step(None, None, None, None, None, None) < 0
    assert b"not bound" in lib.umlh_last_error()


def test_no_gpu_is_a_loud_error():
    import torch
    import umlh
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(umlh.UmlhError):
        umlh.HeadEngine(8, 8, 4, device="cpu")


def test_encoder_ops_validate_arguments_before_touching_the_gpu(lib):
    """The MultiBench encoder ops reject bad arguments with a message (no HIP call is made on these paths)."""
    fake = C.c_void_p(64)                                   # never dereferenced: the shape checks come first
    assert lib.umlh_gemm_f32(None, None, None, 4, 4, 4, 4, 4, 4, 0, 0, None, None, C.c_float(1.0), 1, None, None) != 0
    assert b"umlh_gemm_f32" in lib.umlh_last_error()
    assert lib.umlh_gemm_f32(fake, fake, fake, 4, 4, 4, 4, 4, 4, 0, 0, None, None, C.c_float(1.0), 3, None, None) != 0   # splits without slabs
    rc = lib.umlh_attention_forward(fake, None, 200, 2, 20, 5, C.c_float(0.0), C.c_uint64(0), fake, fake, None)
    assert rc != 0 and b"envelope" in lib.umlh_last_error()
    rc = lib.umlh_attention_backward(fake, None, fake, fake, 16, 2, 330, 5, C.c_float(0.0), C.c_uint64(0), fake, None)   # head dim 66
    assert rc != 0 and b"envelope" in lib.umlh_last_error()
    assert lib.umlh_dropout(fake, 10, C.c_float(1.5), C.c_uint64(1), None) != 0
    assert lib.umlh_eval_rows(None, None, None, None) != 0
