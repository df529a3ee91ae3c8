#!/usr/bin/env python3
"""One encoder layer forward + backward through the C ABI on random tensors (MOSEI sizes), for kernel-level timing under
rocprofv3:  scripts/enc_layer_probe.py [z] [p] [iters]."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
import torch  # noqa: E402
import umlh  # noqa: E402
from umlh._lib import EncLayer, check  # noqa: E402


def main(z=40, p=0.1, iters=50, T=50, B=32, H=5, F=2048):
    dev = "cuda:0"
    lib = umlh.load_library()
    g = torch.Generator(device=dev).manual_seed(0)
    M = T * B
    shapes = [(3 * z, z), (3 * z,), (z, z), (z,), (F, z), (F,), (z, F), (z,), (z,), (z,), (z,), (z,)]
    P = [torch.randn(*s, generator=g, device=dev) * 0.1 for s in shapes]
    G = [torch.empty_like(t) for t in P]
    lc = EncLayer(T, B, z, H, F, float(p), 1e-5, 1234)
    saved = torch.empty(int(lib.umlh_encoder_layer_saved_floats(C.byref(lc))), device=dev)
    scratch = torch.empty(int(lib.umlh_encoder_layer_scratch_floats(C.byref(lc))), device=dev)
    h_in = torch.randn(M, z, generator=g, device=dev)
    h_out, dh_out, dh_in = torch.empty_like(h_in), torch.randn(M, z, generator=g, device=dev), torch.empty_like(h_in)
    lens = torch.randint(5, T + 1, (B,), generator=g, device=dev)
    pa = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    vp = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(iters):
        check(lib.umlh_encoder_layer_forward(C.byref(lc), pa(P), vp(h_in), vp(lens), vp(saved), vp(scratch), vp(h_out), st), "fwd")
        check(lib.umlh_encoder_layer_backward(C.byref(lc), pa(P), vp(h_in), vp(lens), vp(saved), vp(dh_out), vp(scratch), pa(G), vp(dh_in), st), "bwd")
    torch.cuda.synchronize()
    print("ok", float(h_out.abs().mean()), float(dh_in.abs().mean()))


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 40, float(a[1]) if len(a) > 1 else 0.1, int(a[2]) if len(a) > 2 else 50)
