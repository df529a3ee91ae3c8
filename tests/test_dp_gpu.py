"""GPU: two ranks sharing the one MI355X of the test box (gloo transport, HIP engine):
umlh_grad_step -> all_reduce -> umlh_apply_update equals the single-process fused step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case():
    rng = np.random.default_rng(3)
    d, C, B = 64, 100, 96
    xi = rng.standard_normal((B, d)).astype(np.float32)
    xt = rng.standard_normal((B, d)).astype(np.float32)
    xi /= np.linalg.norm(xi, axis=1, keepdims=True)
    xt /= np.linalg.norm(xt, axis=1, keepdims=True)
    w = rng.standard_normal((C, d)).astype(np.float32)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    return d, C, B, xi, rng.integers(0, C, B), xt, rng.integers(0, C, B), w


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import umlh
        d, C, B, xi, yi, xt, yt, w = _case()
        dev = "cuda:0"
        e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=128, max_rows_txt=128, device=dev)
        e.w_head.copy_(torch.from_numpy(w))
        e.scales.fill_(50.0)
        T = lambda a, t: torch.as_tensor(a).to(dev, t).contiguous()
        idx = torch.arange(rank, B, world, device=dev)
        st = umlh.DataParallelStepper(e)
        scal = torch.zeros(umlh.N_SCALARS, device=dev)
        for k in range(2):
            st.step(umlh.RowBatch(T(xi, torch.float32), T(yi, torch.int64), idx),
                    umlh.RowBatch(T(xt, torch.float32), T(yt, torch.int64), idx), lr=1e-3, step=k + 1, scalars_out=scal)
        torch.cuda.synchronize()
        q.put((rank, e.w_head.cpu().numpy(), scal.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_matches_single_rank():
    import torch.multiprocessing as mp
    import umlh
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, C, B, xi, yi, xt, yt, w = _case()
    dev = "cuda:0"
    e = umlh.HeadEngine(d, d, C, optimizer="adamw", weight_decay=0.01, max_rows_img=128, max_rows_txt=128, device=dev)
    e.w_head.copy_(torch.from_numpy(w))
    e.scales.fill_(50.0)
    T = lambda a, t: torch.as_tensor(a).to(dev, t).contiguous()
    scal = torch.zeros(umlh.N_SCALARS, device=dev)
    for k in range(2):
        e.train_step(umlh.RowBatch(T(xi, torch.float32), T(yi, torch.int64)),
                     umlh.RowBatch(T(xt, torch.float32), T(yt, torch.int64)), lr=1e-3, step=k + 1, scalars_out=scal)
    ref = e.w_head.cpu().numpy()
    for rank, wr, sc in res:
        diff = np.abs(wr - ref)
        assert (diff > 1e-6 + 1e-5 * np.abs(ref)).mean() < 1e-3 and diff.max() < 5e-3   # Adam sign flips at |g|~eps
        np.testing.assert_allclose(sc[:4], scal.cpu().numpy()[:4], atol=1e-5)
    np.testing.assert_array_equal(res[0][1], res[1][1])


def _worker_indexed(rank, world, port, q, precision):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import umlh
        rng = np.random.default_rng(8)
        d, C, n = 128, 50, 256
        x = rng.standard_normal((n, d)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        y = rng.integers(0, C, n)
        w = (rng.standard_normal((C, d)) * 0.1).astype(np.float32)
        dev = "cuda:0"
        e = umlh.HeadEngine(d, d, C, optimizer="sgd", weight_decay=0.0, max_rows_img=128, max_rows_txt=128,
                            precision=precision, device=dev)
        e.w_head.copy_(torch.from_numpy(w))
        e.scales.fill_(20.0)
        xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
        tab = (xt, yt, umlh.to_bf16(xt)) if precision == "bf16" else (xt, yt)
        e.bind_tables(tab, tab)
        st = umlh.DataParallelStepper(e)
        for k in range(3):     # rank r takes rows r::world of a fixed 128-row batch; weights_unchanged path after step 0
            idx = torch.arange(rank + 64 * k, 64 * k + 128, world, device=dev)
            st.step_indexed(idx, idx, lr=0.05, step=k + 1)
        torch.cuda.synchronize()
        q.put((rank, e.w_head.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_indexed_dp_path_two_ranks_vs_single(precision):
    """Lean step_indexed path (bind_tables, shadow reuse between steps) on two ranks == one rank."""
    import torch.multiprocessing as mp
    res = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker_indexed, args=(r, world, port, q, precision)) for r in range(world)]
        for p in procs:
            p.start()
        out = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        res[world] = sorted(out, key=lambda t: t[0])
    ref = res[1][0][1]
    tol = 1e-6 if precision == "fp32" else 3e-3
    for rank, w in res[2]:
        np.testing.assert_allclose(w, ref, atol=tol * max(1.0, np.abs(ref).max()), rtol=0)


def test_unequal_shards_rank_without_image_rows_applies_the_same_update():
    """A rank whose image shard is empty (uneven shards / global batch < world size) must still step img_proj and
    img_scale from the all-reduced gradient: the update is gated on the GLOBAL row counts.  Two ranks are emulated by
    two engines in one process, the all-reduce by a sum of their flat gradient buffers."""
    import umlh
    rng = np.random.default_rng(21)
    dv, dt, C, Bi, Bt = 48, 64, 10, 24, 40
    xi = rng.standard_normal((Bi, dv)).astype(np.float32)
    xt = rng.standard_normal((Bt, dt)).astype(np.float32)
    yi, yt = rng.integers(0, C, Bi), rng.integers(0, C, Bt)
    wp = (rng.standard_normal((dt, dv)) * 0.1).astype(np.float32)
    wh = (rng.standard_normal((C, dt)) * 0.1).astype(np.float32)
    dev = "cuda:0"
    T = lambda a, t: torch.as_tensor(a).to(dev, t).contiguous()

    def engine():
        e = umlh.HeadEngine(dv, dt, C, has_proj=True, learnable_temp=True, optimizer="adamw", weight_decay=0.01,
                            max_rows_img=64, max_rows_txt=64, device=dev)
        e.w_head.copy_(torch.from_numpy(wh)); e.w_proj.copy_(torch.from_numpy(wp)); e.scales.fill_(3.0)
        return e

    ref = engine()
    ref.train_step(umlh.RowBatch(T(xi, torch.float32), T(yi, torch.int64)), umlh.RowBatch(T(xt, torch.float32), T(yt, torch.int64)),
                   lr=1e-2, step=1)
    r0, r1 = engine(), engine()
    half = Bt // 2
    idx0, idx1 = torch.arange(0, half, device=dev), torch.arange(half, Bt, device=dev)
    g0 = r0.grad_step(umlh.RowBatch(T(xi, torch.float32), T(yi, torch.int64), global_rows=Bi),
                      umlh.RowBatch(T(xt, torch.float32), T(yt, torch.int64), idx0, global_rows=Bt))
    g1 = r1.grad_step(umlh.RowBatch(T(xi, torch.float32), T(yi, torch.int64), rows=0, global_rows=Bi),
                      umlh.RowBatch(T(xt, torch.float32), T(yt, torch.int64), idx1, global_rows=Bt))
    tot = g0 + g1
    g0.copy_(tot); g1.copy_(tot)
    r0.apply_update(lr=1e-2, step=1); r1.apply_update(lr=1e-2, step=1)
    torch.cuda.synchronize()
    for name in ("w_head", "w_proj", "scales", "m_proj", "v_proj", "m_scales", "v_scales"):
        a, b, c = (getattr(e, name).cpu().numpy() for e in (r0, r1, ref))
        np.testing.assert_array_equal(a, b, err_msg=name)                       # replicas stay identical
        assert np.abs(a - c).max() < 2e-3 * max(1.0, np.abs(c).max()), name      # == the single-GPU step (Adam sign flips at |g|~eps aside)
    assert np.abs(r1.w_proj.cpu().numpy() - wp).max() > 1e-4                     # the rank without image rows did step img_proj

    # a rank with no local row at all contributes zeros and still applies the update
    r2 = engine()
    g2 = r2.grad_step(umlh.RowBatch(T(xi, torch.float32), T(yi, torch.int64), rows=0, global_rows=Bi),
                      umlh.RowBatch(T(xt, torch.float32), T(yt, torch.int64), rows=0, global_rows=Bt))
    assert float(g2.abs().max()) == 0.0


def _worker_c_loop(rank, world, port, q, precision, proj, transport="gloo"):
    """C-level data-parallel loop: umlh_train_steps with a transport attached runs grad -> all-reduce -> update per step
    from C; the transport here is gloo through the all-reduce callback (RCCL refuses two ranks on one GPU)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import umlh
        rng = np.random.default_rng(8)
        dv, dt, C, n, B, steps = 128, (256 if proj else 128), 50, 512, 64, 4
        xi = rng.standard_normal((n, dv)).astype(np.float32); xi /= np.linalg.norm(xi, axis=1, keepdims=True)
        xt = rng.standard_normal((n, dt)).astype(np.float32); xt /= np.linalg.norm(xt, axis=1, keepdims=True)
        yi, yt = rng.integers(0, C, n), rng.integers(0, C, n)
        wh = (rng.standard_normal((C, dt)) * 0.1).astype(np.float32)
        wp = (rng.standard_normal((dt, dv)) / np.sqrt(dv)).astype(np.float32)
        dev = "cuda:0"
        e = umlh.HeadEngine(dv, dt, C, has_proj=proj, optimizer="sgd", weight_decay=0.0, max_rows_img=B, max_rows_txt=B,
                            precision=precision, device=dev)
        e.w_head.copy_(torch.from_numpy(wh)); e.scales.fill_(10.0)
        if proj:
            e.w_proj.copy_(torch.from_numpy(wp))
        e.enable_diagnostics(True)
        Xi, Yi, Xt, Yt = (torch.from_numpy(a).to(dev) for a in (xi, yi, xt, yt))
        ti = (Xi, Yi, umlh.to_bf16(Xi)) if precision == "bf16" else (Xi, Yi)
        tt = (Xt, Yt, umlh.to_bf16(Xt)) if precision == "bf16" else (Xt, Yt)
        # global batch k = rows k*B .. k*B + B of a fixed order; rank r takes every world-th row of it
        bi = [torch.arange(k * B + rank, (k + 1) * B, world, device=dev) for k in range(steps)]
        bt = [torch.arange(k * B + rank, (k + 1) * B, world, device=dev) for k in range(steps)]
        if world > 1 and transport == "p2p":
            e.init_p2p()                              # direct peer-to-peer all-reduce over IPC-mapped regions (csrc/umlh_p2p.hip)
        elif world > 1:
            e.set_allreduce(lambda t: dist.all_reduce(t), world)
        sc = torch.zeros(steps, umlh.N_SCALARS, device=dev)
        e.train_steps(ti, bi, tt, bt, [0.05] * steps, first_step=1, alpha=0.5, scalars_out=sc)
        torch.cuda.synchronize()
        assert e.step_status()[0] == 0
        q.put((rank, e.w_head.cpu().numpy(), e.w_proj.cpu().numpy() if proj else None, sc.cpu().numpy()))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("precision,proj", [("fp32", False), ("fp32", True), ("bf16", True)])
def test_c_level_dp_loop_two_ranks_vs_single(precision, proj):
    """umlh_train_steps with two ranks (transport through the all-reduce callback, the 2-layer head's head-gradient
    all-reduce on the second stream) == the single-GPU fused steps on the concatenated batches: weights, step scalars and
    the per-modality gradient diagnostics (formed from the all-reduced pair)."""
    import torch.multiprocessing as mp
    res = {}
    for world in (1, 2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker_c_loop, args=(r, world, port, q, precision, proj)) for r in range(world)]
        for p in procs:
            p.start()
        out = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        res[world] = sorted(out, key=lambda t: t[0])
    _, w1, p1, s1 = res[1][0]
    tol = 2e-6 if precision == "fp32" else 4e-3
    for rank, w, pj, sc in res[2]:
        np.testing.assert_allclose(w, w1, atol=tol * max(1.0, np.abs(w1).max()), rtol=0)
        if proj:
            np.testing.assert_allclose(pj, p1, atol=tol * max(1.0, np.abs(p1).max()), rtol=0)
        np.testing.assert_allclose(sc[:, :4], s1[:, :4], atol=1e-5 if precision == "fp32" else 2e-3)
        d2, d1 = sc[:, 8:12], s1[:, 8:12]                      # dot, |g_img|^2, |g_txt|^2, sign agreements
        np.testing.assert_allclose(d2[:, :3], d1[:, :3], rtol=2e-3 if precision == "fp32" else 5e-2, atol=1e-9)
        assert np.abs(d2[:, 3] - d1[:, 3]).max() <= (0.002 if precision == "fp32" else 0.05) * w1.size
        assert d1[:, 1].min() > 0 and d1[:, 2].min() > 0
    np.testing.assert_array_equal(res[2][0][1], res[2][1][1])


def test_rccl_communicator_single_rank_round_trip(monkeypatch):
    """librccl is loaded at run time and an RCCL communicator of ONE rank drives the C-level data-parallel loop
    (UMLH_FORCE_DP=1 makes a lone rank take the split path): ncclAllReduce over one rank is the identity, so the result
    equals the fused steps -- checks the loader, ncclCommInitRank, the enqueue order on the step's stream."""
    import ctypes as C
    import umlh
    from umlh import _lib
    rng = np.random.default_rng(4)
    d, Cc, n, B, steps = 128, 40, 300, 96, 3
    x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rng.integers(0, Cc, n)
    w = (rng.standard_normal((Cc, d)) * 0.1).astype(np.float32)
    dev = "cuda:0"
    X, Y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    bi = [torch.randperm(n, generator=torch.Generator().manual_seed(k))[:B].to(dev) for k in range(steps)]
    out = []
    for use_comm in (False, True):
        if use_comm:
            monkeypatch.setenv("UMLH_FORCE_DP", "1")          # read at umlh_create
        e = umlh.HeadEngine(d, d, Cc, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B, device=dev)
        e.w_head.copy_(torch.from_numpy(w)); e.scales.fill_(10.0)
        if use_comm:
            idb = (C.c_ubyte * _lib.COMM_ID_BYTES)()
            _lib.check(e.lib.umlh_comm_unique_id(idb), "umlh_comm_unique_id")
            _lib.check(e.lib.umlh_comm_init_rank(e.handle, idb, 1, 0), "umlh_comm_init_rank")
        sc = torch.zeros(steps, umlh.N_SCALARS, device=dev)
        e.train_steps((X, Y), bi, (X, Y), bi, [1e-3] * steps, first_step=1, scalars_out=sc)
        torch.cuda.synchronize()
        assert e.micro_launches() == 0
        out.append((e.w_head.cpu().numpy(), sc.cpu().numpy()))
        e.close()
    diff = np.abs(out[1][0] - out[0][0])
    assert (diff > 2e-6).mean() < 1e-3 and diff.max() < 5e-3          # Adam sign flips at |g| ~ eps aside
    np.testing.assert_allclose(out[1][1][:, :4], out[0][1][:, :4], atol=1e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_p2p_allreduce_two_ranks_equals_gloo_transport_bit_for_bit(precision):
    """The direct peer-to-peer all-reduce (reduce-scatter + all-gather over hipIpc-mapped exchange regions, csrc/umlh_p2p.hip)
    drives the C-level data-parallel loop with two ranks (two processes on ONE GPU: the protocol, the IPC plumbing and the
    epoch-tagged flags are exercised; xGMI is not): after four steps both ranks hold the weights and scalars of the gloo-callback
    transport BIT for bit (two terms added in rank order = a + b), equal on both ranks, and the status word is clean."""
    import torch.multiprocessing as mp
    res = {}
    for transport in ("gloo", "p2p"):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker_c_loop, args=(r, 2, port, q, precision, False, transport)) for r in range(2)]
        for p in procs:
            p.start()
        out = [q.get(timeout=300) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        res[transport] = sorted(out, key=lambda t: t[0])
    for r in range(2):
        np.testing.assert_array_equal(res["p2p"][r][1], res["gloo"][r][1])
        np.testing.assert_array_equal(res["p2p"][r][3], res["gloo"][r][3])
    np.testing.assert_array_equal(res["p2p"][0][1], res["p2p"][1][1])
