"""world_size-2 gloo test of the data-parallel layer on CPU: the all-reduced step on
two half-batches equals the single-process step on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_case(seed=0, proj=True, learn=True):
    from oracle import uml_oracle as O
    rng = np.random.default_rng(seed)
    di, ds, C, B = 12, 16 if proj else 12, 7, 24
    xi = rng.standard_normal((B, di)).astype(np.float32)
    xt = rng.standard_normal((B, ds)).astype(np.float32)
    yi, yt = rng.integers(0, C, B), rng.integers(0, C, B)
    st = O.HeadState(rng.standard_normal((C, ds)).astype(np.float32) * 0.3,
                     (rng.standard_normal((ds, di)) * 0.3).astype(np.float32) if proj else None, 1.5, 0.8, learn)
    return st, xi, yi, xt, yt


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import umlh
        from _oracle_engine import OracleEngine
        st, xi, yi, xt, yt = _make_case()
        eng = OracleEngine(st.copy(), "adamw", 0.01)
        stepper = umlh.DataParallelStepper(eng)
        assert stepper.world == world
        # rank r owns rows r::world of each modality; ragged on purpose for text (13 vs 11 rows)
        sel_i = np.arange(rank, len(yi), world)
        sel_t = np.arange(0, 13) if rank == 0 else np.arange(13, len(yt))
        T = torch.as_tensor
        scal = torch.zeros(12)
        for k in range(3):
            bi = umlh.RowBatch(T(xi), T(yi), T(sel_i), global_rows=len(yi))
            bt = umlh.RowBatch(T(xt), T(yt), T(sel_t), global_rows=len(yt))
            stepper.step(bi, bt, lr=1e-2, step=k + 1, alpha=0.5, scalars_out=scal)
        q.put((rank, eng.state.w_head.copy(), eng.state.w_proj.copy(), eng.state.img_scale, eng.state.txt_scale,
               scal.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_dp_two_ranks_equals_single_process():
    from oracle import uml_oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    st, xi, yi, xt, yt = _make_case()
    opt = O.OptState("adamw", 0.01)
    for k in range(3):
        so = O.step_grads(st, xi, yi, xt, yt, 0.5)
        O.optimizer_step(st, so.grads, opt, 1e-2)
    for rank, w, wp, s0, s1, scal in res:
        np.testing.assert_allclose(w, st.w_head, atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(wp, st.w_proj, atol=2e-6, rtol=1e-5)
        assert abs(s0 - st.img_scale) < 1e-5 and abs(s1 - st.txt_scale) < 1e-5
        assert abs(scal[0] - so.loss_img) < 1e-5 and abs(scal[1] - so.loss_txt) < 1e-5
    np.testing.assert_array_equal(res[0][1], res[1][1])       # replicas stay bit-identical


def test_stepper_world_one_is_plain_train_step():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import umlh
    from _oracle_engine import OracleEngine
    st, xi, yi, xt, yt = _make_case(1, proj=False, learn=False)
    eng = OracleEngine(st.copy(), "sgd", 0.0)
    T = torch.as_tensor
    umlh.DataParallelStepper(eng).step(umlh.RowBatch(T(xi), T(yi)), umlh.RowBatch(T(xt), T(yt)), lr=0.1, step=1)
    from oracle import uml_oracle as O
    so = O.step_grads(st, xi, yi, xt, yt, 1.0)
    O.optimizer_step(st, so.grads, O.OptState("sgd", 0.0), 0.1)
    np.testing.assert_allclose(eng.state.w_head, st.w_head, atol=1e-7)
