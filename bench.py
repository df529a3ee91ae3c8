#!/usr/bin/env python3
"""Benchmark of the UML head fine-tune hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: 4096 image-feature rows + 4096
text-feature rows PER GPU through the shared linear head (fused GEMM + scale +
softmax-CE forward, dW, AdamW), workload = BASELINE config 2 (ImageNet-1k, CLIP ViT-B/16
features d=512, C=1000, unpaired CUPL text).  Inputs are synthetic, resident in HBM
before the timed region.  N>1: plain data parallel, one RCCL all-reduce of the head
gradient per step, weak scaling.  Rank 0 prints ONE JSON line.

Launched without a rendezvous environment (`python bench.py --gpus N`, N > 1) the script
starts its own N ranks as a `torch.distributed.run` child BEFORE anything touches the GPU
and exits with the child's code.

Timing: after `--prime` untimed steps (clocks, caches, allocator) and the W warm-up steps, the
block of EXACTLY K steps is timed `--repeats` times, each bracketed by barrier + synchronize on
both sides; `value` / `ms_per_step` are the MEDIAN block (all blocks are listed in
`block_ms_per_step`).  A single 20-step block is 1 ms of device work: its time is mostly the
fences and the launch ramp, which is why the median of many blocks is reported.
"""
import argparse
import contextlib
import glob
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "unpaired-multimodal-learning_amd"))

METRIC = "samples/sec (img+text feats) UML head fine-tune, batch 4096, 1/2/4/8 GPU"
N_IMG, N_TXT, D, C, BATCH = 1_281_167, 29_940, 512, 1000, 4096
PEAK = {"fp32": 157.3, "bf16": 2500.0}       # dense MFMA TFLOP/s, MI355X_MICROARCH.md
L2_BW = 34.5e12                              # aggregate L2 -> CU bandwidth, MI355X_MICROARCH.md "L2 (per XCD)"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=0, help="timed K-step blocks (0 = auto: about 3000 steps in all, 3..50 blocks)")
    ap.add_argument("--prime", type=int, default=300, help="untimed steps before the warm-up (clock ramp, caches)")
    ap.add_argument("--precision", default=os.environ.get("UMLH_PRECISION", "bf16"), choices=["fp32", "bf16"],
                    help="bf16 = BASELINE config 2 (headline); fp32 = exact-parity mode")
    ap.add_argument("--force-dp-path", action="store_true", help="N=1 only: run the data-parallel split path (grad -> [all-reduce] -> update) to price it")
    ap.add_argument("--dp-host-loop", action="store_true", help="N>1: per-step Python stepping (torch.distributed all_reduce) instead of the C-level RCCL loop")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the additional fp32 parity-mode measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cfg3", action="store_true", help="N=1: also time BASELINE configs[2] (2-layer head 1024 -> 3200, C=1000, 4096+4096 rows) "
                                                       "in both precision modes and report it under the extra key 'cfg3' (off by default)")
    ap.add_argument("--order-rng", default="device", choices=["device", "torch-cpu"],
                    help="epoch permutations drawn on the GPU (default) or by the reference-identical CPU sampler")
    ap.add_argument("--block", type=int, default=25, help="steps per umlh_train_steps call (host prepares the next block meanwhile)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a rendezvous environment: run N ranks as a child
    `torch.distributed.run` (fresh processes: nothing in THIS process has touched the GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def synth_rows(torch, n, d, c, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(n, d, generator=g, device=device, dtype=torch.float32)
    x = torch.nn.functional.normalize(x, dim=1)
    y = torch.randint(0, c, (n,), generator=g, device=device, dtype=torch.int64)
    return x, y


def latest_pmc():
    """HBM bytes per launch from the newest committed PMC summary (FETCH_SIZE x2 + WRITE_SIZE, separate passes; see
    the file's note).  STATIC: collected by scripts/profile_round.sh, not in this run."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bf16_pmc.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1]))["kernels"], os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    # one rank per GPU (RCCL over xGMI).  Rehearsal on a box with fewer GPUs than ranks: ranks share GPUs and the
    # gradient all-reduce goes through gloo (RCCL refuses two ranks on one device) -- functional check only.
    rehearsal = world > n_dev
    dev = torch.device("cuda", local % max(1, n_dev))
    torch.cuda.set_device(dev)
    rccl_ranks = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
            rccl_ranks = dist.get_world_size()
            if rccl_ranks != args.gpus:
                raise SystemExit(f"RCCL world size {rccl_ranks} != --gpus {args.gpus}")

    if args.force_dp_path and not args.dp_host_loop:
        os.environ["UMLH_FORCE_DP"] = "1"               # read at umlh_create: a lone rank takes the C-level split path
    import umlh
    from engine.datasets.utils import FeatureLoader, FeatureTable
    from engine.models.head import UMLClip
    from engine.optimizer.optim import build_optimizer
    from engine.optimizer.scheduler import build_lr_scheduler
    from engine.tools.utils import set_random_seed
    from finetune import _RowSource, _draw_block

    repeats = args.repeats if args.repeats > 0 else max(3, min(50, -(-3000 // max(1, args.steps))))

    # ---- synthetic, HBM-resident workload (image rows sharded over ranks, text replicated) ----
    set_random_seed(0)
    n_img_local = (N_IMG + world - 1) // world
    x_img, y_img = synth_rows(torch, n_img_local, D, C, 100 + rank, dev)
    x_txt, y_txt = synth_rows(torch, N_TXT, D, C, 7, dev)
    model = UMLClip(D, C, logit_scale_init=4.60517).to(dev)            # s = 100 (config/__init__.py:209-216)
    with contextlib.redirect_stdout(sys.stderr):            # (the mirror prints the reference's progress lines: stdout carries the JSON line only)
        model.zero_shot_init(FeatureTableAsText(x_txt, y_txt))
    optimizer = build_optimizer(model.parameters(), "adamw", 1e-3, 0.01)
    scheduler = build_lr_scheduler(optimizer, "cosine", 50, 12800, warmup_type="linear", warmup_lr=1e-5)
    torch.manual_seed(1234 + rank)                                       # per-rank shuffles
    img_src = _RowSource(FeatureLoader(FeatureTable(x_img, y_img, dev), BATCH, shuffle=True, kind="image",
                                       order_rng=args.order_rng), dev, "image", args.precision)
    txt_src = _RowSource(FeatureLoader(FeatureTable(x_txt, y_txt, dev), BATCH, shuffle=True, kind="text",
                                       order_rng=args.order_rng), dev, "text", args.precision)
    ring = 4096                                                          # per-step scalar rows (device ring)
    scal = torch.zeros(ring, umlh.N_SCALARS, device=dev)
    cursor = {"k": 0}
    dp_path = world > 1 or args.force_dp_path

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def measure(precision, steps, warmup, prime):
        """`prime` + `warmup` untimed steps, then `repeats` timed blocks of `steps` steps, each bracketed by
        barrier + synchronize; returns the median block."""
        engine = model.fused_engine(optimizer, BATCH, BATCH, precision=precision)
        stepper = umlh.DataParallelStepper(engine)
        stepper.broadcast_parameters([model.head.weight.data])
        tab_i, tab_t = img_src.table(precision), txt_src.table(precision)
        engine.bind_tables(tab_i, tab_t)
        c_level_dp = world == 1 and args.force_dp_path and not args.dp_host_loop
        watchdog = None
        if world > 1:
            # a communicator init or a collective that never completes (a rank missing) would otherwise hold the job until
            # the caller's own limit: fail fast and say where.  Covers the RCCL attach below and the untimed warm-up.
            import threading
            state = {"loop": "communicator setup"}
            watchdog = threading.Timer(180.0, lambda: (print(f"[rank {rank}] data-parallel {state['loop']} did not finish in 180 s; aborting",
                                                                 file=sys.stderr, flush=True), os._exit(4)))
            watchdog.daemon = True
            watchdog.start()
        p2p_req = os.environ.get("UMLH_DP_P2P", "0") == "1"
        if world > 1 and (not rehearsal or p2p_req) and not args.dp_host_loop:     # (the peer-to-peer transport also works with ranks sharing a GPU)
            # every rank must take the same loop: a rank whose communicator could not be created (library missing, init
            # error) sends the whole job to the per-step torch.distributed path instead of leaving the others in a collective
            try:
                # UMLH_DP_P2P=1: the direct peer-to-peer all-reduce (csrc/umlh_p2p.hip) instead of RCCL -- opt-in, the default
                # transport of the scaling run stays RCCL until the peer-to-peer path has been measured on a multi-GPU node
                if os.environ.get("UMLH_DP_P2P", "0") == "1":
                    ok = 1 if stepper.attach_p2p() else 0
                else:
                    ok = 1 if stepper.attach_rccl() else 0
            except umlh.UmlhError as exc:
                print(f"[rank {rank}] C-level RCCL loop unavailable ({exc}); falling back to per-step torch.distributed", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev if not rehearsal else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            c_level_dp = bool(int(flag.item()))
            if ok and not c_level_dp:
                engine.detach_comm()
        if watchdog is not None:
            state["loop"] = "warm-up (%s)" % ("C-level RCCL loop" if c_level_dp else "per-step torch.distributed")

        def slot(m):
            if cursor["k"] + m > ring:
                cursor["k"] = 0
            k = cursor["k"]
            cursor["k"] += m
            return scal[k:k + m]

        def one_step():
            ii, ti = img_src.next_index(), txt_src.next_index()
            out = slot(1)[0]
            if dp_path:       # grad -> all-reduce -> update, lean host path
                stepper.step_indexed(ii, ti, lr=optimizer.param_groups[0]["lr"], step=optimizer.step_count + 1,
                                     alpha=1.0, scalars_out=out)
            else:
                engine.train_step(umlh.RowBatch(tab_i[0], tab_i[1], ii, feats_bf16=tab_i[2] if len(tab_i) > 2 else None),
                                  umlh.RowBatch(tab_t[0], tab_t[1], ti, feats_bf16=tab_t[2] if len(tab_t) > 2 else None),
                                  lr=optimizer.param_groups[0]["lr"], step=optimizer.step_count + 1, alpha=1.0,
                                  scalars_out=out)
            optimizer.step_count += 1
            scheduler.step()
            return int(ii.numel()) + int(ti.numel())

        ahead = {}          # block length -> index vectors drawn one block ahead (consumed in draw order)

        def run_steps(n, next_n=None):
            """Blocks of `--block` steps through ONE umlh_train_steps call each -- with a communicator attached the same
            call also runs the data-parallel step (grad -> ncclAllReduce -> update, all enqueued from C); otherwise N>1
            steps from Python.  The index vectors of the NEXT block are drawn (and its call marshalled) right after a block is enqueued, i.e. while
            the GPU runs it, as in an unfenced training loop (finetune.train() enqueues blocks back to back); `next_n` tells
            the first block length of the following call.  Every timed block therefore still contains exactly one draw
            (and its shuffle kernels on the stream); only the host latency in front of the first launch is hidden."""
            rows = 0
            if dp_path and not c_level_dp:
                for _ in range(n):
                    rows += one_step()
                return rows
            done = 0
            while done < n:
                m = min(args.block, n - done)
                blk = ahead.pop(m, None)
                if blk is None:
                    ahead.clear()                           # (a look-ahead of another length is dropped: the draws are never reordered)
                    bi, bt = _draw_block(img_src, txt_src, m)   # as finetune.train(): consecutive batches of an epoch as ONE index slice
                    blk = (bi, bt, engine.prepare_steps(tab_i, bi, tab_t, bt, scheduler.lr_table(m)))
                bi, bt, prep = blk
                engine.train_steps_prepared(prep, first_step=optimizer.step_count + 1, alpha=1.0, scalars_out=slot(m))
                optimizer.step_count += m
                scheduler.step(scheduler.last_epoch + m)
                rows += sum(int(b[0].numel()) if isinstance(b, tuple) else int(b.numel()) for b in bi + bt)
                done += m
                m_next = min(args.block, n - done) if done < n else (min(args.block, next_n) if next_n else 0)
                if m_next > 0:                              # (the scheduler already stands at the next block's first step)
                    nbi, nbt = _draw_block(img_src, txt_src, m_next)
                    ahead[m_next] = (nbi, nbt, engine.prepare_steps(tab_i, nbi, tab_t, nbt, scheduler.lr_table(m_next)))
            return rows

        run_steps(prime, warmup)
        run_steps(warmup, steps)
        if watchdog is not None:
            fence()
            watchdog.cancel()
        blocks, enq = [], []
        for _ in range(repeats):
            fence()
            t0 = time.perf_counter()
            rows = run_steps(steps, steps)
            t_enq = time.perf_counter() - t0
            fence()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt, rows], device=dev, dtype=torch.float64)
                tm = t.clone()
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                dt, rows = float(tm[0].item()), int(t[1].item())
            blocks.append((dt, rows))
            enq.append(t_enq)
        order = sorted(range(repeats), key=lambda i: blocks[i][0])
        med = order[(repeats - 1) // 2]                 # the median block itself (lower median: a measured block)
        dt, rows = blocks[med]
        final = scal[cursor["k"] - 1].cpu().tolist()
        # health gate BEFORE any number is reported (two N=2 rehearsals printed NaN losses and 100 ms blocks and
        # exited 0): a bounded in-launch wait that gave up, a non-finite step scalar anywhere in the ring, or a timed block
        # far off the median ends the run with a non-zero exit code and no JSON line
        engine.check_status()
        ring_used = scal[:max(cursor["k"], 1), :4]
        if not bool(torch.isfinite(ring_used).all()) or not all(math.isfinite(v) for v in final[:4]):
            raise SystemExit(f"[rank {rank}] bench: non-finite step scalars ({precision}; final loss img/txt = {final[0]}, {final[1]}): "
                             "the run is invalid, no result line is printed")
        slow = [b[0] for b in blocks if b[0] > 20.0 * dt]
        if slow:
            raise SystemExit(f"[rank {rank}] bench: {len(slow)} of {repeats} timed blocks took more than 20x the median block "
                             f"({max(slow) / steps * 1e3:.3f} vs {dt / steps * 1e3:.4f} ms/step): stalled waits or a starved device; "
                             "the run is invalid, no result line is printed")
        # roofline leg: per-kernel device time from HIP events recorded on the step's stream (RAW intervals: each
        # carries the cost of its two marker packets, so they read a few hundred ns above rocprofv3's kernel trace)
        engine.profile(True)
        acc, nprof = {}, 50
        for _ in range(nprof):
            one_step()
            for name, ms in engine.profile_read().items():
                acc[name] = acc.get(name, 0.0) + ms / nprof
        engine.profile(False)
        flops = {"fwd_ce": 2.0 * 2 * BATCH * C * D, "dw_head": 2.0 * 2 * BATCH * C * D}   # algorithmic, per launch
        # bf16 mode runs forward and dW as ONE launch (fwd_dw_bf16; UMLH_BF16_FUSE=0 / UMLH_WT=0 keep two): the interval
        # mark 1 -> 2 is then empty and mark 2 -> 3 holds the launch, whose algorithmic work is both GEMMs
        fused = precision == "bf16" and os.environ.get("UMLH_BF16_FUSE", "1") != "0" and os.environ.get("UMLH_WT", "1") != "0" \
            and os.environ.get("UMLH_BF16_FWD2D", "0") != "1"
        if fused:
            # UMLH_BF16_FUSE: 2 (default) = the update and the step scalars ride in the same launch too (single-GPU step only:
            # the data-parallel split step keeps the update behind the all-reduce); 1 = forward + dW
            whole = os.environ.get("UMLH_BF16_FUSE", "2") not in ("0", "1")
            # data-parallel split step: the same launch ends with the slab sum into the gradient message (no update)
            name = ("step_grad" if dp_path else "step") if whole else "fwd_dw"
            acc[name] = acc.pop("dw_head")
            acc["empty_interval_fwd"] = acc.pop("fwd_ce")
            if whole and not dp_path:
                acc["empty_interval_update"] = acc.pop("reduce_update")
            flops = {name: 2 * 2.0 * 2 * BATCH * C * D}
        dom = max(flops, key=lambda n: acc[n])
        achieved = flops[dom] / (acc[dom] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK[precision],
                "unit": "TFLOP/s", "frac": round(achieved / PEAK[precision], 4), "traffic": None,
                "kernel_ms": {k: round(v, 4) for k, v in acc.items()},
                "kernel_ms_source": "HIP events on the step's stream, raw intervals, mean of %d instrumented steps" % nprof,
                "step_frac_of_mfma_roof": round((sum(flops.values()) / (PEAK[precision] * 1e12)) / (dt / steps), 4)}
        if precision == "fp32":
            x3 = os.environ.get("UMLH_F32_X3", "1") != "0"
            roof["arith"] = ("fp32 operands split three ways into bf16 pieces, six v_mfma_f32_32x32x16_bf16 piece products per product, fp32 "
                             "accumulation (UMLH_F32_X3=1, the default): 'achieved' counts the fp32 products once, against the fp32 MFMA roof; "
                             "'issued_frac_of_bf16_roof' counts the six bf16 MFMAs against the bf16 roof") if x3 else "v_mfma_f32_32x32x2_f32 (UMLH_F32_X3=0)"
            if x3:
                roof["issued_frac_of_bf16_roof"] = round(6.0 * achieved / PEAK["bf16"], 4)
                roof["step_issued_frac_of_bf16_roof"] = round(6.0 * (sum(flops.values()) / (PEAK["bf16"] * 1e12)) / (dt / steps), 4)
        if "proj_bwd" in acc:
            # cfg2 has no img_proj: the proj_bwd interval holds NO kernel, so it reads what one pair of event markers costs on
            # this stream; every kernel interval above carries about half of it (rocprofv3's trace of the same kernels:
            # profiles/r02_kernel_stats.md).  Reported beside the raw intervals, never subtracted from them.
            roof["empty_interval_ms"] = round(acc["proj_bwd"], 4)
        if precision == "bf16":
            pmc, src = latest_pmc()
            key = {"fwd_ce": "fwd_ce_bf16", "dw_head": "dw_bf16", "fwd_dw": "fwd_dw_bf16", "step": "step_bf16", "step_grad": "step_bf16"}[dom]
            if pmc and key in pmc:
                roof["traffic"] = pmc[key].get("hbm_bytes_corrected")
                roof["traffic_source"] = f"static: {src} (rocprofv3 --pmc passes, not collected in this run)"
            # second roof of the forward's present decomposition: every CU streams all of W from L2 for its 32 rows
            l2_bytes = (8192 / 32) * 1024 * 512 * 2
            roof["l2_stream_bound_ms"] = round(l2_bytes / L2_BW * 1e3, 5)
        return {"value": rows / dt, "dt": dt, "rows": rows, "ms_per_step": dt / steps * 1e3,
                "block_ms_per_step": [round(b[0] / steps * 1e3, 4) for b in blocks],
                "host_enqueue_ms_per_step": statistics.median(enq) / steps * 1e3, "final": final, "roofline": roof,
                "c_level_dp": bool(c_level_dp)}

    head = measure(args.precision, args.steps, args.warmup, args.prime)
    other = None
    if args.precision == "bf16" and not args.no_fp32_leg:
        # the exact-parity fp32 mode on the same workload (fewer steps: it is ~5x slower)
        other = measure("fp32", max(10, args.steps // 4), max(3, args.warmup // 4), max(10, args.prime // 4))
    value, roofline, final = head["value"], head["roofline"], head["final"]
    if other is not None:   # the parity mode's figures as scalars next to the headline's (1e-4 logits/loss parity lives here)
        roofline.update({"fp32_value": round(other["value"], 1), "fp32_ms_per_step": round(other["ms_per_step"], 4),
                         "fp32_kernel": other["roofline"]["kernel"], "fp32_achieved": other["roofline"]["achieved"],
                         "fp32_peak": PEAK["fp32"], "fp32_frac": other["roofline"]["frac"],
                         "fp32_step_frac_of_mfma_roof": other["roofline"]["step_frac_of_mfma_roof"],
                         "fp32_arith": other["roofline"].get("arith"),
                         "fp32_step_issued_frac_of_bf16_roof": other["roofline"].get("step_issued_frac_of_bf16_roof")})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_cpu_loop as cpu_loop
        n_sub = 32768
        xi, yi = x_img[:n_sub].cpu(), y_img[:n_sub].cpu()
        xt, yt = x_txt.cpu(), y_txt.cpu()
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        # a 1-GPU box's CPU share is 16 cores; more torch threads than that only oversubscribes
        threads = min(cores, int(os.environ.get("UMLH_CPU_THREADS", 16)))
        ncpu = int(os.environ.get("UMLH_CPU_STEPS", 150))
        v, secs, n = cpu_loop.reference_shaped_steps(xi, yi, xt, yt, C, BATCH, steps=ncpu, warmup=2, threads=threads)
        vb, _, _ = cpu_loop.bare_math_steps(xi, yi, xt, yt, C, BATCH, steps=ncpu, warmup=2, threads=threads)
        cpu = {"value": round(v, 1), "unit": "samples/s", "cores": threads, "kind": "port",
               "sample": f"{ncpu} timed steps (+2 warm-up) of 4096+4096 rows on a {n_sub}-row image subset, "
                         f"reference-shaped torch-CPU loop (DataLoader collate, 3 backward passes, per-step scalars)",
               "seconds": round(secs, 2), "bare_math_value": round(vb, 1), "host_cores_visible": cores,
               "calibration": cpu_loop.CALIBRATION}

    if rank == 0:
        out = {"metric": METRIC, "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16",
               "data": "synthetic", "repeats": repeats, "prime_steps": args.prime, "rccl_ranks": rccl_ranks,
               "config": {"workload": "cfg2 ImageNet-1k CLIP-ViT-B/16 features (d=512) + unpaired CUPL text, linear head "
                                      "C=1000, 4096 img + 4096 txt rows/step/GPU, scale 100, zero-shot init, AdamW "
                                      "lr 1e-3 wd 0.01, warm-up 50 + cosine 12800",
                          "n_img_rows": N_IMG, "n_txt_rows": N_TXT, "global_batch": 2 * BATCH * world,
                          "parallelism": f"dp{world}" + (" (REHEARSAL: ranks share GPUs, gloo all-reduce -- not a measurement)" if rehearsal else ""),
                          "dp_stepping": (("c-level peer-to-peer (umlh_p2p)" if os.environ.get("UMLH_DP_P2P", "0") == "1" else "c-level rccl") if head["c_level_dp"] else "python per step") if dp_path else "single gpu",
                          "precision_mode": args.precision, "order_rng": args.order_rng, "steps_per_call": args.block,
                          "timing": f"median of {repeats} blocks of {args.steps} steps after {args.prime}+{args.warmup} untimed steps; "
                                    "each block draws and marshals the next block's index vectors after enqueuing its own steps"},
               "block_ms_per_step": head["block_ms_per_step"],
               "final_loss": {"img": round(final[0], 4), "txt": round(final[1], 4)},
               "host_enqueue_ms_per_step": round(head["host_enqueue_ms_per_step"], 4),
               "roofline": roofline, "cpu_baseline": cpu}
        if other is not None:
            out["fp32_parity_mode"] = {"value": round(other["value"], 1), "ms_per_step": round(other["ms_per_step"], 4),
                                       "dtype": "f32", "roofline": other["roofline"]}
        if cpu:
            out["speedup_vs_cpu"] = round(value / cpu["value"], 1)
        if args.cfg3 and world == 1:
            out["cfg3"] = measure_cfg3(torch, umlh, dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measure_cfg3(torch, umlh, device):
    """BASELINE configs[2] (DINOv2-L 1024-d image rows -> img_proj 3200 -> head C=1000, OpenLLaMA-width text rows, 4096+4096 rows per
    step; reference head.py:64-66,79-82): a parity-test case, timed here as an extra key only.  Blocks of consecutive steps through
    umlh_train_steps, index vectors drawn outside the timed region, median of 5 blocks."""
    import statistics
    import time
    d_img, d_sh, C, B, n_img, n_txt = 1024, 3200, 1000, 4096, 65536, 8192
    flop = 2.0 * (B * d_img * d_sh * 2 + 2 * B * d_sh * C * 2 + B * C * d_sh)      # H, dW_proj | fwd, dW_head | dH^T
    g = torch.Generator(device=device).manual_seed(3)
    xi = torch.nn.functional.normalize(torch.randn(n_img, d_img, generator=g, device=device), dim=1)
    xt = torch.nn.functional.normalize(torch.randn(n_txt, d_sh, generator=g, device=device), dim=1)
    yi = torch.randint(0, C, (n_img,), generator=g, device=device)
    yt = torch.randint(0, C, (n_txt,), generator=g, device=device)
    res = {"workload": "cfg3: 2-layer head 1024 -> 3200 -> 1000, 4096 img + 4096 txt rows/step, AdamW", "flop_per_step": flop}
    for precision, steps in (("bf16", 40), ("fp32", 10)):
        e = umlh.HeadEngine(d_img, d_sh, C, has_proj=True, optimizer="adamw", weight_decay=0.01, max_rows_img=B, max_rows_txt=B,
                            precision=precision, device=device)
        e.w_head.normal_(0, 0.02)
        e.w_proj.normal_(0, 0.02)
        ti = (xi, yi, umlh.to_bf16(xi)) if precision == "bf16" else (xi, yi)
        tt = (xt, yt, umlh.to_bf16(xt)) if precision == "bf16" else (xt, yt)
        times, k0 = [], 1
        sc = torch.zeros(steps, umlh.N_SCALARS, device=device)
        for rep in range(6):
            bi = [torch.randint(0, n_img, (B,), generator=g, device=device) for _ in range(steps)]
            bt = [torch.randint(0, n_txt, (B,), generator=g, device=device) for _ in range(steps)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.train_steps(ti, bi, tt, bt, [1e-3] * steps, first_step=k0, scalars_out=sc)
            torch.cuda.synchronize()
            if rep:
                times.append((time.perf_counter() - t0) / steps)
            k0 += steps
        e.check_status()
        dt = statistics.median(times)
        peak = PEAK[precision] * 1e12
        res[precision] = {"ms_per_step": round(dt * 1e3, 4), "samples_per_s": round(2 * B / dt, 1),
                          "step_frac_of_mfma_roof": round(flop / peak / dt, 4),
                          "final_loss": [round(float(v), 4) for v in sc[-1][:2].tolist()]}
        if not all(math.isfinite(v) for v in res[precision]["final_loss"]):
            raise SystemExit(f"bench cfg3 ({precision}): non-finite loss {res[precision]['final_loss']}")
        del e
    return res


class FeatureTableAsText:
    """Minimal text-dataset view (input_tensor / label_tensor) for zero_shot_init."""

    def __init__(self, x, y):
        self.input_tensor, self.label_tensor = x, y


if __name__ == "__main__":
    main()
