"""The UML supervised fine-tune loop on the fused HIP step.

Mirror of the hot path of the reference's ``vision_language/finetune.py``:
``fetch_next`` (:33-39), ``train`` (:120-288), ``validate`` (:291-315) -- same
signatures, step order, loss weighting, evaluation cadence, early stopping,
best-snapshot restore and returned dict -- with the per-step body
(:180-195: forward, 2x cross-entropy, backward, optimizer.step) executed by
``HeadEngine.train_step`` (one launch sequence, logits never leave registers,
no host synchronisation inside a step).

What the reference does per step only for logging (two extra backward passes for
per-modality gradient statistics :190-191,200-206, a second backbone forward :183)
is not on this path.
"""
from __future__ import annotations

import os
import sys
import threading
from typing import Optional

import torch

import umlh
from engine.datasets.utils import FeatureLoader, FeatureTable, TextTensorDataset  # noqa: F401
from engine.models.head import UML, UMLClip  # noqa: F401
from engine.optimizer.default import HYPER_DICT  # noqa: F401
from engine.optimizer.optim import build_optimizer  # noqa: F401
from engine.optimizer.scheduler import build_lr_scheduler  # noqa: F401

EVAL_FREQ = 100   # evaluate on the val set every 100 iterations (early stopping)


def fetch_next(loader, loader_iter):
    """next(loader_iter), restarting the loader when it is exhausted: each modality
    cycles independently of the other (that is the 'unpaired' pairing)."""
    try:
        batch = next(loader_iter)
    except StopIteration:
        loader_iter = iter(loader)
        batch = next(loader_iter)
    return batch, loader_iter


class _RowSource:
    """Adapts a loader to the fused step.  A ``FeatureLoader`` is consumed as index
    vectors into its device-resident table; any other iterable (e.g. a torch
    DataLoader yielding the reference's dict / tuple batches) is consumed as dense
    batches moved to ``device``."""

    def __init__(self, loader, device, kind, precision="fp32"):
        self.loader, self.device, self.kind, self.precision = loader, device, kind, precision
        self.indexed = hasattr(loader, "iter_index")
        self._make_iter = loader.iter_index if self.indexed else loader.__iter__
        self.it = self._make_iter()

    @property
    def capacity(self):
        return int(getattr(self.loader, "batch_size", None) or 4096)

    def next_index(self):
        idx, self.it = fetch_next(_Reiter(self._make_iter), self.it)
        return idx

    def remaining(self):
        return self.it.remaining() if self.indexed and hasattr(self.it, "remaining") else 0

    def take_span(self, k):
        return self.it.take_span(k)

    def table(self, precision):
        t = self.loader.table
        return (t.features, t.labels, t.features_bf16()) if precision == "bf16" else (t.features, t.labels)

    def save(self):
        """Position of the batch stream (iterator, epoch order, offset): ``restore`` rewinds to it."""
        it = self.it
        return it, getattr(it, "order", None), getattr(it, "pos", None)

    def restore(self, st):
        self.it = st[0]
        if st[2] is not None:
            self.it.order, self.it.pos = st[1], st[2]

    def next(self) -> umlh.RowBatch:
        batch, self.it = fetch_next(_Reiter(self._make_iter), self.it)
        if self.indexed:
            t = self.loader.table
            return umlh.RowBatch(t.features, t.labels, batch,
                                 feats_bf16=t.features_bf16() if self.precision == "bf16" else None)
        if self.kind == "image":
            x, y = batch["img"], batch["label"]
        else:
            x, y = batch[0], batch[1]
        return umlh.RowBatch(x.to(self.device, torch.float32).contiguous(), y.to(self.device, torch.int64).contiguous())


class _Reiter:
    def __init__(self, make):
        self.make = make

    def __iter__(self):
        return self.make()


def _rows(rb):
    return rb.feats if rb.index is None else umlh.gather_rows(rb.feats, rb.index)


def feature_direction_sim(model, img_rows, txt_rows):
    """cos(mean image feature after extract_features, mean text feature) of one step's batches
    (finetune.py:182-183,239); 0 without text.  Row gather, projection and the column sums run on the HIP ops;
    the final ratio is formed on the host from the two d-vectors."""
    if img_rows is None or txt_rows is None:
        return 0.0
    n_of = lambda rb: rb.feats.shape[0] if rb.index is None else rb.index.numel()
    with torch.no_grad():
        fi = umlh.column_sums(model.extract_features(_rows(img_rows)).contiguous()).double().cpu() / n_of(img_rows)
        ft = umlh.column_sums(_rows(txt_rows)).double().cpu() / n_of(txt_rows)
    den = float(fi.norm() * ft.norm())
    return float(fi @ ft) / den if den > 0 else float("nan")


def _state_dict_cpu(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def _state_dict_snapshot(model):
    """Best-model snapshot (finetune.py:249) kept ON THE DEVICE: an asynchronous clone instead of a device-to-host copy
    and a host sync at every evaluation point; it is moved to the CPU once, when train() returns it."""
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def _to_cpu(sd):
    return {k: v.cpu() for k, v in sd.items()}


def _check_micro(engine, scalars=None, where=""):
    """Evaluation-point health check.  A bounded in-launch wait that gave up (micro-step kernel, one-launch step) raises
    UmlhError -- the device-side status words are read on every call, not only when the scalars look wrong.  Non-finite step
    scalars with a clean status are a diverged run: the reference prints 'nan' and trains on (finetune.py:236-244; its
    val_acc comparisons then keep the old best and patience ends the run), so this only says so loudly."""
    engine.check_status()
    if scalars is not None and not bool(torch.isfinite(scalars).all()):
        print(f"=> WARNING: non-finite step scalars{where}: the run has diverged (device status clean); continuing as the "
              "reference does", file=sys.stderr)


def _block_end(i, max_iters, eval_freq):
    """Last iteration of the block that starts at ``i``: the next evaluation point (or the final iteration)."""
    return i if i % eval_freq == 0 else min(max_iters - 1, (i // eval_freq + 1) * eval_freq)


def _draw_block(img_src, txt_src, n):
    """Index batches of the next ``n`` steps of both loaders.  Consecutive batches inside both loaders' current epochs are
    taken as ONE index slice each; a step at which a loader starts an epoch goes through next_index() -- image then
    text, the reference's draw order (finetune.py:164-174) -- so the RNG protocol is untouched."""
    bi = [] if img_src is not None else None
    bt = [] if txt_src is not None else None
    k = 0
    while k < n:
        span = min([n - k] + [src.remaining() for src in (img_src, txt_src) if src is not None])
        if span == 0:
            if img_src is not None:
                bi.append(img_src.next_index())
            if txt_src is not None:
                bt.append(txt_src.next_index())
            k += 1
        else:
            if img_src is not None:
                bi.append(img_src.take_span(span))
            if txt_src is not None:
                bt.append(txt_src.take_span(span))
            k += span
    return bi, bt


def _rewind_point(img_src, txt_src, optimizer, scheduler):
    """Everything a training block advances besides the weights, captured before the block is drawn: loader positions, the
    RNG streams the loaders draw their epoch seeds from, optimizer moments and counters, the scheduler's epoch."""
    gens = []
    for src in (img_src, txt_src):
        g = getattr(src.loader, "generator", None) if src is not None else None
        if src is not None and all(g is not q for q, _ in gens):
            gens.append((g, g.get_state() if g is not None else torch.get_rng_state()))
    moments = [(t, t.clone()) for st in optimizer.state.values() for t in st.values() if torch.is_tensor(t)]
    return {"src": [(src, src.save()) for src in (img_src, txt_src) if src is not None], "gens": gens, "moments": moments,
            "step_count": optimizer.step_count, "epoch": scheduler.last_epoch}


def _rewind(r, optimizer, scheduler):
    for src, st in r["src"]:
        src.restore(st)
    for g, st in r["gens"]:
        g.set_state(st) if g is not None else torch.set_rng_state(st)
    for t, old in r["moments"]:
        t.copy_(old)
    optimizer.step_count = r["step_count"]
    scheduler.step(r["epoch"])


def train(model, image_loader, text_loader, val_loader, test_loader, optimizer, scheduler, device="cuda",
          max_iters=1000, alpha=1.0, eval_freq=EVAL_FREQ, patience=5, capture_features_during_training=False,
          features_pth="./", args=None, logger=None, precision="fp32", diagnostics=None):
    out = {"iter": None, "val_acc": None, "model": None, "val_classwise": None, "val_loss": None,
           "model_records": []}
    model.train()
    assert image_loader is not None or text_loader is not None, "At least one of the loaders should be provided"
    if capture_features_during_training:
        print("=> capture_features_during_training is a logging-only diagnostic; not on the fused path (ignored)")
    dev = model.head.weight.device
    img_src = _RowSource(image_loader, dev, "image", precision) if image_loader is not None else None
    txt_src = _RowSource(text_loader, dev, "text", precision) if text_loader is not None else None
    engine = model.fused_engine(optimizer, max_rows_img=img_src.capacity if img_src else 32,
                                max_rows_txt=txt_src.capacity if txt_src else 32, precision=precision)
    # per-step gradient diagnostics (finetune.py:190-191,203-206): the reference computes them on every
    # step; here they cost ~2.5 us per step, so they are on when a logger asks for them (or on request)
    want_diag = bool(diagnostics) if diagnostics is not None else logger is not None
    # (heads with bias: the engine restricts them to the weight columns of the packed [weight | bias | padding] rows -- the
    # reference's diagnostics are over head.weight only, finetune.py:190-191)
    engine.enable_diagnostics(want_diag)
    scalars = torch.zeros(max_iters, umlh.N_SCALARS, dtype=torch.float32, device=dev)
    no_improve = 0
    img_alpha = 1.0
    last_i = -1
    blockwise = logger is None and (img_src is None or img_src.indexed) and (txt_src is None or txt_src.indexed)

    # Evaluation points (finetune.py:247-271).  Blockwise runs read an evaluation's results one block LATE: the evaluation and
    # its device-to-host copy are enqueued, then the next training block, and only then does the host wait for the copy --
    # the GPU goes from the evaluation straight into the next block instead of idling while the host reads, decides and
    # prepares.  If the late result says "early stop", everything the extra block advanced is rewound to the evaluation
    # point (loader positions and RNG streams, optimizer / scheduler counters and moments; the weights are replaced by the
    # best snapshot anyway), so train() leaves every object exactly as the reference's loop would.
    def eval_begin(i):
        snap = _state_dict_snapshot(model)                    # device-side clone; moved to the CPU on return
        model.eval()
        h = validate_many_begin([(model, val_loader)] + ([(model, test_loader)] if test_loader is not None else []),
                                extra=[scalars[i]])
        model.train()
        return {"i": i, "snap": snap, "h": h}

    def eval_end(p):
        """Bookkeeping of the evaluation at iteration p["i"]; True = stop training."""
        nonlocal no_improve
        i = p["i"]
        res, (s,) = validate_many_end(p["h"])
        val_loss, val_acc = res[0]
        testlog = f" | Test Acc: {res[1][1]:.4f}" if test_loader is not None else ""
        if out["val_acc"] is None or val_acc > out["val_acc"]:
            out.update(iter=i, val_acc=val_acc, val_loss=val_loss, model=p["snap"])
            no_improve = 0
        else:
            no_improve += 1
        if logger is not None:
            logger.log({"val/val_loss": val_loss, "val/val_acc": val_acc, "iter": i})
        _check_micro(engine, s, f" at iter {i}")
        print(f"Iter {i} | Img Loss: {s[umlh.S_LOSS_IMG]:.4f} | Text Loss: {s[umlh.S_LOSS_TXT]:.4f} | "
              f"Img Acc: {s[umlh.S_ACC_IMG]:.4f} | Text Acc: {s[umlh.S_ACC_TXT]:.4f} | Val Loss: {val_loss:.4f} | "
              f"Val Acc {val_acc:.4f}{testlog} | Count {no_improve}/{patience}")
        if no_improve >= patience:
            print(f"=> Early stopping at Iter {i}")
            return True
        return False

    def rewind_point():
        return _rewind_point(img_src, txt_src, optimizer, scheduler)

    def rewind(r):
        _rewind(r, optimizer, scheduler)

    pending = None            # evaluation whose results have not been read yet (blockwise runs only)
    stopped = False
    i = 0
    while i < max_iters:
        if blockwise:
            # all steps up to and including the next evaluation point in ONE C call:
            # no Python, no host sync between steps
            back = rewind_point() if pending is not None else None
            i_end = _block_end(i, max_iters, eval_freq)
            n = i_end - i + 1
            bi, bt = _draw_block(img_src, txt_src, n)
            lrs = scheduler.lr_table(n)
            engine.train_steps(img_src.table(precision) if img_src else None, bi,
                               txt_src.table(precision) if txt_src else None, bt, lrs,
                               first_step=optimizer.step_count + 1, alpha=alpha, img_alpha=img_alpha,
                               scalars_out=scalars[i:i + n])
            optimizer.step_count += n
            scheduler.step(scheduler.last_epoch + n)
            if pending is not None:
                p, pending = pending, None
                if eval_end(p):                               # the block just enqueued ran past an early stop: take it back
                    rewind(back)
                    last_i = p["i"]
                    stopped = True
                    break
            i = i_end
        else:
            img_rows = img_src.next() if img_src is not None else None
            txt_rows = txt_src.next() if txt_src is not None else None
            if logger is not None:                   # with the pre-update projection, as the reference (:182-183)
                feat_sim = feature_direction_sim(model, img_rows, txt_rows)
            engine.train_step(img_rows, txt_rows, lr=optimizer.param_groups[0]["lr"], step=optimizer.step_count + 1,
                              alpha=alpha, img_alpha=img_alpha, scalars_out=scalars[i])
            optimizer.step_count += 1
            scheduler.step()
        last_i = i
        if logger is not None:
            s = scalars[i].cpu()                      # host sync: only when a logger asks for per-step values
            # finetune.py:236-240 (the CKA / mutual-kNN / in-class-distance entries belong to the
            # feature-capture mode, which is outside this path)
            gd = umlh.grad_diagnostics(s, model.head.weight.numel(), 0 if img_src is None else 1, 0 if txt_src is None else 1)
            logger.log({"train/image_loss": float(s[umlh.S_LOSS_IMG]), "train/text_loss": float(s[umlh.S_LOSS_TXT]),
                        "train/image_acc": float(s[umlh.S_ACC_IMG]), "train/text_acc": float(s[umlh.S_ACC_TXT]),
                        "train/lr": scheduler.get_last_lr()[0],
                        "train/grad_direction_sim": gd["grad_direction_sim"], "train/img_grad_norm": gd["img_grad_norm"],
                        "train/txt_grad_norm": gd["txt_grad_norm"], "train/grad_agreement_rate": gd["grad_agreement_rate"],
                        "train/feature_direction_sim": feat_sim})
        if i % eval_freq == 0:
            p = eval_begin(i)
            if blockwise and i + 1 < max_iters:
                pending = p                                   # read after the next block has been enqueued
            elif eval_end(p):
                stopped = True
                break
        i += 1
    if pending is not None and not stopped:
        eval_end(pending)
    _check_micro(engine)
    model.load_state_dict(out["model"])
    out["model"] = _to_cpu(out["model"])
    val_loss, val_acc = validate(model, val_loader, device=device)
    if logger is not None:
        logger.log({"val/best_val_loss": val_loss, "val/best_val_acc": val_acc, "iter": out["iter"]})
    print(f"=> Best Val Loss {val_loss:.4f}, Val Acc {val_acc:.4f} at Iter {out['iter']}")
    out["train_scalars"] = scalars[:last_i + 1].cpu()   # per-step losses / accuracies (extension)
    return out


def train_grouped(runs, device="cuda", eval_freq=EVAL_FREQ, precision="fp32"):
    """``train`` for MANY independent heads in lockstep: the grid points of a sweep (finetune.py:406-448) share the
    feature tables and differ in hyper-parameters, seeds and batch orders.  Between two evaluation points all live heads
    advance inside the same grouped persistent launches (``umlh_train_steps_grouped``); at an evaluation point every
    head's validation / test passes are enqueued and read back with one host synchronisation.  Each head keeps the
    exact semantics of ``train`` (evaluation cadence, strict-improvement best snapshot, patience, restored weights,
    returned dict); ``runs`` = dicts with the arguments of ``train`` (model, image_loader, text_loader, val_loader,
    test_loader, optimizer, scheduler, max_iters, alpha, patience)."""
    st = []
    for r in runs:
        model = r["model"]
        model.train()
        assert r.get("image_loader") is not None or r.get("text_loader") is not None, "At least one of the loaders should be provided"
        dev = model.head.weight.device
        img_src = _RowSource(r["image_loader"], dev, "image", precision) if r.get("image_loader") is not None else None
        txt_src = _RowSource(r["text_loader"], dev, "text", precision) if r.get("text_loader") is not None else None
        if not all(src is None or src.indexed for src in (img_src, txt_src)):
            raise umlh.UmlhError("train_grouped needs FeatureLoader inputs (device-resident tables)")
        engine = model.fused_engine(r["optimizer"], max_rows_img=img_src.capacity if img_src else 32,
                                    max_rows_txt=txt_src.capacity if txt_src else 32, precision=precision)
        engine.enable_diagnostics(False)
        st.append(dict(r, img_src=img_src, txt_src=txt_src, engine=engine, i=0, last_i=-1, no_improve=0, live=True,
                       scalars=torch.zeros(r["max_iters"], umlh.N_SCALARS, dtype=torch.float32, device=dev),
                       out={"iter": None, "val_acc": None, "model": None, "val_classwise": None, "val_loss": None,
                            "model_records": []}))
    # Evaluation results are read one round LATE, as in train(): a round enqueues the next block of every live head first and
    # only then waits for the previous round's evaluation copy, so the GPU runs while the host reads, decides and prepares.
    # A head whose late result is an early stop has its extra block rewound (loader / RNG / optimizer state; the weights are
    # replaced by its best snapshot at the end).  Needs private loader generators (a generator shared between heads cannot be
    # rewound for one of them): otherwise every round reads its own evaluation before the next block is drawn.
    def private(h):
        return all(src is None or getattr(src.loader, "generator", None) is not None for src in (h["img_src"], h["txt_src"]))
    pipelined = all(private(h) for h in st)

    def settle(due, at, handle):
        """Bookkeeping of the evaluations `handle` holds (head h evaluated at iteration at[k]): best snapshot, patience; a head
        that stops is rewound to its evaluation point if a block was enqueued for it since."""
        res, ex = validate_many_end(handle)
        k = 0
        for j, (h, i_eval) in enumerate(zip(due, at)):
            if j < len(ex) and not bool(torch.isfinite(ex[j]).all()):   # (status words are read for every head when the sweep ends)
                _check_micro(h["engine"], ex[j], f" at iter {i_eval} of grid point {h.get('name', j)}")
            val_loss, val_acc = res[k]
            k += 2 if h.get("test_loader") is not None else 1
            out = h["out"]
            if out["val_acc"] is None or val_acc > out["val_acc"]:
                out.update(iter=i_eval, val_acc=val_acc, val_loss=val_loss, model=h.pop("snap"))
                h["no_improve"] = 0
            else:
                h["no_improve"] += 1
                h.pop("snap")
            if h["no_improve"] >= h["patience"]:                    # early stopping of this head at i_eval
                if h["live"] and h.get("back") is not None:         # ... one block ago: take that block back
                    _rewind(h["back"], h["optimizer"], h["scheduler"])
                h["live"] = False
                h["i"] = h["last_i"] = i_eval

    pending = None                      # (due heads, their evaluation iterations, validate_many_begin handle)
    while any(h["live"] for h in st) or pending is not None:
        # ---- one block per live head: all steps up to and including its next evaluation point ----
        by_n = {}
        awaited = {id(h) for h in pending[0]} if pending is not None else set()
        for h in st:
            if not h["live"]:
                continue
            opt, sch = h["optimizer"], h["scheduler"]
            h["back"] = _rewind_point(h["img_src"], h["txt_src"], opt, sch) if id(h) in awaited else None
            i_end = _block_end(h["i"], h["max_iters"], eval_freq)
            n = i_end - h["i"] + 1
            bi, bt = _draw_block(h["img_src"], h["txt_src"], n)
            job = dict(engine=h["engine"], img_table=h["img_src"].table(precision) if h["img_src"] else None, img_index_batches=bi,
                       txt_table=h["txt_src"].table(precision) if h["txt_src"] else None, txt_index_batches=bt,
                       lrs=sch.lr_table(n), first_step=opt.step_count + 1, alpha=h["alpha"], img_alpha=1.0,
                       scalars_out=h["scalars"][h["i"]:h["i"] + n])
            by_n.setdefault(n, []).append(job)
            opt.step_count += n
            sch.step(sch.last_epoch + n)
            h["i"] = h["last_i"] = i_end
        for n, jobs in by_n.items():
            umlh.train_steps_grouped(jobs, n)
        # ---- the previous round's evaluations (the blocks above are already running) ----
        if pending is not None:
            p, pending = pending, None
            settle(*p)
        # ---- evaluation points (every live head sits on one, or on its final iteration) ----
        due = [h for h in st if h["live"] and h["i"] % eval_freq == 0]
        if due:
            pairs, extra = [], []
            for h in due:
                h["snap"] = _state_dict_snapshot(h["model"])
                h["model"].eval()
                pairs.append((h["model"], h["val_loader"]))
                if h.get("test_loader") is not None:
                    pairs.append((h["model"], h["test_loader"]))
                extra.append(h["scalars"][h["i"]])
            handle = validate_many_begin(pairs, extra=extra)
            for h in due:
                h["model"].train()
                h["back"] = None
            pending = (due, [h["i"] for h in due], handle)
            if not pipelined:                                       # shared generators: decide before the next block is drawn
                p, pending = pending, None
                settle(*p)
        for h in st:
            if h["live"]:
                h["i"] += 1
                if h["i"] >= h["max_iters"]:
                    h["live"] = False
    outs = []
    for h in st:
        _check_micro(h["engine"])
        h["model"].load_state_dict(h["out"]["model"])
        h["out"]["model"] = _to_cpu(h["out"]["model"])
        h["out"]["train_scalars"] = h["scalars"][:h["last_i"] + 1].cpu()
        outs.append(h["out"])
    return outs


EVAL_SLAB = 4096     # rows per forward launch of a whole-table evaluation


def _slab_evaluable(loader):
    return hasattr(loader, "iter_index") and not loader.shuffle and not loader.drop_last and len(loader.table) > 0


def _eval_enqueue(model, val_loader):
    """Enqueue the whole-table evaluation of an unshuffled ``FeatureLoader`` (one ``umlh_eval_rows`` launch per
    4096-row slab) and return the device tensor of per-row {CE, correct} -- no host synchronisation."""
    dev = model.head.weight.device
    val_loader.iter_index()                       # iter(loader): the iterator's base seed is drawn as in the reference
    t = val_loader.table
    n = len(t)
    stats = torch.empty(n, 2, dtype=torch.float32, device=dev)
    engine = model._infer_engine(min(n, EVAL_SLAB))
    for s0 in range(0, n, EVAL_SLAB):
        m = min(EVAL_SLAB, n - s0)
        engine.eval_rows(umlh.RowBatch(t.features[s0:s0 + m], t.labels[s0:s0 + m]), stats[s0:s0 + m])
    return stats


def _eval_finish(stats_cpu, bs):
    """(val_loss, val_acc) from per-row stats: mean over batches of the per-batch mean CE, global mean of correct."""
    st = stats_cpu.double().numpy()
    n = st.shape[0]
    nb = (n + bs - 1) // bs
    import numpy as np
    sums = np.add.reduceat(st[:, 0], np.arange(0, n, bs))           # per-batch CE sums, in row order
    counts = np.full(nb, bs, dtype=np.float64)
    counts[-1] = n - (nb - 1) * bs
    return float((sums / counts).sum() / nb), float(st[:, 1].sum() / n)


_PINNED = threading.local()   # per thread (sweep_farm runs train() on worker threads): ring of pinned host buffers for the
                              # evaluation read-backs (hipHostMalloc per call would cost more than the copy)


def _pinned(n):
    ring = getattr(_PINNED, "ring", None)
    if ring is None:
        ring = _PINNED.ring = []
    for k, buf in enumerate(ring):
        if buf.numel() >= n:
            ring.append(ring.pop(k))                       # least recently handed out first
            return ring[-1][:n]
    if len(ring) >= 4:
        ring.pop(0)
    ring.append(torch.empty(max(n, 1 << 16), dtype=torch.float32, pin_memory=True))
    return ring[-1][:n]


def validate_many_begin(pairs, extra=None):
    """First half of ``validate_many``: enqueue every evaluation and ONE asynchronous device-to-host copy of all per-row
    statistics (+ ``extra``) into pinned memory; returns a handle for ``validate_many_end``.  Nothing waits for the GPU:
    the caller can enqueue the next training block before it looks at the results."""
    extra = list(extra or [])
    if not all(_slab_evaluable(ld) for _, ld in pairs):
        return {"done": ([validate(m, ld) for m, ld in pairs], [e.cpu() for e in extra])}
    stats = [_eval_enqueue(m, ld) for m, ld in pairs]
    dev_flat = torch.cat([st.reshape(-1) for st in stats] + [e.reshape(-1).to(torch.float32) for e in extra])
    host = _pinned(dev_flat.numel())
    host.copy_(dev_flat, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev_flat.device))
    return {"host": host, "event": ev, "keep": dev_flat, "pairs": pairs, "sizes": [st.numel() for st in stats], "extra": extra}


def validate_many_end(h):
    """(results, extras on the CPU) of a ``validate_many_begin`` handle; waits for its copy only."""
    if "done" in h:
        return h["done"]
    h["event"].synchronize()
    flat = h["host"].clone()                               # the pinned buffer goes back to the ring
    out, pos = [], 0
    for (m, ld), n in zip(h["pairs"], h["sizes"]):
        out.append(_eval_finish(flat[pos:pos + n].reshape(-1, 2), ld.batch_size))
        pos += n
    ex = []
    for e in h["extra"]:
        ex.append(flat[pos:pos + e.numel()].reshape(e.shape))
        pos += e.numel()
    return out, ex


def validate_many(pairs, extra=None):
    """``validate`` for several (model, loader) pairs with ONE host synchronisation: every evaluation is enqueued
    first (``_eval_enqueue``), then all per-row statistics come back in a single device-to-host copy.  Used at the
    evaluation points of ``train`` (val + test) and of the grouped sweep (every head's val + test).  ``extra``: device
    float tensors (e.g. the step's scalar rows) that ride on the same copy; returns (results, extras on the CPU)."""
    extra = list(extra or [])
    if not all(_slab_evaluable(ld) for _, ld in pairs):
        return [validate(m, ld) for m, ld in pairs], [e.cpu() for e in extra]
    stats = [_eval_enqueue(m, ld) for m, ld in pairs]
    flat = torch.cat([st.reshape(-1) for st in stats] + [e.reshape(-1).to(torch.float32) for e in extra]).cpu()
    out, pos = [], 0
    for (m, ld), st in zip(pairs, stats):
        out.append(_eval_finish(flat[pos:pos + st.numel()].reshape(-1, 2), ld.batch_size))
        pos += st.numel()
    ex = []
    for e in extra:
        ex.append(flat[pos:pos + e.numel()].reshape(e.shape))
        pos += e.numel()
    return out, ex


def validate(model, val_loader, device="cuda"):
    """val_acc = mean over all rows of (argmax == label); val_loss = mean over batches
    of the per-batch mean CE (not sample weighted) -- finetune.py:291-315.  One host read at the end.

    An unshuffled ``FeatureLoader`` is evaluated as whole 4096-row slabs of its device table
    (``umlh_eval_rows`` keeps the per-row CE / correct flag, the per-batch means are formed from them), i.e. one
    launch per slab instead of one per batch: at the reference's batch size of 32 a validation pass over
    Caltech101's val + test rows is 1 + 1 launches instead of 13 + 78."""
    dev = model.head.weight.device
    model.eval()
    src_indexed = hasattr(val_loader, "iter_index")
    if _slab_evaluable(val_loader):
        out = _eval_finish(_eval_enqueue(model, val_loader).cpu(), val_loader.batch_size)
        model.train()
        return out
    rows, slots = [], []
    it = val_loader.iter_index() if src_indexed else iter(val_loader)
    engine = None
    for batch in it:
        if src_indexed:
            t = val_loader.table
            rb = umlh.RowBatch(t.features, t.labels, batch)
        else:
            rb = umlh.RowBatch(batch["img"].to(dev, torch.float32).contiguous(),
                               batch["label"].to(dev, torch.int64).contiguous())
        n = rb.n_rows()
        if engine is None or engine.cfg.max_rows_img < n:
            engine = model._infer_engine(n)
        slot = torch.zeros(umlh.N_SCALARS, dtype=torch.float32, device=dev)
        engine.eval_batch(rb, slot)
        rows.append(n)
        slots.append(slot)
    s = torch.stack(slots).cpu()
    counts = torch.tensor(rows, dtype=torch.float32)
    val_acc = (s[:, umlh.S_CORRECT].sum() / counts.sum()).item()
    val_loss = (s[:, umlh.S_LOSS_SUM] / counts).mean().item()
    model.train()
    return val_loss, val_acc


def hparam_str(optim, lr, wd, batch_size, iters, dropout, learnable_temp):
    base = f"optim_{optim}-lr_{lr}-wd_{wd}-bs_{batch_size}-iters_{iters}"
    if dropout is not None:
        base += f"-dropout_{dropout}"
    if learnable_temp is True:
        base += "-learnable_temp"
    return base


def feature_tables(img_train, img_val, img_test, text_ds, device):
    """The four device-resident tables of one dataset (uploaded once; a sweep shares them)."""
    return {"train": FeatureTable(*img_train, device), "val": FeatureTable(*img_val, device),
            "test": FeatureTable(*img_test, device),
            "text": FeatureTable(text_ds.input_tensor, text_ds.label_tensor, device, text_ds.eot_indices)}


def wants_zero_shot_init(classifier_init, modality, common_dim, text_indim):
    """reference finetune.py:362-363: the text-derived head only for crossmodal runs, or an image-only run whose
    ``common_dim`` equals the text width; every other unimodal run (default ``common_dim`` = 0, and all text-only
    runs) keeps nn.Linear's random init."""
    return classifier_init == "zeroshot" and (modality == "crossmodal" or
                                              (modality == "image" and int(common_dim or 0) == int(text_indim)))


def setup_feature_run(img_train, img_val, img_test, text_ds, hparams, *, num_classes, modality="crossmodal",
                      alpha=1.0, classifier_init="zeroshot", use_clip=False, clip_logit=4.60517, text_indim=None,
                      device="cuda:0", eval_test=True, precision="fp32", eval_freq=EVAL_FREQ, tables=None,
                      generator=None, order_rng="torch-cpu", model=None, common_dim=None):
    """``setup()`` (finetune.py:323-404) for pre-extracted features: builds the model,
    optimizer, scheduler and the four loaders with the reference's wiring, trains, tests.

    img_* are (features [N,d], labels [N]) pairs; text_ds a TextTensorDataset.  ``tables`` (from
    ``feature_tables``), ``generator`` (private seed source of the loaders) and a pre-built ``model``
    are what the concurrent sweep passes in.  ``text_indim`` is ``args.text_indim`` in crossmodal mode and
    ``args.common_dim`` otherwise (the second argument of ``UML(...)``, reference :343-346); ``common_dim``
    overrides the latter."""
    d_img = img_train[0].shape[1]
    d_txt = text_ds.input_tensor.shape[1]
    if model is not None:
        pass
    elif use_clip:
        model = UMLClip(d_img, num_classes, logit_scale_init=clip_logit, bias=False,
                        learnable_temp=hparams["learnable_temp"])
    else:
        tin = (d_txt if text_indim is None else text_indim) if modality == "crossmodal" else (text_indim or 0)
        model = UML(d_img, tin, num_classes, bias=False, learnable_temp=hparams["learnable_temp"])
    if tables is None:
        tables = feature_tables(img_train, img_val, img_test, text_ds, device)
    model.to(device)
    if common_dim is None:
        common_dim = 0 if modality == "crossmodal" else (text_indim or 0)
    if wants_zero_shot_init(classifier_init, modality, common_dim, d_txt):
        model.zero_shot_init(text_ds)
    optimizer = build_optimizer(model.parameters(), hparams["optim"], hparams["lr"], hparams["weight_decay"])
    scheduler = build_lr_scheduler(optimizer, hparams["lr_scheduler"], hparams["warmup_iter"], hparams["max_iter"],
                                   warmup_type=hparams["warmup_type"], warmup_lr=hparams["warmup_min_lr"])
    bs = hparams["batch_size"]
    kw = {"order_rng": order_rng, "generator": generator}
    image_loader = FeatureLoader(tables["train"], bs, shuffle=True, kind="image", **kw)
    text_loader = FeatureLoader(tables["text"], bs, shuffle=True, kind="text", **kw)
    if modality == "image":
        text_loader = None
    elif modality == "text":
        image_loader = None
    val_loader = FeatureLoader(tables["val"], bs, shuffle=False, kind="image", **kw)
    test_loader = FeatureLoader(tables["test"], bs, shuffle=False, kind="image", **kw)
    result = train(model, image_loader, text_loader, val_loader, test_loader if eval_test else None, optimizer,
                   scheduler, device=device, max_iters=hparams["max_iter"], alpha=alpha, eval_freq=eval_freq,
                   patience=hparams["patience"], precision=precision)
    test_loss, test_acc = validate(model, test_loader, device=device)
    return {"test_acc": test_acc, "test_loss": test_loss, "val_acc": result["val_acc"], "model": result["model"],
            "iter": result["iter"], "train_scalars": result["train_scalars"]}


# --------------------------------------------------------------------------------------------- #
# sweep / main over feature files  (reference: finetune.py:58-77 naming, :323-448 setup/sweep,
# :451-510 main).  Same flag names on `args`; wandb / Tee logging and raw-image loading are out
# of scope -- images come from the feature files features.py wrote.
# --------------------------------------------------------------------------------------------- #
FLAG = 0   # run despite an existing result file when set to 1 (reference finetune.py:31)


def savedir(outdir, dataset, encoder, train_shot, seed, text_type, text_shots, image_augmentation, mode,
            init_mode="zeroshot", alpha=0.0, text_bs=0, custom_name="", args=None):
    from features import get_few_shot_setup_name
    benchname = "-".join([dataset, get_few_shot_setup_name(train_shot, seed)])
    text_name = f"text_{text_type}" + (f"_n_{text_shots}" if text_shots is not None else "")
    image_name = f"image_{image_augmentation}_{custom_name}"
    mod_name = (f"finetune-{text_name}-{image_name}" if mode == "crossmodal"
                else f"finetune-{image_name}" if mode == "image" else text_name)
    if mode == "crossmodal":
        mod_name = f"{mod_name}-alpha_{alpha}"
    if text_bs > 0:
        mod_name = f"{mod_name}-text_bs_{text_bs}"
    if args is not None and mode != "crossmodal":
        mod_name = f"{mod_name}-common_dim_{getattr(args, 'common_dim', 0)}"
    return os.path.join(outdir, benchname, encoder.replace("/", "-"), mod_name, init_mode)


def setup(datasets, hparams, args):
    """One hyper-parameter point: build model / optimizer / scheduler / loaders, train, test, save
    ``test_result.pth`` (skipped when it exists and FLAG is 0, reference :330-333)."""
    ckpt_dir = os.path.join(args.savepath, hparam_str(hparams["optim"], hparams["lr"], hparams["weight_decay"],
                                                      hparams["batch_size"], hparams["max_iter"], hparams["dropout"],
                                                      hparams["learnable_temp"]))
    os.makedirs(ckpt_dir, exist_ok=True)
    test_path = os.path.join(ckpt_dir, "test_result.pth")
    if os.path.exists(test_path) and not FLAG:
        print(f"=> Skipping {ckpt_dir} as it already exists!")
        return torch.load(test_path, map_location="cpu", weights_only=True)
    res = setup_feature_run(datasets["img_tr"], datasets["img_val"], datasets["img_te"], datasets["text_ds"], hparams,
                            num_classes=args.nclasses, modality=args.modality, alpha=args.alpha,
                            classifier_init=args.classifier_init, use_clip=args.use_clip, clip_logit=args.logit,
                            text_indim=getattr(args, "text_indim", None) if args.modality == "crossmodal"
                            else getattr(args, "common_dim", 0), device=args.device,
                            eval_test=getattr(args, "eval_test", True), precision=getattr(args, "precision", "fp32"),
                            tables=datasets.get("tables"))
    test_dict = {"test_acc": res["test_acc"], "val_acc": res["val_acc"], "model": res["model"], "iter": res["iter"]}
    print(f"=> Test Acc: {res['test_acc']:.4f}")
    torch.save(test_dict, test_path)
    return test_dict


def _grid(hyperparams):
    from itertools import product
    grid = {k: (v if isinstance(v, list) else [v]) for k, v in hyperparams.items()}
    keys = list(grid)
    return [dict(zip(keys, combo)) for combo in product(*[grid[k] for k in keys])]


def _report(results, args):
    torch.save(results, os.path.join(args.savepath, "results.pth"))
    best = int(torch.argmax(torch.tensor(results["val_acc"])))
    print(f"=> [FINAL] Best Val Acc: {results['val_acc'][best]:.4f} | Best Test Acc: {results['test_acc'][best]:.4f}")
    print(f"=> [FINAL] Best Hyperparameters: {results['hparams'][best]}")
    return results, results["val_acc"][best], results["test_acc"][best]


def sweep(datasets, hyperparams, args):
    """Cartesian product over the list-valued entries of a HYPER_DICT grid, in key order
    (reference :406-448); returns (results, best_val_acc, best_test_acc).

    ``args.sweep_workers > 1`` runs the grid points concurrently on one GPU (``sweep_farm``)."""
    if getattr(args, "sweep_mode", "") == "grouped":
        return sweep_grouped(datasets, hyperparams, args)
    if int(getattr(args, "sweep_workers", 1) or 1) > 1:
        return sweep_farm(datasets, hyperparams, args, int(args.sweep_workers))
    results = {"test_acc": [], "val_acc": [], "hparams": [], "model_records": []}
    # the four feature tables are uploaded once for the whole sweep (every grid point reads the same rows)
    if "tables" not in datasets and isinstance(datasets.get("img_tr"), (tuple, list)):
        datasets = dict(datasets, tables=feature_tables(datasets["img_tr"], datasets["img_val"], datasets["img_te"],
                                                        datasets["text_ds"], torch.device(getattr(args, "device", "cuda:0"))))
    for idx, hp in enumerate(_grid(hyperparams)):
        print(f"=> Running {idx + 1}: {hp}")
        out = setup(datasets, hp, args)
        results["test_acc"].append(out["test_acc"])
        results["val_acc"].append(out["val_acc"])
        results["hparams"].append(hp)
    return _report(results, args)


def farm_seed(base_seed, idx):
    """Seed of grid point ``idx``'s private generator (loader orders) and of its model init."""
    return (int(base_seed) if base_seed is not None and int(base_seed) >= 0 else 0) * 100003 + 7919 * (idx + 1)


def sweep_grouped(datasets, hyperparams, args):
    """The sweep as ONE grouped job (``args.sweep_mode = "grouped"``): every grid point becomes a head of the same
    persistent launches (``train_grouped`` / ``umlh_train_steps_grouped``), all reading the same device-resident
    feature tables -- the [G, C, d] grouped multi-head farm of SURVEY 8(f) rank 2.  Like ``sweep_farm`` every point
    draws its loader orders from a private generator seeded by ``farm_seed(args.seed, k)`` and initialises its model
    under that seed, so a point's result is independent of which other points run beside it: it equals the isolated run
    of that point (``setup_feature_run`` with the same generator) bit for bit."""
    from engine.models.head import UML, UMLClip
    points = _grid(hyperparams)
    dev = torch.device(args.device)
    precision = getattr(args, "precision", "fp32")
    tables = feature_tables(datasets["img_tr"], datasets["img_val"], datasets["img_te"], datasets["text_ds"], dev)
    if precision == "bf16":
        for t in tables.values():
            t.features_bf16()
    d_img, d_txt = tables["train"].features.shape[1], tables["text"].features.shape[1]
    text_indim = getattr(args, "text_indim", None) if args.modality == "crossmodal" else getattr(args, "common_dim", 0)
    common_dim = 0 if args.modality == "crossmodal" else (text_indim or 0)
    runs, slots = [], []
    results = [None] * len(points)
    for idx, hp in enumerate(points):
        ckpt_dir = os.path.join(args.savepath, hparam_str(hp["optim"], hp["lr"], hp["weight_decay"], hp["batch_size"],
                                                          hp["max_iter"], hp["dropout"], hp["learnable_temp"]))
        os.makedirs(ckpt_dir, exist_ok=True)
        test_path = os.path.join(ckpt_dir, "test_result.pth")
        if os.path.exists(test_path) and not FLAG:
            print(f"=> Skipping {ckpt_dir} as it already exists!")
            results[idx] = torch.load(test_path, map_location="cpu", weights_only=True)
            continue
        torch.manual_seed(farm_seed(args.seed, idx))
        if args.use_clip:
            model = UMLClip(d_img, args.nclasses, logit_scale_init=args.logit, bias=False, learnable_temp=hp["learnable_temp"])
        else:
            tin = (d_txt if text_indim is None else text_indim) if args.modality == "crossmodal" else (text_indim or 0)
            model = UML(d_img, tin, args.nclasses, bias=False, learnable_temp=hp["learnable_temp"])
        model.to(dev)
        if wants_zero_shot_init(args.classifier_init, args.modality, common_dim, d_txt):
            model.zero_shot_init(datasets["text_ds"])
        optimizer = build_optimizer(model.parameters(), hp["optim"], hp["lr"], hp["weight_decay"])
        scheduler = build_lr_scheduler(optimizer, hp["lr_scheduler"], hp["warmup_iter"], hp["max_iter"],
                                       warmup_type=hp["warmup_type"], warmup_lr=hp["warmup_min_lr"])
        gen = torch.Generator()
        gen.manual_seed(farm_seed(args.seed, idx))
        kw = {"order_rng": getattr(args, "order_rng", "torch-cpu"), "generator": gen}
        bs = hp["batch_size"]
        image_loader = FeatureLoader(tables["train"], bs, shuffle=True, kind="image", **kw)
        text_loader = FeatureLoader(tables["text"], bs, shuffle=True, kind="text", **kw)
        if args.modality == "image":
            text_loader = None
        elif args.modality == "text":
            image_loader = None
        val_loader = FeatureLoader(tables["val"], bs, shuffle=False, kind="image", **kw)
        test_loader = FeatureLoader(tables["test"], bs, shuffle=False, kind="image", **kw)
        runs.append(dict(model=model, image_loader=image_loader, text_loader=text_loader, val_loader=val_loader,
                         test_loader=test_loader if getattr(args, "eval_test", True) else None, optimizer=optimizer,
                         scheduler=scheduler, max_iters=hp["max_iter"], alpha=args.alpha, patience=hp["patience"],
                         final_test_loader=test_loader))
        slots.append((idx, test_path))
    outs = train_grouped(runs, device=dev, precision=precision)
    finals, _ = validate_many([(r["model"], r["final_test_loader"]) for r in runs]) if runs else ([], [])
    for (idx, test_path), out, (test_loss, test_acc) in zip(slots, outs, finals):
        test_dict = {"test_acc": test_acc, "val_acc": out["val_acc"], "model": out["model"], "iter": out["iter"]}
        torch.save(test_dict, test_path)
        results[idx] = test_dict
    res = {"test_acc": [o["test_acc"] for o in results], "val_acc": [o["val_acc"] for o in results],
           "hparams": points, "model_records": []}
    return _report(res, args)


def sweep_farm(datasets, hyperparams, args, workers):
    """The sweep as a farm: the grid points of one dataset are independent head fine-tunes over the
    SAME feature tables (18 AdamW points in ``clip_linear``), each far too small to fill 256 CUs
    (batch 32: a handful of workgroups per kernel).  The tables are uploaded once; every grid point
    gets its own engine, its own HIP stream and a host thread that enqueues whole evaluation
    intervals through ``umlh_train_steps`` (a C loop, no GIL), so the small kernels of different
    points can overlap on the chip.  No collective, no shared mutable state.  Measured on MI355X this
    does not beat the sequential sweep for batch-32 steps -- the dispatch rate of dependent small kernels is the
    limit -- so ``sweep_workers`` defaults to 1; the farm is for grids whose steps are large enough to be GPU-bound
    individually yet too small to fill the chip.

    Unlike the sequential sweep -- where point k's shuffles depend on how much global RNG the
    points before it consumed -- every point draws from a private generator seeded by
    ``farm_seed(args.seed, k)``: results are reproducible and independent of scheduling, but not
    batch-for-batch those of the sequential order (``sweep_workers=1`` keeps that)."""
    from concurrent.futures import ThreadPoolExecutor
    from engine.models.head import UML, UMLClip
    points = _grid(hyperparams)
    dev = torch.device(args.device)
    tables = feature_tables(datasets["img_tr"], datasets["img_val"], datasets["img_te"], datasets["text_ds"], dev)
    if getattr(args, "precision", "fp32") == "bf16":
        for t in tables.values():
            t.features_bf16()
    d_img, d_txt = tables["train"].features.shape[1], tables["text"].features.shape[1]
    text_indim = getattr(args, "text_indim", None) if args.modality == "crossmodal" else getattr(args, "common_dim", 0)
    jobs = []
    for idx, hp in enumerate(points):                      # models built here, in order: deterministic inits
        ckpt_dir = os.path.join(args.savepath, hparam_str(hp["optim"], hp["lr"], hp["weight_decay"], hp["batch_size"],
                                                          hp["max_iter"], hp["dropout"], hp["learnable_temp"]))
        os.makedirs(ckpt_dir, exist_ok=True)
        test_path = os.path.join(ckpt_dir, "test_result.pth")
        if os.path.exists(test_path) and not FLAG:
            jobs.append((idx, hp, test_path, None))
            continue
        torch.manual_seed(farm_seed(args.seed, idx))
        if args.use_clip:
            model = UMLClip(d_img, args.nclasses, logit_scale_init=args.logit, bias=False, learnable_temp=hp["learnable_temp"])
        else:
            tin = (d_txt if text_indim is None else text_indim) if args.modality == "crossmodal" else (text_indim or 0)
            model = UML(d_img, tin, args.nclasses, bias=False, learnable_temp=hp["learnable_temp"])
        jobs.append((idx, hp, test_path, model))
    main_stream = torch.cuda.current_stream(dev)

    def run(job):
        idx, hp, test_path, model = job
        if model is None:
            print(f"=> Skipping {os.path.dirname(test_path)} as it already exists!")
            return torch.load(test_path, map_location="cpu", weights_only=True)
        gen = torch.Generator()
        gen.manual_seed(farm_seed(args.seed, idx))
        stream = torch.cuda.Stream(dev)
        stream.wait_stream(main_stream)                    # the shared tables were uploaded there
        with torch.cuda.stream(stream):
            res = setup_feature_run(datasets["img_tr"], datasets["img_val"], datasets["img_te"], datasets["text_ds"], hp,
                                    num_classes=args.nclasses, modality=args.modality, alpha=args.alpha,
                                    classifier_init=args.classifier_init, use_clip=args.use_clip, clip_logit=args.logit,
                                    text_indim=text_indim, device=dev, eval_test=getattr(args, "eval_test", True),
                                    precision=getattr(args, "precision", "fp32"), tables=tables, generator=gen,
                                    order_rng=getattr(args, "order_rng", "torch-cpu"), model=model)
            stream.synchronize()
        test_dict = {"test_acc": res["test_acc"], "val_acc": res["val_acc"], "model": res["model"], "iter": res["iter"]}
        torch.save(test_dict, test_path)
        return test_dict

    import threading
    import umlh.head_engine as _he
    _he.ENQUEUE_LOCK = threading.Lock()
    try:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            outs = list(pool.map(run, jobs))
    finally:
        _he.ENQUEUE_LOCK = None
    results = {"test_acc": [o["test_acc"] for o in outs], "val_acc": [o["val_acc"] for o in outs],
               "hparams": points, "model_records": []}
    return _report(results, args)


def main(args):
    """``finetune.main`` (reference :451-510) over pre-extracted feature files: seeds, resolves the
    feature paths with the reference's scheme, builds the text dataset, sweeps HYPER_DICT[args.hyperparams]."""
    import features as F_
    from engine.tools.utils import set_random_seed
    if args.seed >= 0:
        set_random_seed(args.seed)
    args.device = getattr(args, "device", None) or "cuda:0"
    args.use_clip = getattr(args, "vision_model", "") == "" and getattr(args, "language_model", "") == ""
    enc_img = args.clip_encoder if args.use_clip else args.vision_model
    enc_txt = args.clip_encoder if args.use_clip else args.language_model
    encoder_name = enc_img if args.use_clip else f"{args.vision_model}-{args.language_model}"
    args.savepath = savedir(args.result_dir, args.dataset, encoder_name, args.train_shot, args.seed, args.text_type,
                            args.text_shot, args.image_augmentation, args.modality, args.classifier_init, args.alpha,
                            getattr(args, "text_batch_size", 0), getattr(args, "custom_name", ""), args)
    os.makedirs(args.savepath, exist_ok=True)
    text = F_.load_text_features(F_.text_outdir(args.feature_dir, enc_txt, args.dataset, args.text_type))
    shots = args.text_shot
    text_ds = TextTensorDataset(text["features"], text["labels"], text["eot_indices"],
                                n_shots=int(shots) if (shots != "average" and shots is not None) else shots)
    tr = F_.load_image_train_features(F_.img_outdir(args.feature_dir, enc_img, args.dataset, args.image_augmentation,
                                                    args.train_shot, args.seed, "train"))
    te = F_.load_image_test_features(F_.img_outdir(args.feature_dir, enc_img, args.dataset, args.image_augmentation,
                                                   args.train_shot, args.seed, "test"))
    args.img_indim, args.text_indim = tr["train"][0].shape[1], text["features"].shape[1]
    lab2cname = tr.get("lab2cname") or text.get("lab2cname")
    args.nclasses = len(lab2cname) if lab2cname else int(max(tr["train"][1].max(), te["test"][1].max())) + 1
    datasets = {"img_tr": tr["train"], "img_val": tr["val"], "img_te": te["test"], "text_ds": text_ds}
    return sweep(datasets, HYPER_DICT[args.hyperparams] if isinstance(args.hyperparams, str) else args.hyperparams, args)
