"""Host-side profile of ONE sequential sweep point (cProfile over finetune.setup_feature_run): scripts/prof_seq.py [iters] [precision]."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
iters = sys.argv[1] if len(sys.argv) > 1 else "4000"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
sys.argv = ["bench_farm.py", "--iters", iters, "--workers", "1", "--precision", prec]
import bench_farm  # noqa: E402
import finetune  # noqa: E402

orig = finetune.setup_feature_run
state = {"n": 0}


def wrapped(*a, **k):
    state["n"] += 1
    if state["n"] != 30:                    # a point of the second (timed) pass
        return orig(*a, **k)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    r = orig(*a, **k)
    pr.disable()
    print("setup_feature_run wall %.4f s" % (time.perf_counter() - t0), file=sys.stderr)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(26)
    print(s.getvalue()[:6000], file=sys.stderr)
    return r


finetune.setup_feature_run = wrapped
bench_farm.main()
