"""Host-side profile of the grouped sweep (cProfile over finetune.sweep_grouped only): scripts/prof_farm.py [iters]."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.argv = ["bench_farm.py", "--iters", sys.argv[1] if len(sys.argv) > 1 else "4000", "--workers", "1", "--grouped"]
import bench_farm  # noqa: E402
import finetune  # noqa: E402

orig = finetune.sweep_grouped


def wrapped(*a, **k):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    r = orig(*a, **k)
    pr.disable()
    print("sweep_grouped wall %.3f s" % (time.perf_counter() - t0), file=sys.stderr)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
    print(s.getvalue()[:7000], file=sys.stderr)
    return r


finetune.sweep_grouped = wrapped
bench_farm.main()
