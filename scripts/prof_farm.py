import cProfile, pstats, sys, io
sys.argv = ["bench_farm.py", "--iters", "300", "--workers", "1", "--grouped"]
sys.path.insert(0, "scripts")
import bench_farm
pr = cProfile.Profile()
pr.enable()
bench_farm.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(45)
print(s.getvalue()[:9000])
