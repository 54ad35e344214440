"""cProfile of the host side of a training step at B = 1 (launch-bound: shows where Python spends the per-launch time)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--batch", "1"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
