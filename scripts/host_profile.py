"""Host-side (Python) cost of one bench step: cProfile over many steps, plus wall time per step with the GPU
queue left to run ahead (a step whose host time exceeds its device time is host-bound)."""
import cProfile, pstats, os, sys, time, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.Workload(torch.device("cuda", 0), torch.bfloat16, "nhwc")
for _ in range(5):
    bench.disconnected_step(w)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    bench.disconnected_step(w)
torch.cuda.synchronize()
print("ms/step (wall)", round((time.perf_counter() - t0) / 50 * 1e3, 4))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    bench.disconnected_step(w)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:60]))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:50]))
