"""How far the host runs ahead of the GPU in the eager iteration: time until the Python loop has ENQUEUED K iterations
vs time until the GPU has finished them.  enqueue ~ total means the launch loop, not the GPU, sets the rate.
  python tools/cpu_ahead.py [config]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
import __graft_entry__ as ge  # noqa: E402
ge.build()
dev = torch.device("cuda:0")
fit = bench.Fit(cfg, 0, dev, "fused")
for _ in range(10):
    fit.step()
torch.cuda.synchronize()
for K in (20, 50):
    t0 = time.perf_counter()
    for _ in range(K):
        fit.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{cfg}: K={K}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/iteration, GPU done {1e3 * (t2 - t0) / K:.3f} ms/iteration "
          f"(host ahead by {1e3 * (t2 - t1):.2f} ms at the end)", flush=True)
# host cost of the launch loop alone: same iteration with the GPU idle in between
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit.step()
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print(f"{cfg}: host time of one iteration's launches (GPU idle at the start): {1e3 * min(ts):.3f} ms min, {1e3 * sorted(ts)[5]:.3f} ms median")
