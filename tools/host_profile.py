"""cProfile of the host side of the iteration (no synchronisation inside the window): where the Python thread's ~2 ms go.
    python tools/host_profile.py [config]          measurement tool, not part of the product path"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import __graft_entry__ as ge
ge.build()
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
f = bench.Fit(cfg, 0, dev, "fused")
for _ in range(10):
    f.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
n = 8
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    f.step()
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{cfg}: {1e3 * (t1 - t0) / n:.3f} ms host per iteration (under cProfile)")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
