import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
import __graft_entry__ as ge
ge.build()
dev = torch.device("cuda:0")
for cfg in ("library", "default", "snail"):
    g = bench.grouped_fits(cfg, [0], dev)
    for graph in (False, True):
        t, _ = bench.timed_run_grouped(g, 100, 10, graph, lambda: None)
        print(cfg, "grouped x1", "graph" if graph else "eager", round(100 / t, 1), "it/s", flush=True)
    del g
    torch.cuda.empty_cache()
