"""Host-side contention of 8 ranks on one node, measured on a 1-GPU box (no 8-GPU node is available to the build):
the REAL rank runs the headline fit (default skip-net, 512x512, fused closure, eager launches) while 7 DUMMY ranks load
the host the way 7 more ranks would.  Two kinds of dummies, both paced at the real rank's cadence (one iteration per
~6 ms) and each confined, like `bench.pin_to_gpu_numa` does for N ranks, to its 1/8 share of this process' cores:

  cpu     the bench's DIP_BENCH_SELFTEST-style step: pure host work (numpy), no HIP calls -- what 7 other ranks cost in
          cores / caches / memory bandwidth;
  launch  the SAME engine code path (Python planner walk + ctypes + ~250 hipLaunchKernel per iteration) on the default net
          at 64x64, i.e. the full host cost of a rank, but with GPU work of a few us per kernel.  On an 8-GPU node each
          rank owns its GPU; here the dummies' kernels land on the one GPU the real rank uses, so this arm ALSO contains
          command-processor / GPU sharing that the real node does not have: it is an upper bound on the loss.

Reported for the real rank: it/s over K iterations, host time until the K iterations are ENQUEUED (ms/iteration), and
the host time of one iteration's launch loop with the GPU idle (min / median), solo vs each arm.

  python tools/host_contention.py [--k 200] > profiles/r0N_host_contention.txt"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PERIOD = 0.006


def share_of_cores(rank, world=8):
    allowed = sorted(os.sched_getaffinity(0))
    share = allowed[rank * len(allowed) // world:(rank + 1) * len(allowed) // world] or allowed
    os.sched_setaffinity(0, share)
    return len(share), len(allowed)


def dummy(kind, rank):
    import numpy as np
    share_of_cores(rank)
    step = None
    if kind == "cpu":
        def step():
            float(np.sum(np.arange(20000, dtype=np.float64) * (rank + 1)))      # bench.selftest_rank's step
        sync = lambda: None
    else:
        import torch
        import bench
        import __graft_entry__ as ge
        ge.build()
        torch.set_num_threads(1)
        bench.CONFIGS["default"] = dict(size=(64, 64), desc="dummy rank")
        fit = bench.Fit("default", rank, torch.device("cuda:0"), "fused")
        for _ in range(5):
            fit.step()
        torch.cuda.synchronize()
        step, sync = fit.step, torch.cuda.synchronize
    print("READY", flush=True)
    n, t_next = 0, time.perf_counter()
    while True:
        step()
        n += 1
        if n % 8 == 0:
            sync()
        t_next += PERIOD
        d = t_next - time.perf_counter()
        if d > 0:
            time.sleep(d)
        else:
            t_next = time.perf_counter()


def measure(fit, K):
    import torch
    for _ in range(10):
        fit.step()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(K):
            fit.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = (K / (t2 - t0), 1e3 * (t1 - t0) / K)
        best = r if best is None or r[0] > best[0] else best
    ts = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit.step()
        ts.append(1e3 * (time.perf_counter() - t0))
        torch.cuda.synchronize()
    ts.sort()
    return {"it_s": best[0], "enqueue_ms": best[1], "loop_min_ms": ts[0], "loop_med_ms": ts[len(ts) // 2]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--dummy", default=None)
    ap.add_argument("--rank", type=int, default=0)
    args = ap.parse_args()
    if args.dummy:
        return dummy(args.dummy, args.rank)

    import torch
    import bench
    import __graft_entry__ as ge
    ge.build()
    torch.set_num_threads(1)
    mine, total = share_of_cores(0)
    fit = bench.Fit("default", 0, torch.device("cuda:0"), "fused")
    print(f"# host: {total} usable hardware threads; every rank confined "
          f"to {mine} of them (1/8 share, as bench.pin_to_gpu_numa does for 8 ranks); real rank = default skip-net 512x512, "
          f"fused closure, eager launches, K = {args.k} iterations, best of 3; dummies paced at one iteration per {1e3 * PERIOD:.0f} ms")
    rows = [("solo (no other rank)", measure(fit, args.k))]
    for kind in ("cpu", "launch"):
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dummy", kind, "--rank", str(r)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(1, 8)]
        try:
            for p in procs:
                while True:
                    line = p.stdout.readline()
                    assert line, "dummy rank died"
                    if line.strip() == "READY":
                        break
            time.sleep(1.0)
            rows.append((f"+ 7 '{kind}' dummy ranks", measure(fit, args.k)))
        finally:
            for p in procs:
                p.terminate()
            for p in procs:
                p.wait(timeout=20)
    base = rows[0][1]["it_s"]
    print(f"{'arm':28s} {'it/s':>8s} {'vs solo':>8s} {'enqueue ms/it':>14s} {'launch loop, GPU idle: min / median ms':>40s}")
    for name, r in rows:
        print(f"{name:28s} {r['it_s']:8.2f} {100 * (r['it_s'] / base - 1):+7.2f}% {r['enqueue_ms']:14.3f} "
              f"{r['loop_min_ms']:20.3f} / {r['loop_med_ms']:.3f}")
    print("# 'cpu' arm = what 7 more ranks cost the host (cores, caches); 'launch' arm adds 7 x ~250 launches / 6 ms on the SAME\n"
          "# GPU's command processor, which an 8-GPU node does not have: an upper bound, not an estimate.")


if __name__ == "__main__":
    main()
