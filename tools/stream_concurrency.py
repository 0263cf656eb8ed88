"""Do independent small-kernel chains on different HIP streams overlap on this runtime?
Chains of dip_conv_igemm launches on a 32x32 / 64x64 128->128 3x3 layer (latency-bound kernels),
(a) eager on 1 / 2 / 4 / 8 streams, (b) each chain captured into its own hipGraph and replayed on its
own stream, (c) all chains captured into ONE graph as parallel branches."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
import dip_native as N
import hipops as H
from dip_native import round_up
dev = torch.device("cuda:0")
lib = N.lib()


def make_chain(hw):
    x = torch.randn(hw, hw, 128, device=dev)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    packed, fo, _ = H.pack(w)
    y = torch.empty(hw * hw * 128, device=dev)
    ksplit, ntiles, wsf = N.conv_plan(hw, hw, 128, 128, 3, 1)
    ws = torch.empty(max(wsf, 4), device=dev)
    d = N.DipConvDesc(x.data_ptr(), hw, hw, 128, 128, N.DipTransform(None, None, 1.0), packed.data_ptr(), None, y.data_ptr(),
                      hw, hw, 128, 128, 0, 3, 1, N.PAD_REFLECT, 1, 1, 0, None, ksplit, ws.data_ptr() if ksplit > 1 else None)
    return d, (x, w, packed, y, ws)


def run_chain(d, stream, n):
    for _ in range(n):
        lib.dip_conv_igemm(C.byref(d), stream.cuda_stream)


for hw in (32, 64):
    NK = 40
    for ns in (1, 2, 4, 8):
        chains = [make_chain(hw) for _ in range(ns)]
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        for (d, _), s in zip(chains, streams):
            run_chain(d, s, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for (d, _), s in zip(chains, streams):
            run_chain(d, s, NK)
        t_launch = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_eager = time.perf_counter() - t0
        # one graph per chain
        graphs = []
        for (d, _), s in zip(chains, streams):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                run_chain(d, torch.cuda.current_stream(dev), NK)
            graphs.append(g)
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for g, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    g.replay()
            torch.cuda.synchronize()
            t_graphs = time.perf_counter() - t0
        print(f"hw={hw} chains={ns}: eager {1e6*t_eager/NK:7.1f} us per chain-step (launch side {1e6*t_launch/NK:6.1f}), "
              f"per-chain graphs {1e6*t_graphs/NK:7.1f} us per step  [1 chain alone would be x1, perfect overlap keeps it flat]", flush=True)
