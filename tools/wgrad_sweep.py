"""Micro-benchmark of dip_conv_wgrad (+ dip_wgrad_reduce) on the net's big layers; variants of the
kernel are selected through DIP_WGRAD_* environment switches read at every launch (debug builds)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
import dip_native as N
import hipops as H
from dip_native import round_up
dev = torch.device("cuda:0")
lib = N.lib()
st = H.stream(dev)


def bench(Cin, Cout, ks, stride, Hh, Ww, use_tr=True, reps=10):
    P = (ks - 1) // 2
    Ho, Wo = Hh // stride, Ww // stride
    x = torch.randn(Hh, Ww, round_up(Cin, 4), device=dev)
    dy = torch.randn(Ho, Wo, round_up(Cout, 4), device=dev)
    CinP, CoutP = round_up(Cin, 32), round_up(Cout, 32)
    nsplit = N.wgrad_plan(Ho, Wo, Cin, Cout, ks, stride)
    partial = torch.empty(nsplit * ks * ks * CinP * CoutP, device=dev)
    bpart = torch.empty(nsplit * CoutP, device=dev)
    a = torch.rand(round_up(Cin, 4), device=dev) + 0.5
    b = torch.randn(round_up(Cin, 4), device=dev) * 0.3
    tr = N.DipTransform(a.data_ptr(), b.data_ptr(), 0.2) if use_tr else N.DipTransform(None, None, 1.0)
    d = N.DipWgradDesc(x.data_ptr(), Hh, Ww, round_up(Cin, 4), Cin, tr, dy.data_ptr(), Ho, Wo, round_up(Cout, 4),
                       Cout, ks, stride, N.PAD_REFLECT if P else N.PAD_ZERO, P, partial.data_ptr(), bpart.data_ptr(), nsplit)
    for _ in range(2):
        N.check(lib.dip_conv_wgrad(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.dip_conv_wgrad(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gf = 2.0 * Cout * Ho * Wo * Cin * ks * ks / 1e9
    return nsplit, us, gf / us * 1e-3


if __name__ == "__main__":
    variants = ["0"]
    for rep in range(2):
        for v in variants:
            os.environ["DIP_WGRAD_VARIANT"] = v
            out = []
            for (Cin, Cout, ks, s, Hh, Ww) in ((132, 128, 3, 1, 512, 512), (128, 128, 3, 1, 512, 512), (132, 128, 3, 1, 256, 256),
                                               (128, 128, 3, 1, 256, 256), (128, 128, 1, 1, 512, 512), (128, 128, 3, 2, 256, 256)):
                ns, us, tf = bench(Cin, Cout, ks, s, Hh, Ww)
                out.append(f"{Cin}>{Cout} k{ks}s{s} {Hh}: {us:7.1f}us {tf:5.1f}TF n={ns}")
            print(f"variant {v}: " + " | ".join(out), flush=True)
