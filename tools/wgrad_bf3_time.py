"""Times dip_conv_wgrad (+ dip_wgrad_reduce excluded) of the big 3x3 layers on the fp32 MFMA and on the bf16 matrix pipe."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as G
G.add_to_path()
import torch
import dip_native as N
from dip_native import round_up
import hipops as H

lib = N.lib()
dev = torch.device("cuda:0")
st = H.stream(dev)
for (Cin, Cout, Hh, Ww) in ((132, 128, 512, 512), (128, 128, 512, 512), (132, 128, 256, 256), (128, 128, 256, 256)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, Cin, Hh, Ww, generator=g).to(dev)
    dy = torch.randn(1, Cout, Hh, Ww, generator=g).to(dev)
    a = (torch.rand(Cin, generator=g) + 0.5).to(dev); b = (torch.randn(Cin, generator=g) * 0.3).to(dev)
    xb, dyb = H.to_nhwc(x), H.to_nhwc(dy)
    trd, keep = H.transform(a, b, 0.2)
    CinP, CoutP = round_up(Cin, 32), round_up(Cout, 32)
    n, tg, cb = N.wgrad_plan2(Hh, Ww, Cin, Cout, 3, 1)
    partial = torch.zeros(n * 9 * CinP * CoutP, device=dev)
    bpart = torch.zeros(n * CoutP, device=dev)
    d = N.DipWgradDesc(xb.data_ptr(), Hh, Ww, round_up(Cin, 4), Cin, trd, dyb.data_ptr(), Hh, Ww, round_up(Cout, 4), Cout, 3, 1,
                       N.PAD_REFLECT, 1, partial.data_ptr(), bpart.data_ptr(), n, tg, cb)
    gf = 2.0 * Cin * Cout * 9 * Hh * Ww / 1e9
    out = []
    for terms in (0, 9, 6):
        lib.dip_conv_bf3_set_terms(terms)
        el = lib.dip_wgrad_bf3_eligible(C.byref(d))
        for _ in range(3):
            N.check(lib.dip_conv_wgrad(C.byref(d), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            N.check(lib.dip_conv_wgrad(C.byref(d), st))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        out.append(f"terms {terms} (bf3 {el}): {us:7.1f} us = {gf / us * 1e-3:6.1f} TF")
    lib.dip_conv_bf3_set_terms(-1)
    print(f"{Cin}->{Cout} @ {Hh}x{Ww} ({gf:.1f} GFLOP, nsplit {n}): " + " | ".join(out), flush=True)
