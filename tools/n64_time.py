import ctypes as C, os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as G
G.add_to_path()
import torch
import dip_native as N
from dip_native import round_up
import hipops as H
lib = N.lib(); dev = torch.device("cuda:0"); st = H.stream(dev)
for (Cin, Cout, Hh, Ww) in ((128, 128, 128, 128), (128, 128, 128, 240), (128, 128, 64, 128), (128,128,128,256)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, Cin, Hh, Ww, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev)
    a = (torch.rand(Cin, generator=g) + 0.5).to(dev); b = (torch.randn(Cin, generator=g) * 0.3).to(dev)
    xb = H.to_nhwc(x); packed, fo, do = H.pack(w); p3, fo3, do3 = H.pack_bf3(w)
    trd, keep = H.transform(a, b, 0.2)
    Cy, CoutP = round_up(Cout, 4), round_up(Cout, 32)
    y = torch.empty(Hh * Ww * Cy, device=dev); nt = lib.dip_conv_ntiles(Hh, Ww)
    stats = torch.empty(nt * 3 * CoutP, device=dev)
    d = N.DipConvDesc(xb.data_ptr(), Hh, Ww, round_up(Cin, 4), round_up(Cin, 4), trd, packed.data_ptr() + 4 * fo, None, y.data_ptr(),
                      Hh, Ww, Cy, Cout, 0, 3, 1, N.PAD_REFLECT, 1, 1, 0, stats.data_ptr(), 1, None)
    d.wp3 = p3.data_ptr() + 2 * fo3
    gf = 2.0 * Cin * Cout * 9 * Hh * Ww / 1e9
    v = lib.dip_conv_variant(C.byref(d))
    for _ in range(3): N.check(lib.dip_conv_igemm(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.dip_conv_igemm(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{Hh}x{Ww}: tiles {nt} variant {v} {us:.1f} us {gf/us:.1f} GF/us=TF*1e-3 -> {gf/us*1e3/1e3:.1f} TF", flush=True)
