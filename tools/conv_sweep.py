"""Micro-benchmark of the implicit-GEMM conv kernels: time vs number of 8x16 tiles, DMA vs
register-staged variant (selected by DIP_CONV_NO_DMA in the environment of this process)."""
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

def bench(Cin, Cout, ks, Hh, Ww, use_tr, reps=20):
    x = torch.randn(Hh, Ww, round_up(Cin, 4), device=dev)
    w = torch.randn(Cout, Cin, ks, ks, device=dev) / (Cin * ks * ks) ** 0.5
    packed, fo, _ = H.pack(w)
    y = torch.empty(Hh * Ww * round_up(Cout, 4), device=dev)
    a = torch.rand(round_up(Cin, 4), device=dev) + 0.5
    b = torch.randn(round_up(Cin, 4), device=dev) * 0.3
    tr = N.DipTransform(a.data_ptr(), b.data_ptr(), 0.2) if use_tr else N.DipTransform(None, None, 1.0)
    ntiles = lib.dip_conv_ntiles(Hh, Ww)
    stats = torch.empty(ntiles * 3 * round_up(Cout, 32), device=dev)
    P = (ks - 1) // 2
    d = N.DipConvDesc(x.data_ptr(), Hh, Ww, round_up(Cin, 4), round_up(Cin, 4), tr, packed.data_ptr(), None, y.data_ptr(),
                      Hh, Ww, round_up(Cout, 4), Cout, 0, ks, 1, N.PAD_REFLECT if P else N.PAD_ZERO, P, 1, 0, stats.data_ptr(), 1, None)
    for _ in range(3):
        N.check(lib.dip_conv_igemm(C.byref(d), st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.dip_conv_igemm(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gf = 2.0 * Cout * Hh * Ww * Cin * ks * ks / 1e9
    return ntiles, us, gf / us * 1e-3

def main():
  tag = "regstage" if os.environ.get("DIP_CONV_NO_DMA") else "dma"
  for (Cin, Cout, ks) in ((128, 128, 3), (132, 128, 3), (128, 128, 1)):
      for use_tr in (False, True):
          for (Hh, Ww) in ((64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512)):
              nt, us, tf = bench(Cin, Cout, ks, Hh, Ww, use_tr)
              print(f"{tag:9s} Cin={Cin:3d} k={ks} tr={int(use_tr)} {Hh}x{Ww} tiles={nt:5d} {us:8.1f} us  {tf:6.1f} TF/s  us/tile-slot={us/max(1,(nt+255)//256):7.1f}")

if __name__ == "__main__":
    main()
