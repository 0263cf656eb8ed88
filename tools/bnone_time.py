"""Timing of the one-launch BatchNorm backward (dip_bn_bwd_one) against the three-launch form on one MI355X.
Measurement tool, not part of the product path."""
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

def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for (Cc, Hh, Ww, P) in [(128, 16, 16, 1), (128, 32, 32, 1), (128, 64, 64, 1), (128, 128, 128, 1), (128, 28, 44, 2), (64, 56, 88, 2), (4, 64, 64, 0), (132, 64, 64, 1)]:
    Cs = round_up(Cc, 4)
    y = torch.randn(Hh * Ww * Cs, device=dev)
    G = torch.randn((Hh + 2 * P) * (Ww + 2 * P) * Cs, device=dev)
    state = torch.rand(4 * Cs, device=dev) + 0.5
    src = N.DipGradSrc(G.data_ptr(), P, 1 if P else 0, Cs, 0)
    dy = torch.empty(Hh * Ww * Cs, device=dev)
    dg, db, coef = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(2 * Cs, device=dev)
    nblk = lib.dip_bn_bwd_nblk(Hh, Ww, Cc)
    part = torch.empty(nblk * 2 * Cs, device=dev)
    one = lambda: lib.dip_bn_bwd_one(C.byref(src), y.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, 0.2, dy.data_ptr(), Cs, dg.data_ptr(), db.data_ptr(), coef.data_ptr(), st)
    def three():
        lib.dip_bn_bwd_stats(C.byref(src), y.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, 0.2, None, Cs, part.data_ptr(), nblk, st)
        lib.dip_bn_bwd_finalize(part.data_ptr(), nblk, Cs, Cc, Hh * Ww, dg.data_ptr(), db.data_ptr(), coef.data_ptr(), st)
        lib.dip_bn_bwd_apply_src(C.byref(src), y.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, 0.2, coef.data_ptr(), dy.data_ptr(), Cs, st)
    print(f"C={Cc:3d} {Hh}x{Ww} pad {P}: one launch {timeit(one):7.1f} us   three launches {timeit(three):7.1f} us (nblk {nblk})", flush=True)
