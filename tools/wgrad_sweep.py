"""Micro-benchmark of dip_conv_wgrad + dip_wgrad_reduce on the net's layer shapes.
  python tools/wgrad_sweep.py            planned launch shape of every layer shape
  python tools/wgrad_sweep.py small      (tap_groups, nsplit) sweep of the <= 128x128 shapes (calibrates dip_wgrad_plan2)
"""
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


def bench(Cin, Cout, ks, stride, Hh, Ww, plan=None, use_tr=True, reps=10):
    P = (ks - 1) // 2
    Ho, Wo = Hh // stride, Ww // stride
    x = torch.randn(Hh, Ww, round_up(Cin, 4), device=dev)
    dy = torch.randn(Ho, Wo, round_up(Cout, 4), device=dev)
    CinP, CoutP = round_up(Cin, 32), round_up(Cout, 32)
    nsplit, tg, cb = plan or N.wgrad_plan2(Ho, Wo, Cin, Cout, ks, stride)
    partial = torch.empty(nsplit * ks * ks * CinP * CoutP, device=dev)
    bpart = torch.empty(nsplit * CoutP, device=dev)
    dw = torch.empty(Cout * Cin * ks * ks, device=dev)
    db = torch.empty(Cout, device=dev)
    a = torch.rand(round_up(Cin, 4), device=dev) + 0.5
    b = torch.randn(round_up(Cin, 4), device=dev) * 0.3
    tr = N.DipTransform(a.data_ptr(), b.data_ptr(), 0.2) if use_tr else N.DipTransform(None, None, 1.0)
    d = N.DipWgradDesc(x.data_ptr(), Hh, Ww, round_up(Cin, 4), Cin, tr, dy.data_ptr(), Ho, Wo, round_up(Cout, 4),
                       Cout, ks, stride, N.PAD_REFLECT if P else N.PAD_ZERO, P, partial.data_ptr(), bpart.data_ptr(), nsplit,
                       tg, cb)

    def go():
        lib.dip_conv_wgrad(C.byref(d), st)

    def red():
        lib.dip_wgrad_reduce(partial.data_ptr(), bpart.data_ptr(), nsplit, ks, Cin, Cout, dw.data_ptr(), db.data_ptr(), st)

    for _ in range(2):
        N.check(lib.dip_conv_wgrad(C.byref(d), st)); red()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(reps):
        go()
    e[1].record()
    for _ in range(reps):
        red()
    e[2].record(); torch.cuda.synchronize()
    us, rus = e[0].elapsed_time(e[1]) * 1e3 / reps, e[1].elapsed_time(e[2]) * 1e3 / reps
    gf = 2.0 * Cout * Ho * Wo * Cin * ks * ks / 1e9
    return (nsplit, tg, cb), us, rus, gf / us * 1e-3


SHAPES = ((132, 128, 3, 1, 512), (128, 128, 3, 1, 512), (132, 128, 3, 1, 256), (128, 128, 3, 1, 256), (128, 128, 1, 1, 512),
          (128, 128, 3, 2, 256), (32, 128, 3, 2, 512), (132, 128, 3, 1, 128), (128, 128, 3, 1, 128), (128, 128, 3, 2, 128),
          (132, 128, 3, 1, 64), (128, 128, 3, 1, 64), (128, 128, 3, 2, 64), (132, 128, 3, 1, 32), (128, 128, 3, 1, 32),
          (128, 128, 3, 2, 32), (128, 128, 3, 1, 16), (128, 128, 1, 1, 256), (128, 128, 1, 1, 128), (128, 128, 1, 1, 64),
          (128, 128, 1, 1, 32), (256, 128, 3, 1, 512))

K5 = ((16, 16, 5, 1, 224, 352), (16, 32, 5, 2, 224, 352), (32, 32, 5, 1, 112, 176), (32, 64, 5, 2, 112, 176), (64, 64, 5, 1, 56, 88),
      (64, 128, 5, 2, 56, 88), (128, 128, 5, 1, 28, 44), (128, 128, 5, 2, 28, 44), (128, 128, 5, 1, 14, 22), (128, 128, 5, 2, 14, 22),
      (128, 128, 5, 1, 7, 11))       # the 'library' inpainting net at 448 x 704 (inpainting.ipynb:222-232): Cin, Cout, ks, stride, Hin, Win

LIB3 = ((128, 64, 3, 1, 112, 176), (64, 32, 3, 1, 224, 352), (32, 16, 3, 1, 448, 704))      # the library net's 3x3 decoder convs

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lib":       # planned launch of every library-net layer (A/B: DIP_WGRAD_NO_THIN=1)
        for (Cin, Cout, ks, s, Hh, Ww) in K5 + LIB3:
            plan, us, rus, tf = bench(Cin, Cout, ks, s, Hh, Ww, reps=10)
            print(f"{Cin}>{Cout} k{ks}s{s} {Hh}x{Ww}: {us:7.1f}us +reduce {rus:5.1f}us {tf:5.1f}TF plan={plan}", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "k5":       # (tap_groups, nsplit) sweep of the 5x5 layers (dip_wgrad_plan2's 5x5 table)
        for (Cin, Cout, ks, s, Hh, Ww) in K5:
            Ho, Wo = Hh // s, Ww // s
            nt = lib.dip_conv_wgrad_ntiles(Ho, Wo)
            planned = N.wgrad_plan2(Ho, Wo, Cin, Cout, ks, s)
            kw = 4 if Cout <= 32 else (2 if Cout <= 64 else 1)
            rows = []
            for g in (5, 25):
                n = 1
                while n <= nt:
                    try:
                        _, us, rus, tf = bench(Cin, Cout, ks, s, Hh, Ww, plan=(n * kw, g, 1), reps=5)
                        rows.append((us + rus, g, n, us, rus))
                    except Exception as e:
                        print("fail", g, n, e)
                    n *= 2
                if n // 2 != nt:
                    _, us, rus, tf = bench(Cin, Cout, ks, s, Hh, Ww, plan=(nt * kw, g, 1), reps=5)
                    rows.append((us + rus, g, nt, us, rus))
            _, us0, rus0, _ = bench(Cin, Cout, ks, s, Hh, Ww, plan=planned, reps=5)
            rows.sort()
            best = " ".join(f"g{g}n{n}:{us:.0f}+{rus:.0f}" for _, g, n, us, rus in rows[:6])
            print(f"{Cin}>{Cout} k{ks}s{s} out{Ho}x{Wo} nt={nt} planned={planned} {us0:.0f}+{rus0:.0f} | {best}", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        for (Cin, Cout, ks, s, Hh) in SHAPES:
            Ho = Hh // s
            nt = lib.dip_conv_wgrad_ntiles(Ho, Ho)
            if nt > 256 or ks != 3:
                continue
            planned = N.wgrad_plan2(Ho, Ho, Cin, Cout, ks, s)
            rows = []
            for g in (1, 3, 9):
                n = 1
                while n <= nt:
                    _, us, rus, tf = bench(Cin, Cout, ks, s, Hh, Hh, plan=(n, g, 1), reps=5)
                    rows.append((us + rus, g, n, us, rus))
                    n *= 2
            rows.sort()
            best = " ".join(f"g{g}n{n}:{us:.0f}+{rus:.0f}" for _, g, n, us, rus in rows[:6])
            print(f"{Cin}>{Cout} k{ks}s{s} out{Ho} nt={nt} planned={planned} | {best}", flush=True)
        sys.exit(0)
    for (Cin, Cout, ks, s, Hh) in SHAPES:
        plan, us, rus, tf = bench(Cin, Cout, ks, s, Hh, Hh)
        print(f"{Cin}>{Cout} k{ks}s{s} {Hh}: {us:7.1f}us +reduce {rus:5.1f}us {tf:5.1f}TF plan={plan}", flush=True)
