"""Per-layer A/B on one MI355X: the tiled implicit-GEMM launch (dip_conv_igemm with dip_conv_plan's split-K + finish) against
dip_conv_small for the conv shapes of the 'library' inpainting net (inpainting.ipynb:222-232 of the reference: depth 6,
channels 16/32/64/128/128/128, 5x5 down filters, 3x3 up filters, 448 x 704) -- forward descriptors (producer BatchNorm +
LeakyReLU in the loader, BatchNorm partials out) and data-gradient descriptors as dip_engine builds them.

    python tools/thin_sweep.py [fwd|dgrad|all]

Measurement tool (tools/), not part of the product path."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
import dip_native as N  # noqa: E402
import hipops as H  # noqa: E402
from dip_native import round_up  # noqa: E402

dev = torch.device("cuda:0")
lib = N.lib()
st = H.stream(dev)

# (name, Cin, Cout, ks, stride, Hin, Win): the library net at 448 x 704
LAYERS = [
    ("s0.down_a", 1, 16, 5, 2, 448, 704), ("s0.down_b", 16, 16, 5, 1, 224, 352),
    ("s1.down_a", 16, 32, 5, 2, 224, 352), ("s1.down_b", 32, 32, 5, 1, 112, 176),
    ("s2.down_a", 32, 64, 5, 2, 112, 176), ("s2.down_b", 64, 64, 5, 1, 56, 88),
    ("s3.down_a", 64, 128, 5, 2, 56, 88), ("s3.down_b", 128, 128, 5, 1, 28, 44),
    ("s4.down_a", 128, 128, 5, 2, 28, 44), ("s4.down_b", 128, 128, 5, 1, 14, 22),
    ("s5.down_a", 128, 128, 5, 2, 14, 22), ("s5.down_b", 128, 128, 5, 1, 7, 11),
    ("s5.up", 128, 128, 3, 1, 14, 22), ("s4.up", 128, 128, 3, 1, 28, 44), ("s3.up", 128, 128, 3, 1, 56, 88),
    ("s2.up", 128, 64, 3, 1, 112, 176), ("s1.up", 64, 32, 3, 1, 224, 352), ("s0.up", 32, 16, 3, 1, 448, 704),
]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def fwd_desc(Cin, Cout, ks, stride, Hin, Win, keep):
    P = (ks - 1) // 2
    Ho, Wo = (Hin + 2 * P - ks) // stride + 1, (Win + 2 * P - ks) // stride + 1
    Cx = round_up(Cin, 4)
    x = torch.randn(Hin * Win * Cx, device=dev)
    w = torch.randn(Cout, Cin, ks, ks, device=dev) / (Cin * ks * ks) ** 0.5
    packed, fo, _ = H.pack(w)
    Cy = round_up(Cout, 4)
    y = torch.empty(Ho * Wo * Cy, device=dev)
    a = torch.rand(Cx, device=dev) + 0.5
    b = torch.randn(Cx, device=dev) * 0.3
    bias = torch.randn(Cout, device=dev)
    tr = N.DipTransform(a.data_ptr(), b.data_ptr(), 0.2)
    d = N.DipConvDesc(x.data_ptr(), Hin, Win, Cx, Cx, tr, packed.data_ptr() + 4 * fo, bias.data_ptr(), y.data_ptr(), Ho, Wo, Cy,
                      Cout, 0, ks, stride, N.PAD_REFLECT, P, 1, 0, None, 1, None)
    keep += [x, w, packed, y, a, b, bias]
    return d, Ho, Wo


def dgrad_desc(Cin, Cout, ks, stride, Hin, Win, keep):
    """The engine's descriptor of the data gradient wrt the (reflection-padded) input: dy [Ho][Wo][Cout] -> g on the padded
    domain [(Hin + 2P)][(Win + 2P)][Cin]."""
    P = (ks - 1) // 2
    Ho, Wo = (Hin + 2 * P - ks) // stride + 1, (Win + 2 * P - ks) // stride + 1
    Hg, Wg = Hin + 2 * P, Win + 2 * P
    Cg = round_up(Cin, 4)
    dy = torch.randn(Ho * Wo * round_up(Cout, 4), device=dev)
    w = torch.randn(Cout, Cin, ks, ks, device=dev) / (Cin * ks * ks) ** 0.5
    packed, _, do = H.pack(w)
    g = torch.empty(Hg * Wg * Cg, device=dev)
    d = N.DipConvDesc(dy.data_ptr(), Ho, Wo, round_up(Cout, 4), round_up(Cout, 4), N.DipTransform(None, None, 1.0),
                      packed.data_ptr() + 4 * do, None, g.data_ptr(), Hg, Wg, Cg, Cin, 0, ks, 1, N.PAD_ZERO, ks - 1, stride, 0,
                      None, 1, None)
    keep += [dy, w, packed, g]
    return d, Hg, Wg


def run(kind):
    for name, Cin, Cout, ks, stride, Hin, Win in LAYERS:
        keep = []
        if kind == "fwd":
            d, Ho, Wo = fwd_desc(Cin, Cout, ks, stride, Hin, Win, keep)
            ksplit, rows, wsf = N.conv_plan(Ho, Wo, round_up(Cin, 4), Cout, ks, stride)
            ncol = Cout
        else:
            if Cin < 4:
                continue
            d, Ho, Wo = dgrad_desc(Cin, Cout, ks, stride, Hin, Win, keep)
            if stride == 2:
                ksplit, rows, wsf = N.conv_plan_dil2(Ho, Wo, round_up(Cout, 4), Cin, ks)
            else:
                ksplit, rows, wsf = N.conv_plan(Ho, Wo, round_up(Cout, 4), Cin, ks, 1)
            ncol = Cin
        gf = 2.0 * Cout * Cin * ks * ks * ((Hin + 2 * ((ks - 1) // 2) - ks) // stride + 1) * ((Win + 2 * ((ks - 1) // 2) - ks) // stride + 1) / 1e9
        res = {}
        # tiled kernel with the planner's split-K
        stats = torch.empty(max(rows, 1) * 3 * round_up(ncol, 32), device=dev)
        ws = torch.empty(max(wsf, 4), device=dev)
        d.stats = stats.data_ptr() if kind == "fwd" else None
        d.ksplit, d.ws = ksplit, (ws.data_ptr() if ksplit > 1 else None)
        res["igemm"] = timeit(lambda: N.check(lib.dip_conv_igemm(C.byref(d), st), "igemm"))
        # conv_small
        rows_s = lib.dip_conv_small_rows(C.byref(d))
        if rows_s > 0:
            stats_s = torch.empty(rows_s * 3 * round_up(ncol, 32), device=dev)
            d.stats = stats_s.data_ptr() if kind == "fwd" else None
            d.ksplit, d.ws = 1, None
            res["small"] = timeit(lambda: N.check(lib.dip_conv_small(C.byref(d), st), "small"))
        print(f"{kind:5s} {name:10s} {Cin:3d}->{Cout:3d} k{ks} s{stride} out {Ho}x{Wo} {gf:6.3f} GF  ksplit={ksplit:2d} "
              + "  ".join(f"{k}={v:7.1f}us ({gf / v * 1e3:5.1f} TF)" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for kind in (("fwd", "dgrad") if which == "all" else (which,)):
        run(kind)
