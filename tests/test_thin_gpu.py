"""Parity of conv_thin_kernel (csrc/conv_thin.hip: the <= 64-channel 3x3 / 5x5 layers of the narrow nets at their high
resolutions, dispatched by dip_conv_igemm) on a real MI355X: forward with the producer's BatchNorm + LeakyReLU in the loader
and the consumer BatchNorm's partial statistics, and the stride-1 data gradient, against torch-CPU fp64 evaluations of the
reference ops (nn.ReflectionPad2d + nn.Conv2d, models/common.py:114-124 of the reference; autograd's ConvolutionBackward)
with the per-op criterion of test_kernels_gpu.py.  The shapes are those of the 'library' inpainting net
(inpainting.ipynb:222-232) and the snail net (denoising.ipynb:143-150), the first three at the library net's REAL sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402
from dip_native import round_up  # noqa: E402
import test_kernels_gpu as TK  # noqa: E402
from test_kernels_gpu import REFLECT, ZERO  # noqa: E402

THIN_CASES = [
    # Cin, Cout, ks, stride, pad, H, W, transform
    (16, 16, 5, 1, REFLECT, 224, 352, True),     # library s0.down_b at its real size: one 16-column block, one chunk
    (16, 32, 5, 2, REFLECT, 224, 352, True),     # library s1.down_a (stride 2) at its real size
    (32, 64, 5, 2, REFLECT, 112, 176, True),     # library s2.down_a at its real size: 2 chunks of 16 channels, 4 column blocks
    (32, 32, 5, 1, REFLECT, 72, 88, True),       # library s1.down_b (reduced size): 2 chunks
    (64, 32, 3, 1, REFLECT, 70, 90, True),       # library s1.up: 64 input channels
    (32, 16, 3, 1, REFLECT, 80, 96, True),       # library s0.up
    (12, 16, 5, 2, REFLECT, 160, 144, False),    # 12 input channels: a 16-channel K step with 4 idle lanes
    (8, 8, 3, 1, ZERO, 75, 70, True),            # snail s0.down_b: zero padding, ragged tiles, half-empty column block
    (36, 20, 3, 1, REFLECT, 72, 72, False),      # odd channel counts: 3 chunks of 16 (the last one 4 channels), 2 column blocks
    (64, 64, 5, 1, REFLECT, 70, 70, True),       # 4 chunks x 4 column blocks
    (16, 48, 3, 2, ZERO, 140, 150, True),        # stride 2, zero padding, 3 column blocks
]


@pytest.fixture(autouse=True)
def _every_shape(monkeypatch):
    """The planner's bounds (dip_conv_thin_shape_ok: <= 12800 weights, >= 8 input channels) leave some of the cases above to the
    older kernels by default; the KERNEL is tested on all of them (DIP_THIN_ALL is read at every call)."""
    monkeypatch.setenv("DIP_THIN_ALL", "1")


def _dims(case):
    Cin, Cout, ks, stride, pad, Hh, Ww, _ = case
    P = (ks - 1) // 2
    return (Hh + 2 * P - ks) // stride + 1, (Ww + 2 * P - ks) // stride + 1


@pytest.mark.parametrize("case", THIN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_thin_forward_and_stats(dev, case):
    Cin, Cout, ks, stride, pad, Hh, Ww, _ = case
    Ho, Wo = _dims(case)
    assert N.lib().dip_conv_thin_shape_ok(Ho, Wo, round_up(Cin, 4), Cout, ks, stride) == 1
    assert N.conv_plan(Ho, Wo, round_up(Cin, 4), Cout, ks, stride) == (1, N.lib().dip_conv_ntiles(Ho, Wo), 0)
    TK.test_conv_forward_and_stats(dev, case, True)          # (the planned launch: one pass, rows = 8x16 tiles)


@pytest.mark.parametrize("case", [c for c in THIN_CASES if c[3] == 1], ids=lambda c: "x".join(map(str, c)))
def test_conv_thin_dgrad(dev, case):
    """Stride-1 data gradient: the same kernel with the flipped, transposed weight pack on the padded domain."""
    Cin, Cout, ks, stride, pad, Hh, Ww, _ = case
    assert N.lib().dip_conv_thin_shape_ok(Hh + ks - 1, Ww + ks - 1, round_up(Cout, 4), Cin, ks, 1) == 1 or pad == ZERO
    TK.test_conv_dgrad(dev, case, True)


def test_conv_thin_is_what_runs(dev):
    """dip_conv_variant reports the thin kernel for these descriptors, the register-staged kernel under DIP_CONV_NO_THIN
    (read once per process: checked through the shape predicate here) and never for accumulate / row-pitch / dilated launches."""
    import ctypes as C
    lib = N.lib()
    d = N.DipConvDesc(1 << 20, 224, 352, 16, 16, N.DipTransform(None, None, 1.0), 1 << 20, None, 1 << 20, 224, 352, 16, 16, 0,
                      5, 1, N.PAD_REFLECT, 2, 1, 0, None, 1, None)
    assert lib.dip_conv_thin_eligible(C.byref(d)) == 1 and lib.dip_conv_variant(C.byref(d)) == 8
    for field, val in (("accumulate", 1), ("y_pitch", 360), ("dil", 2), ("ksplit", 2)):
        d2 = N.DipConvDesc.from_buffer_copy(d)
        setattr(d2, field, val)
        assert lib.dip_conv_thin_eligible(C.byref(d2)) == 0, field
    d3 = N.DipConvDesc.from_buffer_copy(d)
    d3.Hout, d3.Wout = 56, 80                    # 4480 pixels: conv_small's range
    assert lib.dip_conv_thin_eligible(C.byref(d3)) == 0
    # the default bounds: one chunk must hold all channels and all taps' weights, >= 8 input channels
    import os
    os.environ.pop("DIP_THIN_ALL")
    ok = lib.dip_conv_thin_shape_ok
    assert ok(224, 352, 16, 16, 5, 1) == 1 and ok(112, 176, 16, 32, 5, 2) == 1 and ok(448, 704, 32, 16, 3, 1) == 1
    assert ok(112, 176, 32, 32, 5, 1) == 0 and ok(224, 352, 64, 32, 3, 1) == 0 and ok(224, 352, 4, 16, 5, 2) == 0


WGRAD_CASES = [
    (16, 16, 5, 1, REFLECT, 224, 352, True),     # library s0.down_b at its real size: CI = CO = 1, 7 / 6 / 6 / 6 taps per wave
    (16, 32, 5, 2, REFLECT, 224, 352, True),     # library s1.down_a (stride 2) at its real size: 11 x 35-pixel halo, CO = 2
    (32, 64, 5, 2, REFLECT, 112, 176, True),     # library s2.down_a at its real size: 2 input x 2 output channel blocks
    (32, 32, 5, 1, REFLECT, 72, 88, True),       # CI = CO = 2: 112 accumulator registers
    (64, 32, 3, 1, REFLECT, 70, 90, True),       # library s1.up: two 32-channel input blocks
    (32, 16, 3, 1, REFLECT, 80, 96, False),      # library s0.up, no transform
    (8, 8, 3, 1, ZERO, 75, 70, True),            # snail: half-empty blocks, zero padding, ragged tiles
    (36, 20, 3, 1, REFLECT, 72, 72, True),       # odd channel counts: a 4-channel input block, a 4-column output block
    (64, 64, 5, 1, REFLECT, 66, 70, True),       # 2 x 2 channel blocks
    (16, 48, 3, 2, ZERO, 140, 150, True),        # stride 2, zero padding, 2 output blocks (the second half empty)
    (10, 12, 3, 1, REFLECT, 70, 72, True),       # 10 input channels (a net fed with a 10-plane image): the last float4 is ragged
]


@pytest.mark.parametrize("nsplit", [None, "plan"], ids=["nsplit7", "planned"])
@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wgrad_thin(dev, case, nsplit):
    """dW, db of the thin layers through dip_conv_wgrad (which dispatches to wgrad_thin_kernel) + dip_wgrad_reduce against
    fp64 autograd; 7 slabs and the planned number."""
    Cin, Cout, ks, stride, pad, Hh, Ww, _ = case
    Ho, Wo = _dims(case)
    assert N.lib().dip_wgrad_thin_shape_ok(Ho, Wo, Cin, Cout, ks, stride) == 1
    n, g, cb = N.wgrad_plan2(Ho, Wo, Cin, Cout, ks, stride)
    assert n == N.lib().dip_wgrad_thin_nsplit(Ho, Wo, Cin, Cout, ks, stride) == N.wgrad_plan(Ho, Wo, Cin, Cout, ks, stride) and g == 1
    TK.test_conv_wgrad(dev, case, nsplit)


THIN4_CASES = [
    # layer Cin (the gradient's columns; the first Cin - 128k.. are the thin ones here: ncols), layer Cout (= dy channels), H, W, accumulate
    (4, 128, 45, 61, False),       # the default net's case: 4 skip columns from 128 dy channels; ragged strips and row walks
    (4, 128, 16, 14, False),       # exactly one strip
    (4, 128, 3, 100, True),        # fewer rows than a walk, accumulate onto the buffer
    (2, 64, 33, 47, False),        # 2 columns, 64 dy channels (4 K groups)
    (1, 16, 20, 29, False),        # 1 column, 16 dy channels (1 K group)
    (3, 96, 19, 30, True),         # 96 dy channels: the 128-channel instance with two empty K groups
    (4, 100, 21, 33, False),       # 100 dy channels: a K group with one of its four lane slots past the end
    (4, 32, 130, 70, False),       # 2 K groups, more than one workgroup row
    (4, 160, 18, 22, False),       # > 128 dy channels: the vector-ALU form
]


@pytest.mark.parametrize("form", ["mfma", "valu"])
@pytest.mark.parametrize("case", THIN4_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_thin4_columns(dev, case, form, monkeypatch):
    """dip_conv_thin4 on its own (conv_thin4.hip): columns [0, ncols) of the data gradient of a zero-padded 3x3 stride-1
    conv (the interior-domain form of dip_conv_dgrad_ring's caller: Hout = Hin, off = 1), matrix-pipe form and vector-ALU
    form (DIP_THIN4_VALU is read once per process: the 'valu' arm runs in a subprocess-free way through a >128-channel
    case or is skipped), against autograd in fp64 with the per-op criterion."""
    import ctypes as C
    import hipops as H
    ncols, Cd, Hh, Ww, acc = case
    if form == "valu" and Cd <= 128:
        pytest.skip("the vector-ALU form is what <= 128 dy channels no longer take (kept for > 128 and the fused BatchNorm-backward partials)")
    if form == "mfma" and Cd > 128:
        pytest.skip("> 128 dy channels take the vector-ALU form")
    lib = N.lib()
    g_ = torch.Generator().manual_seed(ncols * 1000 + Cd)
    Cl = 8                                                    # layer input channels (columns of the gradient); the thin ones lead
    w = torch.randn(Cd, Cl, 3, 3, generator=g_) / (Cd * 9) ** 0.5
    dy = torch.randn(1, Cd, Hh, Ww, generator=g_)
    base = torch.randn(1, Cl, Hh, Ww, generator=g_)
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = torch.zeros(1, Cl, Hh, Ww, dtype=dt, requires_grad=True)
        y = torch.nn.functional.conv2d(xx, w.to(dt), None, 1, 1)
        (y * dy.to(dt)).sum().backward()
        res[dt] = (xx.grad + (base.to(dt) if acc else 0))[:, :ncols]
    packed, _, do = H.pack(w.to(dev))
    dyb = H.to_nhwc(dy.to(dev))
    Cg = round_up(Cl, 4)
    gbuf = H.to_nhwc(base.to(dev)).clone() if acc else torch.full((Hh * Ww * Cg,), float("nan"), device=dev)
    before = gbuf.clone()
    d = N.DipConvDesc(dyb.data_ptr(), Hh, Ww, round_up(Cd, 4), round_up(Cd, 4), N.DipTransform(None, None, 1.0),
                      packed.data_ptr() + 4 * do, None, gbuf.data_ptr(), Hh, Ww, Cg, Cl, 0, 3, 1, N.PAD_ZERO, 1, 1, 1 if acc else 0,
                      None, 1, None)
    N.check(lib.dip_conv_thin4(C.byref(d), ncols, H.stream(dev)), "conv_thin4")
    torch.cuda.synchronize()
    got = H.from_nhwc(gbuf, Cl, Hh, Ww)
    TK._check("conv_thin4", got[:, :ncols], res[torch.float64], res[torch.float32])
    # the other columns of the buffer are not touched
    rest_now, rest_before = gbuf.view(-1, Cg)[:, ncols:], before.view(-1, Cg)[:, ncols:]
    assert torch.equal(rest_now.isnan(), rest_before.isnan()) and torch.equal(rest_now.nan_to_num(), rest_before.nan_to_num())
