"""Per-kernel parity on a real MI355X: every libdip_hip.so kernel against a torch-CPU fp64
reference of the same op, at the layer shapes of the reference nets plus ragged / odd cases.
Tolerance: error vs fp64 no worse than 3x the error of torch's own fp32 CPU kernel (plus an
fp32-roundoff floor) -- i.e. fp32-class numerics, no reduced precision anywhere."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402
from dip_native import round_up  # noqa: E402
import hipops as H  # noqa: E402

REFLECT, ZERO, REPLICATE = N.PAD_REFLECT, N.PAD_ZERO, N.PAD_REPLICATE


def _ref_conv(x, w, b, stride, pad_mode, dtype):
    x, w = x.to(dtype), w.to(dtype)
    b = b.to(dtype) if b is not None else None
    P = (w.shape[-1] - 1) // 2
    if pad_mode in (REFLECT, REPLICATE) and P:
        return F.conv2d(F.pad(x, (P,) * 4, mode="reflect" if pad_mode == REFLECT else "replicate"), w, b, stride=stride)
    return F.conv2d(x, w, b, stride=stride, padding=P)


# per-op criterion: error vs the fp64 evaluation <= PER_OP_RATIO x the error of torch's own fp32 CPU kernel (+ a relative
# floor): SURVEY 8c (1) asks for 2x; rounds 1-2 ran with 3x, all 216 cases pass with 2x (DIP_TEST_RATIO overrides it)
PER_OP_RATIO = float(os.environ.get("DIP_TEST_RATIO", "2"))


def _check(name, got, ref64, ref32, floor=2e-6):
    got = got.detach().cpu().double()
    e = (got - ref64).abs().max().item()
    e32 = (ref32.double() - ref64).abs().max().item()
    scale = ref64.abs().max().item() + 1e-30
    tol = max(PER_OP_RATIO * e32, floor * scale)
    assert np.isfinite(e), f"{name}: non-finite output"
    assert e <= tol, f"{name}: max|err| {e:.3e} > tol {tol:.3e} (torch-fp32 err {e32:.3e}, scale {scale:.3e})"


CONV_CASES = [
    # Cin, Cout, ks, stride, pad, H, W, transform
    (32, 128, 3, 1, REFLECT, 16, 32, False),
    (132, 128, 3, 1, REFLECT, 24, 40, True),     # decoder conv: 4x32 + merged 4-channel tail
    (128, 128, 3, 2, REFLECT, 32, 32, True),     # encoder stride-2
    (128, 128, 3, 1, REFLECT, 16, 16, True),
    (128, 128, 1, 1, REFLECT, 24, 24, True),     # 1x1 decoder conv
    (32, 4, 1, 1, REFLECT, 19, 37, False),       # skip conv, ragged size
    (128, 3, 1, 1, REFLECT, 16, 48, True),       # output conv
    (3, 8, 3, 2, REFLECT, 32, 48, False),        # snail net first conv (Cin 3 -> stride 4)
    (16, 32, 5, 2, REFLECT, 32, 32, True),       # library net 5x5 stride 2
    (32, 64, 5, 1, REFLECT, 16, 24, True),
    (1, 16, 5, 2, REFLECT, 32, 32, False),
    (8, 16, 3, 1, ZERO, 16, 16, True),
    (8, 16, 3, 2, ZERO, 16, 32, False),
    (256, 128, 3, 1, REFLECT, 16, 16, True),     # kate net decoder conv
    (48, 128, 3, 1, REFLECT, 8, 16, True),       # 32 + 16 chunks
    (72, 64, 3, 1, REFLECT, 8, 16, True),        # 32 + merged 40
    (64, 160, 3, 1, REFLECT, 8, 16, False),      # N = 128 + 32 split launch
    (36, 64, 3, 2, ZERO, 16, 32, True),          # 32 + 4-channel tail, stride 2 (packed weight-gradient chunk)
    (2, 16, 7, 1, ZERO, 24, 32, False),          # feature_inversion.ipynb: 7x7 filters, zero pad, meshgrid input
    (20, 16, 7, 1, ZERO, 16, 16, True),          # 7x7 decoder conv on [4 skip | 16] channels
    (16, 32, 7, 2, REFLECT, 32, 32, True),       # 7x7 stride 2
    # <= 4 input channels: the thin weight-gradient kernel of a net's first conv (thin_cin_wgrad_kernel)
    (4, 32, 3, 1, REFLECT, 19, 27, True),
    (1, 128, 5, 2, ZERO, 40, 36, False),         # inpainting 'library': 1 input plane, 5x5 stride 2
    (3, 200, 3, 1, ZERO, 16, 16, True),          # 50 output groups: 5 pixel rows per block step (not a power of two)
    # >= 65536 pixels, 128 input channels, 1x1: the weights-resident persistent kernel (conv1x1_res.hip)
    (128, 128, 1, 1, REFLECT, 256, 256, True),
    (128, 100, 1, 1, REFLECT, 128, 512, False),  # 100 output columns (CoutP = 128, 28 idle), no transform
    # conv(..., downsample_mode='lanczos2' | 'lanczos3'): the Downsampler's dense k x k stride-2 conv behind
    # nn.ReplicationPad2d((k - 2) / 2) (models/downsampler.py:66-101 of the reference), k = 8 / 12
    (16, 16, 8, 2, REPLICATE, 32, 32, False),
    (12, 12, 12, 2, REPLICATE, 24, 40, False),
    (128, 128, 8, 2, REPLICATE, 32, 48, False),
    (64, 64, 12, 2, REPLICATE, 24, 24, False),   # K = 9216: sliced by dip_conv_plan's accuracy rule
]


def _mk(case, seed=0):
    Cin, Cout, ks, stride, pad, Hh, Ww, use_tr = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(1, Cin, Hh, Ww, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    a = bb = None
    if use_tr:
        a = torch.rand(Cin, generator=g) + 0.5
        bb = torch.randn(Cin, generator=g) * 0.3
    return x, w, b, a, bb


def _apply_tr(x, a, b, slope, dtype):
    if a is None:
        return x.to(dtype)
    t = x.to(dtype) * a.to(dtype).view(1, -1, 1, 1) + b.to(dtype).view(1, -1, 1, 1)
    return torch.maximum(t, slope * t)


@pytest.mark.parametrize("split", [False, True], ids=["onepass", "splitk"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_forward_and_stats(dev, case, split):
    Cin, Cout, ks, stride, pad, Hh, Ww, use_tr = case
    if ks * ks * Cin > 4608 and not split:
        # dip_conv_plan's accuracy rule: reductions longer than 4608 products never run in one pass (at any image size)
        assert N.conv_plan(512, 512, Cin, Cout, ks, stride)[0] >= 4
        pytest.skip("one-pass evaluation of a K > 4608 reduction is never planned")
    x, w, b, a, bb = _mk(case)
    slope = 0.2
    ref64 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float64), w, b, stride, pad, torch.float64)
    ref32 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float32), w, b, stride, pad, torch.float32)
    tr = (a.to(dev), bb.to(dev), slope) if use_tr else (None, None, 1.0)
    y, stats = H.conv_fwd(x.to(dev), w.to(dev), b.to(dev), stride, pad, tr, want_stats=True, split=split)
    _check("conv_fwd", y, ref64, ref32)
    # BatchNorm partials -> mean / biased variance per channel
    st = stats.cpu().double().numpy()
    n = st[:, 0, :Cout]; m = st[:, 1, :Cout]; M2 = st[:, 2, :Cout]
    N_ = n.sum(0)
    mean = (n * m).sum(0) / N_
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / N_
    r = ref64[0].reshape(Cout, -1)
    assert np.allclose(N_, r.shape[1])
    assert np.allclose(mean, r.mean(1).numpy(), rtol=1e-5, atol=1e-5 * float(r.std()))
    assert np.allclose(var, r.var(1, unbiased=False).numpy(), rtol=2e-5)


@pytest.mark.parametrize("split", [False, True], ids=["onepass", "splitk"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad(dev, case, split):
    Cin, Cout, ks, stride, pad, Hh, Ww, _ = case
    x, w, b, _, _ = _mk(case, 1)
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = x.to(dt).requires_grad_(True)
        y = _ref_conv(xx, w, None, stride, pad, dt)
        g = torch.Generator().manual_seed(7)
        dy = torch.randn(y.shape, generator=g)
        (y * dy.to(dt)).sum().backward()
        res[dt] = xx.grad
    gx = H.conv_dgrad(dy.to(dev), w.to(dev), stride, pad, Hh, Ww, split=split)
    _check("conv_dgrad", gx, res[torch.float64], res[torch.float32])


PHASE_CASES = [
    # Cin (= columns of the gradient, whole 128-blocks), Cout, pad, H, W
    (128, 128, REFLECT, 32, 32),
    (128, 128, ZERO, 16, 32),
    (128, 256, REFLECT, 19, 27),      # odd sizes: ragged phase sub-grids
    (128, 128, REFLECT, 18, 22),      # even size: the last padded row / column gets no contribution
    (256, 36, REFLECT, 18, 22),       # two column blocks; 32 + 4-channel ragged K chunk
    (128, 64, ZERO, 40, 72),
]


@pytest.mark.parametrize("split", [False, True, 2, 4], ids=["onepass", "planned", "k2", "k4"])
@pytest.mark.parametrize("case", PHASE_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad_stride2_phase_mode(dev, case, split):
    """Data gradient of the stride-2 3x3 convolutions: dip_conv_variant == 4, four dense sub-filter
    convolutions (one per output parity) on the LDS-DMA kernel instead of a 3x3 over the dilated dy."""
    Cin, Cout, pad, Hh, Ww = case
    full = (Cin, Cout, 3, 2, pad, Hh, Ww, False)
    x, w, b, _, _ = _mk(full, 1)
    if isinstance(split, int) and not isinstance(split, bool) and split > (round_up(Cout, 4) + 31) // 32:
        pytest.skip("split-K factor above the channel chunks of the 1-tap phase")
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = x.to(dt).requires_grad_(True)
        y = _ref_conv(xx, w, None, 2, pad, dt)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
        (y * dy.to(dt)).sum().backward()
        res[dt] = xx.grad
    # the descriptor H.conv_dgrad builds must be one the phase mode takes
    Ho, Wo = dy.shape[2:]
    P = 1 if pad == REFLECT else 0
    d = N.DipConvDesc(None, Ho, Wo, round_up(Cout, 4), round_up(Cout, 4), N.DipTransform(None, None, 1.0), None, None,
                      None, Hh + 2 * P, Ww + 2 * P, Cin, Cin, 0, 3, 1, N.PAD_ZERO, 2 if pad == REFLECT else 1, 2, 0, None,
                      1, None)
    assert N.lib().dip_conv_variant(C.byref(d)) == 4
    gx = H.conv_dgrad(dy.to(dev), w.to(dev), 2, pad, Hh, Ww, split=split)
    _check("conv_dgrad(phase)", gx, res[torch.float64], res[torch.float32])


def test_conv_dgrad_phase_mode_equals_dilated_evaluation(dev, tmp_path):
    """The phase mode keeps the non-zero terms of the dilated evaluation in their order, so one pass of
    it is bit-identical to the old kernel (run in a subprocess with DIP_CONV_NO_PHASE=1)."""
    import subprocess
    import sys
    case = (128, 128, 3, 2, REFLECT, 32, 48, False)
    x, w, b, _, _ = _mk(case, 1)
    y = _ref_conv(x, w, None, 2, REFLECT, torch.float32)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
    gx = H.conv_dgrad(dy.to(dev), w.to(dev), 2, REFLECT, 32, 48, split=False).cpu()
    torch.save({"dy": dy, "w": w}, tmp_path / "in.pt")
    code = ("import sys, torch; sys.path[:0] = [%r, %r]; import hipops as H, dip_native as N\n"
            "d = torch.load(%r); dev = torch.device('cuda:0')\n"
            "g = H.conv_dgrad(d['dy'].to(dev), d['w'].to(dev), 2, N.PAD_REFLECT, 32, 48, split=False)\n"
            "torch.save(g.cpu(), %r)\n") % (os.path.dirname(__file__), os.path.dirname(N.__file__),
                                             str(tmp_path / "in.pt"), str(tmp_path / "out.pt"))
    env = dict(os.environ, DIP_CONV_NO_PHASE="1")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
    old = torch.load(tmp_path / "out.pt")
    assert torch.equal(gx, old)


@pytest.mark.parametrize("nsplit", [None, "plan"], ids=["nsplit7", "planned"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad(dev, case, nsplit):
    Cin, Cout, ks, stride, pad, Hh, Ww, use_tr = case
    x, w, b, a, bb = _mk(case, 2)
    slope = 0.2
    res = {}
    for dt in (torch.float64, torch.float32):
        ww = w.to(dt).requires_grad_(True)
        bias = b.to(dt).requires_grad_(True)
        y = _ref_conv(_apply_tr(x, a, bb, slope, dt), ww, bias, stride, pad, dt)
        g = torch.Generator().manual_seed(9)
        dy = torch.randn(y.shape, generator=g)
        (y * dy.to(dt)).sum().backward()
        res[dt] = (ww.grad, bias.grad)
    tr = (a.to(dev), bb.to(dev), slope) if use_tr else (None, None, 1.0)
    dw, db = H.conv_wgrad(x.to(dev), dy.to(dev), ks, stride, pad, tr, nsplit=nsplit)
    _check("conv_wgrad.dw", dw, res[torch.float64][0], res[torch.float32][0], floor=4e-6)
    _check("conv_wgrad.db", db, res[torch.float64][1], res[torch.float32][1], floor=4e-6)


@pytest.mark.parametrize("cfg", [(128, 128, 3, 1, 16, 16, 3, 0, 4), (128, 128, 3, 1, 16, 16, 9, 0, 3),
                                 (132, 128, 3, 1, 16, 32, 9, 0, 8), (132, 128, 3, 1, 32, 32, 3, 0, 5),
                                 (128, 128, 3, 2, 32, 32, 9, 0, 4), (32, 128, 3, 2, 32, 64, 3, 0, 2),
                                 (128, 128, 1, 1, 32, 32, 0, 1, 16), (128, 64, 1, 1, 16, 48, 0, 1, 3),
                                 # ragged sizes / channel tails through the sliding-window loop and its packed tail phase
                                 (132, 128, 3, 1, 24, 40, 0, 0, 5), (260, 96, 3, 1, 12, 20, 0, 0, 2),
                                 (192, 256, 3, 1, 8, 16, 0, 0, 2), (129, 128, 3, 1, 19, 27, 0, 0, 7),
                                 (132, 128, 3, 1, 19, 27, 3, 0, 7),
                                 # round 6: 5x5 layers with one TAP per workgroup (25 groups) / one filter row with one tile per walker
                                 (128, 128, 5, 1, 7, 11, 25, 0, 2), (128, 128, 5, 2, 14, 22, 25, 0, 2),
                                 (64, 128, 5, 2, 28, 44, 5, 0, 4), (16, 32, 5, 2, 24, 40, 25, 0, 12)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_wgrad_tap_groups_and_channel_blocks(dev, cfg):
    """The low-resolution launch shapes of dip_wgrad_plan2: the 9 taps of a 3x3 weight gradient spread
    over 3 / 9 workgroups, 32-channel blocks for the 1x1 MFMA kernel."""
    Cin, Cout, ks, stride, Hh, Ww, tg, cb, nsplit = cfg
    case = (Cin, Cout, ks, stride, REFLECT, Hh, Ww, True)
    x, w, b, a, bb = _mk(case, 2)
    slope = 0.2
    res = {}
    for dt in (torch.float64, torch.float32):
        ww = w.to(dt).requires_grad_(True)
        bias = b.to(dt).requires_grad_(True)
        y = _ref_conv(_apply_tr(x, a, bb, slope, dt), ww, bias, stride, REFLECT, dt)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9))
        (y * dy.to(dt)).sum().backward()
        res[dt] = (ww.grad, bias.grad)
    dw, db = H.conv_wgrad(x.to(dev), dy.to(dev), ks, stride, REFLECT, (a.to(dev), bb.to(dev), slope), nsplit=nsplit,
                          tap_groups=tg, chan_block=cb)
    _check("conv_wgrad.dw", dw, res[torch.float64][0], res[torch.float32][0], floor=4e-6)
    _check("conv_wgrad.db", db, res[torch.float64][1], res[torch.float32][1], floor=4e-6)
    n, g, c = N.wgrad_plan2(Hh // stride, Ww // stride, Cin, Cout, ks, stride)
    assert n >= 1 and g in (1, 3, 9, 5, 25) and c in (1, 4)


def test_mfma_layout_asymmetric(dev):
    """A = one-hot pixel/channel probes with an asymmetric weight: catches a transposed or
    permuted MFMA fragment mapping that symmetric data would hide (1x1 conv == plain GEMM)."""
    Cin, Cout, Hh, Ww = 32, 128, 8, 16
    x = torch.zeros(1, Cin, Hh, Ww)
    for p in range(Hh * Ww):
        x[0, p % Cin, p // Ww, p % Ww] = 1.0 + p
    w = (torch.arange(Cout * Cin, dtype=torch.float32).view(Cout, Cin, 1, 1) % 97) / 97.0
    y = H.conv_fwd(x.to(dev), w.to(dev), None, 1, REFLECT)
    ref = F.conv2d(x.double(), w.double())
    assert torch.allclose(y.cpu().double(), ref, rtol=1e-6, atol=1e-6)


# ----------------------------------------------------------------------------- BatchNorm
@pytest.mark.parametrize("Cc,Hh,Ww,P,slope", [(128, 16, 32, 1, 0.2), (4, 19, 23, 0, 0.2), (132, 12, 20, 1, 1.0),
                                               (16, 8, 8, 2, 0.2)])
def test_bn_forward_backward_chain(dev, Cc, Hh, Ww, P, slope):
    """conv-output y -> BN(train)+LeakyReLU -> reflection pad -> random linear functional.
    Checks dip_bn_finalize (via conv-style partials built by a 1x1 identity conv), and the three
    backward phases incl. the reflection fold, against autograd in fp64."""
    lib = N.lib()
    g = torch.Generator().manual_seed(3)
    y = torch.randn(1, Cc, Hh, Ww, generator=g) * 2.0 + torch.randn(1, Cc, 1, 1, generator=g) * 5.0
    gamma = torch.rand(Cc, generator=g) + 0.5
    beta = torch.randn(Cc, generator=g)
    G = torch.randn(1, Cc, Hh + 2 * P, Ww + 2 * P, generator=g)

    def ref(dt):
        yy = y.to(dt).requires_grad_(True)
        ga, be = gamma.to(dt).requires_grad_(True), beta.to(dt).requires_grad_(True)
        u = F.batch_norm(yy, None, None, ga, be, True, 0.1, 1e-5)
        u = torch.maximum(u, slope * u)
        if P:
            u = F.pad(u, (P,) * 4, mode="reflect")
        (u * G.to(dt)).sum().backward()
        return yy.grad, ga.grad, be.grad

    r64, r32 = ref(torch.float64), ref(torch.float32)
    # forward statistics through an identity 1x1 conv (exercises the conv epilogue partials)
    eye = torch.eye(Cc).view(Cc, Cc, 1, 1)
    yv, stats = H.conv_fwd(y.to(dev), eye.to(dev), None, 1, REFLECT, want_stats=True)
    assert torch.equal(yv.cpu(), y)
    Cs = round_up(Cc, 4)
    state = torch.zeros(4 * Cs, device=dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    gam, bet = gamma.to(dev), beta.to(dev)
    st = H.stream(dev)
    N.check(lib.dip_bn_finalize(stats.data_ptr(), stats.shape[0], stats.shape[2], Cc, gam.data_ptr(), bet.data_ptr(),
                                1e-5, 0.1, state.data_ptr(), Cs, rm.data_ptr(), rv.data_ptr(), st))
    torch.cuda.synchronize()
    s = state.view(4, Cs).cpu().double()
    y64 = y.double()[0].reshape(Cc, -1)
    mean, var = y64.mean(1), y64.var(1, unbiased=False)
    assert torch.allclose(s[0, :Cc], mean, rtol=1e-6, atol=1e-6)
    assert torch.allclose(s[1, :Cc], 1 / torch.sqrt(var + 1e-5), rtol=2e-6)
    assert torch.allclose(rm.cpu().double(), 0.1 * mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.cpu().double(), 0.9 + 0.1 * y64.var(1, unbiased=True), rtol=1e-5)
    # backward
    yb = H.to_nhwc(y.to(dev))
    Gb = H.to_nhwc(G.to(dev))
    src = N.DipGradSrc(Gb.data_ptr(), P, 1 if P else 0, Cs, 0)
    nblk = lib.dip_bn_bwd_nblk(Hh, Ww, Cc)
    dz = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    part = torch.full((nblk * 2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_stats(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope,
                                 dz.data_ptr(), Cs, part.data_ptr(), nblk, st))
    dgam, dbet = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    coef = torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_finalize(part.data_ptr(), nblk, Cs, Cc, Hh * Ww, dgam.data_ptr(), dbet.data_ptr(),
                                    coef.data_ptr(), st))
    N.check(lib.dip_bn_bwd_apply(dz.data_ptr(), Cs, yb.data_ptr(), Cs, Hh * Ww, Cc, state.data_ptr(), Cs,
                                 coef.data_ptr(), st))
    torch.cuda.synchronize()
    dy = H.from_nhwc(dz, Cc, Hh, Ww)
    _check("bn_bwd.dy", dy, r64[0], r32[0], floor=5e-6)
    # the recomputing variant the engine uses: phase 1 without dz, phase 3 from the gradient source
    part2 = torch.full((nblk * 2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_stats(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope,
                                 None, Cs, part2.data_ptr(), nblk, st))
    dy2 = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_apply_src(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope,
                                     coef.data_ptr(), dy2.data_ptr(), Cs, st))
    torch.cuda.synchronize()
    assert torch.equal(part2, part)
    assert torch.equal(H.from_nhwc(dy2, Cc, Hh, Ww), dy)
    _check("bn_bwd.dgamma", dgam, r64[1], r32[1], floor=5e-6)
    _check("bn_bwd.dbeta", dbet, r64[2], r32[2], floor=5e-6)


@pytest.mark.parametrize("Cc,Co,Hh,Ww,slope", [(128, 3, 40, 72, 0.2), (16, 1, 33, 47, 0.2), (64, 4, 128, 128, 1.0)])
def test_bn_backward_from_the_thin_output_conv(dev, Cc, Co, Hh, Ww, slope):
    """DipGradSrc.tw: conv-output y -> BN(train)+LeakyReLU -> the net's last 1x1 conv (Co <= 4 channels, models/skip.py:98)
    -> random linear functional.  The BatchNorm backward evaluates du = W^T dout per pixel instead of reading a data
    gradient from memory; dy, dgamma, dbeta against autograd in fp64 (tolerance: the fp32 reference's own distance)."""
    lib = N.lib()
    g = torch.Generator().manual_seed(5)
    y = torch.randn(1, Cc, Hh, Ww, generator=g) * 2.0 + torch.randn(1, Cc, 1, 1, generator=g) * 3.0
    gamma = torch.rand(Cc, generator=g) + 0.5
    beta = torch.randn(Cc, generator=g)
    w = torch.randn(Co, Cc, 1, 1, generator=g) / Cc ** 0.5
    G = torch.randn(1, Co, Hh, Ww, generator=g)

    def ref(dt):
        yy = y.to(dt).requires_grad_(True)
        ga, be = gamma.to(dt).requires_grad_(True), beta.to(dt).requires_grad_(True)
        u = F.batch_norm(yy, None, None, ga, be, True, 0.1, 1e-5)
        u = torch.maximum(u, slope * u)
        (F.conv2d(u, w.to(dt)) * G.to(dt)).sum().backward()
        return yy.grad, ga.grad, be.grad

    r64, r32 = ref(torch.float64), ref(torch.float32)
    Cs = round_up(Cc, 4)
    y64 = y.double()[0].reshape(Cc, -1)
    mean, rstd = y64.mean(1), 1 / torch.sqrt(y64.var(1, unbiased=False) + 1e-5)
    a = gamma.double() * rstd
    state = torch.zeros(4, Cs, dtype=torch.float64)
    state[0, :Cc], state[1, :Cc], state[2, :Cc], state[3, :Cc] = mean, rstd, a, beta.double() - mean * a
    state = state.float().to(dev).contiguous()
    st = H.stream(dev)
    yb = H.to_nhwc(y.to(dev))
    Gb = H.to_nhwc(G.to(dev))                        # [H*W][4], pad channels zero
    if Co < 4:
        Gb.view(-1, 4)[:, Co:] = float("nan")        # ... or anything else: they are not part of the sum
    wd = w.to(dev).contiguous()
    src = N.DipGradSrc(Gb.data_ptr(), 0, 0, 4, 0)
    src.tw, src.tn, src.tcw = wd.data_ptr(), Co, Cc
    nblk = lib.dip_bn_bwd_nblk(Hh, Ww, Cc)
    part = torch.full((nblk * 2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_stats(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope,
                                 None, Cs, part.data_ptr(), nblk, st))
    dgam, dbet = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    coef = torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_finalize(part.data_ptr(), nblk, Cs, Cc, Hh * Ww, dgam.data_ptr(), dbet.data_ptr(),
                                    coef.data_ptr(), st))
    dy = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_apply_src(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope,
                                     coef.data_ptr(), dy.data_ptr(), Cs, st))
    torch.cuda.synchronize()
    _check("thin_src.dy", H.from_nhwc(dy, Cc, Hh, Ww), r64[0], r32[0], floor=5e-6)
    _check("thin_src.dgamma", dgam, r64[1], r32[1], floor=5e-6)
    _check("thin_src.dbeta", dbet, r64[2], r32[2], floor=5e-6)
    # the explicit form (du written by the conv's data gradient, then read) gives the same dy to rounding
    du = H.conv_dgrad(G.to(dev), wd, 1, REFLECT, Hh, Ww)
    dub = H.to_nhwc(du)
    src2 = N.DipGradSrc(dub.data_ptr(), 0, 0, Cs, 0)
    dy2 = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    part2 = torch.full((nblk * 2 * Cs,), float("nan"), device=dev)
    coef2 = torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_stats(C.byref(src2), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope,
                                 None, Cs, part2.data_ptr(), nblk, st))
    N.check(lib.dip_bn_bwd_finalize(part2.data_ptr(), nblk, Cs, Cc, Hh * Ww, dgam.data_ptr(), dbet.data_ptr(),
                                    coef2.data_ptr(), st))
    N.check(lib.dip_bn_bwd_apply_src(C.byref(src2), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope,
                                     coef2.data_ptr(), dy2.data_ptr(), Cs, st))
    torch.cuda.synchronize()
    d1, d2 = H.from_nhwc(dy, Cc, Hh, Ww).double(), H.from_nhwc(dy2, Cc, Hh, Ww).double()
    assert (d1 - d2).norm() <= 2e-6 * d2.norm()


@pytest.mark.parametrize("Cin,Cout,ks,Hh,Ww,slope,mode", [
    (128, 128, 3, 24, 40, 0.2, "reflect"),      # LDS-DMA kernel, padded domain with a reflected ring, ragged tiles
    (132, 128, 3, 16, 32, 1.0, "reflect"),      # 132 columns: conv_thin4 (4 columns) + 128 columns, no activation (concat BN)
    (128, 128, 1, 16, 48, 0.2, "reflect"),      # 1x1 (no padding)
    (128, 128, 1, 256, 256, 0.2, "reflect"),    # 1x1 at >= 65536 pixels: the weights-resident kernel's epilogue
    (64, 96, 3, 19, 23, 0.2, "zero"),           # zero padding, 64-column block
    (128, 8, 1, 16, 16, 0.2, "reflect"),        # data gradient of a thin conv (K = 8), like the RGB output conv
    (32, 32, 5, 12, 20, 0.2, "reflect")])       # 5x5: the register-staged kernel's epilogue
def test_bn_backward_statistics_fused_into_the_data_gradient(dev, Cin, Cout, ks, Hh, Ww, slope, mode):
    """DipConvDesc.bnb_*: phase 1 of the BatchNorm(+LeakyReLU) backward of a conv's INPUT activation computed in the
    epilogue of the data-gradient launch (per tile, ring positions weighted with the activation of their mirror
    pixel) + dip_bn_bwd_finalize2, against the separate pass dip_bn_bwd_stats + dip_bn_bwd_finalize over the same
    gradient buffer, and against autograd in fp64."""
    lib = N.lib()
    g = torch.Generator().manual_seed(Cin + ks)
    P = (ks - 1) // 2
    pad_mode = REFLECT if mode == "reflect" else N.PAD_ZERO
    y = torch.randn(1, Cin, Hh, Ww, generator=g) * 1.5 + torch.randn(1, Cin, 1, 1, generator=g)
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (ks * Cin ** 0.5)
    dy = torch.randn(1, Cout, Hh, Ww, generator=g)

    def ref(dt):
        yy = y.to(dt).requires_grad_(True)
        ga, be = gamma.to(dt).requires_grad_(True), beta.to(dt).requires_grad_(True)
        u = F.batch_norm(yy, None, None, ga, be, True, 0.1, 1e-5)
        u = torch.maximum(u, slope * u) if slope < 1.0 else u
        o = _ref_conv(u, w.to(dt), None, 1, pad_mode, dt)
        (o * dy.to(dt)).sum().backward()
        return ga.grad, be.grad

    r64, r32 = ref(torch.float64), ref(torch.float32)
    st = H.stream(dev)
    Cs = round_up(Cin, 4)
    # forward state of the BatchNorm (mean, rstd, a, b) from the host (the finalisation kernel has its own test)
    y64 = y.double()[0].reshape(Cin, -1)
    mean, var = y64.mean(1), y64.var(1, unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    state = torch.zeros(4, Cs, dtype=torch.float64)
    state[0, :Cin], state[1, :Cin] = mean, rstd
    state[2, :Cin], state[3, :Cin] = gamma.double() * rstd, beta.double() - mean * gamma.double() * rstd
    state = state.float().to(dev).contiguous()
    yb = H.to_nhwc(y.to(dev))
    reflect = pad_mode == REFLECT and P > 0
    pad = P if reflect else 0
    Hg, Wg = Hh + 2 * pad, Ww + 2 * pad
    off = (ks - 1) if reflect else (ks - 1 - P)
    packed, _, do = H.pack(w.to(dev))
    dyb = H.to_nhwc(dy.to(dev))
    gbuf = torch.full((Hg * Wg * Cs,), float("nan"), device=dev)
    rows, rows_lo = lib.dip_conv_ntiles(Hg, Wg), lib.dip_conv_thin4_ntiles(Hg, Wg)
    part = torch.full((rows * 2 * Cs,), float("nan"), device=dev)
    part_lo = torch.full((rows_lo * 2 * Cs,), float("nan"), device=dev)
    d = N.DipConvDesc(dyb.data_ptr(), Hh, Ww, round_up(Cout, 4), round_up(Cout, 4), N.DipTransform(None, None, 1.0),
                      packed.data_ptr() + 4 * do, None, gbuf.data_ptr(), Hg, Wg, Cs, Cin, 0, ks, 1, N.PAD_ZERO, off,
                      1, 0, None, 1, None)
    assert lib.dip_conv_bnb_fusable(C.byref(d)) == 1
    variant = lib.dip_conv_variant(C.byref(d))
    c_lo = Cin - 128 if variant == 3 else 0
    d.bnb_y, d.bnb_state, d.bnb_partials, d.bnb_partials_thin = yb.data_ptr(), state.data_ptr(), part.data_ptr(), part_lo.data_ptr()
    d.bnb_Cy, d.bnb_Cs, d.bnb_pad, d.bnb_slope = Cs, Cs, pad, slope
    N.check(lib.dip_conv_igemm(C.byref(d), st), "conv_igemm(dgrad + bnb)")
    dgam, dbet, coef = (torch.full((n,), float("nan"), device=dev) for n in (Cin, Cin, 2 * Cs))
    N.check(lib.dip_bn_bwd_finalize2(part.data_ptr(), rows, part_lo.data_ptr() if c_lo else None, rows_lo, c_lo, Cs, Cin,
                                     Hh * Ww, dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(), st))
    # the separate pass over the same gradient buffer
    src = N.DipGradSrc(gbuf.data_ptr(), pad, 1 if pad else 0, Cs, 0)
    nblk = lib.dip_bn_bwd_nblk(Hh, Ww, Cin)
    part2 = torch.full((nblk * 2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_stats(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cin, state.data_ptr(), Cs, slope, None, Cs,
                                 part2.data_ptr(), nblk, st))
    dgam2, dbet2, coef2 = (torch.full((n,), float("nan"), device=dev) for n in (Cin, Cin, 2 * Cs))
    N.check(lib.dip_bn_bwd_finalize(part2.data_ptr(), nblk, Cs, Cin, Hh * Ww, dgam2.data_ptr(), dbet2.data_ptr(),
                                    coef2.data_ptr(), st))
    torch.cuda.synchronize()
    assert variant == (3 if Cin == 132 else (0 if ks == 5 else (6 if Hh * Ww >= 65536 else 1))), variant
    for a, b, name in ((dgam, dgam2, "dgamma"), (dbet, dbet2, "dbeta")):
        scale = b.abs().max().item()
        assert torch.allclose(a, b, rtol=0, atol=2e-6 * scale + 1e-9), (name, (a - b).abs().max().item(), scale)
    cv, cv2 = coef.view(2, Cs)[:, :Cin], coef2.view(2, Cs)[:, :Cin]
    assert torch.allclose(cv, cv2, rtol=0, atol=2e-6 * cv2.abs().max().item() + 1e-12)
    _check("bnb_fused.dgamma", dgam, r64[0], r32[0], floor=5e-6)
    _check("bnb_fused.dbeta", dbet, r64[1], r32[1], floor=5e-6)


# ----------------------------------------------------------------------------- AvgPool2d(2, 2)
@pytest.mark.parametrize("Cc,Hh,Ww", [(16, 8, 12), (128, 16, 32), (30, 10, 6)])
def test_avgpool2_forward_stats_backward(dev, Cc, Hh, Ww):
    """conv(..., downsample_mode='avg') pooling stage (reference models/common.py:101-104)."""
    lib = N.lib()
    torch.manual_seed(Cc)
    x = torch.randn(1, Cc, Hh, Ww)
    Cs = round_up(Cc, 4)
    xb = H.to_nhwc(x.to(dev))
    nblk = lib.dip_upcat_nblk(Hh // 2, Ww // 2, Cc)
    y = torch.full(((Hh // 2) * (Ww // 2) * Cs,), float("nan"), device=dev)
    stats = torch.full((nblk * 3 * Cs,), float("nan"), device=dev)
    st = H.stream(dev)
    N.check(lib.dip_avgpool2_fwd(xb.data_ptr(), Hh, Ww, Cs, Cc, y.data_ptr(), Cs, stats.data_ptr(), nblk, st))
    ref = F.avg_pool2d(x.double(), 2, 2)
    got = H.from_nhwc(y, Cc, Hh // 2, Ww // 2).cpu().double()
    assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6)
    # partials -> batch statistics through the BatchNorm finalisation
    state = torch.zeros(4 * Cs, device=dev)
    gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    N.check(lib.dip_bn_finalize(stats.data_ptr(), nblk, Cs, Cc, gam.data_ptr(), bet.data_ptr(), 1e-5, 0.1,
                                state.data_ptr(), Cs, None, None, st))
    torch.cuda.synchronize()
    s = state.view(4, Cs).cpu().double()
    r = ref[0].reshape(Cc, -1)
    assert torch.allclose(s[0, :Cc], r.mean(1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(s[1, :Cc], 1 / torch.sqrt(r.var(1, unbiased=False) + 1e-5), rtol=1e-5)
    # adjoint
    g = torch.randn(1, Cc, Hh // 2, Ww // 2)
    gb = H.to_nhwc(g.to(dev))
    dx = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    N.check(lib.dip_avgpool2_bwd(gb.data_ptr(), Hh, Ww, Cs, Cc, dx.data_ptr(), Cs, st))
    torch.cuda.synchronize()
    xr = x.double().requires_grad_(True)
    (F.avg_pool2d(xr, 2, 2) * g.double()).sum().backward()
    assert torch.allclose(H.from_nhwc(dx, Cc, Hh, Ww).cpu().double(), xr.grad, rtol=1e-6, atol=1e-7)


# ----------------------------------------------------------------------------- upsample + concat
@pytest.mark.parametrize("ns,nd,Hh,Ww,mode", [(4, 128, 16, 32, "bilinear"), (0, 32, 8, 8, "nearest"),
                                               (128, 128, 8, 16, "nearest"), (4, 16, 12, 20, "bilinear"),
                                               (0, 8, 4, 6, "bilinear")])
def test_upcat_forward_backward(dev, ns, nd, Hh, Ww, mode):
    lib = N.lib()
    g = torch.Generator().manual_seed(5)
    Hl, Wl = Hh // 2, Ww // 2
    s = torch.randn(1, max(ns, 1), Hh, Ww, generator=g)
    d = torch.randn(1, nd, Hl, Wl, generator=g)
    a_s, b_s = torch.rand(max(ns, 1), generator=g) + 0.5, torch.randn(max(ns, 1), generator=g) * 0.3
    a_d, b_d = torch.rand(nd, generator=g) + 0.5, torch.randn(nd, generator=g) * 0.3
    Gc = torch.randn(1, ns + nd, Hh, Ww, generator=g)
    slope = 0.2

    def ref(dt):
        dd = d.to(dt).requires_grad_(True)
        ud = _apply_tr(dd, a_d, b_d, slope, dt)
        up = F.interpolate(ud, scale_factor=2, mode=mode)
        parts = [up]
        if ns:
            parts = [_apply_tr(s, a_s, b_s, slope, dt), up]
        cat = torch.cat(parts, 1)
        (cat * Gc.to(dt)).sum().backward()
        return cat.detach(), dd.grad

    (c64, g64), (c32, g32) = ref(torch.float64), ref(torch.float32)
    st = H.stream(dev)
    sb = H.to_nhwc(s.to(dev)) if ns else None
    db = H.to_nhwc(d.to(dev))
    ts, k1 = H.transform(a_s.to(dev), b_s.to(dev), slope) if ns else (N.DipTransform(None, None, 1.0), None)
    td, k2 = H.transform(a_d.to(dev), b_d.to(dev), slope)
    Ccat = ns + nd
    Cs_cat = round_up(Ccat, 4)
    cat = torch.full((Hh * Ww * Cs_cat,), float("nan"), device=dev)
    nblk = lib.dip_upcat_nblk(Hh, Ww, Ccat)
    stats = torch.full((nblk * 3 * Cs_cat,), float("nan"), device=dev)
    desc = N.DipUpcatDesc(sb.data_ptr() if ns else None, round_up(max(ns, 1), 4), ns, ts, db.data_ptr(),
                          round_up(nd, 4), nd, td, Hh, Ww, N.UP_BILINEAR if mode == "bilinear" else N.UP_NEAREST,
                          cat.data_ptr(), Cs_cat, stats.data_ptr(), nblk)
    N.check(lib.dip_upcat_fwd(C.byref(desc), st))
    torch.cuda.synchronize()
    got = H.from_nhwc(cat, Ccat, Hh, Ww)
    _check("upcat_fwd", got, c64, c32)
    stn = stats.view(nblk, 3, Cs_cat).cpu().double().numpy()
    n, m, M2 = stn[:, 0, :Ccat], stn[:, 1, :Ccat], stn[:, 2, :Ccat]
    Nt = n.sum(0)
    mean = (n * m).sum(0) / Nt
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / Nt
    r = c64[0].reshape(Ccat, -1)
    assert np.allclose(mean, r.mean(1).numpy(), rtol=1e-5, atol=1e-6)
    assert np.allclose(var, r.var(1, unbiased=False).numpy(), rtol=2e-5)
    # backward of the deeper branch: dz = up^T(dcat) * lrelu'(a*y+b); (mean, rstd) arbitrary here
    Cs = round_up(nd, 4)
    state = torch.zeros(4, Cs)
    state[0, :nd] = 0.1
    state[1, :nd] = 1.3
    state[2, :nd] = a_d
    state[3, :nd] = b_d
    state = state.to(dev).contiguous()
    Gb = H.to_nhwc(Gc.to(dev), Cs_cat)
    nb2 = lib.dip_bn_bwd_nblk(Hl, Wl, nd)
    dz = torch.full((Hl * Wl * Cs,), float("nan"), device=dev)
    part = torch.full((nb2 * 2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_upsample_bwd_stats(Gb.data_ptr(), Cs_cat, ns, Hh, Ww,
                                       N.UP_BILINEAR if mode == "bilinear" else N.UP_NEAREST, db.data_ptr(), Cs, nd,
                                       state.data_ptr(), Cs, slope, dz.data_ptr(), Cs, part.data_ptr(), nb2, st))
    torch.cuda.synchronize()
    # reference dz = d(loss)/d(z) where z = a*y+b  ->  grad wrt d divided by a
    _check("upsample_bwd.dz", H.from_nhwc(dz, nd, Hl, Wl) * a_d.view(1, -1, 1, 1).to(dev), g64, g32, floor=5e-6)
    p = part.view(nb2, 2, Cs).cpu().double().sum(0)
    dz64 = (g64 / a_d.double().view(1, -1, 1, 1))[0].reshape(nd, -1)
    xh = ((d.double()[0].reshape(nd, -1)) - 0.1) * 1.3
    assert torch.allclose(p[0, :nd], dz64.sum(1), rtol=1e-4, atol=1e-4)
    assert torch.allclose(p[1, :nd], (dz64 * xh).sum(1), rtol=1e-4, atol=1e-4)


# ----------------------------------------------------------------------------- small kernels
def test_layout_head_roundtrip(dev):
    lib = N.lib()
    st = H.stream(dev)
    for Cc, HW in ((32, 64 * 48), (3, 1000), (1, 77), (40, 500), (30, 257)):
        x = torch.randn(Cc, HW, device=dev)
        Cs = round_up(Cc, 4)
        nh = torch.full((HW * Cs,), float("nan"), device=dev)
        N.check(lib.dip_nchw_to_nhwc(x.data_ptr(), nh.data_ptr(), Cc, HW, Cs, st))
        back = torch.empty_like(x)
        N.check(lib.dip_nhwc_to_nchw(nh.data_ptr(), back.data_ptr(), Cc, HW, Cs, 0, st))
        out = torch.empty_like(x)
        N.check(lib.dip_head_fwd(nh.data_ptr(), out.data_ptr(), Cc, HW, Cs, 1, st))
        gout = torch.randn_like(x)
        dy = torch.full((HW * Cs,), float("nan"), device=dev)
        N.check(lib.dip_head_bwd(gout.data_ptr(), out.data_ptr(), dy.data_ptr(), Cc, HW, Cs, 1, st))
        torch.cuda.synchronize()
        assert torch.equal(back, x)
        assert torch.all(nh.view(HW, Cs)[:, Cc:] == 0)
        ref = torch.sigmoid(x.cpu().double())
        assert torch.allclose(out.cpu().double(), ref, rtol=2e-6, atol=1e-7)
        refg = gout.cpu().double() * ref * (1 - ref)
        assert torch.allclose(dy.view(HW, Cs)[:, :Cc].t().cpu().double(), refg, rtol=1e-5, atol=1e-7)


def test_adam_matches_torch(dev):
    """dip_adam_step vs torch.optim.Adam (CPU, the oracle's optimiser) on identical gradients.
    The kernel evaluates the update with ATen's own rounding sequence (fma lerp, fma addcmul,
    (-step*m)/denom), so exp_avg / exp_avg_sq agree BITWISE; the parameters agree to <= 2 ulp of
    max(|p_old|, |p_new|, |update|): the residual is the oracle's, whose `exp_avg_sq.sqrt()` runs MKL
    vsSqrt (<= 1 ulp, not correctly rounded) while v_sqrt + fix-up on the GPU is correctly rounded."""
    lib = N.lib()
    g = torch.Generator().manual_seed(11)
    n = 100003
    p0 = torch.randn(n, generator=g)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=0.01)
    p = p0.to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    eps32 = torch.finfo(torch.float32).eps
    worst = 0.0
    bitwise = True
    for step in range(1, 9):
        gr = torch.randn(n, generator=g) * (10.0 ** torch.randint(-6, 1, (n,), generator=g).float())
        p_old = pt.detach().clone()
        pt.grad = gr.clone()
        # both arms start every step from the oracle's state: the check is per step, not cumulative
        st0 = opt.state[pt]
        m_old = st0["exp_avg"].clone() if "exp_avg" in st0 else torch.zeros(n)
        v_old = st0["exp_avg_sq"].clone() if "exp_avg_sq" in st0 else torch.zeros(n)
        opt.step()
        st = opt.state[pt]
        p.copy_(p_old.to(dev))
        m.copy_(m_old.to(dev))
        v.copy_(v_old.to(dev))
        gd = gr.to(dev)
        N.check(lib.dip_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 0.01, 0.9, 0.999, 1e-8,
                                  step, H.stream(dev)))
        torch.cuda.synchronize()
        # moments: bitwise on an AVX-512 host; <= 1 ulp of the larger operand otherwise (another ISA path of
        # ATen may contract differently; exp_avg can cancel, so the unit is max(|old|, |g|), not |new|)
        assert ((m.cpu() - st["exp_avg"]).abs() <= eps32 * torch.maximum(m_old.abs(), gr.abs())).all(), step
        assert ((v.cpu() - st["exp_avg_sq"]).abs() <= eps32 * torch.maximum(v_old, gr * gr)).all(), step
        bitwise = bitwise and torch.equal(m.cpu(), st["exp_avg"]) and torch.equal(v.cpu(), st["exp_avg_sq"])
        p_new = pt.detach()
        d = (p.cpu() - p_new).abs()
        scale = torch.maximum(torch.maximum(p_old.abs(), p_new.abs()), (p_new - p_old).abs())
        r = (d / (eps32 * scale)).max().item()
        worst = max(worst, r)
        assert r <= 2.0, (step, r)
    print(f"adam: worst parameter deviation {worst:.2f} ulp-units; moments bitwise equal: {bitwise}")


def test_noise_axpy_statistics(dev):
    lib = N.lib()
    n = 1 << 22
    z = torch.rand(n, device=dev)
    out = torch.empty_like(z)
    N.check(lib.dip_noise_axpy(z.data_ptr(), out.data_ptr(), n, 0.5, 1234, 0, H.stream(dev)))
    out2 = torch.empty_like(z)
    N.check(lib.dip_noise_axpy(z.data_ptr(), out2.data_ptr(), n, 0.5, 1234, n // 4, H.stream(dev)))
    torch.cuda.synchronize()
    e = ((out - z) / 0.5).cpu().double()
    assert abs(e.mean().item()) < 3e-3 and abs(e.std().item() - 1) < 3e-3
    assert abs((e ** 4).mean().item() - 3) < 0.05                      # kurtosis of a normal
    assert not torch.equal(out, out2)                                    # offset advances the stream
    c = np.corrcoef(e[:-1].numpy(), e[1:].numpy())[0, 1]
    assert abs(c) < 3e-3


def test_lanczos_downsampler_golden(dev):
    """Downsampler forward/backward against vectors produced by the real reference."""
    import os
    from conftest import GOLDEN
    from models.downsampler import Downsampler
    gold = np.load(os.path.join(GOLDEN, "downsampler.npz"))
    for factor in (4, 2, 8):
        tag = f"lanczos2_f{factor}"
        dmod = Downsampler(n_planes=3, factor=factor, kernel_type="lanczos2", phase=0.5, preserve_size=True).to(dev)
        assert np.array_equal(dmod.kernel, gold[tag + "/kernel"])
        x = torch.from_numpy(gold[tag + "/x"]).to(dev).requires_grad_(True)
        y = dmod(x)
        (y * torch.from_numpy(gold[tag + "/gy"]).to(dev)).sum().backward()
        assert torch.allclose(y.detach().cpu(), torch.from_numpy(gold[tag + "/y"]), rtol=1e-5, atol=2e-6)
        assert torch.allclose(x.grad.cpu(), torch.from_numpy(gold[tag + "/gx"]), rtol=1e-5, atol=2e-6)
