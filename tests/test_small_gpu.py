"""Parity of the low-resolution path on a real MI355X: dip_conv_small (one launch = convolution + in-workgroup split-K +
BatchNorm partials + in-launch finalisation; data gradients with phases 1 and 2 of the producer BatchNorm's backward) and
the in-launch finalisations of dip_upcat_fwd_fin / dip_bn_bwd_stats_fin / dip_upsample_bwd_stats_crop_fin, each against a
torch-CPU fp64 evaluation of the same reference ops (nn.Conv2d behind nn.ReflectionPad2d, nn.BatchNorm2d in training mode,
nn.LeakyReLU, nn.Upsample + Concat: models/common.py:95-124, models/skip.py:55-91 of the reference) with the per-op criterion
of test_kernels_gpu.py."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402
from dip_native import round_up  # noqa: E402
import hipops as H  # noqa: E402
from test_kernels_gpu import _apply_tr, _check, _mk, _ref_conv, REFLECT, ZERO, REPLICATE  # noqa: E402

SMALL_CASES = [
    # Cin, Cout, ks, stride, pad, H, W, transform  -- the low-resolution layers of the notebooks' nets + ragged shapes
    (128, 128, 3, 1, REFLECT, 32, 32, True),     # s3.down_b: 32 groups x 4 blocks, 4 K slices
    (128, 128, 3, 1, REFLECT, 16, 16, True),     # s4.down_b
    (128, 128, 3, 1, REFLECT, 64, 64, True),     # s2.down_b: 128 rows of partials
    (128, 128, 3, 2, REFLECT, 64, 64, True),     # s3.down_a (stride 2)
    (132, 128, 3, 1, REFLECT, 32, 32, True),     # s4.up: 16 full K steps + the 4-channel tail
    (128, 128, 1, 1, REFLECT, 32, 32, True),     # s4.up1 (1x1): 16 K steps, no split
    (128, 4, 1, 1, REFLECT, 64, 64, True),       # s3.skip_conv: one column block, 28 idle columns
    (128, 4, 1, 1, REFLECT, 19, 27, False),      # ragged size, no transform
    (32, 128, 3, 2, REFLECT, 48, 40, False),     # first conv of a small image (stride 2, no transform)
    (36, 64, 3, 1, ZERO, 21, 13, True),          # zero padding, ragged everything, 2 column blocks
    (8, 16, 3, 2, ZERO, 16, 32, False),
    (256, 128, 3, 1, REFLECT, 16, 16, True),     # kate net decoder conv (256 input channels)
    (128, 132, 3, 1, REFLECT, 24, 24, False),    # 5 column blocks (the data gradient's shape as a forward)
    (16, 16, 3, 1, REPLICATE, 12, 20, True),     # replication padding
    # 5x5 / 7x7 filters (round 6: filter_size_down = 5 of the 'library' inpainting net, inpainting.ipynb:222-232)
    (16, 16, 5, 1, REFLECT, 40, 56, True),       # library s0.down_b at reduced size: 50 K steps, 8 waves
    (32, 64, 5, 2, REFLECT, 24, 36, True),       # library s2.down_a (stride 2)
    (128, 128, 5, 1, REFLECT, 14, 22, True),     # library s4.down_b at its real size: 400 K steps, 16 waves, 3 chunks
    (128, 128, 5, 2, REFLECT, 14, 22, True),     # library s5.down_a at its real size
    (4, 16, 5, 2, REFLECT, 32, 48, False),       # first conv of the net (<= 4 input channels)
    (16, 24, 7, 1, ZERO, 12, 20, True),          # 7x7 (feature_inversion.ipynb:169), zero padding
]


@pytest.mark.parametrize("case", SMALL_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_small_forward_and_stats(dev, case):
    Cin, Cout, ks, stride, pad, Hh, Ww, use_tr = case
    x, w, b, a, bb = _mk(case)
    slope = 0.2
    ref64 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float64), w, b, stride, pad, torch.float64)
    ref32 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float32), w, b, stride, pad, torch.float32)
    tr = (a.to(dev), bb.to(dev), slope) if use_tr else (None, None, 1.0)
    g = torch.Generator().manual_seed(5)
    bn = dict(gamma=torch.rand(Cout, generator=g) + 0.5, beta=torch.randn(Cout, generator=g), eps=1e-5, momentum=0.1,
              running_mean=torch.randn(Cout, generator=g), running_var=torch.rand(Cout, generator=g) + 0.5)
    y, stats, out = H.conv_small_fwd(x.to(dev), w.to(dev), b.to(dev), stride, pad, tr, bn=bn)
    _check("conv_small_fwd", y, ref64, ref32)
    # partial rows -> mean / biased variance per channel
    st = stats.cpu().double().numpy()
    n = st[:, 0, :Cout]; m = st[:, 1, :Cout]; M2 = st[:, 2, :Cout]
    N_ = n.sum(0)
    mean = (n * m).sum(0) / N_
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / N_
    r = ref64[0].reshape(Cout, -1)
    assert np.allclose(N_, r.shape[1])
    assert np.allclose(mean, r.mean(1).numpy(), rtol=1e-5, atol=1e-5 * float(r.std()))
    assert np.allclose(var, r.var(1, unbiased=False).numpy(), rtol=2e-5)
    # dip_bn_finalize over the launch's rows == nn.BatchNorm2d (training) statistics of the launch's own output
    yc = y.cpu().double()[0].reshape(Cout, -1)
    mu, vb = yc.mean(1), yc.var(1, unbiased=False)
    rstd = 1.0 / torch.sqrt(vb + 1e-5)
    state = out["state"].cpu().double()
    A = bn["gamma"].double() * rstd
    scale = float(yc.std()) + 1e-30
    assert torch.allclose(state[0, :Cout], mu, rtol=1e-5, atol=2e-6 * scale)
    assert torch.allclose(state[1, :Cout], rstd, rtol=2e-5)
    assert torch.allclose(state[2, :Cout], A, rtol=2e-5)
    assert torch.allclose(state[3, :Cout], bn["beta"].double() - mu * A, rtol=2e-5, atol=1e-5 * float(A.abs().max()) * scale)
    npix = yc.shape[1]
    rm = 0.9 * bn["running_mean"].double() + 0.1 * mu
    rv = 0.9 * bn["running_var"].double() + 0.1 * vb * npix / (npix - 1)
    assert torch.allclose(out["running_mean"].cpu().double(), rm, rtol=1e-5, atol=2e-6 * scale)
    assert torch.allclose(out["running_var"].cpu().double(), rv, rtol=2e-5)


DGRAD_CASES = [
    # Cin (columns of the gradient), Cout, ks, stride, pad, H, W (input size of the forward conv)
    (128, 128, 3, 1, REFLECT, 32, 32),           # s3.down_b data gradient on the 34 x 34 padded domain
    (132, 128, 3, 1, REFLECT, 32, 32),           # towards the concat: 5 column blocks, no conv_thin4 side launch
    (128, 128, 1, 1, REFLECT, 64, 64),           # 1x1
    (128, 128, 3, 2, REFLECT, 32, 32),           # stride-2 conv: four output-parity classes, 4 / 2 / 2 / 1 taps
    (128, 128, 3, 2, REFLECT, 19, 27),           # odd sizes: ragged parity classes
    (128, 64, 3, 2, ZERO, 18, 22),
    (36, 64, 3, 1, ZERO, 21, 13),
    (160, 128, 3, 1, REFLECT, 16, 16),           # 5 full column blocks
    (16, 16, 3, 1, REPLICATE, 12, 20),
    (16, 16, 5, 1, REFLECT, 24, 40),             # 5x5: gradient on the domain padded by 2
    (32, 64, 5, 2, REFLECT, 24, 36),             # stride-2 5x5: parity classes of 9 / 6 / 6 / 4 taps
    (128, 128, 5, 2, REFLECT, 14, 22),           # library s5.down_a at its real size
    (64, 64, 5, 1, ZERO, 13, 9),
    (16, 8, 7, 2, REFLECT, 16, 20),
]


def _dgrad_ref(case, seed=1):
    Cin, Cout, ks, stride, pad, Hh, Ww = case
    x, w, b, _, _ = _mk((Cin, Cout, ks, stride, pad, Hh, Ww, False), seed)
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = x.to(dt).requires_grad_(True)
        y = _ref_conv(xx, w, None, stride, pad, dt)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
        (y * dy.to(dt)).sum().backward()
        res[dt] = xx.grad
    return x, w, dy, res


@pytest.mark.parametrize("case", DGRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_small_dgrad(dev, case):
    Cin, Cout, ks, stride, pad, Hh, Ww = case
    x, w, dy, res = _dgrad_ref(case)
    gx, _ = H.conv_small_dgrad(dy.to(dev), w.to(dev), stride, pad, Hh, Ww)
    _check("conv_small_dgrad", gx, res[torch.float64], res[torch.float32])


@pytest.mark.parametrize("case", [c for c in DGRAD_CASES if c[4] != REPLICATE], ids=lambda c: "x".join(map(str, c)))
def test_conv_small_dgrad_with_fused_batchnorm_backward(dev, case):
    """Phases 1 and 2 of the backward of the BatchNorm + LeakyReLU that produced the conv's input, inside the data-gradient
    launch: dgamma, dbeta, k1 = sum(dz) / N, k2 = sum(dz * xhat) / N with dz = g * lrelu'(a y + b), against autograd."""
    Cin, Cout, ks, stride, pad, Hh, Ww = case
    x, w, dy, res = _dgrad_ref(case)
    g0 = torch.Generator().manual_seed(11)
    yraw = torch.randn(1, Cin, Hh, Ww, generator=g0)              # raw output of the producer conv
    gamma = torch.rand(Cin, generator=g0) + 0.5
    beta = torch.randn(Cin, generator=g0) * 0.3
    slope = 0.2
    out = {}
    for dt in (torch.float64, torch.float32):
        yy = yraw.to(dt)
        mu = yy.mean((0, 2, 3)); var = yy.var((0, 2, 3), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        a = gamma.to(dt) * rstd
        bcoef = beta.to(dt) - mu * a
        z = yy * a.view(1, -1, 1, 1) + bcoef.view(1, -1, 1, 1)
        gfold = res[dt]
        dz = gfold * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))
        xh = (yy - mu.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1)
        out[dt] = dict(s1=dz.sum((0, 2, 3)), s2=(dz * xh).sum((0, 2, 3)), state=torch.stack([mu, rstd, a, bcoef]))
    Cs = round_up(Cin, 4)
    state = torch.zeros(4, Cs)
    state[:, :Cin] = out[torch.float32]["state"]
    gx, fin = H.conv_small_dgrad(dy.to(dev), w.to(dev), stride, pad, Hh, Ww,
                                 bnb=dict(y=yraw, state=state, slope=slope))
    _check("conv_small_dgrad", gx, res[torch.float64], res[torch.float32])
    npix = Hh * Ww
    # the state handed to the kernel is the fp32 one: compare with the fp64 sums evaluated on the same branch pattern
    yy = yraw.double()
    st = state[:, :Cin].double()
    z = yy * st[2].view(1, -1, 1, 1) + st[3].view(1, -1, 1, 1)
    dz = res[torch.float64] * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))
    xh = (yy - st[0].view(1, -1, 1, 1)) * st[1].view(1, -1, 1, 1)
    s1, s2 = dz.sum((0, 2, 3)), (dz * xh).sum((0, 2, 3))
    scale = float(dz.abs().sum((0, 2, 3)).max())              # sums of npix terms of mixed sign: bound relative to sum |dz|
    for name, got, ref in (("dbeta", fin["dbeta"], s1), ("dgamma", fin["dgamma"], s2),
                           ("k1", fin["coef"][0, :Cin] * npix, s1), ("k2", fin["coef"][1, :Cin] * npix, s2)):
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 3e-6 * scale, f"{name}: {err:.3e} vs scale {scale:.3e}"


def test_upcat_and_backward_statistics_with_in_launch_finalisation(dev):
    """dip_upcat_fwd_fin, dip_bn_bwd_stats_fin and dip_upsample_bwd_stats_crop_fin (opt-in, DIP_TICKET_FIN=1) produce what the
    two-launch forms (partials + dip_bn_finalize / dip_bn_bwd_finalize) produce: the same rows reduced in fp64 (another
    pairing of the rows: equal to ~1e-7, the tensors bit for bit)."""
    lib = N.lib()
    st = H.stream(dev)
    g = torch.Generator().manual_seed(3)
    Hh, Ww, ns, nd = 32, 48, 4, 128
    Ccat, Cs_cat = ns + nd, round_up(ns + nd, 4)
    s = torch.randn(Hh * Ww * 4, generator=g).to(dev)
    dlow = torch.randn((Hh // 2) * (Ww // 2) * nd, generator=g).to(dev)
    ts = torch.stack([torch.rand(4, generator=g) + 0.5, torch.randn(4, generator=g)]).to(dev).contiguous()
    td = torch.stack([torch.rand(nd, generator=g) + 0.5, torch.randn(nd, generator=g)]).to(dev).contiguous()
    gamma = (torch.rand(Ccat, generator=g) + 0.5).to(dev)
    beta = torch.randn(Ccat, generator=g).to(dev)
    nblk = lib.dip_upcat_nblk(Hh, Ww, Ccat)
    assert lib.dip_fin_rows_ok(nblk, Ccat) == 1
    res = []
    for fused in (False, True):
        cat = torch.full((Hh * Ww * Cs_cat,), float("nan"), device=dev)
        stats = torch.full((nblk * 3 * Cs_cat,), float("nan"), device=dev)
        state = torch.full((4 * Cs_cat,), float("nan"), device=dev)
        rm, rv = torch.zeros(Ccat, device=dev), torch.ones(Ccat, device=dev)
        tickets = torch.zeros(8, dtype=torch.int32, device=dev)
        d = N.DipUpcatDesc(s.data_ptr(), 4, ns, N.DipTransform(ts[0].data_ptr(), ts[1].data_ptr(), 0.2),
                           dlow.data_ptr(), nd, nd, N.DipTransform(td[0].data_ptr(), td[1].data_ptr(), 0.2), Hh, Ww,
                           N.UP_BILINEAR, cat.data_ptr(), Cs_cat, stats.data_ptr(), nblk)
        if fused:
            fin = N.DipBnFin(gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, state.data_ptr(), Cs_cat, Ccat, rm.data_ptr(),
                             rv.data_ptr(), tickets.data_ptr())
            N.check(lib.dip_upcat_fwd_fin(C.byref(d), C.byref(fin), st), "upcat_fwd_fin")
        else:
            N.check(lib.dip_upcat_fwd(C.byref(d), st), "upcat_fwd")
            N.check(lib.dip_bn_finalize(stats.data_ptr(), nblk, Cs_cat, Ccat, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1,
                                        state.data_ptr(), Cs_cat, rm.data_ptr(), rv.data_ptr(), st), "bn_finalize")
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        res.append((cat.clone(), state.view(4, Cs_cat)[:, :Ccat].clone(), rm.clone(), rv.clone()))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):            # (the in-launch fp64 tree pairs the rows in another order)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    cat, state = res[0][0], torch.zeros(4, Cs_cat, device=dev)
    state[:, :Ccat] = res[0][1]
    state = state.contiguous()
    # backward statistics of the concat BatchNorm (identity activation) from a gradient of the same shape
    gbuf = torch.randn(Hh * Ww * Cs_cat, generator=g).to(dev)
    nb = lib.dip_bn_bwd_nblk(Hh, Ww, Ccat)
    out = []
    for fused in (False, True):
        part = torch.full((nb * 2 * Cs_cat,), float("nan"), device=dev)
        coef = torch.full((2 * Cs_cat,), float("nan"), device=dev)
        dga, dbe = torch.full((Ccat,), float("nan"), device=dev), torch.full((Ccat,), float("nan"), device=dev)
        tickets = torch.zeros(8, dtype=torch.int32, device=dev)
        src = N.DipGradSrc(gbuf.data_ptr(), 0, 0, Cs_cat, 0)
        if fused and lib.dip_fin_rows_ok(nb, Ccat):
            fin = N.DipBnbFin(dga.data_ptr(), dbe.data_ptr(), coef.data_ptr(), Ccat, Hh * Ww, tickets.data_ptr())
            N.check(lib.dip_bn_bwd_stats_fin(C.byref(src), cat.data_ptr(), Hh, Ww, Cs_cat, Ccat, state.data_ptr(), Cs_cat,
                                             1.0, None, Cs_cat, part.data_ptr(), nb, C.byref(fin), st), "bn_bwd_stats_fin")
        else:
            N.check(lib.dip_bn_bwd_stats(C.byref(src), cat.data_ptr(), Hh, Ww, Cs_cat, Ccat, state.data_ptr(), Cs_cat, 1.0,
                                         None, Cs_cat, part.data_ptr(), nb, st), "bn_bwd_stats")
            N.check(lib.dip_bn_bwd_finalize(part.data_ptr(), nb, Cs_cat, Ccat, Hh * Ww, dga.data_ptr(), dbe.data_ptr(),
                                            coef.data_ptr(), st), "bn_bwd_finalize")
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        out.append((coef.view(2, Cs_cat)[:, :Ccat].clone(), dga.clone(), dbe.clone()))
    for a, b in zip(*out):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
    # adjoint of the up-sampling + statistics of the deeper branch's BatchNorm
    Hd, Wd = Hh // 2, Ww // 2
    ylow = torch.randn(Hd * Wd * nd, generator=g).to(dev)
    sd = torch.stack([torch.randn(nd, generator=g), torch.rand(nd, generator=g) + 0.5, torch.rand(nd, generator=g) + 0.5,
                      torch.randn(nd, generator=g)]).to(dev).contiguous()
    nb = lib.dip_bn_bwd_nblk(Hd, Wd, nd)
    assert lib.dip_fin_rows_ok(nb, nd) == 1
    out = []
    for fused in (False, True):
        part = torch.full((nb * 2 * nd,), float("nan"), device=dev)
        coef = torch.full((2 * nd,), float("nan"), device=dev)
        dga, dbe = torch.full((nd,), float("nan"), device=dev), torch.full((nd,), float("nan"), device=dev)
        dz = torch.full((Hd * Wd * nd,), float("nan"), device=dev)
        tickets = torch.zeros(8, dtype=torch.int32, device=dev)
        if fused:
            fin = N.DipBnbFin(dga.data_ptr(), dbe.data_ptr(), coef.data_ptr(), nd, Hd * Wd, tickets.data_ptr())
            N.check(lib.dip_upsample_bwd_stats_crop_fin(gbuf.data_ptr(), Cs_cat, ns, Hh, Ww, Hd, Wd, 0, 0, N.UP_BILINEAR,
                                                        ylow.data_ptr(), nd, nd, sd.data_ptr(), nd, 0.2, dz.data_ptr(), nd,
                                                        part.data_ptr(), nb, C.byref(fin), st), "upsample_bwd_stats_crop_fin")
        else:
            N.check(lib.dip_upsample_bwd_stats(gbuf.data_ptr(), Cs_cat, ns, Hh, Ww, N.UP_BILINEAR, ylow.data_ptr(), nd, nd,
                                               sd.data_ptr(), nd, 0.2, dz.data_ptr(), nd, part.data_ptr(), nb, st),
                    "upsample_bwd_stats")
            N.check(lib.dip_bn_bwd_finalize(part.data_ptr(), nb, nd, nd, Hd * Wd, dga.data_ptr(), dbe.data_ptr(),
                                            coef.data_ptr(), st), "bn_bwd_finalize")
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0
        out.append((coef.clone(), dga.clone(), dbe.clone(), dz.clone()))
    assert torch.equal(out[0][3], out[1][3])
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))


RING_CASES = [
    # Cin (columns of the gradient), Cout, H, W
    (128, 128, 32, 32),
    (132, 128, 24, 40),          # 132 columns: conv_thin4 + the 128-column launch inside dip_conv_igemm, 5 column blocks in the ring
    (64, 32, 4, 4),              # rows 1 and H-2 adjacent: every frame pixel is a corner case
    (36, 64, 21, 13),            # ragged
    (128, 128, 5, 64),
]


@pytest.mark.parametrize("case", RING_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_dgrad_interior_plus_ring(dev, case):
    """Adjoint of nn.ReflectionPad2d(1) + Conv2d(3x3) without the padded domain: interior correlation + the frame launch."""
    Cin, Cout, Hh, Ww = case
    x, w, dy, res = _dgrad_ref((Cin, Cout, 3, 1, REFLECT, Hh, Ww))
    gx = H.conv_dgrad_ring(dy.to(dev), w.to(dev), Hh, Ww)
    _check("conv_dgrad_ring", gx, res[torch.float64], res[torch.float32])
