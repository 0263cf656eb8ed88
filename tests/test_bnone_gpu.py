"""Parity of the one-launch BatchNorm backward of the low-resolution scales (csrc/bn_bwd_one.hip: dip_bn_bwd_one,
dip_upsample_bwd_one) on a real MI355X against torch-CPU fp64 autograd of the reference ops -- nn.BatchNorm2d in training
mode + nn.LeakyReLU (models/common.py:82,95-96 of the reference) behind nn.ReflectionPad2d (models/common.py:116-118) or
nn.Upsample(scale_factor=2) (models/skip.py:81) -- with the per-op criterion of test_kernels_gpu.py (error vs fp64 <= 2 x the
torch-fp32 error + a roundoff floor), and against the three-launch form it replaces."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402
from dip_native import round_up  # noqa: E402
import hipops as H  # noqa: E402
from test_kernels_gpu import _check  # noqa: E402


def _state(y, gamma, beta):
    """[4][Cs] state block (mean, rstd, a, b) as dip_bn_finalize computes it (fp64, rounded once)."""
    Cc = y.shape[1]
    Cs = round_up(Cc, 4)
    y64 = y.double()[0].reshape(Cc, -1)
    mean, var = y64.mean(1), y64.var(1, unbiased=False)
    rstd = (1.0 / torch.sqrt(var + 1e-5)).float()
    a = gamma * rstd
    st = torch.zeros(4, Cs)
    st[0, :Cc], st[1, :Cc], st[2, :Cc] = mean.float(), rstd, a
    st[3, :Cc] = beta - mean.float() * a
    return st


@pytest.mark.parametrize("Cc,Hh,Ww,P,slope", [
    (128, 16, 32, 1, 0.2),        # 512 pixels: half of the workgroup's threads idle
    (128, 64, 64, 1, 0.2),        # s2.down_b of the default net: 4 pixels per thread
    (132, 32, 32, 1, 1.0),        # concat BatchNorm (no activation), 33 channel groups
    (4, 64, 64, 0, 0.2),          # 4-channel skip branch: ONE workgroup
    (16, 100, 90, 2, 0.2),        # 5x5 consumer (pad 2), ragged trip count (9000 pixels)
    (64, 56, 88, 2, 0.2),         # 'library' net s2.down_b at its real size
    (128, 128, 128, 1, 0.2),      # the default bound: 16384 pixels, 16 per thread
    (128, 7, 11, 2, 0.2),         # 'library' net s5: 77 pixels
])
def test_bn_bwd_one_matches_autograd(dev, Cc, Hh, Ww, P, slope):
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
    Cs = round_up(Cc, 4)
    state = _state(y, gamma, beta).to(dev).contiguous()
    st = H.stream(dev)
    yb = H.to_nhwc(y.to(dev))
    Gb = H.to_nhwc(G.to(dev))
    src = N.DipGradSrc(Gb.data_ptr(), P, 1 if P else 0, Cs, 0)
    assert lib.dip_bn_bwd_one_ok(32 * 32, Cc) == 1      # (the engine's bound; the kernel itself serves any size)
    dy = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    dgam, dbet = torch.full((Cc,), float("nan"), device=dev), torch.full((Cc,), float("nan"), device=dev)
    coef = torch.full((2 * Cs,), float("nan"), device=dev)
    for _ in range(2):                       # (a second launch must reproduce the first bit for bit)
        N.check(lib.dip_bn_bwd_one(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope, dy.data_ptr(), Cs,
                                   dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(), st), "bn_bwd_one")
        torch.cuda.synchronize()
        if _ == 0:
            first = (dy.clone(), dgam.clone(), dbet.clone(), coef.clone())
    assert all(torch.equal(a, b) for a, b in zip(first, (dy, dgam, dbet, coef)))
    _check("bn_bwd_one.dy", H.from_nhwc(dy, Cc, Hh, Ww), r64[0], r32[0], floor=5e-6)
    _check("bn_bwd_one.dgamma", dgam, r64[1], r32[1], floor=5e-6)
    _check("bn_bwd_one.dbeta", dbet, r64[2], r32[2], floor=5e-6)
    # the three-launch form on the same inputs: same sums up to the summation order
    nblk = lib.dip_bn_bwd_nblk(Hh, Ww, Cc)
    part = torch.full((nblk * 2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_stats(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope, None, Cs,
                                 part.data_ptr(), nblk, st))
    dgam3, dbet3 = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    coef3 = torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_finalize(part.data_ptr(), nblk, Cs, Cc, Hh * Ww, dgam3.data_ptr(), dbet3.data_ptr(), coef3.data_ptr(), st))
    torch.cuda.synchronize()
    scale = float(dgam3.abs().max() + dbet3.abs().max())
    assert torch.allclose(dgam, dgam3, rtol=1e-4, atol=1e-5 * scale) and torch.allclose(dbet, dbet3, rtol=1e-4, atol=1e-5 * scale)
    assert torch.allclose(coef.view(2, Cs)[:, :Cc], coef3.view(2, Cs)[:, :Cc], rtol=1e-4, atol=1e-5 * scale / (Hh * Ww))
    assert torch.all(dy.view(Hh * Ww, Cs)[:, Cc:] == 0), "pad channels must be written as zeros"


def test_bn_bwd_one_crop_window(dev):
    """The gradient source with Concat's crop window (models/common.py:29-37: the skip branch was centre-cropped): zero
    gradient outside the window."""
    lib = N.lib()
    Cc, Hh, Ww, wy, wx, wh, ww = 4, 21, 30, 1, 2, 18, 27
    g = torch.Generator().manual_seed(9)
    y = torch.randn(1, Cc, Hh, Ww, generator=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    Ccat = 12
    Gw = torch.randn(1, Ccat, wh, ww, generator=g)          # the cropped concat gradient; the branch owns channels 4..7
    choff = 4

    def ref(dt):
        yy = y.to(dt).requires_grad_(True)
        ga, be = gamma.to(dt).requires_grad_(True), beta.to(dt).requires_grad_(True)
        u = F.batch_norm(yy, None, None, ga, be, True, 0.1, 1e-5)
        u = torch.maximum(u, 0.2 * u)[:, :, wy:wy + wh, wx:wx + ww]
        (u * Gw[:, choff:choff + Cc].to(dt)).sum().backward()
        return yy.grad, ga.grad, be.grad

    r64, r32 = ref(torch.float64), ref(torch.float32)
    Cs = 4
    state = _state(y, gamma, beta).to(dev).contiguous()
    yb, Gb = H.to_nhwc(y.to(dev)), H.to_nhwc(Gw.to(dev))
    src = N.DipGradSrc(Gb.data_ptr(), 0, 0, Ccat, choff, wy, wx, wh, ww)
    dy = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    dgam, dbet, coef = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_one(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, 0.2, dy.data_ptr(), Cs,
                               dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(), H.stream(dev)), "bn_bwd_one")
    torch.cuda.synchronize()
    _check("bn_bwd_one.window.dy", H.from_nhwc(dy, Cc, Hh, Ww), r64[0], r32[0], floor=5e-6)
    _check("bn_bwd_one.window.dgamma", dgam, r64[1], r32[1], floor=5e-6)
    _check("bn_bwd_one.window.dbeta", dbet, r64[2], r32[2], floor=5e-6)


@pytest.mark.parametrize("ns,nd,Hh,Ww,mode,od", [
    (4, 128, 32, 32, "bilinear", (0, 0)),       # default net, scale 4 -> 3
    (0, 128, 28, 44, "nearest", (0, 0)),        # 'library' net (no skip branch), s4 -> s3
    (4, 16, 25, 39, "bilinear", (0, 0)),        # odd size: the last up-sampled row / column is cropped
    (128, 128, 64, 64, "nearest", (0, 0)),      # kate net: 4096 low-res pixels... (32 x 32 deep branch: 1024)
    (0, 8, 30, 47, "bilinear", (1, 0)),         # centre crop with an offset (a skip-less scale at a non-divisible size)
    (4, 128, 128, 128, "bilinear", (0, 0)),     # 64 x 64 deep branch: 4 pixels per thread, 16 loads each
])
def test_upsample_bwd_one_matches_autograd(dev, ns, nd, Hh, Ww, mode, od):
    lib = N.lib()
    g = torch.Generator().manual_seed(5)
    ody, odx = od
    Hl, Wl = (Hh + ody + 1) // 2 + ody, (Ww + odx + 1) // 2 + odx      # 2 * Hl >= ody + Hh (the crop window fits)
    d = torch.randn(1, nd, Hl, Wl, generator=g) * 1.5 + torch.randn(1, nd, 1, 1, generator=g)
    gamma, beta = torch.rand(nd, generator=g) + 0.5, torch.randn(nd, generator=g) * 0.3
    Gc = torch.randn(1, ns + nd, Hh, Ww, generator=g)
    slope = 0.2

    def ref(dt):
        dd = d.to(dt).requires_grad_(True)
        ga, be = gamma.to(dt).requires_grad_(True), beta.to(dt).requires_grad_(True)
        u = F.batch_norm(dd, None, None, ga, be, True, 0.1, 1e-5)
        u = torch.maximum(u, slope * u)
        up = F.interpolate(u, scale_factor=2, mode=mode)[:, :, ody:ody + Hh, odx:odx + Ww]
        (up * Gc[:, ns:].to(dt)).sum().backward()
        return dd.grad, ga.grad, be.grad

    r64, r32 = ref(torch.float64), ref(torch.float32)
    Cs, Cs_cat = round_up(nd, 4), round_up(ns + nd, 4)
    state = _state(d, gamma, beta).to(dev).contiguous()
    db, Gb = H.to_nhwc(d.to(dev)), H.to_nhwc(Gc.to(dev), Cs_cat)
    dy = torch.full((Hl * Wl * Cs,), float("nan"), device=dev)
    dgam, dbet, coef = torch.zeros(nd, device=dev), torch.zeros(nd, device=dev), torch.zeros(2 * Cs, device=dev)
    m = N.UP_BILINEAR if mode == "bilinear" else N.UP_NEAREST
    N.check(lib.dip_upsample_bwd_one(Gb.data_ptr(), Cs_cat, ns, Hh, Ww, Hl, Wl, ody, odx, m, db.data_ptr(), Cs, nd,
                                     state.data_ptr(), Cs, slope, dy.data_ptr(), Cs, dgam.data_ptr(), dbet.data_ptr(),
                                     coef.data_ptr(), H.stream(dev)), "upsample_bwd_one")
    torch.cuda.synchronize()
    _check("upsample_bwd_one.dy", H.from_nhwc(dy, nd, Hl, Wl), r64[0], r32[0], floor=5e-6)
    _check("upsample_bwd_one.dgamma", dgam, r64[1], r32[1], floor=5e-6)
    _check("upsample_bwd_one.dbeta", dbet, r64[2], r32[2], floor=5e-6)
    # the three-launch form (dip_upsample_bwd_stats_crop + finalize + apply) on the same inputs
    nb = lib.dip_bn_bwd_nblk(Hl, Wl, nd)
    dz = torch.full((Hl * Wl * Cs,), float("nan"), device=dev)
    part = torch.full((nb * 2 * Cs,), float("nan"), device=dev)
    st = H.stream(dev)
    N.check(lib.dip_upsample_bwd_stats_crop(Gb.data_ptr(), Cs_cat, ns, Hh, Ww, Hl, Wl, ody, odx, m, db.data_ptr(), Cs, nd,
                                            state.data_ptr(), Cs, slope, dz.data_ptr(), Cs, part.data_ptr(), nb, st))
    dg3, db3, coef3 = torch.zeros(nd, device=dev), torch.zeros(nd, device=dev), torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_finalize(part.data_ptr(), nb, Cs, nd, Hl * Wl, dg3.data_ptr(), db3.data_ptr(), coef3.data_ptr(), st))
    N.check(lib.dip_bn_bwd_apply(dz.data_ptr(), Cs, db.data_ptr(), Cs, Hl * Wl, nd, state.data_ptr(), Cs, coef3.data_ptr(), st))
    torch.cuda.synchronize()
    scale = float(dz.abs().max())
    assert torch.allclose(dy, dz, rtol=1e-4, atol=2e-5 * scale)


@pytest.mark.parametrize("Cc,Hh,Ww,P,slope", [(128, 28, 44, 2, 0.2), (64, 56, 88, 2, 0.2), (132, 32, 32, 1, 1.0), (4, 64, 64, 0, 0.2),
                                               (16, 19, 23, 1, 0.2)])
def test_bn_bwd_finalise_in_the_apply_prologue(dev, Cc, Hh, Ww, P, slope):
    """dip_bn_bwd_stats + dip_bn_bwd_apply_src_fin (phase 2 in the prologue of phase 3) against fp64 autograd and against
    the three-launch form; dip_bn_bwd_apply_fin (in place on dz) against the same."""
    lib = N.lib()
    g = torch.Generator().manual_seed(4)
    y = torch.randn(1, Cc, Hh, Ww, generator=g) * 2.0 + torch.randn(1, Cc, 1, 1, generator=g) * 3.0
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
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
    Cs = round_up(Cc, 4)
    state = _state(y, gamma, beta).to(dev).contiguous()
    st = H.stream(dev)
    yb, Gb = H.to_nhwc(y.to(dev)), H.to_nhwc(G.to(dev))
    src = N.DipGradSrc(Gb.data_ptr(), P, 1 if P else 0, Cs, 0)
    nblk = lib.dip_bn_bwd_nblk(Hh, Ww, Cc)
    assert lib.dip_bn_bwd_fin_rows_ok(nblk, Cc) == 1, nblk
    part = torch.full((nblk * 2 * Cs,), float("nan"), device=dev)
    dz = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_stats(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope, dz.data_ptr(), Cs,
                                 part.data_ptr(), nblk, st))
    dy = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    dgam, dbet = torch.full((Cc,), float("nan"), device=dev), torch.full((Cc,), float("nan"), device=dev)
    coef = torch.full((2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_apply_src_fin(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope, part.data_ptr(),
                                         nblk, dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(), dy.data_ptr(), Cs, st), "apply_src_fin")
    torch.cuda.synchronize()
    _check("bn_bwd_fin.dy", H.from_nhwc(dy, Cc, Hh, Ww), r64[0], r32[0], floor=5e-6)
    _check("bn_bwd_fin.dgamma", dgam, r64[1], r32[1], floor=5e-6)
    _check("bn_bwd_fin.dbeta", dbet, r64[2], r32[2], floor=5e-6)
    # three launches on the same partial rows: the same fp64 sums up to their order -> coefficients within an ulp or two
    dg3, db3, coef3 = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_finalize(part.data_ptr(), nblk, Cs, Cc, Hh * Ww, dg3.data_ptr(), db3.data_ptr(), coef3.data_ptr(), st))
    dy3 = torch.full((Hh * Ww * Cs,), float("nan"), device=dev)
    N.check(lib.dip_bn_bwd_apply_src(C.byref(src), yb.data_ptr(), Hh, Ww, Cs, Cc, state.data_ptr(), Cs, slope, coef3.data_ptr(),
                                     dy3.data_ptr(), Cs, st))
    torch.cuda.synchronize()
    assert torch.allclose(coef.view(2, Cs)[:, :Cc], coef3.view(2, Cs)[:, :Cc], rtol=3e-7, atol=0)
    assert torch.allclose(dgam, dg3, rtol=3e-7, atol=0) and torch.allclose(dbet, db3, rtol=3e-7, atol=0)
    assert torch.allclose(dy, dy3, rtol=1e-5, atol=1e-6 * float(dy3.abs().max()))
    # in place on dz (the form behind dip_upsample_bwd_stats)
    dgam2, dbet2, coef2 = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(2 * Cs, device=dev)
    N.check(lib.dip_bn_bwd_apply_fin(dz.data_ptr(), Cs, yb.data_ptr(), Cs, Hh * Ww, Cc, state.data_ptr(), Cs, part.data_ptr(), nblk,
                                     dgam2.data_ptr(), dbet2.data_ptr(), coef2.data_ptr(), st), "apply_fin")
    torch.cuda.synchronize()
    assert torch.equal(coef2, coef) and torch.equal(dgam2, dgam) and torch.equal(dbet2, dbet)
    assert torch.allclose(dz, dy, rtol=1e-5, atol=1e-6 * float(dy.abs().max()))
