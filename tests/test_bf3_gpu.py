"""The bf16-matrix-pipe convolution (csrc/conv_bf3.hip: every fp32 operand split exactly into three bf16 terms, the
cross products accumulated in fp32 by v_mfma_f32_32x32x16_bf16) against a torch-CPU fp64 evaluation of nn.ReflectionPad2d +
nn.Conv2d (models/common.py:114-124 of the reference) and its autograd data gradient (terms = 8: without the lo x lo product,
< 2^-32 of a product; terms = 6: the six largest) -- with the SAME per-op criterion as
the fp32-MFMA kernels (error vs fp64 <= 2 x the error of torch's own fp32 CPU kernel): the scheme is fp32-accurate, not a
reduced-precision mode.  Also asserted: its error is no larger than 1.5 x the fp32-MFMA kernel's own on the same inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402
import hipops as H  # noqa: E402
from test_kernels_gpu import _apply_tr, _check, _mk, _ref_conv, REFLECT, ZERO  # noqa: E402

BF3_CASES = [
    # Cin, Cout, pad, H, W, transform   (>= 256 tiles of 8 x 16 pixels)
    (128, 128, REFLECT, 128, 256, True),      # 128 -> 128 encoder / decoder conv
    (132, 128, REFLECT, 128, 256, True),      # decoder conv on the concat: 8 full 16-channel chunks + a 4-channel one
    (32, 128, ZERO, 136, 248, False),         # ragged tiles, zero padding, 2 chunks
    (128, 256, REFLECT, 128, 256, True),      # two 128-column blocks
    (20, 160, REFLECT, 128, 256, False),      # a partial second column block, partial second chunk
]


@pytest.mark.parametrize("terms", [9, 8, 6])
@pytest.mark.parametrize("case", BF3_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf3_forward_and_stats(dev, case, terms):
    Cin, Cout, pad, Hh, Ww, use_tr = case
    full = (Cin, Cout, 3, 1, pad, Hh, Ww, use_tr)
    x, w, b, a, bb = _mk(full)
    slope = 0.2
    ref64 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float64), w, b, 1, pad, torch.float64)
    ref32 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float32), w, b, 1, pad, torch.float32)
    tr = (a.to(dev), bb.to(dev), slope) if use_tr else (None, None, 1.0)
    y, stats = H.conv_bf3(x.to(dev), w.to(dev), b.to(dev), pad, tr, terms=terms)
    _check(f"conv_bf3[{terms}]", y, ref64, ref32)
    y32 = H.conv_fwd(x.to(dev), w.to(dev), b.to(dev), 1, pad, tr)                   # the fp32-MFMA kernel, same inputs
    e3 = (y.cpu().double() - ref64).pow(2).sum().sqrt().item()
    e32 = (y32.cpu().double() - ref64).pow(2).sum().sqrt().item()
    assert e3 <= 1.5 * e32, f"bf16-pipe error {e3:.3e} vs fp32-MFMA error {e32:.3e}"
    st = stats.cpu().double().numpy()
    n = st[:, 0, :Cout]; m = st[:, 1, :Cout]; M2 = st[:, 2, :Cout]
    N_ = n.sum(0)
    mean = (n * m).sum(0) / N_
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / N_
    r = ref64[0].reshape(Cout, -1)
    assert np.allclose(N_, r.shape[1])
    assert np.allclose(mean, r.mean(1).numpy(), rtol=1e-5, atol=1e-5 * float(r.std()))
    assert np.allclose(var, r.var(1, unbiased=False).numpy(), rtol=2e-5)


@pytest.mark.parametrize("terms", [9, 8, 6])
@pytest.mark.parametrize("case", [BF3_CASES[0], BF3_CASES[1], (256, 128, ZERO, 136, 248, False)], ids=lambda c: "x".join(map(str, c)))
def test_conv_bf3_dgrad(dev, case, terms):
    Cin, Cout, pad, Hh, Ww, _ = case
    x, w, b, _, _ = _mk((Cin, Cout, 3, 1, pad, Hh, Ww, False), 1)
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = x.to(dt).requires_grad_(True)
        y = _ref_conv(xx, w, None, 1, pad, dt)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
        (y * dy.to(dt)).sum().backward()
        res[dt] = xx.grad
    gx = H.conv_bf3(dy.to(dev), w.to(dev), None, pad, terms=terms, dgrad_of=(Hh, Ww))
    _check(f"conv_bf3_dgrad[{terms}]", gx, res[torch.float64], res[torch.float32])


def test_eight_products_are_as_accurate_as_nine(dev):
    """terms = 8 leaves out the lo x lo product alone: < 2^-32 of a product, 2^-8 of the rounding error of ONE fp32
    accumulation step.  Its output must sit on the nine-product output far closer than either sits on the fp64 result, and
    its error against fp64 must be the same."""
    Cin, Cout, pad, Hh, Ww = 128, 128, REFLECT, 128, 256
    x, w, b, a, bb = _mk((Cin, Cout, 3, 1, pad, Hh, Ww, True))
    ref64 = _ref_conv(_apply_tr(x, a, bb, 0.2, torch.float64), w, b, 1, pad, torch.float64)
    tr = (a.to(dev), bb.to(dev), 0.2)
    y9, _ = H.conv_bf3(x.to(dev), w.to(dev), b.to(dev), pad, tr, terms=9)
    y8, _ = H.conv_bf3(x.to(dev), w.to(dev), b.to(dev), pad, tr, terms=8)
    y9, y8 = y9.cpu().double(), y8.cpu().double()
    rms = lambda t: t.pow(2).mean().sqrt().item()
    e9, e8, d89 = rms(y9 - ref64), rms(y8 - ref64), rms(y8 - y9)
    # the two outputs differ only where the missing 2^-32 tips one of the ~650 roundings of an accumulator (about one output
    # in 2^8 per step, by one ulp of the running sum): far below the scheme's own distance to fp64, which does not move
    assert d89 <= 0.5 * e9, (d89, e9)
    assert abs(e8 - e9) <= 0.03 * e9, (e8, e9)


K1_CASES = [
    # Cin, Cout, H, W, transform   (1x1 layers from 256 tiles: conv_bf3_k1_kernel, csrc/conv_bf3.hip)
    (128, 128, 256, 256, True),       # s1.up1 of the default net (need1x1_up, models/skip.py:88-91 of the reference)
    (128, 128, 136, 248, False),      # ragged tiles
    (256, 128, 128, 256, True),       # 16 chunks
    (144, 160, 128, 256, False),      # an odd number of chunks (9); a partial second column block
    (16, 128, 200, 176, True),        # a single chunk: prologue + one unit
]


@pytest.fixture
def _k1_on(monkeypatch):
    """The 1x1 form is opt-in (measured slower per iteration: profiles/r06_conv_bf3_1x1_dead_end.txt); the switch is read at every call."""
    monkeypatch.setenv("DIP_CONV_BF3_1X1", "1")


@pytest.mark.parametrize("terms", [9, 8, 6])
@pytest.mark.parametrize("case", K1_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf3_1x1_forward_and_stats(dev, case, terms, _k1_on):
    """The 1x1 form of the bf16-pipe kernel (one unit per 16-channel chunk, pipelined per chunk) against torch-CPU fp64 with the
    per-op criterion, its error held to 1.5 x the fp32-MFMA kernel's (conv1x1_res / the LDS-DMA kernel) on the same inputs,
    and the consumer BatchNorm's partial statistics."""
    Cin, Cout, Hh, Ww, use_tr = case
    full = (Cin, Cout, 1, 1, REFLECT, Hh, Ww, use_tr)
    x, w, b, a, bb = _mk(full)
    slope = 0.2
    ref64 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float64), w, b, 1, REFLECT, torch.float64)
    ref32 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float32), w, b, 1, REFLECT, torch.float32)
    tr = (a.to(dev), bb.to(dev), slope) if use_tr else (None, None, 1.0)
    y, stats = H.conv_bf3(x.to(dev), w.to(dev), b.to(dev), REFLECT, tr, terms=terms)
    _check(f"conv_bf3_1x1[{terms}]", y, ref64, ref32)
    y32 = H.conv_fwd(x.to(dev), w.to(dev), b.to(dev), 1, REFLECT, tr)             # wp3 is not set: an fp32-MFMA kernel
    e3 = (y.cpu().double() - ref64).pow(2).sum().sqrt().item()
    e32 = (y32.cpu().double() - ref64).pow(2).sum().sqrt().item()
    assert e3 <= 1.5 * e32, f"bf16-pipe error {e3:.3e} vs fp32-MFMA error {e32:.3e}"
    st = stats.cpu().double().numpy()
    n = st[:, 0, :Cout]; m = st[:, 1, :Cout]; M2 = st[:, 2, :Cout]
    N_ = n.sum(0)
    mean = (n * m).sum(0) / N_
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / N_
    r = ref64[0].reshape(Cout, -1)
    assert np.allclose(N_, r.shape[1])
    assert np.allclose(mean, r.mean(1).numpy(), rtol=1e-5, atol=1e-5 * float(r.std()))
    assert np.allclose(var, r.var(1, unbiased=False).numpy(), rtol=2e-5)


@pytest.mark.parametrize("terms", [9, 8])
@pytest.mark.parametrize("case", [K1_CASES[0], K1_CASES[1], K1_CASES[3]], ids=lambda c: "x".join(map(str, c)))
def test_conv_bf3_1x1_dgrad(dev, case, terms, _k1_on):
    Cin, Cout, Hh, Ww, _ = case
    x, w, b, _, _ = _mk((Cin, Cout, 1, 1, REFLECT, Hh, Ww, False), 1)
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = x.to(dt).requires_grad_(True)
        y = _ref_conv(xx, w, None, 1, REFLECT, dt)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
        (y * dy.to(dt)).sum().backward()
        res[dt] = xx.grad
    gx = H.conv_bf3(dy.to(dev), w.to(dev), None, REFLECT, terms=terms, dgrad_of=(Hh, Ww))
    _check(f"conv_bf3_1x1_dgrad[{terms}]", gx, res[torch.float64], res[torch.float32])


N64_CASES = [
    # Cin, Cout, pad, H, W, transform   (96..255 tiles of 8 x 16 pixels: the 64-column form, two workgroups per pixel tile)
    (128, 128, REFLECT, 128, 128, True),      # the 128^2 layers of the default net
    (132, 128, REFLECT, 128, 128, True),
    (128, 160, REFLECT, 96, 128, False),      # 96 tiles; 2.5 column blocks of 64
    (32, 128, ZERO, 100, 120, False),         # ragged tiles
]


@pytest.mark.parametrize("case", N64_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_bf3_n64_forward_dgrad(dev, case):
    """conv_bf3_kernel<*, *, 64> (layers with 96..255 tiles: the 128^2 layers) under the criteria of the 128-column form."""
    Cin, Cout, pad, Hh, Ww, use_tr = case
    assert 96 <= N.lib().dip_conv_ntiles(Hh, Ww) < 256
    x, w, b, a, bb = _mk((Cin, Cout, 3, 1, pad, Hh, Ww, use_tr))
    slope = 0.2
    ref64 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float64), w, b, 1, pad, torch.float64)
    ref32 = _ref_conv(_apply_tr(x, a, bb, slope, torch.float32), w, b, 1, pad, torch.float32)
    tr = (a.to(dev), bb.to(dev), slope) if use_tr else (None, None, 1.0)
    y, stats = H.conv_bf3(x.to(dev), w.to(dev), b.to(dev), pad, tr, terms=8)
    _check("conv_bf3_n64", y, ref64, ref32)
    st = stats.cpu().double().numpy()
    n = st[:, 0, :Cout]; m = st[:, 1, :Cout]
    r = ref64[0].reshape(Cout, -1)
    assert np.allclose(n.sum(0), r.shape[1])
    assert np.allclose((n * m).sum(0) / n.sum(0), r.mean(1).numpy(), rtol=1e-5, atol=1e-5 * float(r.std()))
    if Cin < 128:               # (the data gradient has Cin columns: the bf16 pipe takes >= 128)
        return
    res = {}
    for dt in (torch.float64, torch.float32):
        xx = x.to(dt).requires_grad_(True)
        yy = _ref_conv(xx, w, None, 1, pad, dt)
        dy = torch.randn(yy.shape, generator=torch.Generator().manual_seed(7))
        (yy * dy.to(dt)).sum().backward()
        res[dt] = xx.grad
    gx = H.conv_bf3(dy.to(dev), w.to(dev), None, pad, terms=8, dgrad_of=(Hh, Ww))
    _check("conv_bf3_n64_dgrad", gx, res[torch.float64], res[torch.float32])


def test_split_is_exact(dev):
    """w == w1 + w2 + w3 bit for bit for the planes dip_pack_weights_bf3 writes (incl. tiny and huge magnitudes)."""
    g = torch.Generator().manual_seed(0)
    w = torch.randn(32, 16, 3, 3, generator=g) * torch.logspace(-30, 30, 32).view(32, 1, 1, 1)
    buf, fo, do = H.pack_bf3(w.to(dev))
    CoutP = 32
    planes = buf[:9 * 1 * 3 * CoutP * 16].view(9, 1, 3, CoutP, 16).cpu().to(torch.int32) & 0xFFFF
    as_f32 = (planes << 16).to(torch.int32).view(torch.float32)
    total = as_f32[:, :, 0].double() + as_f32[:, :, 1].double() + as_f32[:, :, 2].double()      # [tap][1][n][k]
    ref = w.permute(2, 3, 0, 1).reshape(9, 32, 16).double()                                     # [tap][o][c]
    assert torch.equal(total[:, 0].float().double(), ref) and torch.equal(total[:, 0], ref)


WGRAD_BF3_CASES = [
    # Cin, Cout, pad, H, W, transform   (>= 2048 tiles of 2 x 16 pixels: the ping-pong form; 512 .. 2047: the 4-wave form)
    (128, 128, REFLECT, 128, 128, True),      # the 128^2 layers of the default net (round 5: on the bf16 pipe too)
    (132, 128, REFLECT, 128, 128, True),
    (128, 128, REFLECT, 256, 256, True),
    (132, 128, REFLECT, 256, 256, True),      # four full chunks on the bf16 pipe + the 4-channel tail (dip_conv_wgrad_tail)
    (48, 160, ZERO, 250, 280, False),         # a 16-channel partial chunk, two 128-column blocks, ragged tiles, zero padding
    (100, 128, REFLECT, 256, 256, True),      # THREE full chunks (group 1 of the last workgroup has none) + a 4-channel tail
    (129, 160, ZERO, 250, 281, False),        # round 6, wgrad_tail_kernel: a ONE-channel tail, zero padding, odd width, two column blocks
    (66, 128, REFLECT, 130, 136, True),       # ... a two-channel tail behind two chunks, 64 slabs
]


@pytest.mark.parametrize("terms", [9, 8, 6])
@pytest.mark.parametrize("case", WGRAD_BF3_CASES, ids=lambda c: "x".join(map(str, c)))
def test_wgrad_bf3(dev, case, terms):
    """Weight + bias gradient of the big 3x3 layers through dip_conv_wgrad -> wgrad_bf3_kernel (+ dip_wgrad_reduce), against
    autograd in fp64, with the criterion of the fp32-MFMA kernel (tests/test_kernels_gpu.py::test_conv_wgrad)."""
    import ctypes as C
    Cin, Cout, pad, Hh, Ww, use_tr = case
    full = (Cin, Cout, 3, 1, pad, Hh, Ww, use_tr)
    x, w, b, a, bb = _mk(full, 2)
    slope = 0.2
    res = {}
    for dt in (torch.float64, torch.float32):
        ww = w.to(dt).requires_grad_(True)
        bias = b.to(dt).requires_grad_(True)
        y = _ref_conv(_apply_tr(x, a, bb, slope, dt), ww, bias, 1, pad, dt)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9))
        (y * dy.to(dt)).sum().backward()
        res[dt] = (ww.grad, bias.grad)
    tr = (a.to(dev), bb.to(dev), slope) if use_tr else (None, None, 1.0)
    lib = N.lib()
    N.check(lib.dip_conv_bf3_set_terms(terms))
    try:
        n, g, cb = N.wgrad_plan2(Hh, Ww, Cin, Cout, 3, 1)
        d = N.DipWgradDesc(None, Hh, Ww, N.round_up(Cin, 4), Cin, N.DipTransform(None, None, 1.0), None, Hh, Ww, N.round_up(Cout, 4),
                           Cout, 3, 1, pad, 1, None, None, n, g, cb)
        assert lib.dip_wgrad_bf3_eligible(C.byref(d)) == 1, "descriptor not taken by the bf16-pipe weight gradient"
        dw, db = H.conv_wgrad(x.to(dev), dy.to(dev), 3, 1, pad, tr, nsplit="plan")
        lib.dip_conv_bf3_set_terms(0)
        dw32, db32 = H.conv_wgrad(x.to(dev), dy.to(dev), 3, 1, pad, tr, nsplit="plan")       # the fp32-MFMA kernel
    finally:
        lib.dip_conv_bf3_set_terms(-1)
    _check(f"wgrad_bf3[{terms}].dw", dw, res[torch.float64][0], res[torch.float32][0], floor=4e-6)
    _check(f"wgrad_bf3[{terms}].db", db, res[torch.float64][1], res[torch.float32][1], floor=4e-6)
    e3 = (dw.cpu().double() - res[torch.float64][0]).pow(2).sum().sqrt().item()
    e32 = (dw32.cpu().double() - res[torch.float64][0]).pow(2).sum().sqrt().item()
    assert e3 <= 1.5 * e32, f"bf16-pipe error {e3:.3e} vs fp32-MFMA error {e32:.3e}"


def test_wgrad_bf3_pingpong_is_bit_identical_to_the_round4_kernel(dev, tmp_path):
    """Round 5: wgrad_bf3_kernel as a ping-pong of two wave groups (one 8-wave workgroup per CU, staging of one group under
    the MFMAs of the other) walks the same tiles in the same order with the same MFMA sequence per accumulator as the
    round-4 kernel (DIP_WGRAD_BF3_V1=1): dW and db must agree BIT FOR BIT on every case, including an odd number of
    32-channel chunks (with and without a 4-channel tail behind them), a partial chunk, two column blocks and ragged tiles
    (tests/wgrad_bf3_probe.py)."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wgrad_bf3_probe.py")
    out = {}
    for tag, env in (("new", {}), ("v1", {"DIP_WGRAD_BF3_V1": "1"})):
        o = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, probe, o], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        out[tag] = np.load(o)
    assert sorted(out["new"].files) == sorted(out["v1"].files) and len(out["new"].files) == 20
    for k in out["new"].files:
        a, b = out["new"][k], out["v1"][k]
        assert np.isfinite(a).all(), k
        assert np.array_equal(a, b), (k, float(np.abs(a.astype(np.float64) - b).max()))
