"""Parity at BASELINE.json's own sizes: the HIP path against the CPU oracle run on the GPU box's
host cores (fp32 = the reference's arithmetic, fp64 = the truth), same thresholds as the
small-size tests (tests/parity.py):

  * default net (denoising.ipynb:160-165 of the reference) at 512x512 (M1, the headline config)
    and 256x256 (M0): iteration-1 output / loss / every gradient tensor;
  * super-resolution closure at 512x512 -> Lanczos x4 -> 128x128 (super-resolution.ipynb:141-153,169-186);
  * text-inpainting nets at their notebook sizes: kate (skip=128, nearest, 256->128 3x3 decoder
    convs) at 512x512 and library (depth 6, twelve 5x5 convs, no skips, no 1x1) at 448x704, masked
    MSE (inpainting.ipynb:192-232,310);
  * per-kernel cases at the layer shapes of SURVEY.md App. A that small shapes cannot reach: the
    132->128 3x3 layer at 512x512 forward / data gradient / weight gradient (2048- and 2145-tile
    grids, the XCD remap, the thin4 + LDS-DMA column split), the stride-2 128->128 layer
    512 -> 256, the 128->128 1x1 layer at 512x512 and upsample+concat at 512x512.
"""
import ctypes as C
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402
from dip_native import round_up  # noqa: E402
import dip_oracle as O  # noqa: E402
import hipops as H  # noqa: E402
import parity as PT  # noqa: E402
from test_kernels_gpu import _apply_tr, _check, _ref_conv  # noqa: E402

REFLECT = N.PAD_REFLECT


def _learnable(net):
    return {k: v.detach().clone() for k, v in net.state_dict().items()
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


def _iter1(dev, net, spec, z, loss_cpu, loss_gpu, tag):
    """One forward/backward of `net` on the GPU vs the oracle (fp32, fp64, fp64 on the HIP branch
    pattern); asserts the iteration-1 criterion of tests/parity.py."""
    sd = _learnable(net)
    t0 = time.time()
    out32, l32, g32 = PT.oracle_grads(spec, sd, z, loss_cpu, torch.float32)
    t32 = time.time() - t0
    zrec = {}
    _, _, g64n = PT.oracle_grads(spec, sd, z, loss_cpu, torch.float64, zrec=zrec)
    net = net.to(dev)
    out = net(z.to(dev))
    loss = loss_gpu(out)
    loss.backward()
    torch.cuda.synchronize()
    hmasks = H.lrelu_masks(net, spec)
    _, _, g64 = PT.oracle_grads(spec, sd, z, loss_cpu, torch.float64, hmasks)
    mrep = PT.mask_report(hmasks, zrec) if spec.act_fun == "LeakyReLU" else None
    del zrec
    psnr = PT.psnr(out.detach().cpu().numpy(), out32.numpy())
    rel = abs(loss.item() - l32) / abs(l32)
    rep = PT.grad_report({k: p.grad for k, p in net.named_parameters()}, g64, g32, g64n, PT.zero_grad_keys(spec, sd))
    print(f"{tag}: out PSNR {psnr:.1f} dB, loss rel {rel:.2e}, {PT.fmt(rep)}; oracle fp32 fwd+bwd {t32:.1f} s "
          f"({torch.get_num_threads()} threads)" + ("; " + PT.fmt_masks(mrep) if mrep else ""))
    assert psnr >= 100.0, psnr
    assert rel <= 1e-5, rel
    PT.check(rep, mrep)
    return net, out


@pytest.mark.parametrize("size", [512, 256])
def test_default_net_iter1_vs_oracle(dev, size):
    """BASELINE configs[1] / M1 (512) and M0 (256): default skip-net, denoising closure."""
    from models import get_net
    from utils.common_utils import get_noise
    torch.manual_seed(0)
    net = get_net(32, "skip", "reflection", skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                  upsample_mode="bilinear")
    z = get_noise(32, "noise", (size, size))
    np.random.seed(0)
    target = torch.from_numpy(np.random.rand(1, 3, size, size).astype(np.float32))
    tg = target.to(dev)
    _iter1(dev, net, O.default_spec(), z, lambda o, dt: F.mse_loss(o, target.to(dt)), lambda o: F.mse_loss(o, tg),
           f"default net {size}x{size}")


def test_sr_closure_512_vs_oracle(dev):
    """BASELINE configs[2]: default net at 512x512, loss through Downsampler(3, 4, 'lanczos2',
    phase=0.5, preserve_size=True) against a 128x128 LR image."""
    from models import get_net
    from models.downsampler import Downsampler
    from utils.common_utils import get_noise
    torch.manual_seed(1)
    net = get_net(32, "skip", "reflection", skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                  upsample_mode="bilinear")
    z = get_noise(32, "noise", (512, 512))
    lr = torch.rand(1, 3, 128, 128)
    down = Downsampler(n_planes=3, factor=4, kernel_type="lanczos2", phase=0.5, preserve_size=True).to(dev)
    lrg = lr.to(dev)
    _iter1(dev, net, O.default_spec(), z,
           lambda o, dt: F.mse_loss(O.downsampler_forward(o, 4, "lanczos2", 0.5, True), lr.to(dt)),
           lambda o: F.mse_loss(down(o), lrg), "SR x4 512x512")


def test_kate_net_512_vs_oracle(dev):
    """BASELINE configs[3], kate: skip(32, 3, [128]*5, [128]*5, [128]*5, nearest, reflection),
    masked MSE (inpainting.ipynb:203-209,310)."""
    from models.skip import skip
    from utils.common_utils import get_noise
    torch.manual_seed(2)
    net = skip(32, 3, num_channels_down=[128] * 5, num_channels_up=[128] * 5, num_channels_skip=[128] * 5,
               filter_size_up=3, filter_size_down=3, upsample_mode="nearest", filter_skip_size=1,
               need_sigmoid=True, need_bias=True, pad="reflection", act_fun="LeakyReLU")
    spec = O.SkipSpec(32, 3, [128] * 5, [128] * 5, [128] * 5, pad="reflection", upsample_mode="nearest")
    z = get_noise(32, "noise", (512, 512))
    img = torch.rand(1, 3, 512, 512)
    mask = (torch.rand(1, 1, 512, 512) > 0.3).float().expand(1, 3, 512, 512).contiguous()
    ig, mg = img.to(dev), mask.to(dev)
    _iter1(dev, net, spec, z, lambda o, dt: F.mse_loss(o * mask.to(dt), img.to(dt) * mask.to(dt)),
           lambda o: F.mse_loss(o * mg, ig * mg), "kate net 512x512")


def test_library_net_448x704_vs_oracle(dev):
    """BASELINE configs[3], library: depth 6, 5x5 down filters, no skips, no 1x1, input_depth 1
    (inpainting.ipynb:222-232), 448x704."""
    from models.skip import skip
    from utils.common_utils import get_noise
    torch.manual_seed(3)
    ch = [16, 32, 64, 128, 128, 128]
    net = skip(1, 3, num_channels_down=ch, num_channels_up=ch, num_channels_skip=[0] * 6, filter_size_up=3,
               filter_size_down=5, filter_skip_size=1, upsample_mode="nearest", need1x1_up=False,
               need_sigmoid=True, need_bias=True, pad="reflection", act_fun="LeakyReLU")
    spec = O.SkipSpec(1, 3, ch, ch, [0] * 6, filter_size_down=5, filter_size_up=3, pad="reflection",
                      upsample_mode="nearest", need1x1_up=False)
    z = get_noise(1, "noise", (448, 704))
    img = torch.rand(1, 3, 448, 704)
    mask = (torch.rand(1, 1, 448, 704) > 0.3).float().expand(1, 3, 448, 704).contiguous()
    ig, mg = img.to(dev), mask.to(dev)
    _iter1(dev, net, spec, z, lambda o, dt: F.mse_loss(o * mask.to(dt), img.to(dt) * mask.to(dt)),
           lambda o: F.mse_loss(o * mg, ig * mg), "library net 448x704")


# ------------------------------------------------------------------------------ App. A layer shapes
APPA_CONVS = [
    # Cin, Cout, ks, stride, H, W, transform           
    (132, 128, 3, 1, 512, 512, True),                  # 3.1 up0: 2048 tiles forward, 2145-tile padded data gradient
    (128, 128, 3, 2, 512, 512, True),                  # stride-2 128->128, 512 -> 256
    (128, 128, 1, 1, 512, 512, True),                  # 6.1 up1x1_0
    (132, 128, 3, 1, 256, 256, True),                  # 1.1.7.3.1 up1: 512 / 561 tiles (exactly one round / spill)
]


@pytest.mark.parametrize("case", APPA_CONVS, ids=lambda c: "x".join(map(str, c)))
def test_appa_conv_fwd_dgrad_wgrad(dev, case):
    Cin, Cout, ks, stride, Hh, Ww, use_tr = case
    g = torch.Generator().manual_seed(Cin + ks + stride)
    x = torch.randn(1, Cin, Hh, Ww, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    a = torch.rand(Cin, generator=g) + 0.5
    bb = torch.randn(Cin, generator=g) * 0.3
    slope = 0.2
    res = {}
    dy = None
    for dt in (torch.float64, torch.float32):
        u = _apply_tr(x, a, bb, slope, dt).detach().requires_grad_(True)
        ww = w.to(dt).requires_grad_(True)
        bias = b.to(dt).requires_grad_(True)
        y = _ref_conv(u, ww, bias, stride, REFLECT, dt)
        if dy is None:
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(9))
        (y * dy.to(dt)).sum().backward()
        res[dt] = (y.detach(), u.grad, ww.grad, bias.grad)
    tr = (a.to(dev), bb.to(dev), slope)
    y, stats = H.conv_fwd(x.to(dev), w.to(dev), b.to(dev), stride, REFLECT, tr, want_stats=True)
    _check("conv_fwd", y, res[torch.float64][0], res[torch.float32][0])
    st = stats.cpu().double().numpy()
    n, m, M2 = st[:, 0, :Cout], st[:, 1, :Cout], st[:, 2, :Cout]
    Nn = n.sum(0)
    mean = (n * m).sum(0) / Nn
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / Nn
    r = res[torch.float64][0][0].reshape(Cout, -1)
    assert np.allclose(Nn, r.shape[1])
    assert np.allclose(mean, r.mean(1).numpy(), rtol=1e-5, atol=1e-5 * float(r.std()))
    assert np.allclose(var, r.var(1, unbiased=False).numpy(), rtol=2e-5)
    del y, stats
    gx = H.conv_dgrad(dy.to(dev), w.to(dev), stride, REFLECT, Hh, Ww)
    _check("conv_dgrad", gx, res[torch.float64][1], res[torch.float32][1])
    del gx
    dw, db = H.conv_wgrad(x.to(dev), dy.to(dev), ks, stride, REFLECT, tr, nsplit="plan")
    _check("conv_wgrad.dw", dw, res[torch.float64][2], res[torch.float32][2], floor=4e-6)
    _check("conv_wgrad.db", db, res[torch.float64][3], res[torch.float32][3], floor=4e-6)


def test_appa_upcat_512(dev):
    """Upsample x2 bilinear + Concat at the top scale: [4 | 128] channels, 256x256 -> 512x512, with
    the BatchNorm partial statistics, and the adjoint of the up-sampling."""
    lib = N.lib()
    ns, nd, Hh, Ww = 4, 128, 512, 512
    g = torch.Generator().manual_seed(5)
    Hl, Wl = Hh // 2, Ww // 2
    s = torch.randn(1, ns, Hh, Ww, generator=g)
    d = torch.randn(1, nd, Hl, Wl, generator=g)
    a_s, b_s = torch.rand(ns, generator=g) + 0.5, torch.randn(ns, generator=g) * 0.3
    a_d, b_d = torch.rand(nd, generator=g) + 0.5, torch.randn(nd, generator=g) * 0.3
    Gc = torch.randn(1, ns + nd, Hh, Ww, generator=g)
    slope = 0.2

    def ref(dt):
        dd = d.to(dt).requires_grad_(True)
        up = F.interpolate(_apply_tr(dd, a_d, b_d, slope, dt), scale_factor=2, mode="bilinear")
        cat = torch.cat([_apply_tr(s, a_s, b_s, slope, dt), up], 1)
        (cat * Gc.to(dt)).sum().backward()
        return cat.detach(), dd.grad

    (c64, g64), (c32, g32) = ref(torch.float64), ref(torch.float32)
    st = H.stream(dev)
    sb, db = H.to_nhwc(s.to(dev)), H.to_nhwc(d.to(dev))
    ts, k1 = H.transform(a_s.to(dev), b_s.to(dev), slope)
    td, k2 = H.transform(a_d.to(dev), b_d.to(dev), slope)
    Ccat = ns + nd
    Cs_cat = round_up(Ccat, 4)
    cat = torch.full((Hh * Ww * Cs_cat,), float("nan"), device=dev)
    nblk = lib.dip_upcat_nblk(Hh, Ww, Ccat)
    stats = torch.full((nblk * 3 * Cs_cat,), float("nan"), device=dev)
    desc = N.DipUpcatDesc(sb.data_ptr(), round_up(ns, 4), ns, ts, db.data_ptr(), round_up(nd, 4), nd, td, Hh, Ww,
                          N.UP_BILINEAR, cat.data_ptr(), Cs_cat, stats.data_ptr(), nblk)
    N.check(lib.dip_upcat_fwd(C.byref(desc), st))
    torch.cuda.synchronize()
    _check("upcat_fwd", H.from_nhwc(cat, Ccat, Hh, Ww), c64, c32)
    stn = stats.view(nblk, 3, Cs_cat).cpu().double().numpy()
    n, m, M2 = stn[:, 0, :Ccat], stn[:, 1, :Ccat], stn[:, 2, :Ccat]
    Nt = n.sum(0)
    mean = (n * m).sum(0) / Nt
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / Nt
    r = c64[0].reshape(Ccat, -1)
    assert np.allclose(mean, r.mean(1).numpy(), rtol=1e-5, atol=1e-6)
    assert np.allclose(var, r.var(1, unbiased=False).numpy(), rtol=2e-5)
    Cs = round_up(nd, 4)
    state = torch.zeros(4, Cs)
    state[0, :nd], state[1, :nd], state[2, :nd], state[3, :nd] = 0.1, 1.3, a_d, b_d
    state = state.to(dev).contiguous()
    Gb = H.to_nhwc(Gc.to(dev), Cs_cat)
    nb2 = lib.dip_bn_bwd_nblk(Hl, Wl, nd)
    dz = torch.full((Hl * Wl * Cs,), float("nan"), device=dev)
    part = torch.full((nb2 * 2 * Cs,), float("nan"), device=dev)
    N.check(lib.dip_upsample_bwd_stats(Gb.data_ptr(), Cs_cat, ns, Hh, Ww, N.UP_BILINEAR, db.data_ptr(), Cs, nd,
                                       state.data_ptr(), Cs, slope, dz.data_ptr(), Cs, part.data_ptr(), nb2, st))
    torch.cuda.synchronize()
    _check("upsample_bwd.dz", H.from_nhwc(dz, nd, Hl, Wl) * a_d.view(1, -1, 1, 1).to(dev), g64, g32, floor=5e-6)
