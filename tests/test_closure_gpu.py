"""Closure-side device helpers on a real MI355X: the fused loss head (utils.loss_head.MSEHead), the
Philox reg-noise (utils.reg_noise.RegNoise), hipGraph capture of one iteration and of a group of
independent fits (dip_optim.GraphedIteration), the device-side Adam step count, torch's
accumulate-on-second-backward semantics at the autograd boundary, and optimize('LBFGS')."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402
import dip_oracle as O  # noqa: E402
import hipops as H  # noqa: E402


def _small_net(seed=0, nout=3):
    from models.skip import skip
    torch.manual_seed(seed)
    return skip(8, nout, num_channels_down=[16, 32, 32], num_channels_up=[16, 32, 32], num_channels_skip=[4, 4, 4],
                upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")


def _same_grads(got, ref, spec):
    """Two HIP evaluations of the same gradient (fused vs unfused head): relative 2e-5 per tensor plus the
    fp32 roundoff floor of back-propagation, eps * the largest gradient norm (the tiny BatchNorm-gamma
    gradients of the deep scales are sums of O(gmax) terms); the analytically-zero tensors
    (tests/parity.py) hold only such roundoff in both."""
    import parity as PT
    zero = PT.zero_grad_keys(spec)
    gmax = max(v.double().norm().item() for k, v in ref.items() if k not in zero)
    for k, g in got.items():
        if k in zero:
            assert g.double().norm().item() <= 4 * ref[k].double().norm().item() + 1e-6 * gmax, k
        else:
            e = (g.double() - ref[k].double()).norm().item()
            assert e <= 2e-5 * ref[k].double().norm().item() + 1e-7 * gmax, (k, e)


def _spec_small(nout=3):
    return O.SkipSpec(8, nout, [16, 32, 32], [16, 32, 32], [4, 4, 4], pad="reflection", upsample_mode="bilinear")


@pytest.mark.parametrize("mask_c,nout,hw", [(0, 3, (64, 96)), (1, 3, (64, 64)), (3, 3, (32, 96)), (1, 1, (64, 64))])
def test_loss_head_matches_unfused_and_oracle(dev, mask_c, nout, hw):
    """MSEHead (1x1 conv + sigmoid + mask + MSE in one launch, analytic backward) against the plain
    `out = net(x); mse(out*mask, img*mask)` spelling on the same weights, and against the CPU oracle
    (inpainting.ipynb:310 / denoising.ipynb:219 of the reference)."""
    from utils.loss_head import MSEHead
    net = _small_net(3, nout)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()
          if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    torch.manual_seed(4)
    z = torch.rand(1, 8, *hw) * 0.1
    img = torch.rand(1, nout, *hw)
    mask = None if mask_c == 0 else (torch.rand(1, mask_c, *hw) > 0.4).float()

    def ref_loss(o, dt):
        if mask is None:
            return F.mse_loss(o, img.to(dt))
        return F.mse_loss(o * mask.to(dt), img.to(dt) * mask.to(dt))

    onet = O.OracleNet(_spec_small(nout), {k: v.double() for k, v in sd.items()})
    lo = ref_loss(onet(z.double()), torch.float64)
    lo.backward()
    g64 = {k: p.grad for k, p in zip(onet.names, onet.params)}

    net = net.to(dev)
    zg, ig = z.to(dev), img.to(dev)
    mg = None if mask is None else mask.to(dev)
    # unfused
    out_u = net(zg)
    loss_u = F.mse_loss(out_u, ig) if mg is None else F.mse_loss(out_u * mg, ig * mg)
    loss_u.backward()
    torch.cuda.synchronize()
    gu = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    # fused (BatchNorm running stats differ by one more update; they do not enter the train-mode forward)
    head = MSEHead(net, ig, mg)
    loss_f, out_f = head(zg)
    assert not out_f.requires_grad and loss_f.requires_grad
    loss_f.backward()
    torch.cuda.synchronize()
    assert abs(loss_f.item() - loss_u.item()) <= 2e-6 * abs(loss_u.item())
    assert abs(loss_f.item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert (out_f - out_u.detach()).abs().max().item() <= 2e-6
    _same_grads({k: p.grad for k, p in net.named_parameters()}, gu, _spec_small(nout))
    # scaling of the upstream gradient reaches the kernel (loss * 3 -> gradients * 3)
    for p in net.parameters():
        p.grad = None
    l3, _ = head(zg)
    (l3 * 3.0).backward()
    torch.cuda.synchronize()
    k0 = "1.1.1.1.weight"
    ref = dict(net.named_parameters())[k0].grad.double()
    assert (ref - 3.0 * gu[k0].double()).norm().item() <= 1e-4 * ref.norm().item()
    # deterministic: the last-arriving block sums the per-block partials in a fixed order
    la, _ = head(zg)
    lb, _ = head(zg)
    torch.cuda.synchronize()
    assert la.item() == lb.item()


def test_loss_head_fullsize_512(dev):
    """The fused head at the headline size: 128 -> 3 channels, 512x512, against torch ops on the GPU
    output of the unfused path."""
    from models import get_net
    from utils.loss_head import MSEHead
    torch.manual_seed(0)
    net = get_net(32, "skip", "reflection", skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                  upsample_mode="bilinear").to(dev)
    z = (torch.rand(1, 32, 512, 512) * 0.1).to(dev)
    img = torch.rand(1, 3, 512, 512).to(dev)
    out_u = net(z)
    loss_u = F.mse_loss(out_u, img)
    loss_u.backward()
    gu = [p.grad.detach().clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    loss_f, out_f = MSEHead(net, img)(z)
    loss_f.backward()
    torch.cuda.synchronize()
    ref64 = ((out_u.detach().double().cpu() - img.double().cpu()) ** 2).mean().item()
    assert abs(loss_f.item() - ref64) <= 2e-6 * ref64
    assert (out_f - out_u.detach()).abs().max().item() <= 2e-6
    _same_grads({k: p.grad for k, p in net.named_parameters()}, dict(zip([k for k, _ in net.named_parameters()], gu)),
                O.default_spec())


def test_reg_noise_stream(dev):
    from utils.reg_noise import RegNoise
    z = (torch.rand(1, 32, 128, 128) * 0.1).to(dev)
    r = RegNoise(z, 1. / 30., seed=7)
    a = r().clone()
    b = r().clone()
    r2 = RegNoise(z, 1. / 30., seed=7)
    a2 = r2().clone()
    torch.cuda.synchronize()
    assert torch.equal(a, a2) and not torch.equal(a, b)          # same seed -> same stream; calls advance it
    e = ((torch.cat([a, b]) - z) * 30.).double().cpu()
    assert abs(e.mean().item()) < 5e-3 and abs(e.std().item() - 1) < 5e-3
    assert abs((e ** 4).mean().item() - 3) < 0.1
    assert int(r.offset.item()) == 2 * (z.numel() // 4)
    assert RegNoise(z, 0.0)() is not None and torch.equal(RegNoise(z, 0.0)(), z)


def _fused_fit(net, z, img, dev, seed=5):
    from utils.common_utils import get_params
    from utils.loss_head import MSEHead
    from utils.reg_noise import RegNoise
    from dip_optim import FusedAdam
    reg = RegNoise(z, 1. / 30., seed=seed)
    head = MSEHead(net, img)
    st = {"avg": torch.zeros_like(img), "loss": torch.zeros((), device=dev)}

    def closure():
        loss, out = head(reg())
        st["avg"].mul_(0.99).add_(out, alpha=0.01)
        loss.backward()
        st["loss"].copy_(loss.detach())
        return loss

    return FusedAdam(get_params("net", net, z), lr=0.01), closure, st


def test_graphed_iteration_equals_eager(dev):
    """One hipGraph replay == one eager iteration, bit for bit (same kernels, same order, device-side
    step count and Philox offset): 12 iterations eager vs 3 eager + capture + 9 replays."""
    from dip_optim import GraphedIteration
    hw = (64, 96)
    z = (torch.rand(1, 8, *hw) * 0.1).to(dev)
    img = torch.rand(1, 3, *hw).to(dev)
    net_a = _small_net(1).to(dev)
    net_b = copy.deepcopy(net_a)
    opt_a, clo_a, st_a = _fused_fit(net_a, z, img, dev)
    for _ in range(12):
        opt_a.zero_grad()
        clo_a()
        opt_a.step()
    opt_b, clo_b, st_b = _fused_fit(net_b, z, img, dev)
    it = GraphedIteration(opt_b, clo_b, warmup=3)
    it.run(9)
    torch.cuda.synchronize()
    assert it.iterations == 12 and opt_b.device_step_count() == 12
    assert st_a["loss"].item() == st_b["loss"].item()
    for (k, pa), pb in zip(net_a.named_parameters(), net_b.parameters()):
        assert torch.equal(pa, pb), k
    assert torch.equal(st_a["avg"], st_b["avg"])
    for (k, ba), bb in zip(net_a.named_buffers(), net_b.buffers()):
        assert torch.equal(ba, bb), k                                   # BatchNorm running statistics too


def test_grouped_instances_in_one_graph(dev):
    """Grouped multi-instance execution: three independent fits captured as concurrent branches of ONE
    hipGraph give exactly the parameters each fit reaches on its own."""
    from dip_optim import GraphedIteration
    hw = (32, 64)
    fits, solo = [], []
    for k in range(3):
        z = (torch.rand(1, 8, *hw) * 0.1).to(dev)
        img = torch.rand(1, 3, *hw).to(dev)
        net = _small_net(10 + k).to(dev)
        ref = copy.deepcopy(net)
        fits.append((net,) + _fused_fit(net, z, img, dev, seed=k))
        solo.append((ref,) + _fused_fit(ref, z, img, dev, seed=k))
    for ref, opt, clo, st in solo:
        for _ in range(8):
            opt.zero_grad()
            clo()
            opt.step()
    g = GraphedIteration.group([(opt, clo) for _, opt, clo, _ in fits], warmup=3)
    assert len(g.members) == 3
    g.run(5)
    torch.cuda.synchronize()
    for (net, _, _, st), (ref, _, _, st_r) in zip(fits, solo):
        assert st["loss"].item() == st_r["loss"].item()
        for (k, pa), pb in zip(net.named_parameters(), ref.parameters()):
            assert torch.equal(pa, pb), k


def test_grouped_instances_single_graph_subprocess(dev):
    """The opt-in single-graph form of the group (all fits as concurrent branches of one capture), run
    in a subprocess: cross-stream captures of this size have crashed the HIP runtime, and a crash must
    not take the test session with it.  Skips (with the reason) if the runtime cannot do it."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, copy, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import conftest, test_closure_gpu as T
from dip_optim import GraphedIteration
dev = torch.device('cuda:0')
fits, solo = [], []
for k in range(2):
    z = (torch.rand(1, 8, 32, 64) * 0.1).to(dev); img = torch.rand(1, 3, 32, 64).to(dev)
    net = T._small_net(20 + k).to(dev); ref = copy.deepcopy(net)
    fits.append((net,) + T._fused_fit(net, z, img, dev, seed=k)); solo.append((ref,) + T._fused_fit(ref, z, img, dev, seed=k))
for ref, opt, clo, st in solo:
    for _ in range(6):
        opt.zero_grad(); clo(); opt.step()
g = GraphedIteration.group([(o, c) for _, o, c, _ in fits], warmup=3, single_graph=True)
g.run(3); torch.cuda.synchronize()
for (net, _, _, st), (ref, _, _, sr) in zip(fits, solo):
    assert st['loss'].item() == sr['loss'].item()
    assert all(torch.equal(a, b) for a, b in zip(net.parameters(), ref.parameters()))
print('SINGLE_GRAPH_OK')
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    if "SINGLE_GRAPH_OK" not in r.stdout:
        pytest.skip(f"single-graph group capture not available on this runtime (rc={r.returncode}): "
                    + (r.stderr.strip().splitlines() or ["?"])[-1][:200])


def test_optimize_graph_flag(dev):
    from utils.common_utils import get_params, optimize
    hw = (32, 32)
    z = (torch.rand(1, 8, *hw) * 0.1).to(dev)
    img = torch.rand(1, 3, *hw).to(dev)
    net = _small_net(2).to(dev)
    _, clo, st = _fused_fit(net, z, img, dev)
    optimize("adam", get_params("net", net, z), clo, 0.01, 40, graph=True)
    l40 = st["loss"].item()
    net2 = _small_net(2).to(dev)
    _, clo2, st2 = _fused_fit(net2, z, img, dev)
    optimize("adam", get_params("net", net2, z), clo2, 0.01, 40)
    assert l40 == st2["loss"].item()


def test_adam_device_state_equals_host_scalars(dev):
    """dip_adam_tick + dip_adam_step_dev (step count and bias corrections in device memory) ==
    dip_adam_step (host scalars), bit for bit over 6 steps."""
    lib = N.lib()
    g = torch.Generator().manual_seed(3)
    n = 50001
    p0 = torch.randn(n, generator=g)
    pa, pb = p0.to(dev), p0.to(dev)
    ma, va, mb, vb = (torch.zeros(n, device=dev) for _ in range(4))
    st = torch.zeros(16, dtype=torch.uint8, device=dev)
    s = H.stream(dev)
    for step in range(1, 7):
        gr = (torch.randn(n, generator=g) * (10.0 ** torch.randint(-6, 1, (n,), generator=g).float())).to(dev)
        N.check(lib.dip_adam_step(pa.data_ptr(), gr.data_ptr(), ma.data_ptr(), va.data_ptr(), n, 0.01, 0.9, 0.999, 1e-8,
                                  step, s))
        N.check(lib.dip_adam_tick(st.data_ptr(), 0.01, 0.9, 0.999, s))
        N.check(lib.dip_adam_step_dev(pb.data_ptr(), gr.data_ptr(), mb.data_ptr(), vb.data_ptr(), n, 0.9, 0.999, 1e-8,
                                      st.data_ptr(), s))
    torch.cuda.synchronize()
    assert int(st.view(torch.int64)[0].item()) == 6
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)


def test_second_backward_accumulates_like_torch(dev):
    """Two closure evaluations without zero_grad(): .grad holds the SUM (torch semantics); and
    zero_grad(set_to_none=False) followed by a backward gives the plain gradient."""
    from dip_optim import FusedAdam
    hw = (32, 48)
    net = _small_net(6).to(dev)
    z1 = (torch.rand(1, 8, *hw) * 0.1).to(dev)
    z2 = (torch.rand(1, 8, *hw) * 0.1).to(dev)
    img = torch.rand(1, 3, *hw).to(dev)

    def grads(zz):
        for p in net.parameters():
            p.grad = None
        F.mse_loss(net(zz), img).backward()
        return [p.grad.detach().clone() for p in net.parameters()]

    g1, g2 = grads(z1), grads(z2)
    for p in net.parameters():
        p.grad = None
    F.mse_loss(net(z1), img).backward()
    F.mse_loss(net(z2), img).backward()            # no zero_grad in between
    torch.cuda.synchronize()
    for p, a, b in zip(net.parameters(), g1, g2):
        assert torch.allclose(p.grad, a + b, rtol=1e-6, atol=1e-12)
    opt = FusedAdam(list(net.parameters()), lr=0.01)
    opt.zero_grad(set_to_none=False)
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0 for p in net.parameters())
    F.mse_loss(net(z1), img).backward()
    torch.cuda.synchronize()
    for p, a in zip(net.parameters(), g1):
        assert torch.equal(p.grad, a)
    # a stale forward still raises (one forward, one backward per closure evaluation)
    o1 = net(z1)
    o2 = net(z2)
    with pytest.raises(RuntimeError, match="stale"):
        F.mse_loss(o1, img).backward()
    del o2


def test_device_guard_other_current_device(dev):
    """Launches go to the net's device even if another device is current (single-GPU box: only the
    guard's code path is exercised)."""
    net = _small_net(7).to(dev)
    z = (torch.rand(1, 8, 32, 32) * 0.1).to(dev)
    with torch.cuda.device(dev):
        out = net(z)
        out.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_lbfgs_on_the_arena(dev):
    """optimize('LBFGS') (reference utils/common_utils.py:208-221): ArenaLBFGS on the HIP net against
    torch.optim.LBFGS on the CPU oracle from the same start -- the first closure evaluations agree,
    and the whole call (100 Adam steps + LBFGS) lowers the loss."""
    from dip_optim import ArenaLBFGS
    from utils.common_utils import get_params, optimize
    hw = (32, 32)
    torch.manual_seed(9)
    z = torch.rand(1, 8, *hw) * 0.1
    img = torch.rand(1, 3, *hw)
    net = _small_net(8)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()
          if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    onet = O.OracleNet(_spec_small(), sd)
    ref_hist = []
    topt = torch.optim.LBFGS(onet.params, max_iter=6, lr=0.05, tolerance_grad=-1, tolerance_change=-1)

    def oclosure():
        topt.zero_grad()
        l = F.mse_loss(onet(z), img)
        l.backward()
        ref_hist.append(l.item())
        return l

    topt.step(oclosure)
    net = net.to(dev)
    zg, ig = z.to(dev), img.to(dev)
    hist = []
    opt = ArenaLBFGS(get_params("net", net, zg), max_iter=6, lr=0.05, tolerance_grad=-1, tolerance_change=-1)

    def closure():
        opt.zero_grad()
        l = F.mse_loss(net(zg), ig)
        l.backward()
        hist.append(l.item())
        return l

    opt.step(closure)
    assert opt._flat is not None                     # the parameters are one arena: no gather/scatter
    assert len(hist) == len(ref_hist)
    for a, b in zip(hist[:4], ref_hist[:4]):
        assert abs(a - b) <= 2e-3 * abs(b), (hist, ref_hist)
    # the optimiser's arithmetic in isolation: ArenaLBFGS and torch.optim.LBFGS driving the SAME (deterministic) HIP net
    # from the same start -- the parameter vectors after max_iter = 8 steps agree to fp32 roundoff
    finals = []
    for kind in ("arena", "torch"):
        n_ = _small_net(8)
        n_.load_state_dict(sd, strict=False)
        n_ = n_.to(dev)
        ps = get_params("net", n_, zg)
        o_ = ArenaLBFGS(ps, max_iter=8, lr=0.05, tolerance_grad=-1, tolerance_change=-1) if kind == "arena" else \
            torch.optim.LBFGS(ps, max_iter=8, lr=0.05, tolerance_grad=-1, tolerance_change=-1)
        losses = []

        def c_():
            o_.zero_grad()
            l = F.mse_loss(n_(zg), ig)
            l.backward()
            losses.append(l.item())
            return l

        o_.step(c_)
        torch.cuda.synchronize()
        finals.append((torch.cat([p.detach().reshape(-1) for p in n_.parameters()]).cpu().double(), losses))
    (pa, la), (pt, lt) = finals
    # (the two flat-vector layouts sum their fp32 dot products in different orders; the two-loop recursion amplifies that:
    # losses agree to 1e-7 for four evaluations and to ~2e-5 afterwards)
    assert len(la) == len(lt) == 8 and all(abs(a - b) <= 2e-4 * abs(b) for a, b in zip(la, lt)), (la, lt)
    rel = (pa - pt).norm().item() / pt.norm().item()
    moved = (pt - torch.cat([v.reshape(-1) for v in sd.values()]).double()).norm().item() / pt.norm().item()
    print(f"ArenaLBFGS vs torch.optim.LBFGS on the HIP net: |dp|/|p| = {rel:.2e} after 8 steps (the steps moved p by {moved:.2e})")
    assert rel <= 5e-2 * moved, (rel, moved)
    # the full optimize('LBFGS') call
    net2 = _small_net(8).to(dev)
    rec = []

    def closure2():
        l = F.mse_loss(net2(zg), ig)
        l.backward()
        rec.append(l.detach())
        return l

    optimize("LBFGS", get_params("net", net2, zg), closure2, 0.05, 10)
    vals = [float(v) for v in rec]
    assert len(vals) >= 100 + 2 and np.isfinite(vals).all() and vals[-1] < vals[0]


def test_downsampler_optimised_as_dense_conv_golden(dev):
    """opt_over='down' (utils/common_utils.py:44-46 of the reference): the Downsampler's dense Conv2d weight and bias
    get gradients and are trained; vectors from the real reference (oracle/make_golden.py: gen_downsampler_dense)."""
    from conftest import GOLDEN
    from models.downsampler import Downsampler
    from utils.common_utils import get_params, optimize
    gold = np.load(os.path.join(GOLDEN, "downsampler_dense.npz"))
    for factor in (4, 2, 8):
        t = f"f{factor}/"
        d = Downsampler(n_planes=3, factor=factor, kernel_type="lanczos2", phase=0.5, preserve_size=True).to(dev)
        assert not d.downsampler_.weight.requires_grad                 # fixed taps until someone asks to optimise them
        sd = d.state_dict()
        sd["downsampler_.weight"] = torch.from_numpy(gold[t + "w"]).to(dev)
        sd["downsampler_.bias"] = torch.from_numpy(gold[t + "b"]).to(dev)
        d.load_state_dict(sd)                                          # not on the channel diagonal any more -> dense path
        x = torch.from_numpy(gold[t + "x"]).to(dev).requires_grad_(True)
        y = d(x)
        assert torch.allclose(y.detach().cpu(), torch.from_numpy(gold[t + "y"]), rtol=1e-5, atol=5e-6)
        params = get_params("down", None, x, d)
        assert all(p.requires_grad for p in params) and len(params) == 2
        y = d(x)
        (y * torch.from_numpy(gold[t + "gy"]).to(dev)).sum().backward()
        for got, key in ((x.grad, "gx"), (d.downsampler_.weight.grad, "dw"), (d.downsampler_.bias.grad, "db")):
            ref = torch.from_numpy(gold[t + key])
            assert torch.allclose(got.cpu(), ref, rtol=2e-5, atol=2e-6 * float(ref.abs().max())), (factor, key)
        # three optimize('adam') steps over the down-sampler alone
        xin = torch.from_numpy(gold[t + "x"]).to(dev)
        target = torch.from_numpy(gold[t + "target"]).to(dev)
        mse = torch.nn.MSELoss()

        def closure():
            l = mse(d(xin), target)
            l.backward()
            return l

        optimize("adam", params, closure, 0.01, 3)
        for p_, key in ((d.downsampler_.weight, "adam3_w"), (d.downsampler_.bias, "adam3_b")):
            ref = torch.from_numpy(gold[t + key])
            err = float((p_.detach().cpu() - ref).abs().max())
            assert err <= 2e-5, (factor, key, err)                     # steps are 1e-2 each


def test_downsampler_fixed_taps_state_dict_roundtrip(dev):
    from models.downsampler import Downsampler
    d = Downsampler(n_planes=3, factor=4, kernel_type="lanczos2", phase=0.5, preserve_size=True).to(dev)
    x = torch.rand(1, 3, 64, 64, device=dev)
    y = d(x)
    d.load_state_dict(copy.deepcopy(d.state_dict()))
    assert not d._nondiag and torch.equal(d(x), y)
    # the dense evaluation of the same taps agrees with the depth-wise kernel to rounding
    d._nondiag = True
    assert torch.allclose(d(x), y, rtol=1e-5, atol=2e-6)
