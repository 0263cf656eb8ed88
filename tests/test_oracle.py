"""CPU suite: the oracle (oracle/dip_oracle.py) against the committed golden vectors that the REAL
reference produced (oracle/make_golden.py), plus -- when the reference checkout is present (build
container only) -- the bitwise check against the reference itself."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
import dip_oracle as O
from test_net_gpu import NETS


def _spec(cfg):
    kw = cfg["kw"]
    return O.SkipSpec(cfg["args"][0], cfg["args"][1], kw["num_channels_down"], kw["num_channels_up"],
                      kw["num_channels_skip"], kw.get("filter_size_down", 3), kw.get("filter_size_up", 3),
                      kw.get("filter_skip_size", 1), True, True, kw.get("pad", "zero"),
                      kw.get("upsample_mode", "nearest"), kw.get("need1x1_up", True),
                      kw.get("downsample_mode", "stride"), kw.get("act_fun", "LeakyReLU"))


@pytest.mark.parametrize("name", list(NETS))
def test_oracle_reproduces_reference_vectors(name):
    torch.set_num_threads(1)
    gold = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    spec = _spec(NETS[name])
    sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
    learn = {k: v for k, v in sd.items() if k in O.param_shapes(spec)}
    assert set(learn) == set(O.param_shapes(spec))
    onet = O.OracleNet(spec, learn)
    z, target, mask = (torch.from_numpy(gold[k]) for k in ("z", "target", "mask"))
    out = onet(z)
    loss = torch.nn.functional.mse_loss(out * mask, target * mask)
    loss.backward()
    assert torch.equal(out.detach(), torch.from_numpy(gold["out"]))
    assert loss.item() == float(gold["loss"])
    for k, p in zip(onet.names, onet.params):
        assert torch.equal(p.grad, torch.from_numpy(gold["grad/" + k])), k
    # optimize('adam') trajectories: 1 and 3 iterations
    mse = torch.nn.MSELoss()
    for nsteps in (1, 3):
        o2 = O.OracleNet(spec, learn)

        def closure():
            l = mse(o2(z) * mask, target * mask)
            l.backward()
            return l

        O.optimize_adam(o2.params, closure, 0.01, nsteps)
        for k, p in zip(o2.names, o2.params):
            assert torch.equal(p.detach(), torch.from_numpy(gold[f"adam{nsteps}/" + k])), (nsteps, k)
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def test_oracle_downsampler_and_noise_vectors():
    gold = np.load(os.path.join(GOLDEN, "downsampler.npz"))
    for factor in (4, 2, 8):
        tag = f"lanczos2_f{factor}"
        assert np.array_equal(O.lanczos_kernel(factor, 0.5, 4 * factor + 1, 2), gold[tag + "/kernel"])
        x = torch.from_numpy(gold[tag + "/x"]).requires_grad_(True)
        y = O.downsampler_forward(x, factor, "lanczos2", 0.5, True)
        (y * torch.from_numpy(gold[tag + "/gy"])).sum().backward()
        assert torch.equal(y.detach(), torch.from_numpy(gold[tag + "/y"]))
        assert torch.allclose(x.grad, torch.from_numpy(gold[tag + "/gx"]), rtol=0, atol=1e-6)
    gn = np.load(os.path.join(GOLDEN, "get_noise.npz"))
    torch.manual_seed(0)
    assert np.array_equal(O.get_noise(32, "noise", (16, 24)).numpy(), gn["u_s0_32x16x24"])
    torch.manual_seed(7)
    assert np.array_equal(O.get_noise(3, "noise", 8, noise_type="n", var=0.5).numpy(), gn["n_s7_3x8x8"])
    assert np.array_equal(O.get_noise(2, "meshgrid", (8, 12)).numpy(), gn["mesh_8x12"])


def test_oracle_default_net_digest():
    """Full default net built by the PRODUCT's skip() under manual_seed(0) + oracle numerics ==
    digest recorded from the real reference (parameter RNG order, key names, fwd/bwd)."""
    from models import get_net
    from utils.common_utils import get_noise
    dg = json.load(open(os.path.join(GOLDEN, "default64_digest.json")))
    torch.manual_seed(0)
    net = get_net(32, "skip", "reflection", skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                  upsample_mode="bilinear")
    z = get_noise(32, "noise", (64, 64))
    assert list(net.state_dict().keys()) == dg["keys"]
    assert sum(p.numel() for p in net.parameters()) == dg["n_params"] == 2217831
    for k, p in net.named_parameters():
        assert abs(p.detach().double().sum().item() - dg["params"][k]["sum"]) <= 1e-9 * (1 + dg["params"][k]["abssum"])
    sd = {k: v.detach() for k, v in net.state_dict().items() if k in O.param_shapes(O.default_spec())}
    onet = O.OracleNet(O.default_spec(), sd)
    np.random.seed(0)
    target = torch.from_numpy(np.random.rand(1, 3, 64, 64).astype(np.float32))
    out = onet(z)
    loss = torch.nn.functional.mse_loss(out, target)
    assert abs(out.detach().double().sum().item() - dg["out"]["sum"]) <= 1e-6 * dg["out"]["abssum"]
    assert abs(loss.item() - dg["loss"]) <= 1e-6 * dg["loss"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference checkout not mounted")
def test_oracle_bitwise_equal_to_reference():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "verify_against_reference.py")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "oracle pinned against the reference" in r.stdout


def test_closure_bookkeeping_restates_the_notebook_rule():
    """oracle.ClosureBookkeeping (reference denoising.ipynb:214-248): EMA, PSNRs, and back-tracking on
    the iterations with i % show_every != 0 only."""
    rng = np.random.RandomState(0)
    gt = rng.rand(3, 8, 8).astype(np.float32)
    noisy = np.clip(gt + rng.normal(scale=0.1, size=gt.shape), 0, 1).astype(np.float32)
    book = O.ClosureBookkeeping(noisy, gt, exp_weight=0.5, show_every=2)
    params = [torch.zeros(3)]
    outs, fell = [], []
    for it, s in enumerate([0.05, 0.05, 0.9, 0.9, 0.05]):
        out = torch.from_numpy(np.clip(gt + rng.normal(scale=s, size=gt.shape), 0, 1).astype(np.float32))[None]
        params[0].add_(1.0)                       # the "optimizer step" of this iteration
        r = book.step(out, params)
        outs.append(out)
        fell.append(r["fell_back"])
        assert r["psrn_gt"] == pytest.approx(O.psnr(gt, out.numpy()[0]))
    # iteration 0 and 2 are multiples of show_every: never checked; iteration 1 checkpoints (params = 2);
    # iteration 3 sees a > 5 dB drop against iteration 1 and restores the checkpoint
    assert fell == [False, False, False, True, False]
    # (restored to 2 in iteration 3, then stepped once more in iteration 4)
    assert torch.equal(params[0], torch.full((3,), 3.0))
    ema = outs[0]
    for o in outs[1:]:
        ema = ema * 0.5 + o * 0.5
    assert torch.allclose(book.out_avg, ema)


def test_parity_mask_report_and_check_logic():
    """tests/parity.py (the checker of the GPU parity tests) on synthetic data: branch mismatches are counted per
    BatchNorm, their distance from the kink is measured against rms(z), and check() refuses both a mismatch far
    from the kink and a gradient tensor beyond the plain rel-L2 bound."""
    import parity as PT
    torch.manual_seed(0)
    z = torch.randn(1, 8, 64, 64)
    z[0, 0, 0, 0], z[0, 3, 5, 7] = 3e-7, -2e-7
    m = z > 0
    m[0, 0, 0, 0], m[0, 3, 5, 7] = False, True                  # two flips inside the roundoff band
    rep = PT.mask_report({"bn": m}, {"bn": z})
    assert rep["n"] == 2 and rep["numel"] == z.numel() and abs(rep["frac"] - 2 / z.numel()) < 1e-12 and rep["zrel"] < 1e-6
    zr = torch.randn(1, 8, 512, 512)
    zr[0, 0, 0, 0] = 3e-7
    mr = zr > 0
    mr[0, 0, 0, 0] = False
    ok = {"worst": 0.5, "worst_zero": 0.1, "worst_rel": 2e-5, "worst_rel_key": "w", "worst_key": "w", "worst_unmasked": 1.0,
          "worst_unmasked_key": "w", "n_zero": 0, "worst_rel_ref": 1e-5, "worst_rel_excess": 0.2}
    PT.check(ok, PT.mask_report({"bn": mr}, {"bn": zr}))
    mr[0, 1, 0, 0] = ~mr[0, 1, 0, 0]                            # a flip of an element that is NOT near zero
    with pytest.raises(AssertionError):
        PT.check(ok, PT.mask_report({"bn": mr}, {"bn": zr}))
    with pytest.raises(AssertionError):
        PT.check(dict(ok, worst_rel=3e-4, worst_rel_excess=3.0))
    # oracle_grads hands the pre-activations out
    spec = O.SkipSpec(4, 3, [8, 8], [8, 8], [4, 4], pad="reflection", upsample_mode="bilinear")
    sd = {k: torch.randn(s) * 0.2 for k, s in O.param_shapes(spec).items()}
    zrec = {}
    PT.oracle_grads(spec, sd, torch.rand(1, 4, 16, 16), lambda o, dt: o.pow(2).mean(), torch.float64, zrec=zrec)
    keys, _ = O.scale_keys(spec)
    assert set(zrec) == {b for k in keys for b in (k.skip_bn, k.down_a_bn, k.down_b_bn, k.up_bn, k.up1_bn) if b}
