"""FitMonitor (device-side closure bookkeeping) against the oracle's line-by-line
restatement of the reference closure (denoising.ipynb:214-248)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    ge.build()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dip_oracle as O
    from models import get_net
    from utils.fit_monitor import FitMonitor
    return O, get_net, FitMonitor


def test_ema_psnr_and_backtracking_match_reference_closure(env):
    O, get_net, FitMonitor = env
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = get_net(8, 'skip', 'reflection', skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=2,
                  upsample_mode='bilinear').to(dev)
    net(torch.rand(1, 8, 32, 32, device=dev))                # builds the parameter arena
    params = [p for p in net.parameters()]
    ref_params = [p.detach().cpu().clone() for p in params]
    rng = np.random.RandomState(0)
    gt = rng.rand(3, 48, 64).astype(np.float32)
    noisy = np.clip(gt + rng.normal(scale=0.1, size=gt.shape), 0, 1).astype(np.float32)
    mon = FitMonitor(net, torch.from_numpy(noisy)[None].to(dev), torch.from_numpy(gt)[None].to(dev), exp_weight=0.9,
                     show_every=3, capacity=32)
    book = O.ClosureBookkeeping(noisy, gt, exp_weight=0.9, show_every=3)
    # per-iteration output error level: the jumps at iterations 4 and 8 cost > 5 dB of psrn_noisy
    sig = [0.20, 0.15, 0.12, 0.10, 0.60, 0.10, 0.09, 0.08, 0.70, 0.08, 0.07]
    fell = []
    for it, s in enumerate(sig):
        out = np.clip(gt + rng.normal(scale=s, size=gt.shape), 0, 1).astype(np.float32)
        # an "optimizer step": every parameter moves, identically in both arms
        with torch.no_grad():
            for p, q in zip(params, ref_params):
                p.add_(0.01 * (it + 1))
                q.add_(0.01 * (it + 1))
        loss = torch.tensor(float(it) + 0.5, device=dev)
        mon.update(torch.from_numpy(out)[None].to(dev), loss)
        r = book.step(torch.from_numpy(out)[None], ref_params)
        fell.append(r["fell_back"])
        last = mon.last()
        assert last["loss"] == pytest.approx(it + 0.5)
        for k in ("psrn_noisy", "psrn_gt", "psrn_gt_sm"):
            assert last[k] == pytest.approx(r[k], abs=2e-4), (it, k)
        assert bool(last["fell_back"]) == r["fell_back"], it
        for p, q in zip(params, ref_params):
            assert torch.equal(p.detach().cpu(), q), f"parameters differ after iteration {it}"
    assert fell[4] and fell[8] and sum(fell) == 2               # the scenario exercised both branches
    assert np.allclose(mon.out_avg.cpu().numpy()[0], book.out_avg.numpy()[0], atol=1e-6)
    hist = mon.history()
    assert hist.shape == (len(sig), 8) and np.all(hist[:, 7] == np.array(fell, dtype=np.float32))


def test_monitor_without_ground_truth_and_without_backtracking(env):
    O, get_net, FitMonitor = env
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(1)
    noisy = rng.rand(1, 3, 40, 24).astype(np.float32)
    mon = FitMonitor(None, torch.from_numpy(noisy).to(dev), None, backtracking=False, capacity=4)
    out = rng.rand(1, 3, 40, 24).astype(np.float32)
    mon.update(torch.from_numpy(out).to(dev))
    r = mon.last()
    assert r["psrn_noisy"] == pytest.approx(O.psnr(noisy[0], out[0]), abs=2e-4)
    assert r["psrn_gt"] == 0.0 and r["fell_back"] == 0.0
    with pytest.raises(RuntimeError):
        FitMonitor(torch.nn.Sequential(), torch.from_numpy(noisy).to(dev))       # no arena -> no back-tracking
