"""Grouped multi-instance execution on a real MI355X (csrc/dip_group.h, dip_group.GroupedFits):
B independent fits -- B copies of the reference's skip-net (models/skip.py:45-100) under the notebooks' closure and
optimize('adam') (utils/common_utils.py:223-230) -- through ONE launch list.  The bar is bit-exactness: every instance must
reach exactly the parameters, BatchNorm statistics, loss and output average that the same fit reaches on its own (same plans,
same tile walks, same summation orders), in both forms a kernel family can take: one dispatch for all instances
(gridDim.z x B) and the library-side loop of B solo dispatches."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

import dip_native as N  # noqa: E402

ALL = 0x7fffffff


def _net(kind, seed):
    from models.skip import skip
    torch.manual_seed(seed)
    if kind == "skip3":        # three scales with 4-channel skip branches, bilinear up-sampling (the denoising nets' shape)
        return skip(8, 3, num_channels_down=[16, 32, 32], num_channels_up=[16, 32, 32], num_channels_skip=[4, 4, 4],
                    upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")
    if kind == "library":      # inpainting.ipynb:222-232 in small: no skips, 5x5 down filters, nearest, no 1x1 after up
        ch = [16, 32, 64]
        return skip(2, 3, num_channels_down=ch, num_channels_up=ch, num_channels_skip=[0] * 3, filter_size_up=3,
                    filter_size_down=5, filter_skip_size=1, upsample_mode="nearest", need1x1_up=False, need_sigmoid=True,
                    need_bias=True, pad="reflection", act_fun="LeakyReLU")
    if kind == "wide":         # 128-channel layers: the LDS-DMA kernels, split-K, the 132-column data gradients
        return skip(32, 3, num_channels_down=[128, 128], num_channels_up=[128, 128], num_channels_skip=[4, 4],
                    upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")
    raise ValueError(kind)


def _solo_fit(net, z, img, mask, std, seed, ema, dev):
    """The fit on its own: utils.reg_noise.RegNoise + utils.loss_head.MSEHead + dip_optim.FusedAdam, eager launches."""
    from utils.common_utils import get_params
    from utils.loss_head import MSEHead
    from utils.reg_noise import RegNoise
    from dip_optim import FusedAdam
    reg = RegNoise(z, std, seed=seed)
    head = MSEHead(net, img, mask)
    st = {"avg": torch.zeros_like(img), "loss": torch.zeros((), device=dev), "n": 0}

    def closure():
        loss, out = head(reg())
        if ema:
            if st["n"] == 0:
                st["avg"].copy_(out)                                   # denoising.ipynb:214-215
            else:
                st["avg"].mul_(0.99).add_(out, alpha=1 - 0.99)
        st["n"] += 1
        loss.backward()
        st["loss"].copy_(loss.detach())
        return loss

    opt = FusedAdam(get_params("net", net, z), lr=0.01)
    return opt, closure, st


CASES = [
    # kind, (H, W), B, reg-noise std, mask channels, EMA, iterations eager + replayed
    ("skip3", (32, 64), 3, 1. / 30., 0, True, (3, 5)),        # every layer below 68 x 68: conv_small, ring data gradients
    ("skip3", (36, 52), 2, 1. / 30., 0, True, (2, 3)),        # 9 x 13 at the deepest scale: Concat's centre crops
    ("library", (40, 56), 4, 0.0, 1, False, (3, 4)),          # masked loss, no reg-noise, no skips, 5x5 filters
    ("wide", (256, 256), 2, 0.03, 3, True, (2, 2)),           # bf16-pipe conv / weight gradient, persistent 1x1, LDS-DMA
]
IDS = ["skip3-32x64", "skip3-36x52", "library-40x56", "wide-256"]


def _problem(kind, hw, B, mask_c, dev):
    cin = {"skip3": 8, "library": 2, "wide": 32}[kind]
    g = torch.Generator().manual_seed(99)
    zs = [(torch.rand(1, cin, *hw, generator=g) * 0.1).to(dev) for _ in range(B)]
    ts = [torch.rand(1, 3, *hw, generator=g).to(dev) for _ in range(B)]
    ms = None if not mask_c else [(torch.rand(1, mask_c, *hw, generator=g) > 0.3).float().to(dev) for _ in range(B)]
    return zs, ts, ms


@pytest.fixture
def native_mask():
    lib = N.lib()
    prev = lib.dip_group_native(-1)
    yield lib
    lib.dip_group_native(prev)
    assert lib.dip_group_size() == 1


@pytest.mark.parametrize("mask", [ALL, 0], ids=["one-dispatch", "host-loop"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_grouped_fits_bitwise_equal_solo(dev, native_mask, case, mask):
    from dip_group import GroupedFits
    from dip_optim import GraphedIteration
    kind, hw, B, std, mask_c, ema, (n_eager, n_graph) = case
    zs, ts, ms = _problem(kind, hw, B, mask_c, dev)
    nets = [_net(kind, 10 + b).to(dev) for b in range(B)]
    refs = [copy.deepcopy(n) for n in nets]
    # each fit on its own
    solo = []
    for b, ref in enumerate(refs):
        opt, clo, st = _solo_fit(ref, zs[b], ts[b], None if ms is None else ms[b], std, 40 + b, ema, dev)
        for _ in range(n_eager + n_graph):
            opt.zero_grad()
            clo()
            opt.step()
        solo.append((opt, st))
    torch.cuda.synchronize()
    # the same fits through one launch list
    native_mask.dip_group_native(mask)
    g = GroupedFits(nets, zs, ts, masks=ms, reg_noise_std=std, seeds=[40 + b for b in range(B)], lr=0.01,
                    exp_weight=0.99 if ema else None, ema_init="first")
    assert g.pointers_outside_row0() == []
    g.step(n_eager - 1)
    it = GraphedIteration.group(g, warmup=1)                   # one more eager iteration, then ONE hipGraph
    assert it is g and g.graph is not None
    it.run(n_graph)
    torch.cuda.synchronize()
    assert g.iterations == n_eager + n_graph and g.step_counts() == [n_eager + n_graph] * B
    assert native_mask.dip_group_size() == 1
    for b in range(B):
        opt, st = solo[b]
        assert opt.device_step_count() == n_eager + n_graph
        assert g.losses[b].item() == st["loss"].item(), (b, g.losses[b].item(), st["loss"].item())
        for (k, pa), pb in zip(nets[b].named_parameters(), refs[b].parameters()):
            assert torch.equal(pa, pb), (b, k)
        for (k, ba), bb in zip(nets[b].named_buffers(), refs[b].buffers()):
            assert torch.equal(ba, bb), (b, k)                  # BatchNorm running statistics, num_batches_tracked
        if ema:
            assert torch.equal(g.out_avg[b:b + 1], st["avg"]), b
    # the instances really are different fits
    assert len({round(g.losses[b].item(), 9) for b in range(B)}) == B


@pytest.mark.parametrize("fam", [1, 2, 4, 8, 16, 32, 64, 128, 256],
                         ids=["bn", "conv", "dma", "small", "thin", "wgrad", "loss", "misc", "upcat"])
def test_one_family_native_at_a_time(dev, native_mask, fam):
    """Each kernel family's one-dispatch form alone (the others as host loops) against the all-host-loop run of the same
    fits: localises a wrong grouped kernel to its family."""
    from dip_group import GroupedFits
    kind, hw, B = "wide", (128, 128), 2          # scale 0 on the implicit-GEMM kernels, scale 1 on conv_small
    zs, ts, ms = _problem(kind, hw, B, 1, dev)
    res = []
    for mask in (0, fam):
        native_mask.dip_group_native(mask)
        nets = [_net(kind, 20 + b).to(dev) for b in range(B)]
        g = GroupedFits(nets, zs, ts, masks=ms, reg_noise_std=0.03, seeds=[7, 8], exp_weight=0.99)
        g.step(3)
        torch.cuda.synchronize()
        res.append((g.losses.clone(), [p.detach().clone() for n in nets for p in n.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)


def test_group_protocol_errors(dev, native_mask):
    """A pointer outside instance 0's slab fails the launch (rc -1, nothing runs); a second dip_group_begin fails;
    dip_group_end always closes."""
    lib = native_mask
    buf = torch.zeros(4 * 1024, dtype=torch.uint8, device=dev)
    other = torch.zeros(8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    assert lib.dip_group_begin(4, 1024, buf.data_ptr(), 512) == 0
    try:
        assert lib.dip_group_size() == 4
        assert lib.dip_group_begin(2, 1024, buf.data_ptr(), 512) != 0
        rc = lib.dip_counter_add(other.data_ptr(), 5, st)                  # not in the slab: refused
        assert rc != 0 and b"outside" in lib.dip_last_error()
        assert lib.dip_counter_add(buf.data_ptr() + 256, 5, st) == 0       # in the slab: all 4 instances
    finally:
        assert lib.dip_group_end() == 0
    torch.cuda.synchronize()
    assert lib.dip_group_size() == 1
    assert other.sum().item() == 0
    rows = buf.view(4, 1024)[:, 256:264].contiguous().view(torch.int64)
    assert rows.flatten().tolist() == [5, 5, 5, 5]
    assert lib.dip_group_begin(2, 1000, buf.data_ptr(), 512) != 0          # stride not 256-byte aligned
    assert lib.dip_group_size() == 1
