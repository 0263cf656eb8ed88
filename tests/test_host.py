"""CPU suite: host-side logic of the MI355X backend -- C-ABI surface, module tree / state_dict
naming, planner, reference-compatible helper semantics, loud failure without the HIP path, and
the multi-process sharding used by `bench.py --gpus N` (world_size-2 gloo)."""
import ctypes
import copy
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_cabi_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "dip_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(dip_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    import dip_native
    assert declared == set(dip_native.EXPORTS), declared ^ set(dip_native.EXPORTS)
    L = ctypes.CDLL(dip_native.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert built.dip_abi_version() == dip_native.ABI_VERSION == 8
    # struct layouts agree with the header's field order (sizes on LP64)
    assert ctypes.sizeof(dip_native.DipTransform) == 24
    assert ctypes.sizeof(dip_native.DipGradSrc) == 56     # + the crop window of round 3, the thin 1x1 conv of round 6
    assert ctypes.sizeof(dip_native.DipPackRec) == 56


def test_state_dict_names_and_shapes_match_oracle_spec():
    import dip_oracle as O
    from models.skip import skip
    from test_net_gpu import NETS
    from test_oracle import _spec
    for name, cfg in NETS.items():
        net = skip(*cfg["args"], **cfg["kw"])
        shapes = O.param_shapes(_spec(cfg))
        got = {k: tuple(p.shape) for k, p in net.named_parameters()}
        assert got == shapes, name
        gold = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
        assert list(net.state_dict().keys()) == [k[3:] for k in gold.files if k.startswith("sd/")]


def test_planner_launch_lists(built):
    """Sizing pass of the engine runs without a GPU: one launch list per direction."""
    from models import get_net
    torch.manual_seed(0)
    net = get_net(32, "skip", "reflection", skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                  upsample_mode="bilinear")
    eng = net.__dict__["_dip_engine"]
    assert len(eng.convs) == 26 and len(eng.bns) == 30
    eng.lib = built
    eng._sizing = True
    eng._alloc = []
    eng.H = eng.W = 512
    eng._reset_sizing()
    eng.fwd_ops, eng.bwd_ops, eng.bwd_input_ops, eng.keep = [], [], [], []
    from dip_engine import Act
    last = eng._plan_scale(0, Act(None, 512, 512, 32), 512, 512)
    assert (last.H, last.W, last.C) == (512, 512, 128)
    assert eng.stat_need >= 2048 * 3 * 128
    assert eng.ws_need > 4            # the low-resolution scales run split-K
    import dip_native as N
    assert N.conv_plan(512, 512, 132, 128, 3, 1) == (1, 2048, 0)
    k, rows, wsf = N.conv_plan(16, 16, 128, 128, 3, 1)
    assert k == 18 and wsf == k * 16 * 16 * 128
    # sizes that are not divisible by 2^depth: strided convs give ceil(S/2) and Concat's centre crop drops the last
    # row / column of the up-sampled tensor (models/common.py:29-37)
    eng._reset_sizing()
    eng.fwd_ops, eng.bwd_ops, eng.bwd_input_ops, eng.keep = [], [], [], []
    last = eng._plan_scale(0, Act(None, 500, 421, 32), 500, 421)
    assert (last.H, last.W) == (500, 421)
    assert [(sc.st["d2"].H, sc.st["d2"].W) for sc in eng.sc] == [(250, 211), (125, 106), (63, 53), (32, 27), (16, 14)]


def test_launch_plans_of_the_round2_kernels(built):
    """Host-side launch plans (no GPU): phase-mode split-K of the stride-2 data gradients, slabs of the narrow-layer /
    thin-input weight gradients, domain of the 64-channel kernel."""
    import dip_native as N
    # stride-2 data gradient (dil == 2 descriptors): 4 workgroups per 8x16 tile of the half-resolution grid
    assert N.conv_plan_dil2(258, 258, 128, 128, 3) == (1, 17 * 9, 0)            # 612 workgroups: one pass
    k, rows, wsf = N.conv_plan_dil2(130, 130, 128, 128, 3)
    assert 1 < k <= 4 and wsf == k * 130 * 130 * 128                             # bounded by the 1-tap phase's 4 chunks
    assert N.conv_plan_dil2(258, 258, 128, 32, 3)[0] == N.conv_plan(258, 258, 128, 32, 3, 1)[0]   # not whole 128-blocks
    assert N.conv_plan_dil2(260, 260, 128, 128, 5)[0] == N.conv_plan(260, 260, 128, 128, 5, 1)[0]  # 5x5: dilated path
    # split-K fills one round of 2 workgroups per CU; round 5: a 3x3 stride-1 layer with 96 .. 255 tiles and >= 128 columns runs
    # the 64-column form of the bf16-pipe kernel in ONE pass (two workgroups per pixel tile)
    assert N.conv_plan(128, 128, 128, 128, 3, 1)[0] == 1 and N.conv_plan(128, 128, 128, 128, 3, 2)[0] == 4
    assert N.conv_plan(128, 128, 128, 64, 3, 1)[0] == 4             # < 128 columns: the fp32 split-K kernel as before
    # ... and a layer of that shape the bf16-pipe kernel will NOT take (no split weights, fused bnb partials, a transform over
    # > 512 channels) keeps the fp32 kernels' split-K: the engine plans it with dip_conv_plan_fp32
    assert N.conv_plan_fp32(128, 128, 128, 128, 3, 1)[0] == 4
    for shape in ((512, 512, 128, 128, 3, 1), (128, 128, 128, 64, 3, 1), (64, 64, 128, 128, 3, 1), (224, 352, 16, 16, 5, 1)):
        assert N.conv_plan_fp32(*shape) == N.conv_plan(*shape)      # everywhere else the two planners agree
    # weight gradient: nsplit counts SLABS; narrow layers write 4 / 2 per workgroup (waves split the K steps)
    n16 = N.wgrad_plan2(224, 352, 16, 16, 3, 1)[0]
    n64 = N.wgrad_plan2(224, 352, 16, 64, 3, 1)[0]
    assert n16 % 4 == 0 and n64 % 2 == 0 and n16 == 2 * n64
    assert N.wgrad_plan2(512, 512, 132, 128, 3, 1) == (128, 1, 1)
    assert N.wgrad_plan2(16, 16, 128, 16, 3, 1) == (16, 9, 1)                   # 4 walkers x 4 slabs, 9 tap groups
    # <= 4 input channels: one slab per block of the streaming kernel, at most 512
    for args in ((224, 352, 1, 16, 5, 2), (128, 192, 3, 8, 3, 2), (512, 512, 2, 128, 7, 1)):
        n, g, cb = N.wgrad_plan2(*args)
        assert 1 <= n <= 512 and n == N.wgrad_plan(*args)


def test_one_pass_plans_of_the_n64_range_are_bf16_pipe_eligible(built, monkeypatch):
    """dip_conv_plan gives a 3x3 stride-1 layer with 96..255 tiles ONE pass because the 64-column bf16-pipe kernel fills
    the chip with it; a descriptor that kernel refuses (fused BatchNorm-backward partials, no split weights) must not be left with
    that plan.  Planned on CPU memory for the default arithmetic and with DIP_BNB_FUSE=1."""
    import ctypes as C
    from models.skip import skip
    import dip_native as N
    if not built.dip_conv_bf3_terms():
        pytest.skip("DIP_CONV_BF3=0 in the environment")
    for fuse in ("0", "1"):
        monkeypatch.setenv("DIP_BNB_FUSE", fuse)
        net = skip(32, 3, [128] * 3, [128] * 3, [4] * 3, pad="reflection")
        eng = net.__dict__["_dip_engine"]
        assert eng.fuse_bnb == (fuse == "1")
        eng._build_arenas(torch.device("cpu"))
        eng._build_plan(256, 256, 32)
        seen = 0
        for d in eng.keep:
            if not isinstance(d, N.DipConvDesc) or d.ks != 3 or d.stride != 1 or d.dil != 1 or d.Cout < 128 or d.Cin < 16:
                continue
            if not 96 <= built.dip_conv_ntiles(d.Hout, d.Wout) < 256:
                continue
            seen += 1
            if d.ksplit <= 1:
                assert built.dip_conv_bf3_eligible(C.byref(d)), (fuse, d.Hout, d.Wout, d.Cin, d.Cout)
            else:
                assert not built.dip_conv_bf3_eligible(C.byref(d))
        assert seen >= 2, seen


def test_planner_builds_launch_lists_for_every_option(built):
    """Both planner passes (sizing + emission of the descriptors) run on CPU memory -- nothing is
    launched -- for every skip() option the backend accepts: per-scale filter sizes 3/5/7, avg / max
    pooling, Swish / ELU / none, filter_skip_size 3, no skips, no 1x1, zero / reflection padding."""
    from models.skip import skip
    from test_net_gpu import NETS
    import dip_native as N
    for name, cfg in NETS.items():
        net = skip(*cfg["args"], **cfg["kw"])
        eng = net.__dict__["_dip_engine"]
        assert not isinstance(eng, Exception), (name, eng)
        eng._build_arenas(torch.device("cpu"))
        eng._build_plan(64, 96, cfg["args"][0])
        names = [n for _, _, n in eng.fwd_ops]
        assert names[-1] == "conv_fwd:out" and sum(n.startswith("conv_fwd:") for n in names) == len(eng.convs), name
        bnames = [n for _, _, n in eng.bwd_ops]
        assert sum(n.startswith("wgrad:") for n in bnames) == len(eng.convs), name
        assert sum(n.startswith(("bnb_apply:", "bnb_one:", "upb_one:")) for n in bnames) == len(eng.bns), name
        if cfg["kw"].get("downsample_mode") in ("avg", "max"):
            assert any(n.startswith("pool:") for n in names) and any(n.startswith("poolb:") for n in bnames), name
    net = skip(8, 3, [16, 16], [16, 16], [4, 4], act_fun="none", pad="reflection")
    eng = net.__dict__["_dip_engine"]
    eng._build_arenas(torch.device("cpu"))
    eng._build_plan(32, 32, 8)
    assert eng.slope == 1.0
    pooled = skip(8, 3, [16, 16], [16, 16], [4, 4], downsample_mode="avg", pad="reflection")
    e3 = pooled.__dict__["_dip_engine"]
    e3._build_arenas(torch.device("cpu"))
    e3._build_plan(30, 33, 8)                  # pooling floors 15x16 -> 7x8: the concats shrink (centre crops with offsets)
    assert (e3.Hout, e3.Wout) == (28, 32) and e3.sc[0].st["geom"]["os_y"] == 1 and e3.sc[1].st["geom"]["os_y"] == 0
    assert [n for _, _, n in e3.fwd_ops if n.startswith("upcat:")] == ["upcat:s1.cat_bn", "upcat:s0.cat_bn"]
    noskip = skip(8, 3, [16, 16], [16, 16], [4, 0], pad="reflection")
    e4 = noskip.__dict__["_dip_engine"]
    e4._build_arenas(torch.device("cpu"))
    e4._build_plan(32, 48, 8)
    e4._build_plan(30, 47, 8)                  # 15x24 -> 8x12 -> (no Concat at scale 1) 16x24 -> x2 = 32x48 against 30x47
    assert (e4.Hout, e4.Wout) == (30, 47) and e4.sc[0].st["geom"]["od_y"] == 1 and e4.sc[0].st["geom"]["od_x"] == 0
    assert (e4.sc[1].st["Ho"], e4.sc[1].st["Wo"]) == (16, 24) and e4.sc[1].st["geom"] is None
    big = skip(8, 3, [16, 16], [16, 16], [4, 4], filter_skip_size=5, filter_size_down=3, pad="reflection")
    e2 = big.__dict__["_dip_engine"]
    e2._build_arenas(torch.device("cpu"))
    with pytest.raises(NotImplementedError, match="skip filter larger"):
        e2._build_plan(32, 32, 8)


def test_backward_stream_dependencies_follow_the_op_list(built):
    """Two-stream backward: every side-stream "dgthin:X" launch (the thin columns of a 132-column data gradient) must
    be waited for by the main-stream op that follows "dgrad:X" -- for ANY conv X, not only the decoder convs of the
    notebooks' nets (round-2 advisor finding: a 132-channel down_b raced)."""
    from models.skip import skip
    net = skip(8, 3, [132, 132], [128, 128], [4, 4], upsample_mode="bilinear", pad="reflection")
    eng = net.__dict__["_dip_engine"]
    eng._build_arenas(torch.device("cpu"))
    eng._build_plan(512, 512, 8)
    names = [n for _, _, n in eng.bwd_ops]
    thin = [n for n in names if n.startswith("dgthin:")]
    assert "dgthin:s0.down_b" in thin and "dgthin:s0.up" in thin, thin
    deps = eng._backward_deps(eng.bwd_ops)
    waited = {p for ps in deps.values() for p in ps}
    assert set(thin) <= waited, (thin, deps)
    for consumer, prods in deps.items():
        assert not eng._BWD_SIDE(consumer) or consumer.endswith(".skip_bn") or consumer.startswith(("dgrad+", "wgrad:")), consumer
        for p in prods:
            assert names.index(p) < names.index(consumer), (p, consumer)
            if p.startswith("dgthin:"):
                # nothing on the main stream may touch the buffer between the 128-column launch and the wait
                j = names.index("dgrad:" + p[len("dgthin:"):])
                between = [n for n in names[j + 1:names.index(consumer)] if not eng._BWD_SIDE(n)]
                assert between == [], (p, consumer, between)
    # (the data gradients run on the interior domain: the frame launch "dgring:" is the first main-stream op behind the
    # 128-column launch)
    assert deps["dgring:s0.down_b"] == ["dgthin:s0.down_b"] and deps["dgring:s0.up"] == ["dgthin:s0.up"]
    # opt-in (DIP_BNB_FUSE=1): the BatchNorm-backward statistics ride in the data-gradient launches, the finalisation waits
    eng.fuse_bnb = True
    eng._build_plan(512, 512, 8)
    names2 = [n for _, _, n in eng.bwd_ops]
    deps2 = eng._backward_deps(eng.bwd_ops)
    assert deps2["dgring:s0.down_b"] == ["dgthin:s0.down_b"] and "bnb_stats:s0.down_a_bn" in names2     # (no fusion with a frame launch)
    assert deps2["dgring:s0.up"] == ["dgthin:s0.up"] and len(names2) <= len(names)
    assert deps["dgrad+:s1.skip_conv"] == ["bnb_apply:s1.skip_bn"]      # (a stride-2 down_a has no thin launch)
    # the weight gradients run on the bulk stream: the skip conv's waits for the side stream's BatchNorm backward
    assert deps["wgrad:s0.skip_conv"] == ["bnb_apply:s0.skip_bn"] and deps["wgrad:s1.skip_conv"] == ["bnb_apply:s1.skip_bn"]
    cls = [eng._BWD_SIDE(n) for n in names]
    assert set(cls) == {0, 1, 2} and all(c == 2 for n, c in zip(names, cls) if n.startswith(("wgrad:", "wgred:")))


def test_get_noise_get_params_semantics():
    from utils.common_utils import get_noise, get_params, np_to_torch, torch_to_np
    gn = np.load(os.path.join(GOLDEN, "get_noise.npz"))
    torch.manual_seed(0)
    assert np.array_equal(get_noise(32, "noise", (16, 24)).numpy(), gn["u_s0_32x16x24"])
    torch.manual_seed(7)
    assert np.array_equal(get_noise(3, "noise", 8, noise_type="n", var=0.5).numpy(), gn["n_s7_3x8x8"])
    m = get_noise(2, "meshgrid", (8, 12))
    assert m.dtype == torch.float64 and np.array_equal(m.numpy(), gn["mesh_8x12"])
    net = torch.nn.Conv2d(3, 3, 1)
    z = torch.zeros(1, 3, 4, 4)
    p = get_params("net,input", net, z)
    assert len(p) == 3 and p[-1] is z and z.requires_grad
    down = torch.nn.Conv2d(1, 1, 1)
    assert get_params("net,down", net, z, down) == list(down.parameters())      # 'down' REPLACES
    a = np.random.rand(3, 5, 7).astype(np.float32)
    assert np_to_torch(a).shape == (1, 3, 5, 7) and np.array_equal(torch_to_np(np_to_torch(a)), a)


def test_downsampler_taps_match_reference_vectors():
    from models.downsampler import Downsampler, get_kernel
    gold = np.load(os.path.join(GOLDEN, "downsampler.npz"))
    for factor in (4, 2, 8):
        d = Downsampler(n_planes=3, factor=factor, kernel_type="lanczos2", phase=0.5, preserve_size=True)
        assert np.array_equal(d.kernel, gold[f"lanczos2_f{factor}/kernel"])
        assert list(d.state_dict().keys()) == ["downsampler_.weight", "downsampler_.bias"]
        assert d.downsampler_.weight.shape == (3, 3, 4 * factor, 4 * factor)
    assert get_kernel(2, "box", 0.5, 4).shape == (4, 4)


def test_downsampler_switches_to_the_dense_path_when_optimised():
    """Fixed taps (depth-wise kernel, parameters without grad) until get_params('down', ...) asks for the reference's
    behaviour -- the dense Conv2d weight is optimised, utils/common_utils.py:44-46 -- or a trained weight is loaded."""
    from models.downsampler import Downsampler
    from utils.common_utils import get_params
    d = Downsampler(n_planes=3, factor=4, kernel_type="lanczos2", phase=0.5, preserve_size=True)
    assert not any(p.requires_grad for p in d.parameters()) and not d._nondiag
    d.load_state_dict(copy.deepcopy(d.state_dict()))
    assert not d._nondiag
    ps = get_params("down", None, torch.zeros(1, 3, 8, 8), d)
    assert [tuple(p.shape) for p in ps] == [(3, 3, 16, 16), (3,)] and all(p.requires_grad for p in ps)
    gold = np.load(os.path.join(GOLDEN, "downsampler_dense.npz"))
    d2 = Downsampler(n_planes=3, factor=4, kernel_type="lanczos2", phase=0.5, preserve_size=True)
    sd = d2.state_dict()
    sd["downsampler_.weight"] = torch.from_numpy(gold["f4/w"])
    sd["downsampler_.bias"] = torch.from_numpy(gold["f4/b"])
    d2.load_state_dict(sd)
    assert d2._nondiag
    with pytest.raises(RuntimeError, match="MI355X"):
        d2(torch.zeros(1, 3, 16, 16))


def test_no_silent_fallback():
    """CPU tensors and unsupported options raise instead of running an eager path."""
    from models.skip import skip
    from models import get_net
    net = skip(4, 3, [8, 8], [8, 8], [4, 4], pad="reflection", upsample_mode="bilinear")
    with pytest.raises(RuntimeError, match="MI355X"):
        net(torch.zeros(1, 4, 16, 16))
    bad = skip(4, 3, [8, 8], [8, 8], [4, 4], act_fun=torch.nn.Tanh)          # (LeakyReLU / Swish / ELU / none have kernels)
    with pytest.raises(NotImplementedError):
        bad(torch.zeros(1, 4, 16, 16))
    mx = skip(4, 3, [8, 8], [8, 8], [4, 4], downsample_mode="lanczos2")     # has kernels since round 3: same loud CPU error
    with pytest.raises(RuntimeError, match="MI355X"):
        mx(torch.zeros(1, 4, 16, 16))
    with pytest.raises(RuntimeError, match="launch list"):                  # its Downsampler is not a stand-alone module
        [m for m in mx.modules() if type(m).__name__ == "Downsampler"][0](torch.zeros(1, 8, 16, 16))
    with pytest.raises(NotImplementedError):
        get_net(3, "UNet", "zero", "nearest")
    # missing shared library -> loud error
    code = ("import sys; sys.path.insert(0, %r); import dip_native as N; N.LIB_PATH = '/nonexistent/libdip_hip.so';"
            "N._lib = None\ntry:\n    N.lib()\nexcept RuntimeError as e:\n    print('LOUD', e)\n"
            % os.path.join(ROOT, "deep-image-prior_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "LOUD" in r.stdout and "no fallback" in r.stdout


def test_grouped_fits_slab_layout_on_host_memory(built):
    """dip_group.GroupedFits builds its memory model without a GPU (the dry mode of this test only: nothing can be
    launched): one slab per instance, identically laid out, every pointer of the launch list inside instance 0's slab
    (what the library checks per grouped launch), parameters / BatchNorm buffers of every net moved into their rows
    unchanged, per-instance views strided by the slab size."""
    from models.skip import skip
    from dip_group import GroupedFits

    def small(seed):
        torch.manual_seed(seed)
        return skip(8, 3, num_channels_down=[16, 32, 32], num_channels_up=[16, 32, 32], num_channels_skip=[4, 0, 4],
                    upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")

    B = 3
    nets = [small(k) for k in range(B)]
    with torch.no_grad():
        for b, n in enumerate(nets):                          # distinguishable BatchNorm buffers
            for m in n.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.add_(b + 1.0)
                    m.num_batches_tracked.add_(7 * b)
    ref = [{k: v.clone() for k, v in n.state_dict().items()} for n in nets]
    zs = [torch.rand(1, 8, 36, 52) * 0.1 for _ in range(B)]
    ts = [torch.rand(1, 3, 36, 52) for _ in range(B)]
    ms = [(torch.rand(1, 1, 36, 52) > 0.3).float() for _ in range(B)]
    g = GroupedFits(nets, zs, ts, masks=ms, reg_noise_std=1 / 30., seeds=[5, 6, 7], exp_weight=0.99, device="cpu",
                    _dry_cpu=True)
    assert g.stride % 256 == 0 and g.mem.data_ptr() % 256 == 0 and g.mem.numel() == B * g.stride
    assert g.pointers_outside_row0() == []
    ex = g._row0_extra
    for b, n in enumerate(nets):
        lo = g.mem.data_ptr() + b * g.stride
        sd = n.state_dict()
        assert list(sd) == list(ref[b])
        for k in ref[b]:
            assert torch.equal(sd[k], ref[b][k]), (b, k)
            assert lo <= sd[k].data_ptr() < lo + g.stride, (b, k)        # parameters AND buffers live in row b
        assert [p.data_ptr() - lo for p in n.parameters()] == [p.data_ptr() - g.mem.data_ptr() for p in nets[0].parameters()]
        assert torch.equal(g._inst(ex["saved"], b).view(zs[b].shape), zs[b])
        assert torch.equal(g._inst(ex["target"], b).view(ts[b].shape), ts[b])
        assert torch.equal(g._inst(ex["mask"], b).view(ms[b].shape), ms[b])
        assert g._inst(ex["rng"], b).tolist() == [0, 5 + b] and g._inst(ex["gl"], b).item() == 1.0
        # descriptor tables are replicated: the grouped weight-packing kernel reads the table of its own instance
        assert torch.equal(g._inst(g.eng.pack_recs, b), g.eng.pack_recs)
    assert g.losses.shape == (B,) and g.losses.stride() == (g.stride // 4,)
    assert g.out.shape == (B, 3, 36, 52) and g.out.stride() == (g.stride // 4, 36 * 52, 52, 1)
    assert g.out[1].data_ptr() == g._inst(ex["out"], 1).data_ptr()
    assert g._nbt_all[2].data_ptr() == g._inst(g.eng.nbt, 2).data_ptr()
    assert g._nbt_all[:, 0].tolist() == [0, 7, 14]
    with pytest.raises(RuntimeError, match="dry"):
        g.step()
    with pytest.raises(RuntimeError, match="MI355X"):
        GroupedFits([small(0)], zs[:1], ts[:1], device="cpu")
    with pytest.raises(ValueError, match="architecture"):
        GroupedFits([small(0), skip(8, 3, [16, 32], [16, 32], [4, 4])], zs[:2], ts[:2], device="cpu", _dry_cpu=True)
    odd = skip(8, 3, [16, 16], [16, 16], [0, 0], pad="reflection")
    with pytest.raises(ValueError, match="net output"):        # without skips an odd size comes out larger than it went in
        GroupedFits([odd], [torch.rand(1, 8, 18, 18)], [torch.rand(1, 3, 18, 18)], device="cpu", _dry_cpu=True)
    eng = odd.__dict__["_dip_engine"]
    assert eng.slab is None and eng.device is None             # a failed construction leaves no allocator / half-built state behind


def test_padded_skip_conv_next_to_a_stride1_down_conv_builds(built):
    """skip(..., filter_skip_size=3, pad='reflection') with a stride-1 down conv (downsample_mode
    avg / max / lanczos2) at a size where down_a's data gradient would take the interior + ring form (pad 0): the skip
    conv's gradient is accumulated into down_a's buffer on the PADDED domain, so the ring form must not be chosen there
    (dip_engine._emit_dgrad need_pad).  Dry build of the launch list on host memory (nothing is launched); the gradients of
    this configuration are checked on the GPU by tests/test_net_gpu.py::test_golden_reference_vectors[net_tiny_skip3]."""
    from models.skip import skip
    from dip_group import GroupedFits
    for mode in ("avg", "max", "lanczos2", "stride"):
        torch.manual_seed(0)
        net = skip(8, 3, [16, 32], [16, 32], [4, 4], filter_skip_size=3, downsample_mode=mode, pad="reflection",
                   upsample_mode="bilinear")
        g = GroupedFits([net], [torch.rand(1, 8, 128, 128)], [torch.rand(1, 3, 128, 128)], device="cpu", _dry_cpu=True)
        names = [op[2] for op in g.engine.bwd_ops] if hasattr(g, "engine") else []
        assert g.pointers_outside_row0() == []
        del g, names


def test_grouped_iteration_issues_its_launch_list_without_a_gpu(built):
    """The host side of one grouped iteration end to end, without a GPU: every launch of the list goes through the real
    C ABI (argument marshalling, dispatch, the slab range check) and fails at hipLaunchKernel with "no ROCm-capable device"
    (rc 100), which this test -- and only this test -- tolerates; anything else (a bad argument, a refused launch, a Python
    error) fails it.  In a subprocess: it patches torch.cuda's stream accessors."""
    code = r"""
import sys, contextlib, torch
sys.path.insert(0, %r)
import dip_native as N
from models.skip import skip
import dip_group as G
seen = []
real = N.check
def check(rc, what=""):
    seen.append((what, rc))
    if rc not in (0, 100):
        real(rc, what)
N.check = check
class FakeStream:
    cuda_stream = None
torch.cuda.current_stream = lambda d=None: FakeStream()
torch.cuda.device = lambda d: contextlib.nullcontext()
torch.cuda.is_current_stream_capturing = lambda: False
def net(seed):
    torch.manual_seed(seed)
    return skip(8, 3, num_channels_down=[16, 32, 32], num_channels_up=[16, 32, 32], num_channels_skip=[4, 4, 4],
                upsample_mode="bilinear", need_sigmoid=True, need_bias=True, pad="reflection")
B = 3
g = G.GroupedFits([net(k) for k in range(B)], [torch.rand(1, 8, 32, 64) * 0.1 for _ in range(B)],
                  [torch.rand(1, 3, 32, 64) for _ in range(B)], masks=[torch.ones(1, 1, 32, 64)] * B, reg_noise_std=1 / 30.,
                  exp_weight=0.99, device="cpu", _dry_cpu=True)
g._dry = False                      # (launches are attempted from here on)
g.step(2)
failed = [w for w, rc in seen if rc == 100]
for must in ("noise_axpy_dev2", "pack_weights", "nchw_to_nhwc", "loss_head_fwd", "loss_head_bwd", "adam_tick", "adam_step_dev"):
    assert failed.count(must) == 2, (must, failed.count(must))
eng = g.eng
names = [n for _, _, n in eng.fwd_ops[:-1] + eng.bwd_ops]
assert all(failed.count(n) == 2 * names.count(n) for n in set(names)), "an op of the launch list was not issued"
assert N.lib().dip_group_size() == 1 and g.iterations == 2
assert g._nbt_all.unique().tolist() == [2]
print("WALK_OK", len(failed))
""" % os.path.join(ROOT, "deep-image-prior_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DIP_TWO_STREAMS="0", DIP_NO_CLIST="1"))      # (launch by launch: a command list stops at its first failure)
    assert "WALK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_command_list_thunks_agree_with_the_binding(built):
    """csrc/dip_list.hip registers every stream-launching entry point with a thunk that unpacks one 8-byte slot per parameter;
    dip_native.CmdList packs the slots from the ctypes signatures.  The two must agree on the parameter count of every entry
    point -- and every entry point of the binding that takes a trailing stream must be registered."""
    import dip_native as N
    L = built
    streamy = [n for n, (res, args) in N._SIGS.items() if args and args[-1] is ctypes.c_void_p and res is ctypes.c_int
               and n not in ("dip_list_run",) and not n.endswith(("_ok", "_eligible"))]
    host_only = {"dip_events_create", "dip_events_destroy"}
    missing = [n for n in streamy if L.dip_list_fn_id(n.encode()) < 0 and n not in host_only]
    # (entry points whose last parameter is a pointer but not a stream)
    assert set(missing) <= {"dip_conv_plan", "dip_conv_plan_dil2", "dip_wgrad_plan", "dip_wgrad_plan2", "dip_group_begin",
                            "dip_device_pci_bus_id"}, missing
    for n in streamy:
        fid = L.dip_list_fn_id(n.encode())
        if fid >= 0:
            assert L.dip_list_fn_nargs(fid) == len(N._SIGS[n][1]), n
    assert L.dip_list_fn_id(b"dip_conv_plan") == -1 and L.dip_list_fn_id(b"nope") == -1
    # slot packing: ints sign-extended, floats in the low 4 bytes, byref -> the struct's address
    d = N.DipGradSrc(0x1234, 1, 1, 4, 0)
    assert N._slot(ctypes.POINTER(N.DipGradSrc), ctypes.byref(d)) == ctypes.addressof(d)
    assert N._slot(ctypes.c_int, -1) == 0xFFFFFFFFFFFFFFFF and N._slot(ctypes.c_void_p, None) == 0
    import struct
    assert N._slot(ctypes.c_float, 0.2) == struct.unpack("<I", struct.pack("<f", 0.2))[0]
    assert N._slot(ctypes.c_double, 0.1) == struct.unpack("<Q", struct.pack("<d", 0.1))[0]


def test_command_list_reports_the_failing_launch(built):
    """Without a GPU the first LAUNCH of a list fails inside the library (hipErrorNoDevice, rc 100): dip_list_run stops there
    and CmdList.run raises with that op's name; RECORD / WAIT indices out of range are refused (rc -1)."""
    import dip_native as N
    L = built
    src = N.DipGradSrc(None, 0, 0, 4, 0)
    args = (ctypes.byref(src), None, 4, 4, 4, 4, None, 4, 0.2, None, 4, None, 1)
    cl = N.CmdList([("launch", L.dip_bn_bwd_stats, args, 0, "bnb_stats:probe"), ("launch", L.dip_bn_bwd_stats, args, 0, "second")])
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    with pytest.raises(RuntimeError, match="bnb_stats:probe"):
        cl.run([None])
    assert cl._failed.value == 0
    bad = (N.DipCmd * 1)(N.DipCmd(N.CMD_RECORD, -1, 0, 3, None, 0, 0))
    st, ev, failed = (ctypes.c_void_p * 1)(), (ctypes.c_void_p * 1)(), ctypes.c_int(-1)
    assert L.dip_list_run(bad, 1, st, 1, ev, 1, ctypes.byref(failed)) == -1 and failed.value == 0
    with pytest.raises(RuntimeError, match="arguments"):
        N.CmdList([("launch", L.dip_bn_bwd_stats, args[:-1], 0, "short")])


def test_group_pointer_lists_cover_every_descriptor_pointer():
    """csrc/dip_group.h names the pointer fields of every descriptor struct ONCE (dip_ptrs); a grouped launch shifts exactly
    those.  A pointer field missing there would be read at instance 0's address by every instance: compare the lists with
    the struct definitions of the binding (which test_cabi_exports_every_declared_symbol ties to include/dip_hip.h)."""
    import dip_native as N
    src = open(os.path.join(ROOT, "deep-image-prior_amd", "csrc", "dip_group.h")).read()
    lists = {}
    for m in re.finditer(r"dip_ptrs\((Dip\w+)& (\w+), F& f\) \{(.*?)\n?\}", src, re.S):
        name, var, body = m.groups()
        fields = re.findall(r"\bf\(%s\.(\w+)\)" % var, body) + re.findall(r"dip_ptrs\(%s\.(\w+), f\)" % var, body)
        lists[name] = sorted(fields)
    structs = {n: getattr(N, n) for n in ("DipTransform", "DipConvDesc", "DipWgradDesc", "DipGradSrc", "DipBnFin", "DipBnbFin",
                                          "DipUpcatDesc", "DipLossHeadDesc")}
    for name, st in structs.items():
        want = sorted(f for f, t in st._fields_ if t is ctypes.c_void_p or (isinstance(t, type) and issubclass(t, ctypes.Structure)))
        assert lists.get(name) == want, (name, lists.get(name), want)
    # structs without pointers must not appear as launch arguments unless they say so (SmallGeom does, in conv_small.hip)
    for name in ("DipPackRec", "DipPackRec3", "DipIterState"):
        assert not any(t is ctypes.c_void_p for _, t in getattr(N, name)._fields_)


def test_group_protocol_host_state(built):
    """dip_group_begin / dip_group_end / dip_group_native are host-side state (csrc/dip_core.hip): checked without a GPU."""
    L = built
    assert L.dip_group_size() == 1
    buf = (ctypes.c_char * 4096)()
    base = (ctypes.addressof(buf) + 255) & ~255
    prev = L.dip_group_native(-1)
    try:
        assert L.dip_group_begin(0, 1024, base, 512) != 0 and b"instances" in L.dip_last_error()
        assert L.dip_group_begin(2, 1024, None, 512) != 0
        assert L.dip_group_begin(2, 256, base, 512) != 0 and b"overlap" in L.dip_last_error()      # stride < row
        assert L.dip_group_begin(2, 1000, base, 512) != 0                                           # unaligned stride
        assert L.dip_group_begin(2, 1024, base + 8, 512) != 0                                       # unaligned base
        assert L.dip_group_size() == 1
        assert L.dip_group_begin(3, 1024, base, 512) == 0 and L.dip_group_size() == 3
        assert L.dip_group_begin(2, 1024, base, 512) != 0 and b"already open" in L.dip_last_error()
        assert L.dip_group_end() == 0 and L.dip_group_size() == 1
        assert L.dip_group_begin(1, 0, base, 512) == 0           # one instance: stride is irrelevant
        assert L.dip_group_end() == 0
        assert L.dip_group_native(0x15) == prev and L.dip_group_native(-1) == 0x15
    finally:
        L.dip_group_end()
        L.dip_group_native(prev)


def test_fused_adam_grouping_cpu_logic():
    from dip_optim import _split_contiguous
    arena = torch.zeros(64)
    a, b, c = arena[0:10], arena[12:20], arena[20:40]
    other = torch.zeros(5)
    groups = _split_contiguous([a, b, c, other])
    assert [len(g) for g in groups] == [3, 1]


def test_shard_assignment_world_size_2_gloo():
    """bench.py's image->rank partition under torch.distributed (gloo, 2 processes, CPU)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from bench import shard_images, reduce_max_time
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["PORT"],
                        rank=int(os.environ["RANK"]), world_size=2)
mine = shard_images(8, dist.get_rank(), 2)
t = reduce_max_time(1.0 + dist.get_rank(), "cpu")
allm = [None, None]
dist.all_gather_object(allm, mine)
if dist.get_rank() == 0:
    assert sorted(allm[0] + allm[1]) == list(range(8)) and not set(allm[0]) & set(allm[1]), allm
    assert abs(t - 2.0) < 1e-6, t
    print("OK")
dist.destroy_process_group()
''' % ROOT
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0][0]


def _bench_selftest(cmd, extra_env=None):
    env = dict(os.environ, DIP_BENCH_SELFTEST="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


def test_bench_rank_affinity_helpers():
    """bench.py's N-rank host hygiene: cpulist parsing and a pin attempt that must never raise."""
    import bench
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and bench.parse_cpulist("5") == [5]
    before = os.sched_getaffinity(0)
    r = bench.pin_to_gpu_numa(0, 1)
    assert isinstance(r, dict) and r["pinned"] is False and os.sched_getaffinity(0) == before
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` itself starts 2 worker processes (one per GPU), which meet over gloo
    (barrier, per-rank gather, max-over-ranks time) and print ONE JSON line with n_gpus == 2.
    DIP_BENCH_SELFTEST=1 swaps the GPU fit for a dummy CPU step; launcher, rendezvous and reduction
    are the code the real run uses."""
    import json
    r = _bench_selftest([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and len(d["per_rank_it_s"]) == 2 and d["value"] > 0
    # the torch.distributed.run spelling the driver uses
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = _bench_selftest([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                         "--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and len(d["per_rank_it_s"]) == 2
    # a world size that disagrees with --gpus is an error, not a silent single-rank run
    r = _bench_selftest([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                        {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_gpus_8_starts_eight_ranks():
    """The world size the driver's scaling run uses (1 / 2 / 4 / 8 GPUs of one node): `python bench.py --gpus 8` starts eight
    worker processes that meet over gloo on 127.0.0.1 (barrier, per-rank gather, max-over-ranks time) and print ONE line with
    eight per-rank figures; the shards of a batch of 8 x instances images are disjoint and cover it."""
    import json
    r = _bench_selftest([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and len(d["per_rank_it_s"]) == 8 and all(v > 0 for v in d["per_rank_it_s"]) and d["value"] > 0
    sys.path.insert(0, ROOT)
    import bench
    for inst in (1, 3):
        shards = [bench.shard_images(8 * inst, rk, 8) for rk in range(8)]
        assert sorted(i for sh in shards for i in sh) == list(range(8 * inst)) and all(len(sh) == inst for sh in shards)
        assert [sh[0] for sh in shards] == list(range(8))        # round-robin: rank r's first image (= seed) is r


def test_arena_lbfgs_matches_torch_lbfgs():
    """dip_optim.ArenaLBFGS restates torch.optim.LBFGS (no line search, the reference's settings:
    tolerance -1, utils/common_utils.py:218): same iterates on a small smooth problem, for parameters
    that are views of one arena (flat path) and for scattered parameters (gather path)."""
    import dip_optim

    def run(kind, flat):
        torch.manual_seed(0)
        A, b = torch.randn(30, 20), torch.randn(30)
        torch.manual_seed(1)
        arena = torch.randn(32) * 0.1
        if flat:
            w, c = torch.nn.Parameter(arena[:20].view(4, 5)), torch.nn.Parameter(arena[20:32])
        else:
            w, c = torch.nn.Parameter(arena[:20].clone().view(4, 5)), torch.nn.Parameter(arena[20:32].clone())
        params = [w, c]
        if kind == "torch":
            opt = torch.optim.LBFGS(params, max_iter=25, lr=0.5, tolerance_grad=-1, tolerance_change=-1)
        else:
            opt = dip_optim.ArenaLBFGS(params, max_iter=25, lr=0.5, tolerance_grad=-1, tolerance_change=-1,
                                       _allow_cpu=True)
        hist = []

        def closure():
            opt.zero_grad()
            x = w.reshape(-1)
            l = ((A @ x - b) ** 2).mean() + 0.1 * (torch.tanh(c) ** 2).sum() + (x[:12] * c).sum() ** 2
            l.backward()
            hist.append(l.item())
            return l

        opt.step(closure)
        return hist, torch.cat([w.detach().reshape(-1), c.detach()])

    h0, x0 = run("torch", False)
    for flat in (True, False):
        h, x = run("arena", flat)
        assert len(h) == len(h0) == 25
        assert max(abs(a - b) for a, b in zip(h, h0)) < 1e-5 and (x - x0).abs().max() < 1e-5
    with pytest.raises(RuntimeError, match="CUDA"):
        dip_optim.ArenaLBFGS([torch.nn.Parameter(torch.zeros(3))])


def test_sr_and_inpainting_helpers(built, tmp_path):
    """utils/sr_utils.py and utils/inpainting_utils.py: the reference's names with its behaviour
    (reference utils/sr_utils.py:3-94, utils/inpainting_utils.py:7-22)."""
    from PIL import Image
    from utils import sr_utils as S
    from utils import inpainting_utils as I
    rng = np.random.RandomState(0)
    # tv_loss against its definition
    x = torch.rand(1, 3, 7, 9, dtype=torch.float64)
    dh, dw = (x[..., 1:] - x[..., :-1]) ** 2, (x[..., 1:, :] - x[..., :-1, :]) ** 2
    want = sum(float((dh[0, c, i, j] + dw[0, c, i, j]) ** 0.5) for c in range(3) for i in range(6) for j in range(8))
    assert abs(float(S.tv_loss(x)) - want) < 1e-9
    # put_in_center
    img = rng.rand(3, 4, 6)
    out = S.put_in_center(img, (10, 10))
    assert out.shape == (3, 10, 10) and np.array_equal(out[:, 3:7, 2:8], img) and out.sum() == pytest.approx(img.sum())
    # load + crop-to-32 + LR / baselines
    arr = (rng.rand(70, 100, 3) * 255).astype(np.uint8)
    f = tmp_path / "img.png"
    Image.fromarray(arr).save(f)
    d = S.load_LR_HR_imgs_sr(str(f), -1, 4, 'CROP')
    assert d['HR_pil'].size == (96, 64) and d['LR_pil'].size == (24, 16) and d['HR_np'].shape == (3, 64, 96)
    assert np.array_equal(d['HR_np'], d['orig_np'][:, 3:67, 2:98])
    bic, sharp, near = S.get_baselines(d['LR_pil'], d['HR_pil'])
    assert bic.shape == sharp.shape == near.shape == (3, 64, 96)
    # masks
    np.random.seed(0)
    m = I.pil_to_np(I.get_bernoulli_mask(d['HR_pil'], zero_fraction=0.9))
    assert m.shape == (3, 64, 96) and 0.07 < m.mean() < 0.13
    t = I.get_text_mask(Image.fromarray(np.zeros((200, 300, 3), dtype=np.uint8)))
    assert t.size == (300, 200) and np.array(t).max() == 255


def test_build_id_is_the_hash_of_the_sources(built):
    """dip_build_id() (round 5): the library carries the first 16 hex digits of the sha256 over csrc/*.hip, csrc/*.h and
    include/dip_hip.h; build() rebuilds when it differs from the sources on disk (not on mtimes), bench.py prints it."""
    import __graft_entry__ as ge
    sid = ge.source_id()
    assert len(sid) == 16 and int(sid, 16) >= 0
    assert built.dip_build_id().decode() == sid == ge.library_id()
    assert ge.library_id("/nonexistent/libdip_hip.so") is None and not ge._stale()


def test_end_quality_rule_on_synthetic_families():
    """The registered end-quality rule (tests/test_net_gpu._compare_end_quality) on synthetic families:
    duplicates count once, the 3 % loss rule applies where the reference family is tight, Welch on log(loss) at alpha = 0.01
    where it is not (and needs n >= 8), the PSNR thresholds act on the family means."""
    import numpy as np
    from test_net_gpu import _compare_end_quality
    rng = np.random.RandomState(0)

    def fam(n, gt, sm, loss, jitter, tail=None):
        out = []
        for k in range(n):
            a = {"psnr_gt": gt + 0.2 * rng.randn(), "psnr_gt_sm": sm + 0.2 * rng.randn(), "loss": loss * (1 + jitter * rng.randn())}
            if tail is not None:
                a["loss_tail"] = tail * float(np.exp(0.13 * rng.randn()))
            out.append(a)
        return out

    # denoising-like: tight loss -> 3 % rule; a duplicate arm is dropped
    cpu = fam(8, 34.0, 38.0, 0.0089, 0.002)
    hip = fam(6, 34.05, 37.95, 0.0089, 0.002)
    _compare_end_quality("synthetic denoise", hip + [dict(hip[0])], cpu)
    with pytest.raises(AssertionError):
        _compare_end_quality("synthetic denoise, loss 5 % off", [dict(h, loss=h["loss"] * 1.05) for h in hip], cpu)
    with pytest.raises(AssertionError):
        _compare_end_quality("synthetic denoise, 0.6 dB off", [dict(h, psnr_gt=h["psnr_gt"] + 0.6) for h in hip], cpu)
    # SR-like: the tail loss jitters by 13 % -> Welch on the log; equal families pass, a 40 % offset is rejected, n < 8 is refused
    cpu = fam(16, 36.0, 36.6, 5e-5, 0.2, tail=5e-5)
    hip = fam(16, 36.0, 36.6, 5e-5, 0.2, tail=5e-5)
    _compare_end_quality("synthetic sr", hip, cpu)
    with pytest.raises(AssertionError):
        _compare_end_quality("synthetic sr, loss 40 % low", [dict(h, loss_tail=h["loss_tail"] * 0.6) for h in hip], cpu)
    with pytest.raises(AssertionError):
        _compare_end_quality("synthetic sr, too few arms", hip[:5], cpu)
