"""Pin the oracle: run the REAL reference (/root/reference) and the restatement
(oracle/dip_oracle.py) on the same state_dict + input and require torch.equal on the
output, the loss and every gradient; then one Adam step -> identical parameters.  Ten net configurations (the
notebooks' nets, avg / max pooling, centre crops at non-divisible sizes, Lanczos down-sampling inside conv).

Run:  python oracle/verify_against_reference.py        (build container only)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _refload  # noqa: E402
import dip_oracle as O  # noqa: E402

CONFIGS = {
    # name: (ref ctor kwargs for skip(), input_depth, H, W)
    "default": dict(args=(32, 3), kw=dict(num_channels_down=[128] * 5, num_channels_up=[128] * 5,
                                          num_channels_skip=[4] * 5, upsample_mode="bilinear",
                                          need_sigmoid=True, need_bias=True, pad="reflection"), hw=(64, 96)),
    "kate": dict(args=(32, 3), kw=dict(num_channels_down=[128] * 5, num_channels_up=[128] * 5,
                                       num_channels_skip=[128] * 5, filter_size_up=3, filter_size_down=3,
                                       upsample_mode="nearest", filter_skip_size=1,
                                       need_sigmoid=True, need_bias=True, pad="reflection"), hw=(64, 64)),
    "library": dict(args=(1, 3), kw=dict(num_channels_down=[16, 32, 64, 128, 128, 128],
                                         num_channels_up=[16, 32, 64, 128, 128, 128],
                                         num_channels_skip=[0] * 6, filter_size_up=3, filter_size_down=5,
                                         filter_skip_size=1, upsample_mode="nearest", need1x1_up=False,
                                         need_sigmoid=True, need_bias=True, pad="reflection"), hw=(256, 192)),
    "snail": dict(args=(3, 3), kw=dict(num_channels_down=[8, 16, 32, 64, 128],
                                       num_channels_up=[8, 16, 32, 64, 128],
                                       num_channels_skip=[0, 0, 0, 4, 4], upsample_mode="bilinear",
                                       need_sigmoid=True, need_bias=True, pad="reflection"), hw=(64, 96)),
    "zero_pad": dict(args=(2, 3), kw=dict(), hw=(64, 64)),   # skip() defaults: pad='zero', nearest
    # restoration.ipynb:149-160: downsample_mode='avg' (stride-1 conv + AvgPool2d)
    "avg_down": dict(args=(32, 3), kw=dict(num_channels_down=[64] * 4, num_channels_up=[64] * 4,
                                           num_channels_skip=[4] * 4, upsample_mode="bilinear", downsample_mode="avg",
                                           need_sigmoid=True, need_bias=True, pad="reflection"), hw=(64, 96)),
    # round 3: Concat's centre crop beyond one row / column (pooling, skip-less scales at non-divisible sizes,
    # models/common.py:29-37) and the Lanczos Downsampler inside conv (models/common.py:107-108)
    "max_down_crop": dict(args=(16, 3), kw=dict(num_channels_down=[32] * 4, num_channels_up=[32] * 4,
                                                num_channels_skip=[4] * 4, upsample_mode="bilinear", downsample_mode="max",
                                                need_sigmoid=True, need_bias=True, pad="reflection"), hw=(77, 93)),
    "noskip_crop": dict(args=(16, 3), kw=dict(num_channels_down=[32] * 4, num_channels_up=[32] * 4,
                                              num_channels_skip=[4, 0, 4, 0], upsample_mode="nearest",
                                              need_sigmoid=True, need_bias=True, pad="reflection"), hw=(61, 75)),
    "lanczos2_down": dict(args=(16, 3), kw=dict(num_channels_down=[64] * 3, num_channels_up=[64] * 3,
                                                num_channels_skip=[4] * 3, upsample_mode="bilinear",
                                                downsample_mode="lanczos2",
                                                need_sigmoid=True, need_bias=True, pad="reflection"), hw=(64, 96)),
    "lanczos3_down_odd": dict(args=(8, 3), kw=dict(num_channels_down=[16, 32], num_channels_up=[16, 32],
                                                   num_channels_skip=[4, 4], upsample_mode="nearest",
                                                   downsample_mode=["lanczos3", "lanczos2"],
                                                   need_sigmoid=True, need_bias=True, pad="zero"), hw=(45, 38)),
}


def spec_from(cfg) -> O.SkipSpec:
    kw = dict(cfg["kw"])
    return O.SkipSpec(cfg["args"][0], cfg["args"][1],
                      kw.get("num_channels_down", [16, 32, 64, 128, 128]),
                      kw.get("num_channels_up", [16, 32, 64, 128, 128]),
                      kw.get("num_channels_skip", [4, 4, 4, 4, 4]),
                      kw.get("filter_size_down", 3), kw.get("filter_size_up", 3),
                      kw.get("filter_skip_size", 1), kw.get("need_sigmoid", True), kw.get("need_bias", True),
                      kw.get("pad", "zero"), kw.get("upsample_mode", "nearest"), kw.get("need1x1_up", True),
                      kw.get("downsample_mode", "stride"))


def check(name, cfg):
    rm = _refload.load_ref_models()
    torch.manual_seed(0)
    net = rm.skip(*cfg["args"], **cfg["kw"])
    spec = spec_from(cfg)
    sd_ref = net.state_dict()
    shapes = O.param_shapes(spec)
    learn = {k: v for k, v in sd_ref.items() if not (k.endswith("running_mean") or k.endswith("running_var")
                                                      or k.endswith("num_batches_tracked"))}
    assert set(learn) == set(shapes), (sorted(set(learn) ^ set(shapes)))
    for k in shapes:
        assert tuple(learn[k].shape) == shapes[k], k
    H, W = cfg["hw"]
    x = O.get_noise(cfg["args"][0], "noise", (H, W)) if cfg["args"][0] != 2 else \
        O.get_noise(2, "meshgrid", (H, W)).float()
    out_ref = net(x)
    # (with pooling / skip-less scales at non-divisible sizes Concat's crop makes the output smaller than the input)
    target = torch.rand(1, cfg["args"][1], *out_ref.shape[2:])
    loss_ref = torch.nn.functional.mse_loss(out_ref, target)
    loss_ref.backward()

    onet = O.OracleNet(spec, {k: v.detach() for k, v in learn.items()})
    out_o = onet(x)
    loss_o = torch.nn.functional.mse_loss(out_o, target)
    loss_o.backward()

    assert torch.equal(out_ref, out_o), name
    assert torch.equal(loss_ref, loss_o), name
    gref = {k: p.grad for k, p in net.named_parameters()}
    for k, p in zip(onet.names, onet.params):
        assert torch.equal(gref[k], p.grad), (name, k)
    # one Adam step each
    torch.optim.Adam(net.parameters(), lr=0.01).step()
    torch.optim.Adam(onet.params, lr=0.01).step()
    pref = dict(net.named_parameters())
    for k, p in zip(onet.names, onet.params):
        assert torch.equal(pref[k], p), (name, k)
    print(f"[ok] {name}: out/loss/{len(onet.names)} grads/Adam step bitwise equal "
          f"({sum(p.numel() for p in onet.params)} params, input {H}x{W})")


def check_downsampler():
    rm = _refload.load_ref_models()
    for factor, kt in ((4, "lanczos2"), (2, "lanczos2"), (4, "lanczos3")):
        d = rm.downsampler.Downsampler(n_planes=3, factor=factor, kernel_type=kt, phase=0.5, preserve_size=True)
        x = torch.rand(1, 3, 64, 96)
        ref = d(x)
        o = O.downsampler_forward(x, factor, kt, 0.5, True)
        assert torch.equal(ref, o), (factor, kt)
        sup = {"lanczos2": 2, "lanczos3": 3}[kt]
        assert np.array_equal(d.kernel, O.lanczos_kernel(factor, 0.5, 2 * sup * factor + 1, sup))
    print("[ok] Downsampler lanczos2/3 phase 0.5: taps and forward bitwise equal")


def check_get_noise():
    cu = _refload.load_ref_common_utils()
    torch.manual_seed(3); a = cu.get_noise(32, "noise", (16, 24))
    torch.manual_seed(3); b = O.get_noise(32, "noise", (16, 24))
    assert torch.equal(a, b)
    assert torch.equal(cu.get_noise(2, "meshgrid", (8, 12)), O.get_noise(2, "meshgrid", (8, 12)))
    print("[ok] get_noise noise/meshgrid equal")


if __name__ == "__main__":
    assert _refload.available(), "reference checkout not found"
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for n, c in CONFIGS.items():
        check(n, c)
    check_downsampler()
    check_get_noise()
    print("oracle pinned against the reference")
