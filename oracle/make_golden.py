"""Generate the committed golden vectors in tests/golden/ from the REAL reference.

Run:  python oracle/make_golden.py       (build container only; needs /root/reference)

Outputs (all float32 unless noted, produced by /root/reference code on torch CPU):
  tests/golden/net_<name>.npz   small skip() nets: state_dict, input z, target, mask,
                                out, loss, every gradient, parameters after 1 and after 3
                                optimize('adam') iterations (utils/common_utils.py:223-230)
  tests/golden/downsampler.npz  Downsampler(3,4,'lanczos2',0.5,preserve_size) taps, fwd, bwd
  tests/golden/downsampler_dense.npz  the same module as a trainable dense conv (opt_over='down')
  tests/golden/get_noise.npz    get_noise() draws for fixed seeds
  tests/golden/default64_digest.json  digests of the FULL default net (2 217 831 params,
                                torch.manual_seed(0) construction) at 64x64: pins parameter
                                RNG order + state_dict naming + forward/backward numerics
"""
import copy
import functools
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

NETS = {
    # tiny but structurally complete variants of the BASELINE configs
    "tiny_default": dict(args=(8, 3), hw=(32, 48), seed=1,
                         kw=dict(num_channels_down=[16, 32, 32], num_channels_up=[16, 32, 32],
                                 num_channels_skip=[4, 4, 4], upsample_mode="bilinear",
                                 need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_kate": dict(args=(8, 3), hw=(32, 32), seed=2,
                      kw=dict(num_channels_down=[16, 16, 16], num_channels_up=[16, 16, 16],
                              num_channels_skip=[16, 16, 16], upsample_mode="nearest",
                              need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_library": dict(args=(1, 3), hw=(64, 48), seed=3,
                         kw=dict(num_channels_down=[8, 16, 32], num_channels_up=[8, 16, 32],
                                 num_channels_skip=[0, 0, 0], filter_size_up=3, filter_size_down=5,
                                 upsample_mode="nearest", need1x1_up=False,
                                 need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_snail": dict(args=(3, 3), hw=(32, 48), seed=4,
                       kw=dict(num_channels_down=[8, 16, 32], num_channels_up=[8, 16, 32],
                               num_channels_skip=[0, 4, 4], upsample_mode="bilinear",
                               need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_zero": dict(args=(2, 1), hw=(32, 32), seed=5,
                      kw=dict(num_channels_down=[8, 16], num_channels_up=[8, 16],
                              num_channels_skip=[4, 4], need_sigmoid=True, need_bias=True)),
    # restoration.ipynb:149-160: stride-1 convs + AvgPool2d(2, 2) instead of strided convs
    "tiny_avg": dict(args=(8, 3), hw=(32, 48), seed=6,
                     kw=dict(num_channels_down=[16, 32], num_channels_up=[16, 32],
                             num_channels_skip=[4, 4], upsample_mode="bilinear", downsample_mode="avg",
                             need_sigmoid=True, need_bias=True, pad="reflection")),
    # conv(..., downsample_mode='max'), models/common.py:105-106 (no notebook uses it)
    "tiny_max": dict(args=(8, 3), hw=(32, 48), seed=8,
                     kw=dict(num_channels_down=[16, 32], num_channels_up=[16, 32],
                             num_channels_skip=[4, 4], upsample_mode="bilinear", downsample_mode="max",
                             need_sigmoid=True, need_bias=True, pad="reflection")),
    # act_fun='Swish' / 'ELU' (models/common.py:62-92; no notebook uses them)
    "tiny_swish": dict(args=(8, 3), hw=(32, 48), seed=9,
                       kw=dict(num_channels_down=[16, 32], num_channels_up=[16, 32], num_channels_skip=[4, 4],
                               upsample_mode="bilinear", act_fun="Swish",
                               need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_elu": dict(args=(8, 3), hw=(32, 32), seed=10,
                     kw=dict(num_channels_down=[16, 16], num_channels_up=[16, 16], num_channels_skip=[4, 4],
                             upsample_mode="nearest", act_fun="ELU",
                             need_sigmoid=True, need_bias=True, pad="zero")),
    # act_fun as a module CLASS / factory (models/common.py:90-91: `return act_fun()`; round 6)
    "tiny_relu": dict(args=(8, 3), hw=(32, 48), seed=21,
                      kw=dict(num_channels_down=[16, 32], num_channels_up=[16, 32], num_channels_skip=[4, 4],
                              upsample_mode="bilinear", act_fun=torch.nn.ReLU,
                              need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_leaky01": dict(args=(8, 3), hw=(32, 32), seed=22,
                         kw=dict(num_channels_down=[16, 16], num_channels_up=[16, 16], num_channels_skip=[4, 4],
                                 upsample_mode="nearest", act_fun=functools.partial(torch.nn.LeakyReLU, 0.1),
                                 need_sigmoid=True, need_bias=True, pad="zero")),
    # filter_skip_size = 3 (models/skip.py:58; every notebook keeps 1)
    "tiny_skip3": dict(args=(8, 3), hw=(32, 48), seed=11,
                       kw=dict(num_channels_down=[16, 32], num_channels_up=[16, 32], num_channels_skip=[4, 4],
                               filter_skip_size=3, upsample_mode="bilinear",
                               need_sigmoid=True, need_bias=True, pad="reflection")),
    # sizes that are not divisible by 2^depth: Concat's centre crop (models/common.py:29-37) drops the last row /
    # column of the x2 up-sampled tensor (SURVEY 8f n3 "ragged crop")
    "tiny_ragged": dict(args=(8, 3), hw=(37, 50), seed=12,
                        kw=dict(num_channels_down=[16, 32, 32], num_channels_up=[16, 32, 32],
                                num_channels_skip=[4, 4, 4], upsample_mode="bilinear",
                                need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_ragged_nn": dict(args=(8, 3), hw=(45, 39), seed=13,
                           kw=dict(num_channels_down=[16, 16], num_channels_up=[16, 16],
                                   num_channels_skip=[4, 4], upsample_mode="nearest",
                                   need_sigmoid=True, need_bias=True, pad="zero")),
    # Concat's centre crop with OFFSETS (models/common.py:29-37): avg pooling floors the odd sizes 45x38 -> 22x19 -> 11x9, so
    # the x2 up-sampled tensors are smaller than the skip branches (cropped at offsets (2,3) / (1,1) / 0) and the net's
    # output is 40x32; two skip-less scales up-sample 8x10 -> 16x20 -> 32x40 against a 29x37 skip branch (deep branch
    # cropped at offset (1,1))
    "tiny_poolcrop": dict(args=(8, 3), hw=(45, 38), seed=14,
                          kw=dict(num_channels_down=[16, 16, 16], num_channels_up=[16, 16, 16],
                                  num_channels_skip=[4, 4, 4], upsample_mode="bilinear", downsample_mode="avg",
                                  need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_noskipcrop": dict(args=(8, 3), hw=(29, 37), seed=15,
                            kw=dict(num_channels_down=[16, 16, 16], num_channels_up=[16, 16, 16],
                                    num_channels_skip=[4, 0, 0], upsample_mode="bilinear",
                                    need_sigmoid=True, need_bias=True, pad="reflection")),
    # conv(..., downsample_mode='lanczos2' | 'lanczos3') (models/common.py:107-108): a stride-1 conv followed by a
    # Downsampler whose dense 8x8 / 12x12 stride-2 Conv2d is part of net.parameters() and is trained
    "tiny_lanczos2": dict(args=(8, 3), hw=(32, 48), seed=16,
                          kw=dict(num_channels_down=[16, 16], num_channels_up=[16, 16], num_channels_skip=[4, 4],
                                  upsample_mode="bilinear", downsample_mode="lanczos2",
                                  need_sigmoid=True, need_bias=True, pad="reflection")),
    "tiny_lanczos3": dict(args=(8, 3), hw=(32, 32), seed=17,
                          kw=dict(num_channels_down=[8, 16], num_channels_up=[8, 16], num_channels_skip=[4, 4],
                                  upsample_mode="nearest", downsample_mode=["lanczos3", "lanczos2"],
                                  need_sigmoid=True, need_bias=True, pad="zero")),
    # feature_inversion.ipynb:169-174: per-scale filter sizes 7 / 5 / 3, zero padding, avg-pool
    # down-sampling, nearest up-sampling, meshgrid input
    "tiny_feat7": dict(args=(2, 3), hw=(32, 48), seed=7,
                       kw=dict(num_channels_down=[8, 16, 16], num_channels_up=[8, 16, 16],
                               num_channels_skip=[4, 4, 4], filter_size_down=[7, 5, 3], filter_size_up=[7, 5, 3],
                               upsample_mode="nearest", downsample_mode="avg",
                               need_sigmoid=True, need_bias=True, pad="zero")),
}


def gen_net(name, cfg):
    rm = _refload.load_ref_models()
    cu = _refload.load_ref_common_utils()
    torch.manual_seed(cfg["seed"])
    net = rm.skip(*cfg["args"], **cfg["kw"])
    # Non-degenerate BatchNorm affine parameters.  At the default init (gamma=1, beta=0) every
    # BatchNorm that is followed by conv+BatchNorm has an ANALYTICALLY ZERO gamma-gradient (the
    # loss is invariant to a joint positive rescale of (gamma, beta) and beta=0), so the
    # reference's value would be pure roundoff and useless as a golden vector.
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.3)
    H, W = cfg["hw"]
    cin, cout = cfg["args"]
    z = cu.get_noise(cin, "meshgrid" if cin == 2 else "noise", (H, W)).float()
    Ho, Wo = copy.deepcopy(net)(z).shape[2:]        # (the output is smaller than the input when Concat crops a skip branch)
    target = torch.rand(1, cout, Ho, Wo)
    mask = (torch.rand(1, 1, Ho, Wo) > 0.3).float()
    rec = {"z": z.numpy(), "target": target.numpy(), "mask": mask.numpy()}
    for k, v in net.state_dict().items():
        rec["sd/" + k] = v.detach().numpy().copy()

    out = net(z)
    loss = torch.nn.functional.mse_loss(out * mask, target * mask)   # inpainting.ipynb:310 form
    loss.backward()
    rec["out"] = out.detach().numpy().copy()
    rec["loss"] = np.array(loss.item(), dtype=np.float64)
    for k, p in net.named_parameters():
        rec["grad/" + k] = p.grad.numpy().copy()
    for p in net.parameters():
        p.grad = None

    # optimize('adam') trajectory, utils/common_utils.py:223-230, closure in the notebook style
    mse = torch.nn.MSELoss()
    # optimize() creates a fresh Adam every call, so run it once for 1 step on a clone and once for 3
    for m in net.modules():               # Downsampler.forward keeps its padded input on self.x (models/downsampler.py:70):
        if hasattr(m, "x") and torch.is_tensor(m.x):      # a non-leaf tensor that deepcopy refuses
            m.x = None
    for nsteps in (1, 3):
        net2 = copy.deepcopy(net)

        def closure2():
            o = net2(z)
            l = mse(o * mask, target * mask)
            l.backward()
            return l

        cu.optimize("adam", cu.get_params("net", net2, z), closure2, 0.01, nsteps)
        for k, p in net2.named_parameters():
            rec[f"adam{nsteps}/" + k] = p.detach().numpy().copy()
        if nsteps == 3:
            rec["out_after3"] = net2(z).detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"net_{name}.npz"), **rec)
    print(f"net_{name}.npz: {sum(v.size for v in rec.values())} values")
    return cfg


def gen_downsampler():
    rm = _refload.load_ref_models()
    rec = {}
    for factor, kt in ((4, "lanczos2"), (2, "lanczos2"), (8, "lanczos2")):
        d = rm.downsampler.Downsampler(n_planes=3, factor=factor, kernel_type=kt, phase=0.5, preserve_size=True)
        torch.manual_seed(factor)
        x = torch.rand(1, 3, 64, 96, requires_grad=True)
        y = d(x)
        g = torch.rand_like(y)
        (y * g).sum().backward()
        tag = f"{kt}_f{factor}"
        rec[tag + "/kernel"] = d.kernel
        rec[tag + "/x"] = x.detach().numpy()
        rec[tag + "/y"] = y.detach().numpy()
        rec[tag + "/gy"] = g.numpy()
        rec[tag + "/gx"] = x.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "downsampler.npz"), **rec)
    print("downsampler.npz")


def gen_downsampler_dense():
    """opt_over='down' (utils/common_utils.py:44-46): the Downsampler's dense Conv2d weight is what gets optimised.
    Vectors: forward / input gradient / weight and bias gradients at a weight that is NOT on the channel diagonal any
    more, and the parameters after 3 optimize('adam') steps over get_params('down', ...)."""
    rm = _refload.load_ref_models()
    cu = _refload.load_ref_common_utils()
    rec = {}
    for factor, hw in ((4, (64, 96)), (2, (36, 44)), (8, (64, 64))):
        d = rm.downsampler.Downsampler(n_planes=3, factor=factor, kernel_type="lanczos2", phase=0.5, preserve_size=True)
        torch.manual_seed(10 + factor)
        with torch.no_grad():
            d.downsampler_.weight += 0.01 * torch.randn_like(d.downsampler_.weight)
            d.downsampler_.bias += 0.1 * torch.randn_like(d.downsampler_.bias)
        x = torch.rand(1, 3, *hw, requires_grad=True)
        y = d(x)
        g = torch.rand_like(y)
        (y * g).sum().backward()
        tag = f"f{factor}"
        rec[tag + "/w"] = d.downsampler_.weight.detach().numpy().copy()
        rec[tag + "/b"] = d.downsampler_.bias.detach().numpy().copy()
        rec[tag + "/x"] = x.detach().numpy()
        rec[tag + "/y"] = y.detach().numpy()
        rec[tag + "/gy"] = g.numpy()
        rec[tag + "/gx"] = x.grad.numpy().copy()
        rec[tag + "/dw"] = d.downsampler_.weight.grad.numpy().copy()
        rec[tag + "/db"] = d.downsampler_.bias.grad.numpy().copy()
        d.x = None
        d2 = copy.deepcopy(d)
        xin = x.detach().clone()
        target = torch.rand_like(y)
        mse = torch.nn.MSELoss()

        def closure():
            l = mse(d2(xin), target)
            l.backward()
            return l

        cu.optimize("adam", cu.get_params("down", None, xin, d2), closure, 0.01, 3)
        rec[tag + "/target"] = target.numpy()
        rec[tag + "/adam3_w"] = d2.downsampler_.weight.detach().numpy().copy()
        rec[tag + "/adam3_b"] = d2.downsampler_.bias.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "downsampler_dense.npz"), **rec)
    print("downsampler_dense.npz")


def gen_get_noise():
    cu = _refload.load_ref_common_utils()
    rec = {}
    torch.manual_seed(0); rec["u_s0_32x16x24"] = cu.get_noise(32, "noise", (16, 24)).numpy()
    torch.manual_seed(7); rec["n_s7_3x8x8"] = cu.get_noise(3, "noise", 8, noise_type="n", var=0.5).numpy()
    rec["mesh_8x12"] = cu.get_noise(2, "meshgrid", (8, 12)).numpy()
    np.savez_compressed(os.path.join(OUT, "get_noise.npz"), **rec)
    print("get_noise.npz")


def digest(t: torch.Tensor):
    d = t.detach().double().flatten()
    idx = torch.linspace(0, d.numel() - 1, 5).long()
    return {"shape": list(t.shape), "sum": d.sum().item(), "abssum": d.abs().sum().item(),
            "sq": (d * d).sum().item(), "samples": d[idx].tolist()}


def gen_default_digest():
    rm = _refload.load_ref_models()
    cu = _refload.load_ref_common_utils()
    torch.manual_seed(0)
    net = rm.get_net(32, "skip", "reflection", skip_n33d=128, skip_n33u=128, skip_n11=4,
                     num_scales=5, upsample_mode="bilinear")
    z = cu.get_noise(32, "noise", (64, 64))
    np.random.seed(0)
    target = torch.from_numpy(np.random.rand(1, 3, 64, 64).astype(np.float32))
    out = net(z)
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    rec = {"config": "get_net(32,'skip','reflection',skip_n33d=128,skip_n33u=128,skip_n11=4,num_scales=5,"
                     "upsample_mode='bilinear'); torch.manual_seed(0) before construction; "
                     "z=get_noise(32,'noise',(64,64)) drawn right after; target=np.random.seed(0) rand(1,3,64,64)",
           "n_params": sum(p.numel() for p in net.parameters()),
           "keys": list(net.state_dict().keys()),
           "z": digest(z), "out": digest(out), "loss": loss.item(),
           "params": {k: digest(p) for k, p in net.named_parameters()},
           "grads": {k: digest(p.grad) for k, p in net.named_parameters()}}
    with open(os.path.join(OUT, "default64_digest.json"), "w") as f:
        json.dump(rec, f)
    print("default64_digest.json", rec["n_params"], "params", len(rec["keys"]), "state_dict keys")


if __name__ == "__main__":
    assert _refload.available(), "reference checkout not found"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)          # fixed reduction order for the committed vectors
    only = sys.argv[1:]                # optional: regenerate just the named nets
    for n, c in NETS.items():
        if not only or n in only:
            gen_net(n, c)
    if only == ["downsampler_dense"]:
        gen_downsampler_dense()
    if only:
        sys.exit(0)
    gen_downsampler()
    gen_downsampler_dense()
    gen_get_noise()
    gen_default_digest()
