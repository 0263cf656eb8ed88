"""CPU oracle: a plain-torch restatement of the deep-image-prior hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``deep-image-prior_amd/`` imports this
module.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / the timed CPU
baseline -- never as the product path.

What it restates (all citations are into the reference checkout, which is NOT
needed at run time):

* ``skip()``                models/skip.py:5-100      -> :class:`SkipSpec` + :func:`skip_forward`
* ``conv()/bn()/act()``     models/common.py:76-124   -> ``F.pad(reflect)``/``F.conv2d``/``F.batch_norm(training=True)``/``F.leaky_relu``
* ``Concat``                models/common.py:11-42    -> centre-crop + ``torch.cat``
* ``nn.Upsample``           models/skip.py:81         -> ``F.interpolate(scale_factor=2, mode=...)``
* ``Downsampler/get_kernel`` models/downsampler.py:9-135 -> :func:`lanczos_kernel` / :func:`downsampler_forward`
* ``get_noise``             utils/common_utils.py:127-153
* ``optimize('adam')``      utils/common_utils.py:223-230 (torch.optim.Adam defaults)

The arithmetic of the reference lives in PyTorch itself (pinned pytorch=0.4 in
environment.yml:14; this image has torch 2.10 CPU).  Because both the reference
modules and this restatement lower to the same ATen CPU kernels, the
restatement is BITWISE equal to the real reference in this container
(``oracle/verify_against_reference.py`` asserts ``torch.equal`` on output, loss
and every gradient; ``tests/test_oracle.py`` re-checks against the committed
golden vectors in ``tests/golden/``, which were produced by the real reference
with ``oracle/make_golden.py``).

Parity pin: the reference ships no tests or golden vectors of its own
(SURVEY.md section 8c), so the pin is "outputs of the reference itself run
here" (tests/golden/*.npz, generated from /root/reference by
oracle/make_golden.py).

The network is described by a flat parameter dict keyed exactly like the
reference ``state_dict()`` (e.g. ``1.0.1.1.weight``), so a reference checkpoint
drives the oracle directly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# Architecture description (mirrors models/skip.py:45-100)
# --------------------------------------------------------------------------
@dataclass
class SkipSpec:
    num_input_channels: int = 2
    num_output_channels: int = 3
    num_channels_down: Sequence[int] = (16, 32, 64, 128, 128)
    num_channels_up: Sequence[int] = (16, 32, 64, 128, 128)
    num_channels_skip: Sequence[int] = (4, 4, 4, 4, 4)
    filter_size_down: Sequence[int] | int = 3
    filter_size_up: Sequence[int] | int = 3
    filter_skip_size: int = 1
    need_sigmoid: bool = True
    need_bias: bool = True
    pad: str = "zero"
    upsample_mode: Sequence[str] | str = "nearest"
    need1x1_up: bool = True
    downsample_mode: Sequence[str] | str = "stride"      # 'stride' | 'avg' | 'max' | 'lanczos2' | 'lanczos3' (models/common.py:99-112)
    act_fun: object = "LeakyReLU"                        # 'LeakyReLU' | 'Swish' | 'ELU' | 'none' | a module class (models/common.py:76-92)

    def __post_init__(self):
        n = len(self.num_channels_down)
        assert len(self.num_channels_up) == n == len(self.num_channels_skip)
        if isinstance(self.filter_size_down, int):
            self.filter_size_down = [self.filter_size_down] * n
        if isinstance(self.filter_size_up, int):
            self.filter_size_up = [self.filter_size_up] * n
        if isinstance(self.upsample_mode, str):
            self.upsample_mode = [self.upsample_mode] * n
        if isinstance(self.downsample_mode, str):
            self.downsample_mode = [self.downsample_mode] * n

    @property
    def n_scales(self) -> int:
        return len(self.num_channels_down)


def default_spec(input_depth=32, n_channels=3, pad="reflection", upsample_mode="bilinear",
                 skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5) -> SkipSpec:
    """get_net(..., 'skip', ...) expansion, models/__init__.py:12-17."""
    return SkipSpec(input_depth, n_channels,
                    [skip_n33d] * num_scales, [skip_n33u] * num_scales, [skip_n11] * num_scales,
                    upsample_mode=upsample_mode, need_sigmoid=True, need_bias=True, pad=pad)


# --------------------------------------------------------------------------
# state_dict key naming (models/common.py:6-9 'Module.add' + skip.py order)
# --------------------------------------------------------------------------
def _conv_key(prefix: str, slot: int, pad: str) -> str:
    # conv() returns Sequential([padder], conv): the Conv2d is child "1" with a
    # reflection padder (even for k=1: ReflectionPad2d(0)), child "0" without.
    return f"{prefix}{slot}.{1 if pad == 'reflection' else 0}"


@dataclass
class ScaleKeys:
    skip_conv: Optional[str]
    skip_bn: Optional[str]
    down_a: str
    down_a_ds: Optional[str]          # the Downsampler's dense Conv2d behind down_a (downsample_mode 'lanczos2' | 'lanczos3')
    down_a_bn: str
    down_b: str
    down_b_bn: str
    cat_bn: str
    up: str
    up_bn: str
    up1: Optional[str]
    up1_bn: Optional[str]


def scale_keys(spec: SkipSpec) -> (List[ScaleKeys], str):
    """Returns per-scale parameter-key prefixes and the key of the output conv."""
    keys = []
    P = ""
    for i in range(spec.n_scales):
        has_skip = spec.num_channels_skip[i] != 0
        if has_skip:
            sk, dp = P + "1.0.", P + "1.1."
        else:
            sk, dp = None, P + "1."
        k = ScaleKeys(
            skip_conv=_conv_key(sk, 1, spec.pad) if has_skip else None,
            skip_bn=(sk + "2") if has_skip else None,
            down_a=_conv_key(dp, 1, spec.pad),
            # conv() = Sequential([padder], Conv2d, Downsampler): the Downsampler follows the Conv2d (models/common.py:122-123)
            down_a_ds=(f"{dp}1.{2 if spec.pad == 'reflection' else 1}.downsampler_"
                       if spec.downsample_mode[i] in ("lanczos2", "lanczos3") else None),
            down_a_bn=dp + "2",
            down_b=_conv_key(dp, 4, spec.pad), down_b_bn=dp + "5",
            cat_bn=P + "2",
            up=_conv_key(P, 3, spec.pad), up_bn=P + "4",
            up1=_conv_key(P, 6, spec.pad) if spec.need1x1_up else None,
            up1_bn=(P + "7") if spec.need1x1_up else None,
        )
        keys.append(k)
        P = dp + "7."
    n_top = 2 + 3 + (3 if spec.need1x1_up else 0)        # children "1".."n_top" of the top Sequential
    out_key = _conv_key("", n_top + 1, spec.pad)
    return keys, out_key


def lanczos_ds_width(mode: str, factor: int = 2) -> int:
    """Filter width of Downsampler(factor, 'lanczos2' | 'lanczos3', phase=0.5): kernel_width = 4*factor+1 | 6*factor+1
    (models/downsampler.py:14-22), one less for phase 0.5 (get_kernel, :77-78)."""
    return (4 if mode == "lanczos2" else 6) * factor


def param_shapes(spec: SkipSpec) -> Dict[str, tuple]:
    """All learnable tensors (state_dict order is not guaranteed here)."""
    keys, out_key = scale_keys(spec)
    shapes: Dict[str, tuple] = {}

    def conv(key, cin, cout, k):
        shapes[key + ".weight"] = (cout, cin, k, k)
        if spec.need_bias:
            shapes[key + ".bias"] = (cout,)

    def bn(key, c):
        shapes[key + ".weight"] = (c,)
        shapes[key + ".bias"] = (c,)

    cin = spec.num_input_channels
    n = spec.n_scales
    for i, k in enumerate(keys):
        nd, nu, ns = spec.num_channels_down[i], spec.num_channels_up[i], spec.num_channels_skip[i]
        kdeep = spec.num_channels_up[i + 1] if i < n - 1 else nd
        if ns:
            conv(k.skip_conv, cin, ns, spec.filter_skip_size)
            bn(k.skip_bn, ns)
        conv(k.down_a, cin, nd, spec.filter_size_down[i]); bn(k.down_a_bn, nd)
        if k.down_a_ds is not None:          # Downsampler(n_planes=nd, factor=2, ...): dense nd x nd conv, bias always present
            kw = lanczos_ds_width(spec.downsample_mode[i])
            shapes[k.down_a_ds + ".weight"] = (nd, nd, kw, kw)
            shapes[k.down_a_ds + ".bias"] = (nd,)
        conv(k.down_b, nd, nd, spec.filter_size_down[i]); bn(k.down_b_bn, nd)
        bn(k.cat_bn, ns + kdeep)
        conv(k.up, ns + kdeep, nu, spec.filter_size_up[i]); bn(k.up_bn, nu)
        if spec.need1x1_up:
            conv(k.up1, nu, nu, 1); bn(k.up1_bn, nu)
        cin = nd
    conv(out_key, spec.num_channels_up[0], spec.num_output_channels, 1)
    return shapes


# --------------------------------------------------------------------------
# Forward (functional)
# --------------------------------------------------------------------------
def _conv(x, sd, key, k, stride, pad):
    """conv(): models/common.py:99-124 (stride down-sampling only)."""
    to_pad = int((k - 1) / 2)
    w = sd[key + ".weight"]
    b = sd.get(key + ".bias")
    if pad == "reflection":
        if to_pad:
            x = F.pad(x, (to_pad,) * 4, mode="reflect")
        return F.conv2d(x, w, b, stride=stride, padding=0)
    return F.conv2d(x, w, b, stride=stride, padding=to_pad)


def _bn_act(x, sd, key, act=True, eps=1e-5, masks=None, act_fun="LeakyReLU", zrec=None):
    """bn() + act(): BatchNorm2d in TRAIN mode (batch stats), then act(act_fun), models/common.py:76-92:
    LeakyReLU(0.2) | Swish (x * sigmoid(x), :62-73) | nn.ELU() | 'none' (empty nn.Sequential).

    `masks` (test-only): {bn_key: bool tensor} imposes the LeakyReLU branch pattern of ANOTHER
    implementation.  LeakyReLU's derivative jumps at 0, so two correct fp32 implementations whose
    pre-activations differ by roundoff can pick different branches for an element with z ~ 1e-7;
    that single element changes the gradient by O(1/sqrt(numel)) ~ 1e-3 relative.  With the pattern
    imposed, the oracle differentiates exactly the piecewise-linear branch the other side took."""
    x = F.batch_norm(x, None, None, sd[key + ".weight"], sd[key + ".bias"], True, 0.1, eps)
    if zrec is not None and act:
        zrec[key] = x.detach().float()       # (test-only) the pre-activation: tests/parity.mask_report
    if not act or act_fun == "none":
        return x
    if not isinstance(act_fun, str):         # a module class / factory: models/common.py:90-91 `return act_fun()`
        return act_fun()(x)
    if act_fun == "Swish":
        return x * torch.sigmoid(x)
    if act_fun == "ELU":
        return F.elu(x)
    assert act_fun == "LeakyReLU", act_fun
    if masks is not None and key in masks:
        # multiply by a constant slope map (1 on the positive branch, 0.2 on the other)
        slope = masks[key].contiguous().to(x.dtype) * 0.8 + 0.2
        return x * slope
    return F.leaky_relu(x, 0.2)


def _concat(inputs):
    """Concat.forward: models/common.py:19-39 (centre crop to the min H, W)."""
    h = min(t.shape[2] for t in inputs)
    w = min(t.shape[3] for t in inputs)
    outs = []
    for t in inputs:
        d2, d3 = (t.shape[2] - h) // 2, (t.shape[3] - w) // 2
        outs.append(t[:, :, d2:d2 + h, d3:d3 + w])
    return torch.cat(outs, dim=1)


def skip_forward(spec: SkipSpec, sd: Dict[str, torch.Tensor], x: torch.Tensor,
                 taps: Optional[dict] = None, masks: Optional[dict] = None,
                 zrec: Optional[dict] = None) -> torch.Tensor:
    """Forward of the skip encoder-decoder; ``sd`` is keyed like the reference state_dict.

    ``taps`` (optional dict) receives named intermediate tensors for per-layer parity tests;
    ``zrec`` (optional dict) the output of every BatchNorm that is followed by an activation.
    """
    keys, out_key = scale_keys(spec)

    def scale(i, x):
        k = keys[i]
        ns = spec.num_channels_skip[i]
        fd, fu = spec.filter_size_down[i], spec.filter_size_up[i]
        if spec.downsample_mode[i] == "stride":
            d = _conv(x, sd, k.down_a, fd, 2, spec.pad)
        elif spec.downsample_mode[i] == "avg":   # conv(): stride-1 conv followed by nn.AvgPool2d(2, 2), common.py:101-104
            d = F.avg_pool2d(_conv(x, sd, k.down_a, fd, 1, spec.pad), 2, 2)
        elif spec.downsample_mode[i] in ("lanczos2", "lanczos3"):
            # ... or Downsampler(n_planes=out_f, factor=2, kernel_type, phase=0.5, preserve_size=True), common.py:107-108:
            # ReplicationPad2d((k - factor) / 2) + a dense (trainable) Conv2d(out_f, out_f, k, stride=2), downsampler.py:44-71
            d = _conv(x, sd, k.down_a, fd, 1, spec.pad)
            wds = sd[k.down_a_ds + ".weight"]
            p = (wds.shape[-1] - 2) // 2
            d = F.conv2d(F.pad(d, (p,) * 4, mode="replicate"), wds, sd[k.down_a_ds + ".bias"], stride=2)
        else:                               # ... or nn.MaxPool2d(2, 2), common.py:105-106
            assert spec.downsample_mode[i] == "max", spec.downsample_mode[i]
            d = F.max_pool2d(_conv(x, sd, k.down_a, fd, 1, spec.pad), 2, 2)
        d = _bn_act(d, sd, k.down_a_bn, masks=masks, act_fun=spec.act_fun, zrec=zrec)
        d = _conv(d, sd, k.down_b, fd, 1, spec.pad)
        d = _bn_act(d, sd, k.down_b_bn, masks=masks, act_fun=spec.act_fun, zrec=zrec)
        if i < spec.n_scales - 1:
            d = scale(i + 1, d)
        d = F.interpolate(d, scale_factor=2, mode=spec.upsample_mode[i])
        if ns:
            s = _conv(x, sd, k.skip_conv, spec.filter_skip_size, 1, spec.pad)
            s = _bn_act(s, sd, k.skip_bn, masks=masks, act_fun=spec.act_fun, zrec=zrec)
            y = _concat([s, d])
        else:
            y = d
        if taps is not None:
            taps[f"cat{i}"] = y
        y = _bn_act(y, sd, k.cat_bn, act=False)
        y = _conv(y, sd, k.up, fu, 1, spec.pad)
        if taps is not None:
            taps[f"up{i}_raw"] = y
        y = _bn_act(y, sd, k.up_bn, masks=masks, act_fun=spec.act_fun, zrec=zrec)
        if spec.need1x1_up:
            y = _conv(y, sd, k.up1, 1, 1, spec.pad)
            y = _bn_act(y, sd, k.up1_bn, masks=masks, act_fun=spec.act_fun, zrec=zrec)
        return y

    y = scale(0, x)
    y = _conv(y, sd, out_key, 1, 1, spec.pad)
    if spec.need_sigmoid:
        y = torch.sigmoid(y)
    return y


# --------------------------------------------------------------------------
# Deterministic parameter init for tests that have no reference checkout
# --------------------------------------------------------------------------
def init_params(spec: SkipSpec, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Kaiming-uniform-like init (same distribution family as nn.Conv2d defaults).

    NOT the reference's RNG stream -- use a real state_dict for that; this is
    for self-contained parity tests where only 'same weights on both sides' matters.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shp in param_shapes(spec).items():
        if len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bound
        elif name.endswith(".bias") and (name[:-5] + ".weight") in sd and sd[name[:-5] + ".weight"].dim() == 4:
            w = sd[name[:-5] + ".weight"]
            bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
            t = (torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bound
        elif name.endswith(".weight"):      # BN gamma: perturb around 1 so tests see gamma
            t = 1.0 + 0.2 * (torch.rand(shp, generator=g, dtype=torch.float64) - 0.5)
        else:                                # BN beta
            t = 0.2 * (torch.rand(shp, generator=g, dtype=torch.float64) - 0.5)
        sd[name] = t.to(dtype)
    return sd


# --------------------------------------------------------------------------
# Lanczos down-sampler (models/downsampler.py:9-135)
# --------------------------------------------------------------------------
def lanczos_kernel(factor: int, phase: float, kernel_width: int, support: int) -> np.ndarray:
    """get_kernel(factor,'lanczos',phase,kernel_width,support): downsampler.py:73-135 (float64)."""
    assert phase in (0, 0.5)
    n = kernel_width - 1 if phase == 0.5 else kernel_width
    kernel = np.zeros([n, n])
    center = (kernel_width + 1) / 2.0
    for i in range(1, n + 1):
        for j in range(1, n + 1):
            if phase == 0.5:
                di = abs(i + 0.5 - center) / factor
                dj = abs(j + 0.5 - center) / factor
            else:
                di = abs(i - center) / factor
                dj = abs(j - center) / factor
            val = 1
            if di != 0:
                val = val * support * np.sin(np.pi * di) * np.sin(np.pi * di / support)
                val = val / (np.pi * np.pi * di * di)
            if dj != 0:
                val = val * support * np.sin(np.pi * dj) * np.sin(np.pi * dj / support)
                val = val / (np.pi * np.pi * dj * dj)
            kernel[i - 1][j - 1] = val
    kernel /= kernel.sum()
    return kernel


def downsampler_forward(x: torch.Tensor, factor: int = 4, kernel_type: str = "lanczos2",
                        phase: float = 0.5, preserve_size: bool = True) -> torch.Tensor:
    """Downsampler.forward (downsampler.py:65-71) for lanczos2/lanczos3: ReplicationPad2d +
    dense Conv2d(n,n,k,stride=factor) with the taps on the channel diagonal, zero bias."""
    support = {"lanczos2": 2, "lanczos3": 3}[kernel_type]
    kw = 2 * support * factor + 1
    k = lanczos_kernel(factor, phase, kw, support)
    n = x.shape[1]
    w = torch.zeros(n, n, k.shape[0], k.shape[1], dtype=x.dtype)
    kt = torch.from_numpy(k).to(x.dtype)
    for i in range(n):
        w[i, i] = kt
    if preserve_size:
        if k.shape[0] % 2 == 1:
            pad = int((k.shape[0] - 1) / 2.0)
        else:
            pad = int((k.shape[0] - factor) / 2.0)
        x = F.pad(x, (pad,) * 4, mode="replicate")
    return F.conv2d(x, w, torch.zeros(n, dtype=x.dtype), stride=factor)


# --------------------------------------------------------------------------
# get_noise (utils/common_utils.py:127-153)
# --------------------------------------------------------------------------
def get_noise(input_depth, method, spatial_size, noise_type="u", var=1.0 / 10):
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    if method == "noise":
        t = torch.zeros([1, input_depth, spatial_size[0], spatial_size[1]])
        t.uniform_() if noise_type == "u" else t.normal_()
        t *= var
        return t
    if method == "meshgrid":
        assert input_depth == 2
        X, Y = np.meshgrid(np.arange(0, spatial_size[1]) / float(spatial_size[1] - 1),
                           np.arange(0, spatial_size[0]) / float(spatial_size[0] - 1))
        return torch.from_numpy(np.concatenate([X[None, :], Y[None, :]]))[None, :]
    raise AssertionError(method)


# --------------------------------------------------------------------------
# nn.Module wrapper so torch.optim.Adam / the closure loop can drive the oracle
# --------------------------------------------------------------------------
class OracleNet(torch.nn.Module):
    def __init__(self, spec: SkipSpec, sd: Dict[str, torch.Tensor]):
        super().__init__()
        self.spec = spec
        self.names = list(sd.keys())
        self.params = torch.nn.ParameterList([torch.nn.Parameter(sd[k].clone()) for k in self.names])

    def sd(self):
        return {k: p for k, p in zip(self.names, self.params)}

    def forward(self, x, taps=None, masks=None, zrec=None):
        return skip_forward(self.spec, self.sd(), x, taps, masks, zrec)


def optimize_adam(parameters, closure, LR, num_iter):
    """optimize('adam', ...): utils/common_utils.py:223-230."""
    optimizer = torch.optim.Adam(parameters, lr=LR)
    for _ in range(num_iter):
        optimizer.zero_grad()
        closure()
        optimizer.step()


def psnr(a: np.ndarray, b: np.ndarray, data_range: float = 1.0) -> float:
    """skimage<=0.15 compare_psnr for non-negative float images, in float64."""
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * math.log10(data_range ** 2 / mse)


class ClosureBookkeeping:
    """The host-side bookkeeping of the reference's denoising closure, restated line by line
    (denoising.ipynb:214-248): EMA of the output (:214-217), the three PSNRs (:223-225) and the
    back-tracking rule (:238-248; note `if i % show_every:` -- the check runs on every iteration
    that is NOT a multiple of show_every).  `params` plays the role of net.parameters()."""

    def __init__(self, img_noisy_np, img_np, exp_weight=0.99, show_every=100):
        self.noisy, self.gt = img_noisy_np, img_np
        self.exp_weight, self.show_every = exp_weight, show_every
        self.out_avg = None
        self.last_net = None
        self.psrn_noisy_last = 0
        self.i = 0

    def step(self, out: torch.Tensor, params):
        if self.out_avg is None:
            self.out_avg = out.detach()
        else:
            self.out_avg = self.out_avg * self.exp_weight + out.detach() * (1 - self.exp_weight)
        psrn_noisy = psnr(self.noisy, out.detach().cpu().numpy()[0])
        psrn_gt = psnr(self.gt, out.detach().cpu().numpy()[0])
        psrn_gt_sm = psnr(self.gt, self.out_avg.detach().cpu().numpy()[0])
        fell_back = False
        if self.i % self.show_every:
            if psrn_noisy - self.psrn_noisy_last < -5:
                for new_param, net_param in zip(self.last_net, params):
                    net_param.data.copy_(new_param)
                fell_back = True
            else:
                self.last_net = [x.detach().clone() for x in params]
                self.psrn_noisy_last = psrn_noisy
        self.i += 1
        return {"psrn_noisy": psrn_noisy, "psrn_gt": psrn_gt, "psrn_gt_sm": psrn_gt_sm, "fell_back": fell_back}
