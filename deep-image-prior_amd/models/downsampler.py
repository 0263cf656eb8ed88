"""Fixed-kernel down-sampler (Lanczos / Gauss / box), API-compatible with the reference's
models/downsampler.py:9-135, executed by a depth-wise gfx950 kernel.

The reference realises it as ReplicationPad2d + a dense Conv2d(n, n, k, stride=factor) whose
weight is the 2-D tap table on the channel diagonal (2/3 of the MACs multiply zeros for n=3) and
registers the fixed taps as trainable parameters.  Here the same taps (float64 numpy, then fp32)
drive `dip_lanczos_down_fwd/bwd`: one depth-wise stencil with clamped (replicated) borders.
`downsampler_.weight/bias` are kept as parameters so `state_dict()`, `get_params('down', ...)` and
`.type(dtype)` have the reference's shape.  The depth-wise path reads the taps, not the dense weight, so:
  * the parameters do not require grad by default (the fixed-taps kernels return no weight gradient);
  * OPTIMISING the down-sampler (opt_over containing 'down', utils/common_utils.py:44-46 of the reference -- no
    notebook does): `get_params('down', ...)` turns requires_grad on, and forward() then runs the module as what the
    reference's is -- a dense Conv2d(n, n, k, stride=factor) behind the replication padding -- through
    `dip_down_dense_fwd / _bwd_data / _bwd_weight`, with gradients for weight and bias;
  * `load_state_dict()` re-derives the taps from the loaded weight; a weight that is not "one 2-D kernel on the channel
    diagonal, zero bias" any more (a trained down-sampler) switches the module to the dense path as well.
"""
import numpy as np
import torch
import torch.nn as nn


def get_kernel(factor, kernel_type, phase, kernel_width, support=None, sigma=None):
    """2-D resampling taps, normalised to sum 1 (float64).  Half-phase kernels have an even
    size (kernel_width - 1)."""
    assert kernel_type in ('lanczos', 'gauss', 'box')
    n = kernel_width - 1 if (phase == 0.5 and kernel_type != 'box') else kernel_width
    taps = np.zeros([n, n])
    if kernel_type == 'box':
        assert phase == 0.5, 'Box filter is always half-phased'
        taps[:] = 1. / (kernel_width * kernel_width)
    elif kernel_type == 'gauss':
        assert sigma, 'sigma is not specified'
        assert phase != 0.5, 'phase 1/2 for gauss not implemented'
        center = (kernel_width + 1.) / 2.
        s2 = sigma * sigma
        for i in range(1, n + 1):
            for j in range(1, n + 1):
                di, dj = (i - center) / 2., (j - center) / 2.
                taps[i - 1][j - 1] = np.exp(-(di * di + dj * dj) / (2 * s2)) / (2. * np.pi * s2)
    else:
        assert support, 'support is not specified'
        center = (kernel_width + 1) / 2.
        shift = 0.5 if phase == 0.5 else 0.0

        for i in range(1, n + 1):
            for j in range(1, n + 1):
                di = abs(i + shift - center) / factor
                dj = abs(j + shift - center) / factor
                val = 1
                if di != 0:
                    val = val * support * np.sin(np.pi * di) * np.sin(np.pi * di / support)
                    val = val / (np.pi * np.pi * di * di)
                if dj != 0:
                    val = val * support * np.sin(np.pi * dj) * np.sin(np.pi * dj / support)
                    val = val / (np.pi * np.pi * dj * dj)
                taps[i - 1][j - 1] = val
    taps /= taps.sum()
    return taps


_PRESETS = {
    'lanczos2': dict(support=2, width=lambda f: 4 * f + 1, base='lanczos'),
    'lanczos3': dict(support=3, width=lambda f: 6 * f + 1, base='lanczos'),
    'gauss12': dict(sigma=1 / 2, width=lambda f: 7, base='gauss'),
    'gauss1sq2': dict(sigma=1. / np.sqrt(2), width=lambda f: 9, base='gauss'),
}


class _LanczosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, taps, k, factor, pad):
        import dip_native as N
        lib = N.lib()
        _, C, H, W = x.shape
        Ho, Wo = (H + 2 * pad - k) // factor + 1, (W + 2 * pad - k) // factor + 1
        xs = x.detach().contiguous().float()
        y = torch.empty((1, C, Ho, Wo), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = torch.cuda.current_stream(x.device).cuda_stream
            N.check(lib.dip_lanczos_down_fwd(xs.data_ptr(), taps.data_ptr(), y.data_ptr(), C, H, W, k, factor, pad, st),
                    "lanczos_down_fwd")
        ctx.meta = (taps, k, factor, pad, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        import dip_native as N
        lib = N.lib()
        taps, k, factor, pad, C, H, W = ctx.meta
        g = gy.detach().contiguous().float()
        gx = torch.empty((1, C, H, W), dtype=torch.float32, device=gy.device)
        with torch.cuda.device(gy.device):
            st = torch.cuda.current_stream(gy.device).cuda_stream
            N.check(lib.dip_lanczos_down_bwd(g.data_ptr(), taps.data_ptr(), gx.data_ptr(), C, H, W, k, factor, pad, st),
                    "lanczos_down_bwd")
        return gx, None, None, None, None


class _DenseFn(torch.autograd.Function):
    """ReplicationPad2d(pad) + Conv2d(n, n, k, stride=factor) with trainable weight / bias (models/downsampler.py:88-101
    of the reference when its parameters are optimised)."""

    @staticmethod
    def forward(ctx, x, weight, bias, k, factor, pad):
        import dip_native as N
        lib = N.lib()
        _, C, H, W = x.shape
        Ho, Wo = (H + 2 * pad - k) // factor + 1, (W + 2 * pad - k) // factor + 1
        xs = x.detach().contiguous().float()
        w = weight.detach().contiguous().float()
        b = bias.detach().contiguous().float()
        y = torch.empty((1, C, Ho, Wo), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            st = torch.cuda.current_stream(x.device).cuda_stream
            N.check(lib.dip_down_dense_fwd(xs.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C, H, W, k, factor, pad,
                                           st), "down_dense_fwd")
        ctx.save_for_backward(xs, w)
        ctx.meta = (k, factor, pad, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        import dip_native as N
        lib = N.lib()
        xs, w = ctx.saved_tensors
        k, factor, pad, C, H, W = ctx.meta
        g = gy.detach().contiguous().float()
        gx = dw = db = None
        with torch.cuda.device(gy.device):
            st = torch.cuda.current_stream(gy.device).cuda_stream
            if ctx.needs_input_grad[0]:
                gx = torch.empty((1, C, H, W), dtype=torch.float32, device=gy.device)
                N.check(lib.dip_down_dense_bwd_data(g.data_ptr(), w.data_ptr(), gx.data_ptr(), C, H, W, k, factor, pad, st),
                        "down_dense_bwd_data")
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                dw = torch.empty_like(w)
                db = torch.empty((C,), dtype=torch.float32, device=gy.device)
                N.check(lib.dip_down_dense_bwd_weight(g.data_ptr(), xs.data_ptr(), dw.data_ptr(), db.data_ptr(), C, H, W, k,
                                                      factor, pad, st), "down_dense_bwd_weight")
        return gx, dw, db, None, None, None


class Downsampler(nn.Module):
    def __init__(self, n_planes, factor, kernel_type, phase=0, kernel_width=None, support=None, sigma=None,
                 preserve_size=False, _dense=False):
        super().__init__()
        # _dense (models.common.conv only): the module sits INSIDE a skip() net (conv(..., downsample_mode='lanczos2')),
        # where the reference trains its dense n_planes x n_planes x k x k weight with the rest of net.parameters()
        # (models/common.py:107-108): dip_engine runs it as a stride-`factor` convolution with replication padding and
        # returns its weight gradient; the parameters stay trainable and the fixed-taps check does not apply
        self._dense = _dense
        assert phase in [0, 0.5], 'phase should be 0 or 0.5'
        if kernel_type in _PRESETS:
            p = _PRESETS[kernel_type]
            support, sigma = p.get('support', support), p.get('sigma', sigma)
            kernel_width, base = p['width'](factor), p['base']
        elif kernel_type in ('lanczos', 'gauss', 'box'):
            base = kernel_type
        else:
            assert False, 'wrong name kernel'
        self.kernel = get_kernel(factor, base, phase, kernel_width, support=support, sigma=sigma)
        self.factor = factor
        k = self.kernel.shape[0]
        # parameter holder with the reference's shapes/names (downsampler_.weight / .bias)
        holder = nn.Conv2d(n_planes, n_planes, kernel_size=self.kernel.shape, stride=factor, padding=0)
        holder.weight.data[:] = 0
        holder.bias.data[:] = 0
        kt = torch.from_numpy(self.kernel)
        for c in range(n_planes):
            holder.weight.data[c, c] = kt
        # the depth-wise path applies FIXED taps and returns no weight gradient: say so on the parameters themselves.
        # get_params('down', ...) -- or the user -- turns requires_grad on, and forward() then takes the dense path
        self._nondiag = False
        if not _dense:
            holder.weight.requires_grad_(False)
            holder.bias.requires_grad_(False)
        self.downsampler_ = holder
        self.register_buffer('_taps', kt.to(torch.float32).contiguous(), persistent=False)
        self.register_load_state_dict_post_hook(Downsampler._taps_from_weight)
        self.preserve_size = preserve_size
        self._pad = 0
        if preserve_size:
            self._pad = int((k - 1) / 2.) if k % 2 == 1 else int((k - factor) / 2.)
            self.padding = nn.ReplicationPad2d(self._pad)

    def _taps_from_weight(self, *_):
        """load_state_dict post-hook: the native path applies `_taps`, so re-derive them from the loaded
        dense weight and refuse anything that is not one 2-D kernel on the channel diagonal."""
        if getattr(self, "_dense", False):        # inside a skip() net: the engine convolves with the dense weight itself
            return
        w = self.downsampler_.weight.detach()
        b = self.downsampler_.bias.detach()
        n = w.shape[0]
        diag = torch.stack([w[c, c] for c in range(n)])
        off = w.clone()
        for c in range(n):
            off[c, c] = 0
        # a trained down-sampler (anything but one 2-D kernel on the channel diagonal, zero bias): dense path from now on
        self._nondiag = bool(float(off.abs().max()) != 0 or float(b.abs().max()) != 0
                             or float((diag - diag[0]).abs().max()) != 0)
        if not self._nondiag:
            self._taps = diag[0].to(torch.float32).contiguous().to(self._taps.device)

    def forward(self, input):
        if getattr(self, "_dense", False):
            raise RuntimeError("dip-amd: this Downsampler belongs to a skip() net (conv(..., downsample_mode='lanczos*')); "
                               "it runs as part of the net's launch list, not on its own")
        if not input.is_cuda:
            raise RuntimeError("dip-amd: Downsampler runs on an MI355X only (no CPU fallback in this backend)")
        if input.dim() != 4 or input.shape[0] != 1:
            raise NotImplementedError("dip-amd: Downsampler expects a [1,C,H,W] tensor")
        w, b = self.downsampler_.weight, self.downsampler_.bias
        if w.requires_grad or b.requires_grad or getattr(self, "_nondiag", False):
            self._nondiag = True        # once handed to an optimiser the weight is no longer known to equal the taps
            return _DenseFn.apply(input, w, b, self.kernel.shape[0], self.factor, self._pad)
        return _LanczosFn.apply(input, self._taps.to(input.device), self.kernel.shape[0], self.factor, self._pad)
