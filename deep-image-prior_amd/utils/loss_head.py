"""Fused loss head for the notebooks' closures (no counterpart file in the reference, which spells
the same arithmetic out as separate PyTorch ops in every closure):

    denoising.ipynb:212,219      out = net(net_input);            total_loss = mse(out, img_noisy_torch)
    inpainting.ipynb:308-310     out = net(net_input);            total_loss = mse(out * mask_var, img_var * mask_var)

    head = MSEHead(net, img_noisy_torch)                      # or MSEHead(net, img_var, mask=mask_var)
    def closure():
        total_loss, out = head(net_input)                     # out: the network output, detached
        total_loss.backward()
        return total_loss

`head(net_input)` runs the skip-net up to the input of its last conv and then ONE launch
(dip_loss_head_fwd) for the 1x1 output conv + Sigmoid + mask + MSE, with the scalar reduced by an
LDS tree per block and a fixed-order fp64 sum of the per-block partials; `backward()` starts from
dip_loss_head_bwd.  The plain `out = net(x); mse(out, t)` spelling keeps working (the head,
the mask product and the MSE then run as separate kernels); this class is the opt-in fused path.
"""
import ctypes as C

import torch

import dip_native as N


class MSEHead:
    def __init__(self, net, target, mask=None):
        eng = getattr(net, "__dict__", {}).get("_dip_engine")
        if eng is None or isinstance(eng, Exception):
            raise RuntimeError("dip-amd: MSEHead needs a net built by models.skip.skip()")
        if not target.is_cuda:
            raise RuntimeError("dip-amd: MSEHead works on MI355X tensors only (no CPU fallback)")
        self.net, self.engine = net, eng
        oc = eng.out_conv
        if oc.ks != 1 or oc.Cout > 4:
            raise NotImplementedError("dip-amd: the fused loss head covers a 1x1 output conv with <= 4 channels "
                                      "(n_channels 1 or 3 in every reference notebook)")
        if target.dim() != 4 or target.shape[0] != 1 or target.shape[1] != oc.Cout:
            raise ValueError(f"MSEHead: target must be [1,{oc.Cout},H,W], got {tuple(target.shape)}")
        self.target = target.detach().contiguous().float()
        self.mask = None
        self.mask_c = 0
        if mask is not None:
            m = mask.detach().to(target.device).float()
            while m.dim() < 4:
                m = m[None]
            if m.shape[0] != 1 or m.shape[1] not in (1, oc.Cout) or m.shape[2:] != target.shape[2:]:
                raise ValueError(f"MSEHead: mask must be [1,1|{oc.Cout},H,W], got {tuple(m.shape)}")
            self.mask, self.mask_c = m.contiguous(), int(m.shape[1])
        self._scratch = None

    def _descriptor(self, eng, out, loss):
        """DipLossHeadDesc for the engine's current plan (called by SkipEngine.forward)."""
        H, W = eng.Hout, eng.Wout
        if tuple(self.target.shape[2:]) != (H, W):
            raise ValueError(f"MSEHead: target is {tuple(self.target.shape[2:])}, the net output is {(H, W)}")
        a = eng.last_act
        oc = eng.out_conv
        nblk = eng.lib.dip_loss_head_nblk(H * W, oc.Cin)
        dev = out.device
        if self._scratch is None or self._scratch.numel() != nblk or self._scratch.device != dev:
            self._scratch = torch.empty(nblk, dtype=torch.float32, device=dev)
        partials = self._scratch
        tr = a.transform()
        self._keep = (out.detach(), tr)     # (not `loss`: it becomes the autograd output and would pin the graph)
        ptr = lambda t, off=0: None if t is None else t.data_ptr() + 4 * off
        return N.DipLossHeadDesc(ptr(a.buf), a.Cs, oc.Cin, tr, ptr(eng.params, oc.w_off),
                                 ptr(eng.params, oc.b_off) if oc.b_off >= 0 else None, oc.Cout, H * W,
                                 1 if eng.need_sigmoid else 0, ptr(self.target), ptr(self.mask), self.mask_c,
                                 ptr(out), ptr(partials), nblk, ptr(loss))

    def __call__(self, net_input):
        import dip_engine
        if not self.net.training:
            raise NotImplementedError("dip-amd: eval-mode BatchNorm is not implemented")
        loss, out = dip_engine.run_net_loss(self.engine, self, net_input)
        return loss, out
