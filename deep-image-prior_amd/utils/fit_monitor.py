"""Device-side bookkeeping for the notebooks' closures (no counterpart file in the reference: the
reference does this work inline, on the host, in every closure -- denoising.ipynb:214-248,
restoration.ipynb:192-211).

    monitor = FitMonitor(net, img_noisy_torch, img_torch, exp_weight=0.99, show_every=100)
    def closure():
        out = net(net_input_saved + noise.normal_() * reg_noise_std)
        total_loss = mse(out, img_noisy_torch)
        total_loss.backward()
        monitor.update(out, total_loss)          # EMA + 3 PSNRs + back-tracking, no host sync
        return total_loss
    optimize('adam', p, closure, LR, num_iter)
    hist = monitor.history()                     # [iters, 8] numpy, ONE device->host copy
    out_avg = monitor.out_avg                    # the smoothed output (1 x C x H x W, on the GPU)

Record columns: loss, mse_noisy, mse_gt, mse_gt_sm, psrn_noisy, psrn_gt, psrn_gt_sm, fell_back.
The reference's per-iteration cost this replaces: three `.detach().cpu().numpy()` of the output, a
`.item()`, and -- whenever `i % show_every` is non-zero -- a copy of all 2.2 M parameters to the CPU.
"""
import ctypes as C

import numpy as np
import torch

import dip_native as N


class FitMonitor:
    COLUMNS = ("loss", "mse_noisy", "mse_gt", "mse_gt_sm", "psrn_noisy", "psrn_gt", "psrn_gt_sm", "fell_back")

    def __init__(self, net, img_noisy, img_gt=None, exp_weight=0.99, show_every=100, backtrack_db=5.0,
                 backtracking=True, capacity=16384):
        if not img_noisy.is_cuda:
            raise RuntimeError("dip-amd: FitMonitor works on MI355X tensors only (no CPU fallback)")
        self.lib = N.lib()
        self.dev = img_noisy.device
        self.noisy = img_noisy.detach().contiguous().float()
        self.gt = None if img_gt is None else img_gt.detach().to(self.dev).contiguous().float()
        if self.gt is not None and self.gt.shape != self.noisy.shape:
            raise ValueError("FitMonitor: img_gt and img_noisy differ in shape")
        self.n = self.noisy.numel()
        self.exp_weight, self.show_every, self.backtrack_db = float(exp_weight), int(show_every), float(backtrack_db)
        self.capacity = int(capacity)
        self.records = torch.zeros((self.capacity, 8), dtype=torch.float32, device=self.dev)
        self.state = torch.zeros(4, dtype=torch.float32, device=self.dev)
        self.partial = torch.empty(4 * self.lib.dip_fit_monitor_nblk(self.n), dtype=torch.float32, device=self.dev)
        self.out_avg = torch.zeros_like(self.noisy)
        self.i = 0
        self.engine = None
        self.snapshot = None
        if backtracking:
            eng = getattr(net, "__dict__", {}).get("_dip_engine")
            if eng is None:
                raise RuntimeError("dip-amd: back-tracking needs a net built by models.skip.skip() (flat parameter arena)")
            self.engine = eng

    def update(self, out, loss=None):
        """Call once per closure evaluation, after backward() (like the reference, the fall-back
        overwrites the parameters AFTER the gradients of this iteration were computed)."""
        if self.i >= self.capacity:
            raise RuntimeError("FitMonitor: capacity exceeded; construct it with capacity >= num_iter")
        o = out.detach()
        if o.shape != self.noisy.shape or not o.is_cuda:
            raise ValueError("FitMonitor.update: output shape/device does not match the target image")
        o = o.contiguous().float()
        with torch.cuda.device(self.dev):        # raw HIP launches go to the current device's streams
            stream = torch.cuda.current_stream(self.dev).cuda_stream
            lptr = None
            if loss is not None:
                self._loss = loss.detach().reshape(1).float()          # keep alive until the launch has run
                lptr = self._loss.data_ptr()
            check = 1 if (self.engine is not None and self.i % self.show_every) else 0
            N.check(self.lib.dip_fit_monitor(o.data_ptr(), self.noisy.data_ptr(),
                                             self.gt.data_ptr() if self.gt is not None else None,
                                             self.out_avg.data_ptr(), self.n, self.exp_weight, 1 if self.i == 0 else 0,
                                             lptr, self.partial.data_ptr(), self.records[self.i].data_ptr(),
                                             self.state.data_ptr(), check, self.backtrack_db, stream), "fit_monitor")
            if self.engine is not None:
                params = self.engine.params
                if self.snapshot is None or self.snapshot.numel() != params.numel() \
                        or self.snapshot.device != params.device:
                    self.snapshot = torch.empty_like(params)
                N.check(self.lib.dip_arena_backtrack(params.data_ptr(), self.snapshot.data_ptr(), params.numel(),
                                                     self.state.data_ptr(), stream), "arena_backtrack")
        self._keep = o
        self.i += 1

    def history(self):
        """All records so far as a [iters, 8] float32 numpy array (synchronises once)."""
        return self.records[:self.i].cpu().numpy()

    def last(self):
        """The latest record as a dict (synchronises)."""
        r = self.records[self.i - 1].cpu().numpy()
        return dict(zip(self.COLUMNS, (float(x) for x in r)))
