"""Fused Adam over flat arenas (libdip_hip.so: dip_adam_step).

Replaces `torch.optim.Adam(parameters, lr=LR)` of the reference's optimize()
(utils/common_utils.py:225) with torch 2.x Adam semantics (betas (0.9, 0.999), eps 1e-8, no
weight decay; bias corrections computed in double on the host).  Parameters that are views of
one contiguous fp32 CUDA arena (a SkipNet's parameters) are stepped by ONE launch; any other
CUDA fp32 tensor (net_input for opt_over='net,input', Downsampler weights) gets its own launch
of the same kernel.
"""
from __future__ import annotations

import torch

import dip_native as N


class _Group:
    """A maximal run of parameters that is contiguous in device memory."""

    def __init__(self, params):
        self.params = params
        self.base = params[0].data_ptr()
        last = params[-1]
        self.numel = (last.data_ptr() - self.base) // 4 + last.numel()
        dev = params[0].device
        self.m = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.gbuf = None
        self.offsets = [(p.data_ptr() - self.base) // 4 for p in params]


def _split_contiguous(params, max_gap=4):
    groups, run = [], []
    for p in params:
        if run:
            prev = run[-1]
            gap = (p.data_ptr() - (prev.data_ptr() + 4 * prev.numel())) // 4
            same = p.device == prev.device and p.untyped_storage().data_ptr() == prev.untyped_storage().data_ptr()
            if not (same and 0 <= gap < max_gap and (p.data_ptr() - prev.data_ptr()) % 4 == 0):
                groups.append(run)
                run = []
        run.append(p)
    if run:
        groups.append(run)
    return groups


class FusedAdam:
    def __init__(self, parameters, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in parameters]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self._groups = None
        self._sig = None

    # torch.optim API subset used by optimize() and the notebooks
    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.zero_()

    def _signature(self):
        return tuple(p.data_ptr() for p in self.params)

    def _prepare(self):
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("dip-amd FusedAdam: parameters must be contiguous fp32 CUDA tensors "
                                   f"(got {p.dtype} on {p.device}); this backend has no CPU optimiser path")
        old = {}
        if self._groups is not None:          # parameters moved (net re-typed): carry the moments over
            for g in self._groups:
                for p, o in zip(g.params, g.offsets):
                    old[id(p)] = (g.m[o:o + p.numel()].clone(), g.v[o:o + p.numel()].clone())
        self._groups = [_Group(run) for run in _split_contiguous(self.params)]
        for g in self._groups:
            for p, o in zip(g.params, g.offsets):
                if id(p) in old:
                    g.m[o:o + p.numel()].copy_(old[id(p)][0])
                    g.v[o:o + p.numel()].copy_(old[id(p)][1])
        self._sig = self._signature()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._sig != self._signature():
            self._prepare()
        lib = N.lib()
        self.step_count += 1
        for g in self._groups:
            grads = [p.grad for p in g.params]
            if all(gr is None for gr in grads):
                continue
            # fast path: the gradients are views of one arena with the parameters' layout
            g0 = grads[0]
            flat_ptr = None
            if g0 is not None and g0.is_cuda and g0.dtype == torch.float32:
                base = g0.data_ptr() - 4 * g.offsets[0]
                if all(gr is not None and gr.dtype == torch.float32 and gr.is_contiguous()
                       and gr.data_ptr() == base + 4 * o for gr, o in zip(grads, g.offsets)):
                    flat_ptr = base
            if flat_ptr is None:
                if g.gbuf is None:
                    g.gbuf = torch.zeros(g.numel, dtype=torch.float32, device=g.params[0].device)
                for p, gr, o in zip(g.params, grads, g.offsets):
                    if gr is None:
                        # torch skips params without grad; a zero gradient would still decay the moments, so
                        # step such params separately is required -- not supported in a fused run
                        raise RuntimeError("dip-amd FusedAdam: a parameter of a fused group has no gradient")
                    g.gbuf[o:o + p.numel()].copy_(gr.reshape(-1))
                flat_ptr = g.gbuf.data_ptr()
            stream = torch.cuda.current_stream(g.params[0].device).cuda_stream
            N.check(lib.dip_adam_step(g.base, flat_ptr, g.m.data_ptr(), g.v.data_ptr(), g.numel, float(self.lr),
                                      float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_count,
                                      stream), "adam_step")
        return loss
