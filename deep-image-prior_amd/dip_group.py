"""Grouped multi-instance execution: B independent fits of one architecture -- B copies of
the reference's skip-net (models/skip.py:45-100) with own weights, own BatchNorm statistics, own Adam state, own input,
target and reg-noise stream, each advancing through the body of the reference's optimize() loop
(utils/common_utils.py:226-230) with the notebooks' closure (denoising.ipynb:204-221, inpainting.ipynb:295-313) --
through ONE launch list: every kernel launch of an iteration serves all B instances (gridDim.z x B workgroups;
csrc/dip_group.h), instead of B launch lists on B streams.  What fills an MI355X when one image (the 384x256 snail net:
~25 us of math per iteration) cannot: the B-streams form is bound by the dispatch rate of the command processor
(profiles/r03_dispatch_rate.txt), the grouped form issues 1/B of the launches.

Memory: EVERY buffer of a fit -- parameter / gradient / Adam arenas, BatchNorm state, packed weights, activations, scratch,
descriptor tables, net input, noise state, target, mask, output, loss -- is carved from one slab per instance; the B slabs
are the rows of one [B][stride] allocation, laid out identically.  The launch list is compiled for instance 0 (the
SkipEngine of nets[0], with the slab as its allocator) and issued between dip_group_begin / dip_group_end; instance b sees
every pointer advanced by b * stride.  Plans, tile walks and summation orders are those of a solo fit, so every instance
is bit-identical to the same fit run on its own (tests/test_group_gpu.py).

    g = GroupedFits(nets, net_inputs, targets, masks=None, reg_noise_std=1/30, seeds=range(B), lr=0.01, exp_weight=0.99)
    g.capture()                     # 3 eager warm-up iterations, then ONE hipGraph of the grouped iteration
    g.run(num_iter - 3)
    g.losses                        # [B] device tensor: total_loss of the last iteration, per instance
    g.out, g.out_avg                # [B, C, H, W]: network outputs / their exponential moving averages
    nets[b].state_dict()            # the parameters of nets[b] are views of its slab: always current

There is no CPU or per-instance fallback here: the library must be loaded, and an architecture / size mismatch raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

import dip_native as N
from dip_native import round_up

_ALIGN = 256


class Slab:
    """Bump allocator over one row of the [B][stride] allocation.  First pass (no buffer bound): hands out ordinary torch
    tensors and only measures; second pass (bind()): the same sequence of requests returns views of row 0."""

    def __init__(self, device):
        self.device = device
        self.buf = None
        self.off = 0
        self.sizes = []

    def bind(self, row0: torch.Tensor):
        self.buf, self.measured, self.off, self.replay = row0, self.off, 0, 0

    def alloc(self, n, dtype=torch.float32, zero=False):
        nbytes = int(n) * torch.empty((), dtype=dtype).element_size()
        step = round_up(max(nbytes, 1), _ALIGN)
        if self.buf is None:
            self.sizes.append(step)
            self.off += step
            return (torch.zeros if zero else torch.empty)(int(n), dtype=dtype, device=self.device)
        if self.replay >= len(self.sizes) or self.sizes[self.replay] != step or self.off + step > self.buf.numel():
            raise RuntimeError("dip-amd Slab: the second build asked for different buffers than the measuring one")
        self.replay += 1
        t = self.buf[self.off:self.off + nbytes].view(dtype)
        self.off += step
        if zero:
            t.zero_()
        return t


class GroupedFits:
    ADAM_BETAS, ADAM_EPS = (0.9, 0.999), 1e-8

    def __init__(self, nets, net_inputs, targets, masks=None, reg_noise_std=0.0, seeds=None, lr=0.01, exp_weight=None,
                 ema_init="first", device=None, _dry_cpu=False):
        """nets: B nets of models.skip.skip() with identical architecture; net_inputs / targets (/ masks): one tensor per
        instance, identical shapes ([1,C,H,W]; masks [1,1|Cout,H,W] or None).  reg_noise_std / seeds: the closure's input
        noise (utils.reg_noise.RegNoise; seeds default to 0..B-1).  exp_weight: None = no moving average of the output;
        ema_init 'first' = out_avg starts as the first output (denoising.ipynb:214-215), 'zeros' = starts at 0."""
        B = len(nets)
        if B < 1 or len(net_inputs) != B or len(targets) != B or (masks is not None and len(masks) != B):
            raise ValueError("GroupedFits: one net, one input, one target (and one mask) per instance")
        engs = [getattr(n, "__dict__", {}).get("_dip_engine") for n in nets]
        if any(e is None or isinstance(e, Exception) for e in engs):
            raise RuntimeError("dip-amd: GroupedFits needs nets built by models.skip.skip()")
        if device is None:
            device = net_inputs[0].device
        device = torch.device(device)
        # (_dry_cpu: the slab construction alone, on host memory, for the layout unit test -- nothing can be launched)
        self._dry = bool(_dry_cpu)
        if device.type != "cuda" and not self._dry:
            raise RuntimeError("dip-amd: GroupedFits runs on an MI355X only (no CPU fallback in this backend)")
        self.B, self.nets, self.device = B, list(nets), device
        self.eng = eng = engs[0]
        self.lib = N.lib()
        self.lr, self.std = float(lr), float(reg_noise_std)
        self.exp_weight = None if exp_weight is None else float(exp_weight)
        if ema_init not in ("first", "zeros"):
            raise ValueError("GroupedFits: ema_init is 'first' or 'zeros'")
        self.ema_first = ema_init == "first"
        self.seeds = [int(s) for s in (seeds if seeds is not None else range(B))]
        self.iterations = 0
        self.graph = None
        # --- same architecture, same sizes
        sig0 = [(k, tuple(p.shape)) for k, p in nets[0].named_parameters()]
        for b, n in enumerate(nets):
            if [(k, tuple(p.shape)) for k, p in n.named_parameters()] != sig0:
                raise ValueError(f"GroupedFits: net {b} differs from net 0 in architecture")
            if not n.training:
                raise NotImplementedError("dip-amd: eval-mode BatchNorm is not implemented")
        z0, t0 = net_inputs[0], targets[0]
        if z0.dim() != 4 or z0.shape[0] != 1:
            raise ValueError("GroupedFits: net inputs are [1,C,H,W]")
        for b in range(B):
            if net_inputs[b].shape != z0.shape or targets[b].shape != t0.shape:
                raise ValueError(f"GroupedFits: instance {b} differs from instance 0 in input / target shape")
            if masks is not None and (masks[b] is None) != (masks[0] is None):
                raise ValueError("GroupedFits: either every instance has a mask or none has")
        oc = eng.out_conv
        if oc.ks != 1 or oc.Cout > 4:
            raise NotImplementedError("dip-amd: the fused loss head covers a 1x1 output conv with <= 4 channels")
        if t0.dim() != 4 or t0.shape[0] != 1 or t0.shape[1] != oc.Cout:
            raise ValueError(f"GroupedFits: targets must be [1,{oc.Cout},H,W], got {tuple(t0.shape)}")
        self.mask_c = 0
        m0 = None
        if masks is not None and masks[0] is not None:
            m0 = self._mask4(masks[0])
            if m0.shape[1] not in (1, oc.Cout) or m0.shape[2:] != t0.shape[2:]:
                raise ValueError(f"GroupedFits: masks must be [1,1|{oc.Cout},H,W]")
            self.mask_c = int(m0.shape[1])
        _, Cimg, H, W = z0.shape
        # --- the slab: a measuring build, the allocation, the real build into row 0
        with self._devctx():
            slab = Slab(device)
            try:
                self._build_row0(slab, Cimg, H, W, t0, m0)
                self.stride = slab.off                               # a multiple of 256 by construction
                del self._x, self._row0_extra
                self._raw = torch.zeros(B * self.stride + _ALIGN, dtype=torch.uint8, device=device)
                o = (-self._raw.data_ptr()) % _ALIGN                  # (the device allocator aligns to >= 256 anyway)
                self.mem = self._raw[o:o + B * self.stride]
                slab.bind(self.mem[:self.stride])
                self._build_row0(slab, Cimg, H, W, t0, m0)
                if slab.off != self.stride:
                    raise RuntimeError("dip-amd GroupedFits: the slab build is not reproducible")
            finally:
                eng.slab = None            # whatever happened: a later (re-)plan of nets[0] uses torch's allocator
                if not hasattr(self, "mem") or slab.buf is None or slab.off != getattr(self, "stride", -1):
                    eng.device = None      # ... and a half-built engine state is rebuilt by the next forward
            # --- rows 1..B-1: a copy of row 0 (descriptor tables, constants, zeroed state), then what is the instance's own
            rows = self.mem.view(B, self.stride)
            if B > 1:
                rows[1:].copy_(rows[:1].expand(B - 1, self.stride))
            ex = self._row0_extra
            with torch.no_grad():
                for b in range(B):
                    if b > 0:
                        self._adopt_net(b)
                    self._inst(ex["saved"], b).copy_(net_inputs[b].detach().to(device).float().reshape(-1))
                    self._inst(ex["target"], b).copy_(targets[b].detach().to(device).float().reshape(-1))
                    if ex["mask"] is not None:
                        self._inst(ex["mask"], b).copy_(self._mask4(masks[b]).to(device).float().reshape(-1))
                    self._inst(ex["rng"], b).copy_(torch.tensor([0, self.seeds[b]], dtype=torch.int64))
            # --- what the caller reads: strided views over the instances
            HWo = eng.Hout * eng.Wout
            self.losses = self._strided(ex["loss"], (B,), ())
            self.out = self._strided(ex["out"], (B, oc.Cout, eng.Hout, eng.Wout), (HWo, eng.Wout, 1))
            self.out_avg = torch.zeros((B, oc.Cout, eng.Hout, eng.Wout), dtype=torch.float32, device=device) \
                if self.exp_weight is not None else None
            self._nbt_all = self._strided(eng.nbt, (B, eng.nbt.numel()), (1,))
            if not self._dry:
                torch.cuda.synchronize(device)

    # ------------------------------------------------------------------ construction helpers
    def _devctx(self):
        return contextlib.nullcontext() if self._dry else torch.cuda.device(self.device)

    def pointers_outside_row0(self):
        """Self-check of the memory model: every device pointer of the launch list (descriptor fields and pointer
        arguments) must lie inside instance 0's slab -- the library refuses a grouped launch otherwise (rc -1).  Returns
        the offenders as (op name, field) pairs; [] when the list is sound."""
        lo, hi = self.mem.data_ptr(), self.mem.data_ptr() + self.stride
        bad = []

        def visit(name, field, v):
            if isinstance(v, C.Structure):
                for f, _ in v._fields_:
                    visit(name, field + "." + f, getattr(v, f))
            elif isinstance(v, int) and v >= (1 << 32) and not (lo <= v < hi):
                bad.append((name, field))

        eng = self.eng
        for ops in (eng.fwd_ops, eng.bwd_ops):
            for fn, args, name in ops:
                for k, a in enumerate(args):
                    visit(name, f"arg{k}", a._obj if hasattr(a, "_obj") else a)
        visit("loss_head", "desc", self._head)
        for k, t in self._row0_extra.items():
            if t is not None:
                visit("extra", k, t.data_ptr())
        for k in ("params", "grads", "packed", "pack_recs", "x_nhwc", "dy_out", "bnbuf", "nbt"):
            visit("engine", k, getattr(eng, k).data_ptr())
        if eng.bf3:
            visit("engine", "packed3", eng.packed3.data_ptr())
            visit("engine", "pack_recs3", eng.pack_recs3.data_ptr())
        return bad

    @staticmethod
    def _mask4(m):
        m = m.detach().float()
        while m.dim() < 4:
            m = m[None]
        return m.contiguous()

    def _build_row0(self, slab, Cimg, H, W, t0, m0):
        """Everything instance 0 owns, in a fixed order, from `slab`."""
        eng = self.eng
        eng.slab = slab
        dev = self.device
        eng._build_arenas(dev)
        if Cimg != eng.sc[0].down_a.Cin:
            raise RuntimeError(f"dip-amd: input has {Cimg} channels, net expects {eng.sc[0].down_a.Cin}")
        eng._build_plan(H, W, Cimg)
        if (eng.Hout, eng.Wout) != tuple(t0.shape[2:]):
            raise ValueError(f"GroupedFits: targets are {tuple(t0.shape[2:])}, the net output is {(eng.Hout, eng.Wout)}")
        oc = eng.out_conv
        nin = Cimg * H * W
        nout = oc.Cout * eng.Hout * eng.Wout
        ex = {}
        ex["saved"] = slab.alloc(nin)                       # net_input_saved (denoising.ipynb:198)
        ex["noisy"] = slab.alloc(nin) if self.std > 0 else None
        ex["rng"] = slab.alloc(2, torch.int64, zero=True)   # {Philox offset, seed} (dip_noise_axpy_dev2)
        ex["target"] = slab.alloc(nout)
        ex["mask"] = slab.alloc(self.mask_c * eng.Hout * eng.Wout) if m0 is not None else None
        ex["out"] = slab.alloc(nout)
        self.nblk = self.lib.dip_loss_head_nblk(eng.Hout * eng.Wout, oc.Cin)
        ex["partials"] = slab.alloc(self.nblk)
        ex["loss"] = slab.alloc(1, zero=True)
        ex["gl"] = slab.alloc(1)                            # d(total_loss)/d(total_loss) = 1 (total_loss.backward())
        ex["gl"].fill_(1.0)
        ex["m"] = slab.alloc(eng.n_arena, zero=True)        # Adam moments over the parameter arena
        ex["v"] = slab.alloc(eng.n_arena, zero=True)
        ex["iter"] = slab.alloc(16, torch.uint8, zero=True)  # DipIterState: step count, step size, sqrt(bias correction 2)
        self._row0_extra = ex
        self._x = ex["noisy"] if self.std > 0 else ex["saved"]
        a = eng.last_act
        ptr = lambda t, off=0: None if t is None else t.data_ptr() + 4 * off
        self._tr = a.transform()
        self._head = N.DipLossHeadDesc(ptr(a.buf), a.Cs, oc.Cin, self._tr, ptr(eng.params, oc.w_off),
                                       ptr(eng.params, oc.b_off) if oc.b_off >= 0 else None, oc.Cout, eng.Hout * eng.Wout,
                                       1 if eng.need_sigmoid else 0, ptr(ex["target"]), ptr(ex["mask"]), self.mask_c,
                                       ptr(ex["out"]), ptr(ex["partials"]), self.nblk, ptr(ex["loss"]))

    def _off(self, t0):
        o = t0.data_ptr() - self.mem.data_ptr()
        assert 0 <= o and o + t0.numel() * t0.element_size() <= self.stride, "tensor is not in row 0 of the slab"
        return o

    def _inst(self, t0, b):
        """The buffer of instance b that corresponds to t0 (a flat tensor in row 0)."""
        o = self._off(t0) + b * self.stride
        return self.mem[o:o + t0.numel() * t0.element_size()].view(t0.dtype)

    def _strided(self, t0, shape, inner_strides):
        """[B, ...] view over the instances of t0 (row 0), without a copy."""
        es = t0.element_size()
        assert self.stride % es == 0
        flat = self.mem.view(t0.dtype)
        # (as_strided's offset counts from the start of the STORAGE, not of `flat`)
        return torch.as_strided(flat, shape, (self.stride // es,) + tuple(inner_strides),
                                flat.storage_offset() + self._off(t0) // es)

    def _adopt_net(self, b):
        """Parameters and BatchNorm buffers of nets[b] move into row b (same offsets as instance 0) and become views of it,
        like SkipEngine._build_arenas does for a solo net: state_dict() / load_state_dict() / .data.copy_() keep working."""
        eng = self.eng
        net = self.nets[b]
        prow = self._inst(eng.params, b)
        for p, o in zip(net.parameters(), eng.slots):
            n = p.numel()
            prow[o:o + n].copy_(p.detach().reshape(-1).to(device=self.device, dtype=torch.float32))
            p.data = prow[o:o + n].view(p.shape)
            p.grad = None
        bnrow, nbtrow = self._inst(eng.bnbuf, b), self._inst(eng.nbt, b)
        bn_mods = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        by_name = dict(net.named_modules())
        names0 = {id(m): k for k, m in self.nets[0].named_modules()}
        for k, rec in enumerate(eng.bns):
            m = by_name[names0[id(rec.module)]]                 # the same BatchNorm in nets[b]
            if rec.rm_off >= 0:
                bnrow[rec.rm_off:rec.rm_off + rec.C].copy_(m.running_mean.to(self.device))
                bnrow[rec.rv_off:rec.rv_off + rec.C].copy_(m.running_var.to(self.device))
                nbtrow[k] = m.num_batches_tracked.to(self.device)
                m._buffers["running_mean"] = bnrow[rec.rm_off:rec.rm_off + rec.C]
                m._buffers["running_var"] = bnrow[rec.rv_off:rec.rv_off + rec.C]
                m._buffers["num_batches_tracked"] = nbtrow[k]
        assert len(bn_mods) == len(eng.bns)

    # ------------------------------------------------------------------ one iteration
    def _iteration(self):
        """optimizer.zero_grad(); closure(); optimizer.step() for all B instances: ONE launch list."""
        eng, lib, ex = self.eng, self.lib, self._row0_extra
        dev = self.device
        main = torch.cuda.current_stream(dev)
        st = main.cuda_stream
        N.check(lib.dip_group_begin(self.B, self.stride, self.mem.data_ptr(), self.stride), "group_begin")
        try:
            # closure: net_input = net_input_saved + noise.normal_() * reg_noise_std
            if self.std > 0:
                N.check(lib.dip_noise_axpy_dev2(ex["saved"].data_ptr(), ex["noisy"].data_ptr(), ex["saved"].numel(), self.std,
                                                ex["rng"].data_ptr(), st), "noise_axpy_dev2")
            # out = net(net_input); total_loss = mse(out [* mask], target [* mask])
            eng._launch_forward(self._x.data_ptr(), main, with_out_conv=False)
            N.check(lib.dip_loss_head_fwd(C.byref(self._head), st), "loss_head_fwd")
            # total_loss.backward()
            N.check(lib.dip_loss_head_bwd(C.byref(self._head), ex["gl"].data_ptr(), eng.dy_out.data_ptr(),
                                          round_up(eng.n_out, 4), st), "loss_head_bwd")
            eng._launch_backward(main)
            # optimizer.step(): torch.optim.Adam semantics (dip_optim.FusedAdam), step count on the device
            N.check(lib.dip_adam_tick(ex["iter"].data_ptr(), self.lr, self.ADAM_BETAS[0], self.ADAM_BETAS[1], st), "adam_tick")
            N.check(lib.dip_adam_step_dev(eng.params.data_ptr(), eng.grads.data_ptr(), ex["m"].data_ptr(), ex["v"].data_ptr(),
                                          eng.n_arena, self.ADAM_BETAS[0], self.ADAM_BETAS[1], self.ADAM_EPS,
                                          ex["iter"].data_ptr(), st), "adam_step_dev")
        finally:
            lib.dip_group_end()
        # ATen, batched over the instances: BatchNorm's num_batches_tracked and the closure's out_avg
        if len(eng.bns):
            self._nbt_all.add_(1)
        if self.out_avg is not None:
            if self.iterations == 0 and self.ema_first and not torch.cuda.is_current_stream_capturing():
                self.out_avg.copy_(self.out)
            else:
                self.out_avg.mul_(self.exp_weight).add_(self.out, alpha=1 - self.exp_weight)

    def step(self, n=1):
        """n eager iterations (launches on the current stream + the engine's auxiliary streams)."""
        if self._dry:
            raise RuntimeError("dip-amd: a dry (host-memory) GroupedFits cannot launch anything")
        with torch.cuda.device(self.device):
            for _ in range(int(n)):
                self._iteration()
                self.iterations += 1

    def capture(self, warmup=3):
        """`warmup` eager iterations, then the grouped iteration as ONE hipGraph (replayed by run())."""
        dev = self.device
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            self.capture_stream = torch.cuda.Stream(dev)
            self.capture_stream.wait_stream(cur)
            with torch.cuda.stream(self.capture_stream):
                self.step(max(int(warmup), 1))
            cur.wait_stream(self.capture_stream)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.capture_stream):
                self._iteration()
        return self

    def run(self, n=1):
        if self.graph is None:
            return self.step(n)
        for _ in range(int(n)):
            self.graph.replay()
        self.iterations += int(n)

    def step_counts(self):
        """Adam's step count of every instance, as the device holds it."""
        return [int(self._inst(self._row0_extra["iter"], b).view(torch.int64)[0].item()) for b in range(self.B)]
