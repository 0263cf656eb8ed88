"""Execution engine: compiles a `skip()` module tree (models/skip.py) into a static list of
libdip_hip.so kernel launches for forward and backward, and exposes it to autograd as ONE
`torch.autograd.Function`, so the reference's closure-style loop (`out = net(z); loss(...)
.backward(); optimizer.step()`, utils/common_utils.py:223-230 of the reference) runs unchanged.

MI355X-first design:
  * activations live in HBM as NHWC fp32, channel stride padded to 4; every conv is an
    implicit-GEMM on the fp32 MFMA with the PRODUCER's BatchNorm-apply + LeakyReLU and the
    padding fused into its loader and the CONSUMER BatchNorm's statistics fused into its epilogue;
  * parameters, gradients, Adam moments and BatchNorm running stats are flat arenas; the
    nn.Parameters the user sees are views into them (state_dict()/load_state_dict()/.data.copy_()
    keep working), and one fused launch steps Adam over the whole arena;
  * all buffers are allocated once per input size; the launch list is static (hipGraph-capturable).

PyTorch is used for device memory, streams and the autograd boundary only.  There is no eager
fallback: a missing libdip_hip.so or an unsupported option raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

import dip_native as N
from dip_native import round_up


def _ptr(t: Optional[torch.Tensor], off_floats: int = 0) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr() + 4 * off_floats


class BNRec:
    """One BatchNorm2d (train mode): parameters in the arena + a device state block
    [mean, rstd, a, b] x Cs and backward coefficients [k1, k2] x Cs."""

    def __init__(self, module: torch.nn.BatchNorm2d, name: str):
        self.module = module
        self.name = name
        self.C = module.num_features
        self.Cs = round_up(self.C, 4)
        self.gamma_off = self.beta_off = -1
        self.rm_off = self.rv_off = -1
        self.state = None
        self.coef = None


class ConvRec:
    def __init__(self, module: torch.nn.Conv2d, pad_mode: int, name: str):
        self.module = module
        self.name = name
        self.Cin, self.Cout = module.in_channels, module.out_channels
        self.ks = module.kernel_size[0]
        self.stride = module.stride[0]
        self.P = (self.ks - 1) // 2
        self.pad_mode = pad_mode if self.P > 0 else N.PAD_ZERO
        self.w_off = self.b_off = -1
        self.fwd_off = self.dgrad_off = -1
        self.need_dgrad = True

    @property
    def fwd_elems(self):
        return self.ks * self.ks * round_up(self.Cin, 4) * round_up(self.Cout, 32)

    @property
    def dgrad_elems(self):
        return self.ks * self.ks * round_up(self.Cout, 4) * round_up(self.Cin, 32)


class Act:
    """An activation as a consumer sees it: raw NHWC buffer + the (BatchNorm, LeakyReLU slope)
    that the consumer's loader applies on the fly."""

    def __init__(self, buf, H, W, Cch, bn: Optional[BNRec] = None, slope: float = 1.0):
        self.buf, self.H, self.W, self.C = buf, H, W, Cch
        self.Cs = round_up(Cch, 4)
        self.bn, self.slope = bn, slope

    def transform(self, shape_only=False) -> N.DipTransform:
        if self.bn is None:
            return N.DipTransform(None, None, 1.0)
        if shape_only:          # planner's sizing pass: only "there is a transform" matters (never dereferenced)
            return N.DipTransform(1, 1, self.slope)
        assert self.bn.Cs == self.Cs
        return N.DipTransform(_ptr(self.bn.state, 2 * self.Cs), _ptr(self.bn.state, 3 * self.Cs), self.slope)


class ScalePlan:
    """Module handles of one scale of the hour-glass (filled by models.skip.skip())."""

    def __init__(self):
        self.skip_conv = self.skip_bn = None
        self.down_a = self.down_a_bn = self.down_b = self.down_b_bn = None
        self.cat_bn = self.up = self.up_bn = self.up1 = self.up1_bn = None
        self.ns = 0
        self.upsample_mode = "nearest"
        self.pool = None                    # 'avg' | 'max': down_a is a stride-1 conv followed by Avg/MaxPool2d(2, 2);
        self.down_ds = None                 # 'lanczos': ... followed by the Downsampler's dense stride-2 conv (down_ds)


class SkipEngine:
    def __init__(self, net, scales: List[ScalePlan], out_conv: torch.nn.Conv2d, need_sigmoid: bool, pad: str,
                 act_slope: float = 0.2):
        self.net = net
        self.need_sigmoid = need_sigmoid
        self.pad_mode = N.PAD_REFLECT if pad == "reflection" else N.PAD_ZERO
        self.slope = act_slope
        self.nscales = len(scales)
        self.fwd_id = 0
        self._aux = {}                 # (capture | eager, device) -> ([side stream, bulk stream], events)
        self.two_streams = os.environ.get("DIP_TWO_STREAMS", "1") != "0"
        # backward schedule (see _run_two_streams): the weight gradients emitted before the backward walk reaches
        # scale `defer_scale` are held back until then, later ones until the end of their scale's decoder / encoder
        # part (-1: every one is launched where it is emitted)
        self.defer_scale = int(os.environ.get("DIP_DEFER_WGRAD", "2")) if self.two_streams else -1
        # forward: skip-branch convs below this many pixels stay on the main stream (a fork + join costs more
        # than the ~8 us launch it would overlap)
        self.side_min_pixels = int(os.environ.get("DIP_SIDE_MIN_PIXELS", "0"))
        self._fwd_side, self._deferred, self._entered_defer_scale, self._fused_bnb = set(), [], False, {}
        # the launch lists are compiled into command lists (dip_native.CmdList -> dip_list_run: one foreign call per direction
        # and iteration instead of one per launch); DIP_NO_CLIST=1 issues them from Python, launch by launch, as rounds 1-5 did
        self.use_clist = os.environ.get("DIP_NO_CLIST") is None
        # DIP_KNOCKOUT=<regex over op names>: a knock-out timing experiment (what would the iteration gain if these launches
        # cost nothing?); never a valid configuration -- bench.py refuses to print a headline line with it set
        import re
        ko = os.environ.get("DIP_KNOCKOUT")
        self._knockout = re.compile(ko) if ko else None
        self._ko_after, self._ko_seen = int(os.environ.get("DIP_KNOCKOUT_AFTER", "0")), {}
        self._clists = {}
        self._replicate_bufs = set()
        # BatchNorm-backward statistics in the epilogue of the data-gradient launch (DipConvDesc.bnb_*) instead of a pass
        # of their own: 18 launches and one pass over g fewer, but measured (round 3) as a wash -- the epilogue's
        # per-lane reads of y cost the big launches 35..75 us each, as much as the streaming statistics kernels they
        # replace (+0.5 % on a fast-class box, -1.5 % on a slow-class one) -- so it is opt-in
        self.fuse_bnb = os.environ.get("DIP_BNB_FUSE", "0") == "1"
        # DIP_BNB_FUSE=2: only in the epilogue of the 1x1 data gradients (conv1x1_res_kernel: a lane owns a channel, so its reads of
        # y are whole lines) -- an experiment of round 6's last session
        self.fuse_bnb_1x1 = os.environ.get("DIP_BNB_FUSE", "0") == "2"
        self.thin_src = os.environ.get("DIP_NO_THIN_SRC") is None
        # low-resolution layers (<= DIP_SMALL_MAX_PIXELS output pixels): ONE dip_conv_small launch per convolution
        # (conv + in-workgroup split-K + BatchNorm partials; dip_bn_finalize follows) instead of conv + split-K finish +
        # bn_finalize, and data gradient + BatchNorm-backward partials (+ dip_bn_bwd_finalize2) instead of four launches
        # (csrc/conv_small.hip); DIP_CONV_NO_SMALL=1 restores the round-3 lists
        self.use_small = os.environ.get("DIP_CONV_NO_SMALL") is None
        # in-launch BatchNorm finalisation by the last workgroup to arrive (csrc/bn_ticket.h) for up-sample + concat and
        # the backward statistics passes: built, parity-tested, and measured SLOWER than a finalisation launch of its own
        # (the write-through stores, the ticket and the L2-bypassing row loads are three dependent memory round trips at
        # the end of the producer: -1.5 % end to end) -- opt-in
        self.ticket_fin = os.environ.get("DIP_TICKET_FIN") == "1"
        # low-resolution BatchNorm backward as ONE launch (csrc/bn_bwd_one.hip: a workgroup owns four channels of the whole
        # plane, so statistics -> finalise -> apply needs no grid-wide dependency); the library decides per shape
        # (dip_bn_bwd_one_ok: DIP_BNB_ONE_MAX_PIXELS, 0 = off)
        self.bnb_one = os.environ.get("DIP_BNB_NO_ONE") is None
        self.tail_inline = self.two_streams and os.environ.get("DIP_TAIL_INLINE", "1") != "0"
        # phase 2 of a BatchNorm backward in the prologue of its apply launch when phase 1 left few partial rows
        # (dip_bn_bwd_apply_src_fin / _apply_fin; the library decides: dip_bn_bwd_fin_rows_ok, DIP_BNB_FIN_MAX_ROWS)
        # MEASURED (profiles/r06_ab_bnb_fin_fuse.txt): -8 launches per iteration of the default net, -10 of the 'library' net,
        # and -0.4 % / -0.5 % it/s -- a dependent 5 us launch costs the chain ~2.5 us, the prologue's rows cost every block
        # as much.  Opt-in (DIP_BNB_FIN_FUSE=1); parity-tested either way (tests/test_bnone_gpu.py).
        self.bnb_fin_fuse = os.environ.get("DIP_BNB_FIN_FUSE") == "1"
        # second bulk stream (round 6): the weight gradients of layers with <= DIP_BULK2_MAX_PIXELS output pixels alternate
        # between two bulk streams (scratch sets of their own), so that two of these small launches -- none fills the chip --
        # run side by side.  The 'library' net's bulk stream was backlogged: 1.35 ms of weight gradients queued one after the
        # other, the last three still waiting when the main chain had finished (profiles/r06_library_timeline_*.txt)
        # Only for nets WITHOUT big MFMA-bound layers: with them (the default / kate nets: 128 x 128-channel 3x3 convs at
        # >= 256 x 256) the one bulk stream is not backlogged and a second one costs 2 % (more fork events on the main
        # chain, two MFMA-bound launches side by side gain nothing): library 370.6 -> 383.0 it/s, default 182.4 -> 178.1
        # (profiles/r06_ab_bulk2.txt).  DIP_BULK2_MAX_PIXELS overrides (0 = off).
        self.bulk2_max_pixels = 0
        if self.two_streams:
            env = os.environ.get("DIP_BULK2_MAX_PIXELS")
            self.bulk2_max_pixels = int(env) if env is not None else -1        # -1: decided per plan (_build_plan)
        self._bulk2, self._bulk_flip = set(), False
        # bf16 matrix pipe for the big 3x3 layers (csrc/conv_bf3.hip: fp32 operands as three exact bf16 terms, the
        # cross products accumulated in fp32): the library decides per descriptor (DIP_CONV_BF3=8 | 9 | 6 | 0), the engine only
        # keeps the split weight planes up to date
        self.bf3 = False
        self.device = None
        self.shape_key = None
        self.lib = None
        # device-memory source: None = torch's allocator; dip_group.Slab = every buffer of the fit (arenas, activations,
        # scratch, tables) is carved from ONE slab, so that B identically laid out slabs run through one launch list
        # (grouped multi-instance execution, csrc/dip_group.h)
        self.slab = None

        # records in a fixed traversal order
        self.convs: List[ConvRec] = []
        self.bns: List[BNRec] = []
        self.sc = []
        for i, s in enumerate(scales):
            rec = ScalePlan()
            rec.ns, rec.upsample_mode = s.ns, s.upsample_mode
            rec.pool = getattr(s, 'pool', None)
            for attr in ("skip_conv", "down_a", "down_ds", "down_b", "up", "up1"):
                m = getattr(s, attr, None)
                if m is not None:
                    r = ConvRec(m, self.pad_mode, f"s{i}.{attr}")
                    if attr == "down_ds":       # Downsampler(preserve_size=True): nn.ReplicationPad2d((k - factor) / 2)
                        r.P = (r.ks - r.stride) // 2
                        r.pad_mode = N.PAD_REPLICATE
                    self.convs.append(r)
                    setattr(rec, attr, r)
            for attr in ("skip_bn", "down_a_bn", "down_b_bn", "cat_bn", "up_bn", "up1_bn"):
                m = getattr(s, attr)
                if m is not None:
                    r = BNRec(m, f"s{i}.{attr}")
                    self.bns.append(r)
                    setattr(rec, attr, r)
            self.sc.append(rec)
        self.out_conv = ConvRec(out_conv, self.pad_mode, "out")
        self.convs.append(self.out_conv)
        # the two convs reading net_input need no data gradient unless the input is optimised
        self.param_list = list(net.parameters())
        self._check_supported()

    # ------------------------------------------------------------------ support matrix
    def _check_supported(self):
        for r in self.convs:
            lanczos_ds = r.name.endswith(".down_ds") and r.ks in (8, 12) and r.stride == 2
            if not lanczos_ds and (r.ks not in (1, 3, 5, 7) or r.stride not in (1, 2) or (r.ks == 1 and r.stride != 1)):
                raise NotImplementedError(f"dip-amd: conv {r.name} k={r.ks} s={r.stride} has no gfx950 kernel")
            m = r.module
            if m.dilation != (1, 1) or m.groups != 1 or m.kernel_size[0] != m.kernel_size[1]:
                raise NotImplementedError(f"dip-amd: conv {r.name}: dilation/groups/non-square unsupported")
        for b in self.bns:
            if b.module.momentum is None:
                raise NotImplementedError(f"dip-amd: BatchNorm {b.name}: momentum=None (cumulative moving average of "
                                          "the running statistics) is not implemented; the reference uses 0.1")
            if not b.module.affine:
                raise NotImplementedError(f"dip-amd: BatchNorm {b.name}: affine=False is not implemented")
        for i, s in enumerate(self.sc):
            if s.ns % 4 or s.down_b.Cout % 4 or s.up.Cout % 4 or (s.up.Cin % 4):
                raise NotImplementedError("dip-amd: internal channel counts must be multiples of 4")

    # ------------------------------------------------------------------ device memory
    def _dalloc(self, n, dtype=torch.float32, zero=False):
        """n elements of device memory from the slab (when one is set) or from torch."""
        if self.slab is not None:
            return self.slab.alloc(int(n), dtype, zero)
        return (torch.zeros if zero else torch.empty)(int(n), dtype=dtype, device=self.device)

    def _dcopy(self, host_bytes):
        """A host table (ctypes array) as a uint8 device tensor."""
        src = torch.frombuffer(bytearray(host_bytes), dtype=torch.uint8)
        t = self._dalloc(src.numel(), torch.uint8)
        t.copy_(src)
        return t

    # ------------------------------------------------------------------ arenas
    def _build_arenas(self, device):
        self.lib = N.lib()
        self.device = device
        params = self.param_list
        off = 0
        slots = []
        for p in params:
            slots.append(off)
            off += round_up(p.numel(), 4)
        self.n_arena = off
        self.params = self._dalloc(off, zero=True)
        self.grads = self._dalloc(off, zero=True)
        self.slots = slots
        with torch.no_grad():
            for p, o in zip(params, slots):
                n = p.numel()
                self.params[o:o + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
                p.data = self.params[o:o + n].view(p.shape)
        pid = {id(p): o for p, o in zip(params, slots)}
        for r in self.convs:
            r.w_off = pid[id(r.module.weight)]
            r.b_off = pid[id(r.module.bias)] if r.module.bias is not None else -1
        # BatchNorm buffers arena
        nb = sum(2 * b.Cs for b in self.bns)
        self.bnbuf = self._dalloc(max(nb, 4), zero=True)
        self.nbt = self._dalloc(max(len(self.bns), 1), torch.int64, zero=True)
        o = 0
        with torch.no_grad():
            for k, b in enumerate(self.bns):
                m = b.module
                b.gamma_off, b.beta_off = pid[id(m.weight)], pid[id(m.bias)]
                if m.track_running_stats and m.running_mean is not None:
                    b.rm_off, b.rv_off = o, o + b.Cs
                    self.bnbuf[o:o + b.C].copy_(m.running_mean.to(device))
                    self.bnbuf[o + b.Cs:o + b.Cs + b.C].copy_(m.running_var.to(device))
                    m._buffers["running_mean"] = self.bnbuf[o:o + b.C]
                    m._buffers["running_var"] = self.bnbuf[o + b.Cs:o + b.Cs + b.C]
                    self.nbt[k] = m.num_batches_tracked.to(device)
                    m._buffers["num_batches_tracked"] = self.nbt[k]
                o += 2 * b.Cs
                b.state = self._dalloc(4 * b.Cs, zero=True)
                b.coef = self._dalloc(2 * b.Cs, zero=True)
        # packed weights
        off = 0
        recs = (N.DipPackRec * len(self.convs))()
        max_elems = 0
        for k, r in enumerate(self.convs):
            r.fwd_off = off
            off += r.fwd_elems
            r.dgrad_off = off
            off += r.dgrad_elems
            recs[k] = N.DipPackRec(r.w_off, r.fwd_off, r.dgrad_off, r.Cout, r.Cin, r.ks, round_up(r.Cin, 4),
                                   round_up(r.Cout, 32), round_up(r.Cout, 4), round_up(r.Cin, 32))
            max_elems = max(max_elems, r.fwd_elems + r.dgrad_elems)
        self.packed = self._dalloc(off, zero=True)
        # three-bf16-plane copies of the 3x3 stride-1 weights (bf16-pipe convolution), when the library has it switched on
        self.bf3 = bool(self.lib.dip_conv_bf3_terms())
        if self.bf3:
            off3, max3 = 0, 0
            recs3 = (N.DipPackRec3 * len(self.convs))()
            for k, r in enumerate(self.convs):
                r.fwd3_off = r.dgrad3_off = -1
                nchF, nchD = (round_up(r.Cin, 4) + 15) // 16, (round_up(r.Cout, 4) + 15) // 16
                CoutP, CinP = round_up(r.Cout, 32), round_up(r.Cin, 32)
                # (1x1 layers, round 6, opt-in -- measured slower per iteration: conv_bf3_k1_kernel takes them from 256 tiles,
                # whole 16-channel chunks on both sides)
                k1 = (os.environ.get("DIP_CONV_BF3_1X1") == "1" and r.ks == 1 and r.Cin % 16 == 0 and r.Cout % 16 == 0
                      and min(r.Cin, r.Cout) >= 128)
                if (r.ks == 3 or k1) and r.stride == 1 and not r.name.endswith(".down_ds"):
                    kk = r.ks * r.ks
                    r.fwd3_off = off3
                    off3 += kk * nchF * 3 * CoutP * 16
                    r.dgrad3_off = off3
                    off3 += kk * nchD * 3 * CinP * 16
                    max3 = max(max3, kk * 16 * (nchF * CoutP + nchD * CinP))
                recs3[k] = N.DipPackRec3(r.w_off, r.fwd3_off, r.dgrad3_off, r.Cout, r.Cin, r.ks, nchF, CoutP, nchD, CinP)
            self.packed3 = self._dalloc(max(off3, 8), torch.int16, zero=True)
            self.pack_recs3 = self._dcopy(bytes(recs3))
            self.pack_max3 = max3
        self.pack_recs = self._dcopy(bytes(recs))
        self.pack_max = max_elems
        self.shape_key = None

    def _arena_ok(self) -> bool:
        if self.device is None:
            return False
        base = self.params.data_ptr()
        for p, o in zip(self.param_list, self.slots):
            if p.data_ptr() != base + 4 * o:
                return False
        return True

    # ------------------------------------------------------------------ per-shape plan
    def _new(self, n):
        t = self._dalloc(n)
        self._alloc.append(t)          # the launch descriptors hold raw pointers: keep every buffer alive
        return t

    def _reset_sizing(self):
        """Sizes of the shared scratch buffers, accumulated by the sizing pass of the planner."""
        self.stat_need = self.wg_need = self.wgb_need = self.bwdp_need = self.ws_need = 4
        self.wg2_need = self.wgb2_need = 4             # slab scratch of the second bulk stream
        self.stat2_need = self.ws2_need = 4            # scratch of the skip-branch convs (side stream)
        self.bwdp2_need = 4                            # ... and of the skip-branch BatchNorm backward
        self.bwdp3_need = 4                            # fused BatchNorm-backward partials of the thin data-gradient columns
        self.ticket_need = 8                           # arrival counters of the in-launch finalisations (bn_ticket.h)

    def _build_plan(self, H, W, Cin_img):
        if min(H, W) < 2 ** self.nscales:
            raise NotImplementedError(f"dip-amd: input {H}x{W} is too small for {self.nscales} scales")
        self.H, self.W, self.Cimg = H, W, Cin_img
        self._clists = {}
        self._reset_sizing()
        self._alloc = []
        oc = self.out_conv
        self.n_out = oc.Cout
        if getattr(self, "_bulk2_auto", self.bulk2_max_pixels < 0):
            # second bulk stream iff no conv of the net is a big MFMA-bound layer at this input size (see __init__)
            self._bulk2_auto = True
            big, h, w = False, H, W
            for sc in self.sc:
                hl, wl = ((h + 1) // 2, (w + 1) // 2) if sc.pool is None else (h // 2, w // 2)
                for r, px in ((sc.down_a, hl * wl), (sc.down_b, hl * wl), (sc.up, h * w), (sc.up1, h * w)):
                    if r is not None and r.ks >= 3 and r.Cin >= 96 and r.Cout >= 96 and px >= 65536:
                        big = True
                h, w = hl, wl
            self.bulk2_max_pixels = 0 if big else 100000
        # The plan is generated twice: a sizing pass (no allocations, no descriptors) that only
        # measures the shared scratch buffers, then the emitting pass.
        for sizing in (True, False):
            self._sizing = sizing
            self.fwd_ops, self.bwd_ops, self.bwd_input_ops, self.keep = [], [], [], []
            if not sizing:
                self.stats_scratch = self._new(self.stat_need)
                self.bwd_scratch = self._new(self.bwdp_need)
                self.wg_scratch = self._new(self.wg_need)
                self.wgb_scratch = self._new(self.wgb_need)
                self.wg_scratch2 = self._new(self.wg2_need)
                self.wgb_scratch2 = self._new(self.wgb2_need)
                self.ws_scratch = self._new(self.ws_need)
                self.stats_scratch2 = self._new(self.stat2_need)
                self.ws_scratch2 = self._new(self.ws2_need)
                self.bwd_scratch2 = self._new(self.bwdp2_need)
                self.bwd_scratch3 = self._new(self.bwdp3_need)
                # zero-initialised; every launch leaves its counters at zero
                self.tickets = self._dalloc(self.ticket_need, torch.int32, zero=True)
                self._alloc.append(self.tickets)
            self._ticket_off = 0
            self._fused_bnb = {}
            self._deferred = []
            self._replicate_bufs = set()
            self._fwd_side = set()
            self._bulk2, self._bulk_flip = set(), False
            self._entered_defer_scale = False
            self.x_nhwc = self._buf(H * W * round_up(Cin_img, 4))
            xin = Act(self.x_nhwc, H, W, Cin_img)
            last = self._plan_scale(0, xin, H, W)
            # output conv (no BatchNorm) + sigmoid head
            # (the net's output is the size of the top scale's concat: smaller than the input when pooling floors an odd
            # size and Concat crops the skip branch, models/common.py:29-37)
            self.Hout, self.Wout = last.H, last.W
            self.y_out = self._buf(last.H * last.W * round_up(oc.Cout, 4))
            self._emit_conv_fwd(oc, last, self.y_out, None)
            # backward: head, out conv, then the scales from the top
            self.dy_out = self._buf(last.H * last.W * round_up(oc.Cout, 4))
            pre = []
            self._emit_wgrad(oc, last, self.dy_out, pre, scale=0)
            if self._thin_src_ok(oc, last):
                # the 1x1 output conv's data gradient is not a launch: the BatchNorm backward of `last` evaluates
                # du = W^T dy_out per pixel where it reads it (DipGradSrc.tw) -- three passes over a [H][W][C] tensor less
                du_last = (self.dy_out, 0, oc)
            else:
                du_last = self._emit_dgrad(oc, last, self.dy_out, pre, fuse_bn=True)
            dy_last = self._emit_bn_act_bwd(last, du_last, pre)
            self.dbg_top = {"du_last": du_last, "dy_last": dy_last, "last": last}
            self.last_act = last                         # input of the output conv (utils/loss_head.MSEHead)
            self.bwd_ops = pre + self._bwd_scale_ops(0, dy_last)
            self.bwd_ops += self._flush_deferred_wgrads()       # (fewer scales than defer_scale)
        self.shape_key = (H, W, Cin_img)

    def _plan_scale(self, i, xin: Act, H, W):
        s = self.sc[i]
        st = {}
        # strided convs: ceil(S / 2); for an odd S the x2 up-sampled tensor is one larger than the skip branch and
        # Concat's centre crop (offset 0) drops its last row / column -- dip_upcat_fwd never produces it
        Hl, Wl = ((H + 1) // 2, (W + 1) // 2) if s.pool is None else (H // 2, W // 2)
        if s.ns:
            st["s_y"] = self._buf(H * W * round_up(s.ns, 4))
            self._emit_conv_fwd(s.skip_conv, xin, st["s_y"], s.skip_bn)
            st["s_act"] = Act(st["s_y"], H, W, s.ns, s.skip_bn, self.slope)
        st["d1_y"] = self._buf(Hl * Wl * round_up(s.down_a.Cout, 4))
        if s.pool in ('avg', 'max'):
            # conv(..., downsample_mode='avg' | 'max'), models/common.py:101-104: full-resolution conv, then the
            # pooling pass produces the BatchNorm partials of the pooled tensor
            st["d1_full"] = self._buf(H * W * round_up(s.down_a.Cout, 4))
            self._emit_conv_fwd(s.down_a, xin, st["d1_full"], None)
            self._emit_avgpool(st["d1_full"], H, W, s.down_a.Cout, st["d1_y"], s.down_a_bn, s.pool)
        elif s.pool == 'lanczos':
            # conv(..., downsample_mode='lanczos2' | 'lanczos3'), models/common.py:107-108: full-resolution conv (raw: no
            # BatchNorm in between), then the Downsampler's dense k x k stride-2 conv behind nn.ReplicationPad2d
            st["d1_full"] = self._buf(H * W * round_up(s.down_a.Cout, 4))
            self._emit_conv_fwd(s.down_a, xin, st["d1_full"], None)
            st["d1_raw"] = Act(st["d1_full"], H, W, s.down_a.Cout)
            self._emit_conv_fwd(s.down_ds, st["d1_raw"], st["d1_y"], s.down_a_bn)
        else:
            self._emit_conv_fwd(s.down_a, xin, st["d1_y"], s.down_a_bn)
        d1 = Act(st["d1_y"], Hl, Wl, s.down_a.Cout, s.down_a_bn, self.slope)
        st["d2_y"] = self._buf(Hl * Wl * round_up(s.down_b.Cout, 4))
        self._emit_conv_fwd(s.down_b, d1, st["d2_y"], s.down_b_bn)
        d2 = Act(st["d2_y"], Hl, Wl, s.down_b.Cout, s.down_b_bn, self.slope)
        st["d1"], st["d2"] = d1, d2
        deep = d2
        if i < self.nscales - 1:
            deep = self._plan_scale(i + 1, d2, Hl, Wl)
        st["deep"] = deep
        ccat = s.ns + deep.C
        assert ccat == s.cat_bn.C == s.up.Cin, (ccat, s.cat_bn.C, s.up.Cin)
        # Concat (models/common.py:19-39): [skip branch H x W | x2 up-sampled deeper branch Hu x Wu], both centre-cropped
        # to the smaller size; without a skip branch there is no Concat and the up-sampled size goes on as it is.
        Hu, Wu = 2 * deep.H, 2 * deep.W
        if s.ns:
            Ho, Wo = min(H, Hu), min(W, Wu)
        else:
            Ho, Wo = Hu, Wu
        geom = dict(Hs=H, Ws=W, os_y=(H - Ho) // 2 if s.ns else 0, os_x=(W - Wo) // 2 if s.ns else 0,
                    Hd=deep.H, Wd=deep.W, od_y=(Hu - Ho) // 2, od_x=(Wu - Wo) // 2)
        # the geometry of the 2x2-block kernels: nothing cropped but (for an odd size) the last up-sampled row / column
        default = (not s.ns or (Ho, Wo) == (H, W)) and deep.H == (Ho + 1) // 2 and deep.W == (Wo + 1) // 2 \
            and geom["od_y"] == 0 and geom["od_x"] == 0
        st["geom"] = None if default else geom
        st["cat"] = self._buf(Ho * Wo * round_up(ccat, 4))
        self._emit_upcat(s, st.get("s_act"), deep, st["cat"], Ho, Wo, st["geom"])
        cat = Act(st["cat"], Ho, Wo, ccat, s.cat_bn, 1.0)
        st["cat_act"] = cat
        st["u_y"] = self._buf(Ho * Wo * round_up(s.up.Cout, 4))
        self._emit_conv_fwd(s.up, cat, st["u_y"], s.up_bn)
        u = Act(st["u_y"], Ho, Wo, s.up.Cout, s.up_bn, self.slope)
        st["u"] = u
        res = u
        if s.up1 is not None:
            st["u1_y"] = self._buf(Ho * Wo * round_up(s.up1.Cout, 4))
            self._emit_conv_fwd(s.up1, u, st["u1_y"], s.up1_bn)
            res = Act(st["u1_y"], Ho, Wo, s.up1.Cout, s.up1_bn, self.slope)
            st["u1"] = res
        st["xin"], st["H"], st["W"], st["Ho"], st["Wo"] = xin, H, W, Ho, Wo
        s.st = st
        return res

    def _buf(self, n):
        return None if self._sizing else self._new(n)

    # ------------------------------------------------------------------ op emitters
    def _ticket(self, n=8):
        """n arrival counters for one in-launch finalisation (sizing pass: only counts them)."""
        if self._sizing:
            self.ticket_need += n
            return None
        off = self._ticket_off
        self._ticket_off += n
        assert self._ticket_off <= self.tickets.numel()
        return self.tickets.data_ptr() + 4 * off

    def _bn_fin(self, bn: BNRec, ticket) -> N.DipBnFin:
        m = bn.module
        return N.DipBnFin(_ptr(self.params, bn.gamma_off), _ptr(self.params, bn.beta_off), float(m.eps), float(m.momentum),
                          _ptr(bn.state), bn.Cs, bn.C,
                          _ptr(self.bnbuf, bn.rm_off) if bn.rm_off >= 0 else None,
                          _ptr(self.bnbuf, bn.rv_off) if bn.rv_off >= 0 else None, ticket)

    def _bnb_fin(self, bn: BNRec, npix, ticket) -> N.DipBnbFin:
        return N.DipBnbFin(_ptr(self.grads, bn.gamma_off), _ptr(self.grads, bn.beta_off), _ptr(bn.coef), bn.C, npix,
                           ticket)

    def _emit_bn_finalize(self, bn: BNRec, scratch, rows, cstride):
        """Partial rows -> state block + running statistics (dip_bn_finalize).  A separate launch on purpose: finishing
        inside the producer ("last-arriving workgroup") was built and measured twice -- with an agent-scope release fence
        per workgroup (round 2: +200 us on a 2048-tile conv launch, the fence writes back / invalidates L2 on the multi-XCD
        part) and fence-free (round 4, csrc/bn_ticket.h: sc1 write-through stores + a relaxed ticket, which leans on gfx950
        behaviour the HIP memory model does not promise; -1.5 % end to end: the last arriver's serial read of the rows
        costs more than the 5-7 us launch).  The fence-free form stays opt-in (DIP_TICKET_FIN=1) on the up-sample / backward
        statistics kernels only."""
        m = bn.module
        args = (_ptr(scratch), rows, cstride, bn.C, _ptr(self.params, bn.gamma_off), _ptr(self.params, bn.beta_off),
                float(m.eps), float(m.momentum), _ptr(bn.state), bn.Cs,
                _ptr(self.bnbuf, bn.rm_off) if bn.rm_off >= 0 else None,
                _ptr(self.bnbuf, bn.rv_off) if bn.rv_off >= 0 else None)
        self.fwd_ops.append((self.lib.dip_bn_finalize, args, "bn_fin:" + bn.name))

    def _emit_conv_fwd(self, r: ConvRec, x: Act, y, bn: Optional[BNRec]):
        assert x.C == r.Cin, (r.name, x.C, r.Cin)
        Ho = (x.H + 2 * r.P - r.ks) // r.stride + 1
        Wo = (x.W + 2 * r.P - r.ks) // r.stride + 1
        Cy = round_up(r.Cout, 4)
        sizing = self._sizing
        # (sizing pass: the descriptor carries the shape only -- buffers are None)
        d = N.DipConvDesc(_ptr(x.buf), x.H, x.W, x.Cs, round_up(x.C, 4), x.transform(sizing),
                          None if sizing else _ptr(self.packed, r.fwd_off),
                          _ptr(self.params, r.b_off) if (r.b_off >= 0 and not sizing) else None,
                          _ptr(y), Ho, Wo, Cy, r.Cout, 0, r.ks, r.stride, r.pad_mode, r.P, 1, 0, None, 1, None)
        small = self.use_small and bool(self.lib.dip_conv_small_eligible(C.byref(d)))
        if small:
            ksplit, ntiles, wsf = 1, self.lib.dip_conv_small_rows(C.byref(d)), 0
        else:
            # (a layer the bf16-pipe kernel will not take -- no split weights, a transform over > 512 channels -- is planned for
            # the fp32 kernels: dip_conv_plan's "96..255 tiles in one pass" rule is the 64-column bf16 form's)
            bf3_ok = self.bf3 and getattr(r, "fwd3_off", -1) >= 0 and not (x.bn is not None and round_up(x.C, 4) > 512)
            plan = N.conv_plan if bf3_ok else N.conv_plan_fp32
            ksplit, ntiles, wsf = plan(Ho, Wo, round_up(x.C, 4), r.Cout, r.ks, r.stride)
        # the skip-branch convs run on the side stream next to the encoder convs of their scale
        # (_run_two_streams), so they get scratch of their own
        # (a skip conv below DIP_SIDE_MIN_PIXELS stays on the main stream AND on the main stream's scratch: the side
        # stream may still be running the previous scale's skip conv out of scratch set 2 -- round-3 advisor finding)
        side = r.name.endswith("skip_conv") and Ho * Wo >= self.side_min_pixels and bn is not None
        if side:
            self._fwd_side.update(("conv_fwd:" + r.name, "bn_fin:" + bn.name))
        if sizing:
            if side:
                if bn is not None:
                    self.stat2_need = max(self.stat2_need, ntiles * 3 * round_up(r.Cout, 32))
                self.ws2_need = max(self.ws2_need, wsf)
            else:
                if bn is not None:
                    self.stat_need = max(self.stat_need, ntiles * 3 * round_up(r.Cout, 32))
                self.ws_need = max(self.ws_need, wsf)
            return
        stats_scratch = self.stats_scratch2 if side else self.stats_scratch
        ws_scratch = self.ws_scratch2 if side else self.ws_scratch
        d.stats = _ptr(stats_scratch) if bn is not None else None
        if self.bf3 and getattr(r, "fwd3_off", -1) >= 0:
            d.wp3 = self.packed3.data_ptr() + 2 * r.fwd3_off
        d.ksplit = ksplit
        d.ws = _ptr(ws_scratch) if ksplit > 1 else None
        self.keep.append(d)
        lib = self.lib
        # low resolution: one launch for conv + in-workgroup split-K + BatchNorm partials (no workspace, no finish launch)
        self.fwd_ops.append((lib.dip_conv_small if small else lib.dip_conv_igemm, (C.byref(d),), "conv_fwd:" + r.name))
        if bn is not None:
            self._emit_bn_finalize(bn, stats_scratch, ntiles, round_up(r.Cout, 32))

    def _emit_avgpool(self, x, H, W, Cc, y, bn: BNRec, kind='avg'):
        Cs = round_up(Cc, 4)
        nblk = self.lib.dip_upcat_nblk(H // 2, W // 2, Cc)
        if self._sizing:
            self.stat_need = max(self.stat_need, nblk * 3 * Cs)
            return
        fn = self.lib.dip_maxpool2_fwd if kind == 'max' else self.lib.dip_avgpool2_fwd
        self.fwd_ops.append((fn, (_ptr(x), H, W, Cs, Cc, _ptr(y), Cs, _ptr(self.stats_scratch), nblk), "pool:" + bn.name))
        self._emit_bn_finalize(bn, self.stats_scratch, nblk, Cs)

    def _emit_upcat(self, s, s_act: Optional[Act], deep: Act, cat, H, W, geom=None):
        Ccat = s.ns + deep.C
        Cs_cat = round_up(Ccat, 4)
        nblk = self.lib.dip_upcat_nblk(H, W, Ccat)
        fin_ok = self.ticket_fin and bool(self.lib.dip_fin_rows_ok(nblk, Ccat))
        ticket = self._ticket() if fin_ok else None
        if self._sizing:
            self.stat_need = max(self.stat_need, nblk * 3 * Cs_cat)
            return
        mode = N.UP_BILINEAR if s.upsample_mode == "bilinear" else N.UP_NEAREST
        d = N.DipUpcatDesc(_ptr(s_act.buf) if s_act else None, s_act.Cs if s_act else 4, s.ns,
                           s_act.transform() if s_act else N.DipTransform(None, None, 1.0),
                           _ptr(deep.buf), deep.Cs, deep.C, deep.transform(), H, W, mode, _ptr(cat), Cs_cat,
                           _ptr(self.stats_scratch), nblk)
        if geom is not None:                      # general centre crop (DipUpcatDesc: Hs / Hd geometry)
            for k, v in geom.items():
                setattr(d, k, v)
        self.keep.append(d)
        if fin_ok:      # few partial rows: the last block to arrive finalises the concat BatchNorm (bn_ticket.h)
            fin = self._bn_fin(s.cat_bn, ticket)
            self.keep.append(fin)
            self.fwd_ops.append((self.lib.dip_upcat_fwd_fin, (C.byref(d), C.byref(fin)), "upcat:" + s.cat_bn.name))
            return
        self.fwd_ops.append((self.lib.dip_upcat_fwd, (C.byref(d),), "upcat:" + s.cat_bn.name))
        self._emit_bn_finalize(s.cat_bn, self.stats_scratch, nblk, Cs_cat)

    def _flush_deferred_wgrads(self):
        ops, self._deferred = self._deferred, []
        return ops

    def _emit_wgrad(self, r: ConvRec, x: Act, dy, ops, scale=None):
        """Weight (+ bias) gradient of conv r.  scale = index of the scale being walked: the two launches are held
        back (self._deferred) and enter the list in batches -- those of the high-resolution decoder layers when the
        backward walk reaches self.defer_scale, the others at the end of their scale's decoder / encoder part -- so
        that the bulk stream forks off the main stream once per batch (every fork is an event on the main stream: a
        ~6 us bubble in the dependent chain) and the big ones run underneath the low-resolution walk."""
        Ho = (x.H + 2 * r.P - r.ks) // r.stride + 1
        Wo = (x.W + 2 * r.P - r.ks) // r.stride + 1
        CinP, CoutP = round_up(r.Cin, 32), round_up(r.Cout, 32)
        nsplit, tap_groups, chan_block = N.wgrad_plan2(Ho, Wo, r.Cin, r.Cout, r.ks, r.stride)
        if scale is not None and self.defer_scale >= 0:
            ops = self._deferred            # enters the list at the next _flush_deferred_wgrads()
        slab = r.ks * r.ks * CinP * CoutP
        # small layers alternate between the two bulk streams (same decision in both planner passes)
        second = False
        if Ho * Wo <= self.bulk2_max_pixels:
            second = self._bulk_flip
            self._bulk_flip = not self._bulk_flip
        if self._sizing:
            if second:
                self.wg2_need = max(self.wg2_need, nsplit * slab)
                self.wgb2_need = max(self.wgb2_need, nsplit * CoutP)
            else:
                self.wg_need = max(self.wg_need, nsplit * slab)
                self.wgb_need = max(self.wgb_need, nsplit * CoutP)
            return
        has_b = r.b_off >= 0
        wg, wgb = (self.wg_scratch2, self.wgb_scratch2) if second else (self.wg_scratch, self.wgb_scratch)
        if second:
            self._bulk2.update(("wgrad:" + r.name, "wgred:" + r.name))
        d = N.DipWgradDesc(_ptr(x.buf), x.H, x.W, x.Cs, x.C, x.transform(), _ptr(dy), Ho, Wo,
                           round_up(r.Cout, 4), r.Cout, r.ks, r.stride, r.pad_mode, r.P,
                           _ptr(wg), _ptr(wgb) if has_b else None, nsplit, tap_groups,
                           chan_block)
        self.keep.append(d)
        ops.append((self.lib.dip_conv_wgrad, (C.byref(d),), "wgrad:" + r.name))
        ops.append((self.lib.dip_wgrad_reduce,
                    (_ptr(wg), _ptr(wgb) if has_b else None, nsplit, r.ks, r.Cin, r.Cout,
                     _ptr(self.grads, r.w_off), _ptr(self.grads, r.b_off) if has_b else None), "wgred:" + r.name))

    def _emit_dgrad(self, r: ConvRec, x: Act, dy, ops, accumulate_into=None, fuse_bn=False, need_pad=0):
        """Data gradient of conv r wrt its input x.  Returns a DipGradSrc-describing tuple
        (buf, pad, fold).  accumulate_into = (buf, pad): add the gradient into an existing (padded)
        gradient buffer of the same input (skip-branch conv next to down_a).
        fuse_bn: x is consumed by this conv only, so the launch's output is the complete gradient wrt x's BatchNorm
        (+activation) output: phase 1 of that BatchNorm's backward rides in the launch's epilogue (DipConvDesc.bnb_*)
        when the launch is a one-pass one, and _emit_bn_act_bwd skips its statistics pass (self._fused_bnb).
        need_pad: a later accumulate_into launch (the scale's skip conv, filter_skip_size > 1 with reflection /
        replication padding) needs this gradient buffer on a domain padded by at least that much, so the interior +
        ring form (pad 0) must not be chosen."""
        Ho = (x.H + 2 * r.P - r.ks) // r.stride + 1
        Wo = (x.W + 2 * r.P - r.ks) // r.stride + 1
        reflect = r.pad_mode in (N.PAD_REFLECT, N.PAD_REPLICATE) and r.P > 0      # gradient on the PADDED domain, folded later
        pad = r.P if reflect else 0
        Hg, Wg = x.H + 2 * pad, x.W + 2 * pad
        off = (r.ks - 1) if reflect else (r.ks - 1 - r.P)
        Cg = x.Cs
        if accumulate_into is not None:
            # the skip-branch conv (filter_skip_size, 1x1 in every notebook) reads the same input as down_a:
            # its data gradient is ADDED into down_a's (padded) gradient buffer, at the offset that aligns
            # the two padded domains (a reflected ring position of the smaller pad is the same virtual pixel)
            gbuf, gpad = accumulate_into
            if pad > gpad or r.stride != 1:
                raise NotImplementedError(f"dip-amd: {r.name}: a skip filter larger than the down filter (with "
                                          "reflection padding) or a strided skip conv has no gfx950 path")
            if self._sizing:
                return accumulate_into
            o = gpad - pad
            Wg2 = x.W + 2 * gpad
            ybase = _ptr(gbuf, (o * Wg2 + o) * Cg)
            d = N.DipConvDesc(_ptr(dy), Ho, Wo, round_up(r.Cout, 4), round_up(r.Cout, 4),
                              N.DipTransform(None, None, 1.0), _ptr(self.packed, r.dgrad_off), None,
                              ybase, Hg, Wg, Cg, r.Cin, Wg2, r.ks, 1, N.PAD_ZERO, off, 1, 1, None)
            self.keep.append(d)
            small = self.use_small and bool(self.lib.dip_conv_small_eligible(C.byref(d)))
            ops.append((self.lib.dip_conv_small if small else self.lib.dip_conv_igemm, (C.byref(d),), "dgrad+:" + r.name))
            return accumulate_into
        ring = False
        if reflect and r.pad_mode == N.PAD_REFLECT and r.ks == 3 and r.stride == 1 and self.use_small and need_pad == 0:
            # interior domain + a frame launch instead of the padded domain (dip_conv_dgrad_ring): no lonely second round
            # of tiles at 256^2 (561 -> 512), and the gradient buffer needs no fold
            di = N.DipConvDesc(None, Ho, Wo, round_up(r.Cout, 4), round_up(r.Cout, 4), N.DipTransform(None, None, 1.0), None,
                               None, None, x.H, x.W, Cg, r.Cin, 0, r.ks, 1, N.PAD_ZERO, 1, 1, 0, None, 1, None)
            if self.lib.dip_conv_dgrad_ring_ok(C.byref(di)) and not self.lib.dip_conv_small_eligible(C.byref(di)):
                ring, pad, Hg, Wg, off = True, 0, x.H, x.W, 1
        gbuf = self._buf(Hg * Wg * Cg)
        if r.stride == 2:
            ksplit, _, wsf = N.conv_plan_dil2(Hg, Wg, round_up(r.Cout, 4), r.Cin, r.ks)
        else:
            # (fused BatchNorm-backward partials keep the layer off the bf16-pipe kernel: planned for the fp32 kernels then)
            fuse_here = self.fuse_bnb or (self.fuse_bnb_1x1 and r.ks == 1 and r.stride == 1)
            bf3_ok = self.bf3 and getattr(r, "dgrad3_off", -1) >= 0 and not (fuse_bn and fuse_here and not ring and x.bn is not None)
            ksplit, _, wsf = (N.conv_plan if bf3_ok else N.conv_plan_fp32)(Hg, Wg, round_up(r.Cout, 4), r.Cin, r.ks, 1)
        sizing = self._sizing
        d = N.DipConvDesc(None if sizing else _ptr(dy), Ho, Wo, round_up(r.Cout, 4), round_up(r.Cout, 4),
                          N.DipTransform(None, None, 1.0), None if sizing else _ptr(self.packed, r.dgrad_off), None,
                          None if sizing else _ptr(gbuf), Hg, Wg, Cg, r.Cin, 0, r.ks, 1, N.PAD_ZERO, off, r.stride, 0, None,
                          ksplit, (None if sizing else _ptr(self.ws_scratch)) if ksplit > 1 else None)
        if self.bf3 and not sizing and getattr(r, "dgrad3_off", -1) >= 0:
            d.wp3 = self.packed3.data_ptr() + 2 * r.dgrad3_off
        small = self.use_small and bool(self.lib.dip_conv_small_eligible(C.byref(d)))
        if small:
            # low resolution: ONE launch for all (<= 160) columns, no split-K workspace; when x feeds this conv only, phase 1
            # of its BatchNorm(+activation) backward rides in the epilogue
            d.ksplit, d.ws = 1, None
            rows = self.lib.dip_conv_small_rows(C.byref(d))
            fuse = fuse_bn and x.bn is not None and r.pad_mode != N.PAD_REPLICATE and not self._bnb_one_ok(x)
            if sizing:
                if fuse:
                    self.bwdp_need = max(self.bwdp_need, rows * 2 * x.bn.Cs)
                return (gbuf, pad)
            if r.pad_mode == N.PAD_REPLICATE:
                self._replicate_bufs.add(gbuf.data_ptr())
            if fuse:
                bn = x.bn
                d.bnb_y, d.bnb_state = _ptr(x.buf), _ptr(bn.state)
                d.bnb_partials = _ptr(self.bwd_scratch)
                d.bnb_Cy, d.bnb_Cs, d.bnb_pad, d.bnb_slope = x.Cs, bn.Cs, pad, float(x.slope)
                self._fused_bnb[gbuf.data_ptr()] = (rows, 0, 0)        # -> dip_bn_bwd_finalize2 over these rows, then apply
            self.keep.append(d)
            ops.append((self.lib.dip_conv_small, (C.byref(d),), "dgrad:" + r.name))
            return (gbuf, pad)
        variant = self.lib.dip_conv_variant(C.byref(d))
        fused = None
        fuse_here = self.fuse_bnb or (self.fuse_bnb_1x1 and r.ks == 1 and r.stride == 1)
        if fuse_bn and fuse_here and not ring and x.bn is not None and self.lib.dip_conv_bnb_fusable(C.byref(d)):
            bn = x.bn
            rows = self.lib.dip_conv_ntiles(Hg, Wg)
            c_lo = (r.Cin - 128) if variant == 3 else 0          # columns of the conv_thin4 launch (always 4 here: % 4)
            rows_lo = self.lib.dip_conv_thin4_ntiles(Hg, Wg) if c_lo else 0
            if c_lo % 4 == 0:
                fused = (rows, rows_lo, c_lo)
                self.bwdp_need = max(self.bwdp_need, rows * 2 * bn.Cs)
                self.bwdp3_need = max(self.bwdp3_need, rows_lo * 2 * bn.Cs)
        if sizing:
            self.ws_need = max(self.ws_need, wsf)
            return (gbuf, pad)
        if r.pad_mode == N.PAD_REPLICATE:
            self._replicate_bufs.add(gbuf.data_ptr())       # _gradsrc: fold = 2 (adjoint of nn.ReplicationPad2d)
        if fused is not None:
            bn = x.bn
            d.bnb_y, d.bnb_state = _ptr(x.buf), _ptr(bn.state)
            d.bnb_partials = _ptr(self.bwd_scratch)
            d.bnb_partials_thin = _ptr(self.bwd_scratch3) if fused[2] else None
            d.bnb_Cy, d.bnb_Cs, d.bnb_pad, d.bnb_slope = x.Cs, bn.Cs, pad, float(x.slope)
            self._fused_bnb[gbuf.data_ptr()] = fused
        self.keep.append(d)
        if variant == 3 and self.two_streams:
            # 132-column data gradient = 4 thin columns on the vector ALU + 128 columns on the LDS-DMA kernel:
            # two launches that write disjoint columns, so the thin one goes to the side stream
            # (_run_backward_two_streams makes the BatchNorm backward of the concat wait for it)
            ncols = r.Cin - 128
            ops.append((self.lib.dip_conv_thin4, (C.byref(d), ncols), "dgthin:" + r.name))
            ops.append((self.lib.dip_conv_igemm_dma_cols, (C.byref(d), ncols), "dgrad:" + r.name))
        else:
            ops.append((self.lib.dip_conv_igemm, (C.byref(d),), "dgrad:" + r.name))
        if ring:
            # what the reflected ring folds onto the frame rows / columns, accumulated into the interior gradient
            dr = N.DipConvDesc.from_buffer_copy(d)
            dr.ksplit, dr.ws = 1, None
            self.keep.append(dr)
            ops.append((self.lib.dip_conv_dgrad_ring, (C.byref(dr),), "dgring:" + r.name))
        return (gbuf, pad)

    def _thin_src_ok(self, oc: ConvRec, last: Act) -> bool:
        """The net's last conv (models/skip.py:98) as a gradient source of the BatchNorm in front of it: 1x1, <= 4 output
        channels, a whole number of float4 channel groups in, and an activation above the one-launch BatchNorm backward's range
        (DIP_NO_THIN_SRC=1: the data-gradient launch as before round 6's second session)."""
        return (self.thin_src and oc.ks == 1 and oc.stride == 1 and oc.Cout <= 4 and oc.Cin % 4 == 0 and last.bn is not None
                and not self.fuse_bnb and not self._bnb_one_ok(last))

    def _gradsrc(self, g, Cg, choff=0, window=None):
        if len(g) == 3:                           # (dy of the thin 1x1 conv, 0, its ConvRec): _thin_src_ok
            buf, _, oc = g
            d = N.DipGradSrc(_ptr(buf), 0, 0, round_up(oc.Cout, 4), 0)
            d.tw, d.tn, d.tcw = _ptr(self.params, oc.w_off), oc.Cout, oc.Cin
            self.keep.append(d)
            return d
        buf, pad = g
        fold = 0 if pad == 0 else (2 if buf.data_ptr() in self._replicate_bufs else 1)
        d = N.DipGradSrc(_ptr(buf), pad, fold, Cg, choff)
        if window is not None:                    # (win_y, win_x, win_h, win_w): adjoint of Concat's crop of this branch
            d.win_y, d.win_x, d.win_h, d.win_w = window
        self.keep.append(d)
        return d

    def _emit_bn_act_bwd(self, a: Act, g, ops, choff=0, Cg=None, side=False, window=None):
        """BatchNorm(+LeakyReLU) backward of activation `a` given the gradient source g=(buf,pad)
        wrt the activated value.  Returns the dy buffer (grad wrt a.buf, the conv's raw output).
        side=True: the op runs on the side stream (skip branch) and gets partial-sum scratch of its own."""
        bn = a.bn
        nblk = self.lib.dip_bn_bwd_nblk(a.H, a.W, a.C)
        fin_ok = self.ticket_fin and bool(self.lib.dip_fin_rows_ok(nblk, a.C))
        ticket = self._ticket() if fin_ok else None      # (unused when the data gradient carried the statistics)
        if self._sizing:
            if side:
                self.bwdp2_need = max(self.bwdp2_need, nblk * 2 * a.Cs)
            else:
                self.bwdp_need = max(self.bwdp_need, nblk * 2 * a.Cs)
            return None
        scratch = self.bwd_scratch2 if side else self.bwd_scratch
        dz = self._new(a.H * a.W * a.Cs)
        src = self._gradsrc(g, Cg if Cg is not None else a.Cs, choff, window)
        lib = self.lib
        if self._bnb_one_ok(a) and self._fused_bnb.get(g[0].data_ptr()) is None:
            # low resolution: statistics, finalisation and apply in ONE launch (a workgroup owns 4 channels of the plane)
            ops.append((lib.dip_bn_bwd_one, (C.byref(src), _ptr(a.buf), a.H, a.W, a.Cs, a.C, _ptr(bn.state), bn.Cs,
                                             float(a.slope), _ptr(dz), a.Cs, _ptr(self.grads, bn.gamma_off),
                                             _ptr(self.grads, bn.beta_off), _ptr(bn.coef)), "bnb_one:" + bn.name))
            return dz
        # phase 1 only reduces (dz = NULL); phase 3 recomputes the masked gradient from the source:
        # 5 tensor passes per BatchNorm instead of 6
        fused = self._fused_bnb.get(g[0].data_ptr()) if (choff == 0 and not side) else None
        if fused is None and fin_ok:
            # few partial rows: phase 2 rides in the statistics launch (the last block to arrive reduces them)
            fin = self._bnb_fin(bn, a.H * a.W, ticket)
            self.keep.append(fin)
            ops.append((lib.dip_bn_bwd_stats_fin, (C.byref(src), _ptr(a.buf), a.H, a.W, a.Cs, a.C, _ptr(bn.state), bn.Cs,
                                                   float(a.slope), None, a.Cs, _ptr(scratch), nblk, C.byref(fin)),
                        "bnb_stats:" + bn.name))
        elif fused is not None:
            # phase 1 already ran in the epilogue of the data-gradient launch(es) that produced g (_emit_dgrad)
            rows, rows_lo, c_lo = fused
            if c_lo == 0 and self._fin_rows_ok(rows, a.C):
                ops.append(self._apply_src_fin(src, a, bn, self.bwd_scratch, rows, dz))
                return dz
            ops.append((lib.dip_bn_bwd_finalize2, (_ptr(self.bwd_scratch), rows, _ptr(self.bwd_scratch3) if c_lo else None,
                                                   rows_lo, c_lo, bn.Cs, bn.C, a.H * a.W,
                                                   _ptr(self.grads, bn.gamma_off), _ptr(self.grads, bn.beta_off),
                                                   _ptr(bn.coef)), "bnb_fin:" + bn.name))
        else:
            ops.append((lib.dip_bn_bwd_stats, (C.byref(src), _ptr(a.buf), a.H, a.W, a.Cs, a.C, _ptr(bn.state), bn.Cs,
                                               float(a.slope), None, a.Cs, _ptr(scratch), nblk),
                        "bnb_stats:" + bn.name))
            if self._fin_rows_ok(nblk, a.C):
                # few partial rows: every block of the apply launch reduces them in its prologue (no finalisation launch)
                ops.append(self._apply_src_fin(src, a, bn, scratch, nblk, dz))
                return dz
            ops.append((lib.dip_bn_bwd_finalize, (_ptr(scratch), nblk, bn.Cs, bn.C, a.H * a.W,
                                                  _ptr(self.grads, bn.gamma_off), _ptr(self.grads, bn.beta_off),
                                                  _ptr(bn.coef)), "bnb_fin:" + bn.name))
        ops.append((lib.dip_bn_bwd_apply_src, (C.byref(src), _ptr(a.buf), a.H, a.W, a.Cs, a.C, _ptr(bn.state), bn.Cs,
                                               float(a.slope), _ptr(bn.coef), _ptr(dz), a.Cs), "bnb_apply:" + bn.name))
        return dz

    def _fin_rows_ok(self, rows, Cc) -> bool:
        return self.bnb_fin_fuse and bool(self.lib.dip_bn_bwd_fin_rows_ok(rows, Cc))

    def _apply_src_fin(self, src, a: Act, bn: BNRec, scratch, rows, dz):
        return (self.lib.dip_bn_bwd_apply_src_fin,
                (C.byref(src), _ptr(a.buf), a.H, a.W, a.Cs, a.C, _ptr(bn.state), bn.Cs, float(a.slope), _ptr(scratch), rows,
                 _ptr(self.grads, bn.gamma_off), _ptr(self.grads, bn.beta_off), _ptr(bn.coef), _ptr(dz), a.Cs),
                "bnb_apply:" + bn.name)

    def _bnb_one_ok(self, a: Act) -> bool:
        return self.bnb_one and a.bn is not None and bool(self.lib.dip_bn_bwd_one_ok(a.H * a.W, a.C))

    def _emit_up_bwd(self, deep: Act, dcat, Cs_cat, choff, H, W, mode, ops, geom=None):
        bn = deep.bn
        nblk = self.lib.dip_bn_bwd_nblk(deep.H, deep.W, deep.C)
        fin_ok = self.ticket_fin and bool(self.lib.dip_fin_rows_ok(nblk, deep.C))
        ticket = self._ticket() if fin_ok else None
        if self._sizing:
            self.bwdp_need = max(self.bwdp_need, nblk * 2 * deep.Cs)
            return None
        dz = self._new(deep.H * deep.W * deep.Cs)
        lib = self.lib
        m = N.UP_BILINEAR if mode == "bilinear" else N.UP_NEAREST
        if self._bnb_one_ok(deep):
            # low resolution: adjoint of the up-sampling + the three BatchNorm-backward phases of the deeper branch in ONE launch
            gm = geom if geom is not None else dict(Hd=(H + 1) // 2, Wd=(W + 1) // 2, od_y=0, od_x=0)
            ops.append((lib.dip_upsample_bwd_one,
                        (_ptr(dcat), Cs_cat, choff, H, W, gm["Hd"], gm["Wd"], gm["od_y"], gm["od_x"], m, _ptr(deep.buf),
                         deep.Cs, deep.C, _ptr(bn.state), bn.Cs, float(deep.slope), _ptr(dz), deep.Cs,
                         _ptr(self.grads, bn.gamma_off), _ptr(self.grads, bn.beta_off), _ptr(bn.coef)), "upb_one:" + bn.name))
            return dz
        if fin_ok:
            # adjoint of the up-sampling + phases 1 and 2 of the deeper branch's BatchNorm backward in one launch
            fin = self._bnb_fin(bn, deep.H * deep.W, ticket)
            self.keep.append(fin)
            gm = geom if geom is not None else dict(Hd=(H + 1) // 2, Wd=(W + 1) // 2, od_y=0, od_x=0)
            ops.append((lib.dip_upsample_bwd_stats_crop_fin,
                        (_ptr(dcat), Cs_cat, choff, H, W, gm["Hd"], gm["Wd"], gm["od_y"], gm["od_x"], m,
                         _ptr(deep.buf), deep.Cs, deep.C, _ptr(bn.state), bn.Cs, float(deep.slope), _ptr(dz), deep.Cs,
                         _ptr(self.bwd_scratch), nblk, C.byref(fin)), "upb_stats:" + bn.name))
            ops.append((lib.dip_bn_bwd_apply, (_ptr(dz), deep.Cs, _ptr(deep.buf), deep.Cs, deep.H * deep.W, deep.C,
                                               _ptr(bn.state), bn.Cs, _ptr(bn.coef)), "bnb_apply:" + bn.name))
            return dz
        if geom is None:
            ops.append((lib.dip_upsample_bwd_stats, (_ptr(dcat), Cs_cat, choff, H, W, m, _ptr(deep.buf), deep.Cs, deep.C,
                                                     _ptr(bn.state), bn.Cs, float(deep.slope), _ptr(dz), deep.Cs,
                                                     _ptr(self.bwd_scratch), nblk), "upb_stats:" + bn.name))
        else:
            ops.append((lib.dip_upsample_bwd_stats_crop,
                        (_ptr(dcat), Cs_cat, choff, H, W, geom["Hd"], geom["Wd"], geom["od_y"], geom["od_x"], m,
                         _ptr(deep.buf), deep.Cs, deep.C, _ptr(bn.state), bn.Cs, float(deep.slope), _ptr(dz), deep.Cs,
                         _ptr(self.bwd_scratch), nblk), "upb_stats:" + bn.name))
        if self._fin_rows_ok(nblk, deep.C):
            ops.append((lib.dip_bn_bwd_apply_fin, (_ptr(dz), deep.Cs, _ptr(deep.buf), deep.Cs, deep.H * deep.W, deep.C,
                                                   _ptr(bn.state), bn.Cs, _ptr(self.bwd_scratch), nblk,
                                                   _ptr(self.grads, bn.gamma_off), _ptr(self.grads, bn.beta_off),
                                                   _ptr(bn.coef)), "bnb_apply:" + bn.name))
            return dz
        ops.append((lib.dip_bn_bwd_finalize, (_ptr(self.bwd_scratch), nblk, bn.Cs, bn.C, deep.H * deep.W,
                                              _ptr(self.grads, bn.gamma_off), _ptr(self.grads, bn.beta_off),
                                              _ptr(bn.coef)), "bnb_fin:" + bn.name))
        ops.append((lib.dip_bn_bwd_apply, (_ptr(dz), deep.Cs, _ptr(deep.buf), deep.Cs, deep.H * deep.W, deep.C,
                                           _ptr(bn.state), bn.Cs, _ptr(bn.coef)), "bnb_apply:" + bn.name))
        return dz

    def _bwd_scale_ops(self, i, dy_last):
        """Backward of scale i given dy wrt the raw output of its last conv.  For i > 0 returns ops
        that end with the gradient source of the scale's input stored in self.sc[i].gin; for
        i == 0 the input-gradient ops go to self.bwd_input_ops (only run when net_input is optimised)."""
        s = self.sc[i]
        st = s.st
        ops = []
        if i == self.defer_scale:
            self._entered_defer_scale = True
            ops += self._flush_deferred_wgrads()
        H, W, xin = st["H"], st["W"], st["xin"]
        Ho, Wo, geom = st["Ho"], st["Wo"], st["geom"]
        if s.up1 is not None:
            self._emit_wgrad(s.up1, st["u"], dy_last, ops, scale=i)
            g = self._emit_dgrad(s.up1, st["u"], dy_last, ops, fuse_bn=True)
            dy_u = self._emit_bn_act_bwd(st["u"], g, ops)
        else:
            dy_u = dy_last
        cat = st["cat_act"]
        self._emit_wgrad(s.up, cat, dy_u, ops, scale=i)
        g = self._emit_dgrad(s.up, cat, dy_u, ops, fuse_bn=True)
        dcat = self._emit_bn_act_bwd(cat, g, ops)            # grad wrt the concat tensor [H,W,Cs_cat]
        dy_s = None
        if s.ns:
            win = (geom["os_y"], geom["os_x"], Ho, Wo) if (geom is not None and (Ho, Wo) != (H, W)) else None
            dy_s = self._emit_bn_act_bwd(st["s_act"], (dcat, 0), ops, choff=0, Cg=cat.Cs, side=True, window=win)
            self._emit_wgrad(s.skip_conv, xin, dy_s, ops, scale=i)
        deep = st["deep"]
        dy_deep = self._emit_up_bwd(deep, dcat, cat.Cs, s.ns, Ho, Wo, s.upsample_mode, ops, geom)
        if self._entered_defer_scale:
            ops += self._flush_deferred_wgrads()          # this scale's decoder weight gradients, one fork
        if i < self.nscales - 1:
            ops += self._bwd_scale_ops(i + 1, dy_deep)
            gin = self.sc[i + 1].gin                         # gradient source wrt act d2
            dy_d2 = self._emit_bn_act_bwd(st["d2"], gin, ops)
        else:
            dy_d2 = dy_deep
        # the LAST batch of weight gradients (scale 0's encoder convs) is not held back: flushed at the end of the scale they
        # would fork off the main stream behind its last kernel and run as a pure tail (r05 timeline: main ends at +5870 us,
        # wgrad s0.down_b + s0.down_a + reductions until +6134); issued where their dy appears, the first runs next to the
        # scale's data gradient and only the second remains behind the main chain (DIP_TAIL_INLINE=0: the round-5 order)
        inline_tail = i == 0 and self.tail_inline
        self._emit_wgrad(s.down_b, st["d1"], dy_d2, ops, scale=None if inline_tail else i)
        g = self._emit_dgrad(s.down_b, st["d1"], dy_d2, ops, fuse_bn=True)
        dy_d1 = self._emit_bn_act_bwd(st["d1"], g, ops)
        if s.pool == 'lanczos':             # backward of the Downsampler's dense conv: its weight gradient, then the data
            # gradient on the replication-padded domain folded into dy of the full-resolution conv's (raw) output
            Cs1 = round_up(s.down_a.Cout, 4)
            raw = st["d1_raw"]
            self._emit_wgrad(s.down_ds, raw, dy_d1, ops, scale=i)
            g2 = self._emit_dgrad(s.down_ds, raw, dy_d1, ops)
            dy_full = self._buf(H * W * Cs1)
            if not self._sizing:
                src = self._gradsrc(g2, Cs1)
                ops.append((self.lib.dip_fold_to_nhwc, (C.byref(src), H, W, s.down_a.Cout, _ptr(dy_full), Cs1),
                            "fold:" + s.down_ds.name))
            dy_d1 = dy_full
        if s.pool in ('avg', 'max'):        # adjoint of the pooling: dy of the full-resolution conv output
            Cs1 = round_up(s.down_a.Cout, 4)
            dy_full = self._buf(H * W * Cs1)
            if not self._sizing:
                if s.pool == 'avg':
                    ops.append((self.lib.dip_avgpool2_bwd, (_ptr(dy_d1), H, W, Cs1, s.down_a.Cout, _ptr(dy_full), Cs1),
                                "poolb:" + s.down_a_bn.name))
                else:                       # MaxPool2d(2, 2): the arg-max is recomputed from the conv output
                    ops.append((self.lib.dip_maxpool2_bwd, (_ptr(dy_d1), _ptr(st["d1_full"]), H, W, Cs1, Cs1,
                                                            s.down_a.Cout, _ptr(dy_full), Cs1),
                                "poolb:" + s.down_a_bn.name))
            dy_d1 = dy_full
        self._emit_wgrad(s.down_a, xin, dy_d1, ops, scale=None if inline_tail else i)
        tgt = ops if i > 0 else self.bwd_input_ops
        sk = s.skip_conv if s.ns else None
        sk_pad = sk.P if (sk is not None and sk.P > 0 and sk.pad_mode in (N.PAD_REFLECT, N.PAD_REPLICATE)) else 0
        g = self._emit_dgrad(s.down_a, xin, dy_d1, tgt, need_pad=sk_pad)
        if s.ns:
            g = self._emit_dgrad(s.skip_conv, xin, dy_s, tgt, accumulate_into=g)
        s.gin = g
        if self._entered_defer_scale:
            ops += self._flush_deferred_wgrads()          # ... and its encoder ones
        s.dbg = {"dy_last": dy_last, "dy_u": dy_u, "dcat": dcat, "dy_s": dy_s, "dy_deep": dy_deep, "dy_d2": dy_d2,
                 "dy_d1": dy_d1}      # gradient buffers by role (tools/debug_grads.py)
        return ops

    # ------------------------------------------------------------------ run
    def _schedule(self, ops, cls_fn=None, join_before_fn=None, deps=None):
        """The static schedule of a launch list on the main HIP stream + up to three auxiliary streams:
        [("launch", k, c) | ("record", tag, c) | ("wait", c, tag)], k = index into ops, c = stream class (cls_fn(name) ->
        0 main, 1 side, 2 bulk, 3 second bulk; None: everything on the main stream), tag = event name.
        Backward: the weight-gradient kernels (+ their slab reductions) of a layer depend only on that
        layer's dy and the stored activations, and nothing but the optimiser step waits for them: they form the
        BULK stream.  Co-running a big weight gradient with the big data gradient of the same layer buys nothing
        (both are MFMA-bound: 1430 us together = 700 + 780 alone, profiles/r03_timeline_*), whereas the main
        chain's walk through the low-resolution scales leaves the chip ~85 % idle for 1.6 ms (latency-bound
        launches) -- so the launch list holds the big weight gradients back (_flush_deferred_wgrads) and the bulk
        stream runs them underneath that walk (+1.8 %; restricting them to part of the chip so that the walk keeps
        moving -- smaller grids, a CU-masked stream -- was measured and does not pay).  SIDE stream: small work the main chain waits for again -- the BatchNorm backward of the 4-channel
        skip branch, the thin columns of the 132-column data gradients; forward: the 1x1 skip-branch conv (+ its
        BatchNorm finalisation) of a scale next to the encoder convs (scratch of its own).
        Fork: before an auxiliary op, its stream waits for an event recorded on the main stream if the main stream
        has advanced since that stream's last fork; join: main waits for the side stream in front of every op
        `join_before_fn` selects, and for all of them at the end.  `deps` = {consumer op name: [producer op names]}: the
        consumer's stream waits for an event recorded right after the producer (finer than a join)."""
        sched = []
        if cls_fn is None:
            return [("launch", k, 0) for k in range(len(ops))]
        deps = deps or {}
        producers = {p for ps in deps.values() for p in ps}
        main_seq = 0                          # main-stream ops issued so far
        forked = [0, -1, -1, -1]              # main_seq at the last fork of each auxiliary stream
        pending = [False, False, False, False]    # auxiliary work the main stream has not joined yet
        for k, (fn, args, name) in enumerate(ops):
            c = cls_fn(name)
            if c and forked[c] != main_seq:
                sched += [("record", ("fork", k), 0), ("wait", c, ("fork", k))]
                forked[c] = main_seq
            if c == 0 and pending[1] and join_before_fn(name):
                sched += [("record", ("join", k), 1), ("wait", 0, ("join", k))]
                pending[1] = False
            for prod in deps.get(name, ()):
                sched.append(("wait", c, ("dep", prod)))
            sched.append(("launch", k, c))
            if name in producers:
                sched.append(("record", ("dep", name), c))
            if c:
                pending[c] = True
            else:
                main_seq += 1
        for c in (1, 2, 3):
            if pending[c]:
                sched += [("record", ("join", -c), c), ("wait", 0, ("join", -c))]
        return sched

    def _aux_streams(self):
        """Auxiliary streams (side, bulk, second bulk) + the Python path's events of the current mode: captured and eager
        runs own separate sets."""
        slot = "cap" if torch.cuda.is_current_stream_capturing() else "eager"
        st_ = self._aux.get((slot, self.device))
        if st_ is None:
            # (HIP stream priorities for the auxiliary streams were measured in round 4: no effect; default priority)
            st_ = self._aux[(slot, self.device)] = ([torch.cuda.Stream(self.device) for _ in range(3)], {})
        return slot, st_[0], st_[1]

    def _issue(self, ops, main, key, cls_fn=None, join_before_fn=None, deps=None):
        """Issues a launch list: compiled once per (key, length) into a command list and run by ONE dip_list_run call, or
        (DIP_NO_CLIST=1) walked in Python with torch events -- the same schedule either way."""
        multi = cls_fn is not None
        slot, aux, events = self._aux_streams() if multi else ("one", [], None)
        ko = False
        if self._knockout is not None:          # DIP_KNOCKOUT_AFTER=n: the first n issues of a list run whole (buffers hold real values)
            n = self._ko_seen[key] = self._ko_seen.get(key, 0) + 1
            ko = n > self._ko_after
        ck = (key, len(ops), multi, ko)
        ent = self._clists.get(ck)
        if ent is None:
            sched = self._schedule(ops, cls_fn, join_before_fn, deps)
            if ko:                              # timing experiment: these launches are left out (the results are WRONG)
                sched = [c for c in sched if not (c[0] == "launch" and self._knockout.search(ops[c[1]][2]))]
            cl = None
            if self.use_clist:
                evidx = {}
                cmds = []
                for c in sched:
                    if c[0] == "launch":
                        fn, args, name = ops[c[1]]
                        cmds.append(("launch", fn, args, c[2], name))
                    elif c[0] == "record":
                        cmds.append(("record", evidx.setdefault(c[1], len(evidx)), c[2]))
                    else:
                        cmds.append(("wait", c[1], evidx.setdefault(c[2], len(evidx))))
                cl = N.CmdList(cmds)
            ent = self._clists[ck] = (sched, cl)
        sched, cl = ent
        if cl is not None:
            cl.run([main.cuda_stream] + [s_.cuda_stream for s_ in aux], slot)
            return
        streams = [main] + aux
        ptrs = [s_.cuda_stream for s_ in streams]
        check = N.check
        for c in sched:
            if c[0] == "launch":
                fn, args, name = ops[c[1]]
                rc = fn(*args, ptrs[c[2]])
                if rc:
                    check(rc, name)
            elif c[0] == "record":
                ev = events.get((key, c[1]))
                if ev is None:
                    ev = events[(key, c[1])] = torch.cuda.Event()
                ev.record(streams[c[2]])
            else:
                streams[c[1]].wait_event(events[(key, c[2])])

    def _run(self, ops, main, key="one"):
        self._issue(ops, main, key)

    def _run_two_streams(self, ops, main, cls_fn, join_before_fn, key, deps=None):
        self._issue(ops, main, key, cls_fn, join_before_fn, deps)

    # stream class of a backward op: 2 = bulk (weight gradients), 1 = side, 0 = main
    _BWD_SIDE = staticmethod(lambda n: 2 if n.startswith(("wgrad:", "wgred:")) else
                             (1 if (n.startswith("dgthin:") or n.endswith(".skip_bn")) else 0))

    def _backward_deps(self, ops):
        """{consumer op: [producer ops]} of the backward list whose two ends run on different streams.
        * dgrad+ of a skip conv (main stream) and its weight gradient (bulk stream) consume dy of the skip BatchNorm
          backward (side stream);
        * the thin columns of a 132-column data gradient ("dgthin:X", side stream) are written into the same
          gradient buffer as the 128 columns of "dgrad:X" (main): whatever main-stream op follows "dgrad:X" reads
          (or accumulates into) that buffer and must wait for them -- derived from the op list, for ANY conv X
          (decoder convs in the notebooks' nets; a 129..132-channel down_a / down_b is legal too)."""
        names = [name for _, _, name in ops]
        present = set(names)
        deps = {}
        for i, sc in enumerate(self.sc):
            c, p = f"dgrad+:s{i}.skip_conv", f"bnb_apply:s{i}.skip_bn"
            if p not in present:
                p = f"bnb_one:s{i}.skip_bn"            # (low resolution: the one-launch form)
            if sc.ns and c in present and p in present:        # (a wait on a never-recorded event is illegal under capture)
                deps.setdefault(c, []).append(p)
            # ... and so does the skip conv's weight gradient (bulk stream)
            c = f"wgrad:s{i}.skip_conv"
            if sc.ns and c in present and p in present:
                deps.setdefault(c, []).append(p)
        for k, name in enumerate(names):
            if not name.startswith("dgthin:"):
                continue
            main_part = "dgrad:" + name[len("dgthin:"):]
            j = names.index(main_part, k)
            consumer = next((n for n in names[j + 1:] if not self._BWD_SIDE(n)), None)
            if consumer is not None:
                deps.setdefault(consumer, []).append(name)
            # (no main-stream op behind it: the final join of _run_two_streams covers it)
        return deps

    def _run_backward_two_streams(self, ops, main):
        deps = self._bwd_deps if getattr(self, "_bwd_deps_for", None) is ops else None
        if deps is None:
            deps = self._bwd_deps = self._backward_deps(ops)
            self._bwd_deps_for = ops
        # the second bulk stream is an EAGER form: inside a hipGraph every extra fork / join costs what ROCm 7.2's graph
        # executor loses on them (tools/ubench/graph_fork_join.hip) -- snail as ONE graph 562 -> 638 it/s with the weight
        # gradients back on one bulk stream (profiles/r06_ab_bulk2_graph.txt); the scratch sets stay as planned
        capturing = torch.cuda.is_current_stream_capturing()
        bulk2 = self._bulk2 if not capturing else ()
        cls = self._BWD_SIDE if not bulk2 else (lambda n: 3 if n in bulk2 else self._BWD_SIDE(n))
        self._run_two_streams(ops, main, cls, lambda n: False, "bwd_cap" if capturing else "bwd", deps)

    def _run_forward_two_streams(self, ops, main):
        side = self._fwd_side
        self._run_two_streams(ops, main, lambda n: 1 if n in side else 0, lambda n: n.startswith("upcat:"), "fwd")

    def _launch_forward(self, x_ptr, main, with_out_conv=True):
        """The static forward launch list on stream `main` (+ the side stream): weight repack, NCHW -> NHWC of the input
        at x_ptr, every layer up to (with_out_conv: and including) the output conv.  Pointers only -- also what
        dip_group.GroupedFits issues once for B instances."""
        lib, stream = self.lib, main.cuda_stream
        N.check(lib.dip_pack_weights(_ptr(self.params), _ptr(self.packed), self.pack_recs.data_ptr(),
                                     len(self.convs), self.pack_max, stream), "pack_weights")
        if self.bf3:
            N.check(lib.dip_pack_weights_bf3(_ptr(self.params), self.packed3.data_ptr(), self.pack_recs3.data_ptr(),
                                             len(self.convs), self.pack_max3, stream), "pack_weights_bf3")
        N.check(lib.dip_nchw_to_nhwc(x_ptr, _ptr(self.x_nhwc), self.Cimg, self.H * self.W, round_up(self.Cimg, 4), stream),
                "nchw_to_nhwc")
        ops = self.fwd_ops if with_out_conv else self.fwd_ops[:-1]      # the last op is the output conv
        if self.two_streams:
            self._run_forward_two_streams(ops, main)
        else:
            self._run(ops, main, "fwd1")

    def _launch_backward(self, main):
        """The static backward launch list (dy of the output conv already in self.dy_out)."""
        if self.two_streams:
            self._run_backward_two_streams(self.bwd_ops, main)
        else:
            self._run(self.bwd_ops, main, "bwd1")

    def forward(self, x: torch.Tensor, head=None):
        """Runs the forward launch list.  head = None: returns the network output [1,C,H,W].
        head = a utils.loss_head.MSEHead: the output conv + sigmoid + (mask) + MSE run as ONE launch
        (dip_loss_head_fwd) and (loss, out) is returned."""
        if x.dim() != 4 or x.shape[0] != 1:
            raise NotImplementedError("dip-amd: input must be [1,C,H,W] (train-mode BatchNorm couples a batch; "
                                      "independent images run as independent nets)")
        if not self.net.training:
            raise NotImplementedError("dip-amd: eval-mode BatchNorm is not implemented (the reference never "
                                      "calls .eval() on the skip path)")
        dev = x.device
        if self.device != dev or not self._arena_ok():
            self._build_arenas(dev)
        _, Cimg, H, W = x.shape
        if self.shape_key != (H, W, Cimg):
            if Cimg != self.sc[0].down_a.Cin:
                raise RuntimeError(f"dip-amd: input has {Cimg} channels, net expects {self.sc[0].down_a.Cin}")
            self._build_plan(H, W, Cimg)
        lib = self.lib
        with torch.cuda.device(dev):          # raw HIP launches go to the CURRENT device's streams
            main = torch.cuda.current_stream(dev)
            stream = main.cuda_stream
            xs = x.detach()
            if xs.dtype != torch.float32:
                xs = xs.float()
            xs = xs.contiguous()
            self.fwd_id += 1
            self._launch_forward(xs.data_ptr(), main, head is None)
            Ho_, Wo_ = self.Hout, self.Wout
            out = torch.empty((1, self.n_out, Ho_, Wo_), dtype=torch.float32, device=dev)
            loss = None
            if head is None:
                N.check(lib.dip_head_fwd(_ptr(self.y_out), out.data_ptr(), self.n_out, Ho_ * Wo_, round_up(self.n_out, 4),
                                         1 if self.need_sigmoid else 0, stream), "head_fwd")
            else:
                loss = torch.empty((), dtype=torch.float32, device=dev)
                self._head_desc = head._descriptor(self, out, loss)
                N.check(lib.dip_loss_head_fwd(C.byref(self._head_desc), stream), "loss_head_fwd")
            if len(self.bns):
                self.nbt.add_(1)
        # (aliases without autograd history: holding the Function's own output tensors would keep the
        # autograd graph -- and its AccumulateGrad nodes with their recorded stream -- alive until the
        # next forward, which breaks hipGraph capture on another stream)
        self.last_out = out.detach()
        self.last_head = head
        return out if head is None else (loss, out)

    def _detach_stale_grads(self):
        """.grad tensors that still alias the gradient arena (no zero_grad() since the previous
        backward, or zero_grad(set_to_none=False)) would be overwritten by this backward before
        autograd accumulates into them: move them to a second arena first, so that the usual
        `p.grad += new` semantics of a second backward() hold."""
        base = self.grads.data_ptr()
        stale = [k for k, (p, o) in enumerate(zip(self.param_list, self.slots))
                 if p.grad is not None and p.grad.data_ptr() == base + 4 * o]
        if not stale:
            return
        if getattr(self, "grads_alt", None) is None or self.grads_alt.numel() != self.grads.numel() \
                or self.grads_alt.device != self.grads.device:
            self.grads_alt = torch.empty_like(self.grads)
        self.grads_alt.copy_(self.grads)
        for k in stale:
            p, o = self.param_list[k], self.slots[k]
            p.grad = self.grads_alt[o:o + p.numel()].view(p.shape)

    def backward(self, gout, fwd_id: int, need_input_grad: bool, gloss=None):
        if fwd_id != self.fwd_id:
            raise RuntimeError("dip-amd: backward() of a stale forward: the engine keeps the activations of the "
                               "most recent net(x) only (one forward, one backward per closure call)")
        dev = self.device
        lib = self.lib
        H, W = self.H, self.W
        grads = self.grads
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream(dev)
            stream = main.cuda_stream
            self._detach_stale_grads()
            if self.last_head is None:
                g = gout.detach()
                if g.dtype != torch.float32:
                    g = g.float()
                g = g.contiguous()
                N.check(lib.dip_head_bwd(g.data_ptr(), self.last_out.data_ptr(), _ptr(self.dy_out), self.n_out,
                                         self.Hout * self.Wout,
                                         round_up(self.n_out, 4), 1 if self.need_sigmoid else 0, stream), "head_bwd")
            else:
                gl = gloss.detach().reshape(1)
                if gl.dtype != torch.float32:
                    gl = gl.float()
                self._gloss_keep = gl
                N.check(lib.dip_loss_head_bwd(C.byref(self._head_desc), gl.data_ptr(), _ptr(self.dy_out),
                                              round_up(self.n_out, 4), stream), "loss_head_bwd")
            self._launch_backward(main)
            gx = None
            if need_input_grad:
                self._run(self.bwd_input_ops, main, "bwdin")
                gbuf, pad = self.sc[0].gin
                src = N.DipGradSrc(_ptr(gbuf), pad, 1 if pad > 0 else 0, round_up(self.Cimg, 4), 0)
                gx = torch.empty((1, self.Cimg, H, W), dtype=torch.float32, device=dev)
                N.check(lib.dip_fold_to_nchw(C.byref(src), H, W, self.Cimg, gx.data_ptr(), stream), "fold_to_nchw")
        # NOTE: the returned gradients are VIEWS of the gradient arena; the next backward() overwrites
        # them in place (keep a .clone() if a gradient has to survive the next closure evaluation)
        views = [grads[o:o + p.numel()].view(p.shape) for p, o in zip(self.param_list, self.slots)]
        return gx, views


class _SkipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine: SkipEngine, x, *params):
        out = engine.forward(x)
        ctx.engine = engine
        ctx.fwd_id = engine.fwd_id
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        need_x = ctx.needs_input_grad[1]
        gx, views = ctx.engine.backward(gout, ctx.fwd_id, need_x)
        return (None, gx, *views)


class _SkipLossFn(torch.autograd.Function):
    """net + fused loss head (utils/loss_head.MSEHead): returns (loss, out); `out` is not differentiable."""

    @staticmethod
    def forward(ctx, engine: SkipEngine, head, x, *params):
        loss, out = engine.forward(x, head)
        ctx.engine = engine
        ctx.fwd_id = engine.fwd_id
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gloss, gout_unused):
        need_x = ctx.needs_input_grad[2]
        gx, views = ctx.engine.backward(None, ctx.fwd_id, need_x, gloss=gloss)
        return (None, None, gx, *views)


def _check_input(engine: SkipEngine, x: torch.Tensor):
    if not x.is_cuda:
        raise RuntimeError("dip-amd: the skip-net runs on an MI355X only (input tensor is on the CPU); there is "
                           "no CPU fallback in this backend")
    # make sure the parameter arena exists before autograd records the parameter tensors
    if engine.device != x.device or not engine._arena_ok():
        engine._build_arenas(x.device)


def run_net(engine: SkipEngine, x: torch.Tensor) -> torch.Tensor:
    _check_input(engine, x)
    return _SkipFn.apply(engine, x, *engine.param_list)


def run_net_loss(engine: SkipEngine, head, x: torch.Tensor):
    _check_input(engine, x)
    return _SkipLossFn.apply(engine, head, x, *engine.param_list)
