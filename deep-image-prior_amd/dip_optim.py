"""Fused Adam over flat arenas (libdip_hip.so: dip_adam_step).

Replaces `torch.optim.Adam(parameters, lr=LR)` of the reference's optimize()
(utils/common_utils.py:225) with torch 2.x Adam semantics (betas (0.9, 0.999), eps 1e-8, no
weight decay; bias corrections computed in double on the host).  Parameters that are views of
one contiguous fp32 CUDA arena (a SkipNet's parameters) are stepped by ONE launch; any other
CUDA fp32 tensor (net_input for opt_over='net,input', Downsampler weights) gets its own launch
of the same kernel.

The step count lives in device memory (DipIterState, advanced by dip_adam_tick), so `step()` is a
static launch sequence: `GraphedIteration` captures {zero_grad(); closure(); step()} into one
hipGraph and replays it.  Several independent fits: `GraphedIteration.group(dip_group.GroupedFits(...))`
captures ONE launch list that serves all of them (every kernel launch covers all instances), and
`GraphedIteration.group([(optimizer, closure), ...])` -- arbitrary closures -- one graph per fit on its own stream.
"""
from __future__ import annotations

import os

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
        self._iter_state = {}                  # device -> DipIterState bytes (uint8[16])

    # torch.optim API subset used by optimize() and the notebooks
    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:                  # torch.optim.Optimizer.zero_grad: views cannot be detached in place
                    if p.grad.grad_fn is not None:
                        p.grad = p.grad.detach()
                    else:
                        p.grad.requires_grad_(False)
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
        active = []
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
            active.append((g, flat_ptr))
        ticked = set()
        for g, flat_ptr in active:
            dev = g.params[0].device
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                st = self._iter_state.get(dev)
                if st is None:
                    st = self._iter_state[dev] = torch.zeros(16, dtype=torch.uint8, device=dev)
                    st.view(torch.int64)[0] = self.step_count - 1
                if dev not in ticked:          # t <- t + 1, step_size and sqrt(bc2) in double, on the device
                    N.check(lib.dip_adam_tick(st.data_ptr(), float(self.lr), float(self.betas[0]),
                                              float(self.betas[1]), stream), "adam_tick")
                    ticked.add(dev)
                N.check(lib.dip_adam_step_dev(g.base, flat_ptr, g.m.data_ptr(), g.v.data_ptr(), g.numel,
                                              float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              st.data_ptr(), stream), "adam_step_dev")
        return loss

    def device_step_count(self):
        """Adam's step count as the device holds it (after graph replays the host count is stale)."""
        counts = [int(st.view(torch.int64)[0].item()) for st in self._iter_state.values()]
        return max(counts) if counts else self.step_count


_LIVE_GRAPHS = []


class GraphedIteration:
    """One optimisation iteration -- optimizer.zero_grad(); closure(); optimizer.step(), i.e. the body
    of the reference's optimize() loop (utils/common_utils.py:226-230) -- captured ONCE into a
    hipGraph and replayed: ~300 kernel launches, the event fork/joins of the two-stream schedule and
    the ATen ops of the closure become a single graph launch per iteration.

    The closure must be replay-safe: no host synchronisation (.item(), .cpu(), print of a tensor),
    every tensor it keeps across iterations updated IN PLACE, reg-noise from utils.reg_noise.RegNoise
    (or torch's graph-safe device generator).  `lr` is frozen at capture time.

        it = GraphedIteration(optimizer, closure)     # 3 eager warm-up iterations, then the capture
        it.run(num_iter - 3)
    """

    def __init__(self, optimizer, closure, warmup=3, device=None):
        self.fits = [(optimizer, closure)]
        self._capture(warmup, device)

    @classmethod
    def group(cls, fits, warmup=3, device=None, single_graph=False):
        """Grouped multi-instance execution: `fits` = a dip_group.GroupedFits (B fits of one architecture with the
        fused closure: ONE launch list, one hipGraph -- see that module), or [(optimizer, closure), ...] of INDEPENDENT nets
        with arbitrary closures (own weights, own BatchNorm statistics, own Adam state).  In the second form every fit is captured into its own
        hipGraph on its own HIP stream and one run() step replays all of them, so the kernels of
        different instances overlap on the chip -- what fills an MI355X when one image (e.g. the
        384x256 snail net: 25 us of math per iteration) cannot.
        single_graph=True captures all fits as concurrent branches of ONE graph instead (one graph
        launch per iteration; cross-stream capture of this size is fragile in the HIP runtime, so it is
        opt-in)."""
        import dip_group
        if isinstance(fits, dip_group.GroupedFits):
            # ONE launch list for all the fits (every kernel launch serves all instances, csrc/dip_group.h), captured as
            # ONE hipGraph: the form for many small fits -- 1/B of the launches of the per-fit graphs below
            return fits.capture(warmup)
        self = cls.__new__(cls)
        self.fits = list(fits)
        if single_graph or len(self.fits) == 1:
            self._capture(warmup, device)
            return self
        if device is None:
            device = self.fits[0][0].params[0].device
        self.device = device
        self.iterations = 0
        self.graph = None
        self.members = []
        with torch.cuda.device(device):
            for opt, clo in self.fits:
                m = cls.__new__(cls)
                m.fits = [(opt, clo)]
                m._capture(warmup, device)
                self.members.append(m)
        self.iterations = self.members[0].iterations
        return self

    def _one(self, optimizer, closure):
        optimizer.zero_grad()
        closure()
        optimizer.step()

    def _capture(self, warmup, device):
        if device is None:
            device = self.fits[0][0].params[0].device
        self.device = device
        self.iterations = 0
        with torch.cuda.device(device):
            cur = torch.cuda.current_stream(device)
            # The eager warm-up runs on the SAME streams the capture uses: autograd caches the stream of
            # every AccumulateGrad node, and a node created on another stream would pull a dependency on a
            # non-capturing stream into the capture.
            self.capture_stream = torch.cuda.Stream(device)
            self.branch_streams = [torch.cuda.Stream(device) for _ in self.fits[1:]]
            streams = [self.capture_stream] + self.branch_streams
            for s in streams:
                s.wait_stream(cur)
            for _ in range(max(int(warmup), 1)):
                for (opt, clo), s in zip(self.fits, streams):
                    with torch.cuda.stream(s):
                        self._one(opt, clo)
            for s in streams:
                cur.wait_stream(s)
            torch.cuda.synchronize(device)
            self.iterations += max(int(warmup), 1)
            self.graph = torch.cuda.CUDAGraph()
            # ROCm 7.2: per-launch event timing AFTER a captured graph of this iteration had been destroyed aborted
            # 10-25 % of bench.py's small-config runs with glibc heap-corruption errors (bench.py now takes those
            # timings before any graph exists: 0 / 30).  DIP_KEEP_GRAPHS=1 keeps every captured graph alive until the
            # interpreter exits instead -- not the default, because eager iterations run ~11 % slower while a
            # graph of the same net is alive (128 -> 113 it/s at 512x512).
            if os.environ.get("DIP_KEEP_GRAPHS", "0") == "1":
                _LIVE_GRAPHS.append((self.graph, self.capture_stream, self.branch_streams))
            with torch.cuda.graph(self.graph, stream=self.capture_stream):
                main = torch.cuda.current_stream(device)
                start = torch.cuda.Event()
                start.record(main)
                for (opt, clo), s in zip(self.fits[1:], self.branch_streams):
                    s.wait_event(start)                       # fork at the START: the branches run concurrently
                    with torch.cuda.stream(s):
                        self._one(opt, clo)
                self._one(*self.fits[0])
                for s in self.branch_streams:
                    main.wait_stream(s)                       # join

    def run(self, n=1):
        if self.graph is not None:
            for _ in range(int(n)):
                self.graph.replay()
        else:                                   # one graph per instance, each on its own stream
            cur = torch.cuda.current_stream(self.device)
            for m in self.members:
                m.capture_stream.wait_stream(cur)
            for _ in range(int(n)):
                for m in self.members:
                    with torch.cuda.stream(m.capture_stream):
                        m.graph.replay()
            for m in self.members:
                cur.wait_stream(m.capture_stream)
        self.iterations += int(n)


class ArenaLBFGS:
    """torch.optim.LBFGS (no line search: `line_search_fn=None`, the reference's setting,
    utils/common_utils.py:218) restated on FLAT vectors: when all parameters are views of one
    contiguous arena (a SkipNet's parameters; the gaps between tensors hold zeros in both the
    parameter and the gradient arena) the parameter vector and the gradient are the arenas themselves
    -- no per-tensor gather/scatter of 112 tensors per evaluation -- and the two-loop recursion runs
    on 2.2 M-element device vectors.  Other parameter lists are gathered/scattered like torch does.
    Same update rule, same stopping tests (with tolerance -1 the quirk `gtd > -tolerance_change`
    stops the run when the directional derivative exceeds 1), same history handling."""

    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9,
                 history_size=100, _allow_cpu=False):
        self.params = list(params)
        self.lr, self.max_iter = lr, max_iter
        self.max_eval = max_eval if max_eval is not None else max_iter * 5 // 4
        self.tolerance_grad, self.tolerance_change, self.history_size = tolerance_grad, tolerance_change, history_size
        self.state = {"func_evals": 0, "n_iter": 0}
        for p in self.params:          # (_allow_cpu: the algorithm-vs-torch.optim.LBFGS unit test only)
            if (not p.is_cuda and not _allow_cpu) or p.dtype != torch.float32:
                raise RuntimeError("dip-amd ArenaLBFGS: parameters must be fp32 CUDA tensors (no CPU optimiser path)")
        self._flat = None
        self._sig = None

    def _bind(self):
        """(Re)derives the flat view: a SkipNet's parameters become views of one arena at its first
        forward (and again after .type()/.to()), so this is checked at every step()."""
        sig = tuple(p.data_ptr() for p in self.params)
        if sig == self._sig:
            return
        self._sig = sig
        self._flat = None
        if len(_split_contiguous(self.params)) == 1:
            p0, pl = self.params[0], self.params[-1]
            n = (pl.data_ptr() - p0.data_ptr()) // 4 + pl.numel()
            st = p0.untyped_storage()
            off = (p0.data_ptr() - st.data_ptr()) // 4
            self._flat = torch.empty(0, dtype=torch.float32, device=p0.device).set_(st, off, (n,), (1,))
            self._offsets = [(p.data_ptr() - p0.data_ptr()) // 4 for p in self.params]

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None if set_to_none else (p.grad.zero_() if p.grad is not None else None)

    def _gather_flat_grad(self):
        g0 = self.params[0].grad
        if self._flat is not None and g0 is not None:
            base = g0.data_ptr() - 4 * self._offsets[0]
            if all(p.grad is not None and p.grad.is_contiguous() and p.grad.data_ptr() == base + 4 * o
                   for p, o in zip(self.params, self._offsets)):
                st = g0.untyped_storage()
                off = (base - st.data_ptr()) // 4
                return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, off, (self._flat.numel(),),
                                                                                  (1,)).clone()
        views = [(p.grad.reshape(-1) if p.grad is not None else p.new_zeros(p.numel())) for p in self.params]
        if self._flat is not None:
            # same layout as the fast path (arena length, alignment gaps zero): history vectors of both paths mix
            g = torch.zeros(self._flat.numel(), dtype=torch.float32, device=self._flat.device)
            for v, o in zip(views, self._offsets):
                g[o:o + v.numel()].copy_(v)
            return g
        return torch.cat(views, 0)

    def _add_grad(self, t, d):
        if self._flat is not None and d.numel() == self._flat.numel():
            self._flat.add_(d, alpha=t)
            return
        off = 0
        for p in self.params:
            n = p.numel()
            p.data.add_(d[off:off + n].view_as(p), alpha=t)
            off += n

    @torch.no_grad()
    def step(self, closure):
        closure = torch.enable_grad()(closure)
        lr, max_iter, max_eval = self.lr, self.max_iter, self.max_eval
        tg, tc, hs = self.tolerance_grad, self.tolerance_change, self.history_size
        st = self.state
        orig_loss = closure()
        self._bind()
        loss = float(orig_loss)
        current_evals = 1
        st["func_evals"] += 1
        g = self._gather_flat_grad()
        if float(g.abs().max()) <= tg:
            return orig_loss
        d, t = st.get("d"), st.get("t")
        old_dirs, old_stps, ro = st.get("old_dirs"), st.get("old_stps"), st.get("ro")
        H_diag, prev_g, prev_loss = st.get("H_diag"), st.get("prev_flat_grad"), st.get("prev_loss")
        n_iter = 0
        while n_iter < max_iter:
            n_iter += 1
            st["n_iter"] += 1
            if st["n_iter"] == 1:
                d = g.neg()
                old_dirs, old_stps, ro = [], [], []
                H_diag = 1
            else:
                y = g.sub(prev_g)
                s = d.mul(t)
                ys = float(y.dot(s))
                if ys > 1e-10:
                    if len(old_dirs) == hs:
                        old_dirs.pop(0)
                        old_stps.pop(0)
                        ro.pop(0)
                    old_dirs.append(y)
                    old_stps.append(s)
                    ro.append(1.0 / ys)
                    H_diag = ys / float(y.dot(y))
                num_old = len(old_dirs)
                al = [None] * num_old
                q = g.neg()
                for i in range(num_old - 1, -1, -1):
                    al[i] = float(old_stps[i].dot(q)) * ro[i]
                    q.add_(old_dirs[i], alpha=-al[i])
                d = r = torch.mul(q, H_diag)
                for i in range(num_old):
                    be_i = float(old_dirs[i].dot(r)) * ro[i]
                    r.add_(old_stps[i], alpha=al[i] - be_i)
            prev_g = g.clone(memory_format=torch.contiguous_format)
            prev_loss = loss
            t = min(1.0, 1.0 / float(g.abs().sum())) * lr if st["n_iter"] == 1 else lr
            gtd = float(g.dot(d))
            if gtd > -tc:
                break
            ls_func_evals = 0
            self._add_grad(t, d)
            if n_iter != max_iter:
                loss = float(closure())
                g = self._gather_flat_grad()
                ls_func_evals = 1
            current_evals += ls_func_evals
            st["func_evals"] += ls_func_evals
            if n_iter == max_iter or current_evals >= max_eval:
                break
            if float(g.abs().max()) <= tg:
                break
            if float(d.mul(t).abs().max()) <= tc:
                break
            if abs(loss - prev_loss) < tc:
                break
        st.update(d=d, t=t, old_dirs=old_dirs, old_stps=old_stps, ro=ro, H_diag=H_diag, prev_flat_grad=prev_g,
                  prev_loss=prev_loss)
        return orig_loss
