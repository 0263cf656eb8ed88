"""Benchmark of the deep-image-prior hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config default|sr|kate|library|snail]
                  [--closure fused|notebook] [--no-graph] [--instances B [--group both|native|graphs]]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one optimisation iteration (reg-noise perturbation, skip-net forward, MSE, backward,
fused Adam) of ONE fit.  The default config is BASELINE.json's headline: the default net
  get_net(32,'skip','reflection',skip_n33d=128,skip_n33u=128,skip_n11=4,num_scales=5,'bilinear')
on a 512x512 denoising problem (SURVEY.md section 8d "M1").  The other configs are the reference's
remaining notebook set-ups at their own sizes (BASELINE.json configs[2..3]): `sr` (M2: the same net,
loss through the Lanczos x4 down-sampler), `kate` (inpainting net with 128 skip channels, 512x512,
masked MSE), `library` (depth 6, 5x5 filters, 448x704), `snail` (8..128-channel net, 256x384).

With N GPUs every rank optimises its own independent image (no collective on the data path:
train-mode BatchNorm forbids batching images and the path never shards one image -- "replicas
only") -> "scaling": "weak", value = N * K / max-over-ranks time.  `--gpus N` without a torchrun
environment starts the N worker processes itself; the ranks meet over gloo (start barrier and
max-over-ranks time only -- no RCCL anywhere).

Timed region: K iterations with the fused closure (utils.reg_noise.RegNoise +
utils.loss_head.MSEHead + in-place EMA), executed as eager launches on the engine's two HIP streams
or as K replays of the iteration captured into a hipGraph (dip_optim.GraphedIteration);
`--mode auto` (default) times K steps each way and reports the faster as `value`, the other under
"other_mode".  `--closure notebook --mode eager` times the notebook's own torch closure; the default
line reports that figure too (`eager_notebook`) from a short extra run.

`--instances B` (> 1): B independent fits per GPU, timed as ONE launch list that serves all of them
(dip_group.GroupedFits: every kernel launch covers the B instances) and as one hipGraph per
fit on its own stream; the faster is `value`, the others go to "other_mode".  The default line carries a
short run of that form as `grouped_batch_of_4` (4 fits of the headline configuration on the GPU).

The JSON line also carries
  roofline       : the dominant kernel (3x3 stride-1 implicit-GEMM conv, 128-wide N block: forward
                   and data-gradient launches) -- ALGORITHMIC FLOPs (2*Cout*Hout*Wout*
                   Cin*k*k of the layer, no padded ring, no dilation zeros) / the kernel's own
                   HIP-event time (the split-K finish kernel is timed separately), vs the 157.3
                   TFLOP/s fp32 MFMA peak (MI355X_MICROARCH.md chip table);
  roofline_wgrad : the same for conv_wgrad_kernel (3x3), the top line of the rocprof profile;
  cpu_baseline   : the CPU oracle (oracle/dip_oracle.py, a bitwise-verified restatement of the
                   reference's PyTorch-CPU path) timed on this box's host cores on the same workload;
  sustained      : (round 5) BASELINE.md section 4's own protocol next to the driver's short one -- 50 warm-up + 300 timed
                   iterations of the reported mode, with the PPT / sclk samples of that region;
  build_id       : (round 5) dip_build_id(): the sha256 of the sources the loaded libdip_hip.so was built from;
  per_rank_final_loss_hex : every rank's final loss, exact; `--gpus 1 --first-image r` is the solo run of rank r's fit
                   (tests/test_shard_gpu.py compares the two bit for bit).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: BF16 dense (2382 measured); the fp32 products of conv_bf3 / wgrad_bf3 run there
SELFTEST = os.environ.get("DIP_BENCH_SELFTEST") == "1"   # CPU-only plumbing test of the N-rank path (tests/test_host.py)

CONFIGS = {
    "default": dict(size=(512, 512), desc="default skip-net (2 217 831 params) 512x512 denoising fit"),
    "sr": dict(size=(512, 512), desc="default skip-net 512x512, super-resolution x4 closure (Lanczos2 down-sampler)"),
    "kate": dict(size=(512, 512), desc="inpainting 'kate' skip-net (skip=128, nearest) 512x512, masked MSE"),
    "library": dict(size=(448, 704), desc="inpainting 'library' skip-net (depth 6, 5x5 down filters) 448x704, masked MSE"),
    "snail": dict(size=(256, 384), desc="denoising 'snail' skip-net (8..128 channels) 256x384"),
}


# ------------------------------------------------------------------------------ rank plumbing
def shard_images(n_images: int, rank: int, world: int):
    """Static round-robin partition of independent image fits over ranks (no data exchange)."""
    return [i for i in range(n_images) if i % world == rank]


def reduce_max_time(t: float, device=None) -> float:
    """max over ranks of a wall-time (gloo, CPU tensor; not on the data path)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return t
    x = torch.tensor([t], dtype=torch.float64)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return float(x.item())


def gather_floats(v: float):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [v]
    xs = [torch.zeros(1, dtype=torch.float64) for _ in range(dist.get_world_size())]
    dist.all_gather(xs, torch.tensor([v], dtype=torch.float64))
    return [float(x.item()) for x in xs]


def gpu_sysfs_dir(dev_index):
    """sysfs directory of HIP device `dev_index` of this process, found by its PCI address (dip_device_pci_bus_id: the HIP
    runtime libdip_hip.so itself runs on) -- NOT by position among /sys/class/drm/card*: a container sees every card of the
    host there, while HIP enumerates only the GPUs it was given (round 4: the `device` block and the power samples of
    earlier lines came from card0, another tenant's GPU).  None when it cannot be determined."""
    import ctypes
    try:
        import __graft_entry__ as ge
        ge.add_to_path()
        import dip_native
        buf = ctypes.create_string_buffer(64)
        if dip_native.lib().dip_device_pci_bus_id(int(dev_index), buf, 64) != 0:
            return None
        path = "/sys/bus/pci/devices/" + buf.value.decode().strip().lower()
        return path if os.path.isdir(path) else None
    except Exception:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(local_rank, world):
    """N ranks on one host: keep each rank's Python thread (it enqueues ~160 launches per iteration, 2.4 ms of host
    time) on the cores of its GPU's NUMA node and keep torch's intra-op pool at one thread, so that 8 ranks do not
    migrate across sockets or oversubscribe each other.  Best effort (sysfs may be absent); returns a description."""
    import glob
    try:
        import torch
        if world > 1:
            torch.set_num_threads(1)
        ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
        cards = [gpu_sysfs_dir(r) if r < ngpu else None for r in range(max(world, local_rank + 1))]
        if local_rank >= len(cards) or cards[local_rank] is None:
            return {"pinned": False, "why": "no sysfs entry for this GPU"}
        node = int(open(cards[local_rank] + "/numa_node").read().strip())
        if node < 0:
            return {"pinned": False, "why": "numa_node = -1 (single-node host)"}
        cpus = parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return {"pinned": False, "why": "node's cores are outside this process' affinity mask"}
        if world > 1:
            # the ranks that share a node split its cores
            same = [r for r in range(world) if r < len(cards) and cards[r] is not None and
                    open(cards[r] + "/numa_node").read().strip() == str(node)]
            if local_rank not in same:
                same = [local_rank]
            k, n = same.index(local_rank), len(same)
            share = allowed[k * len(allowed) // n:(k + 1) * len(allowed) // n] or allowed
            os.sched_setaffinity(0, share)
            return {"pinned": True, "numa_node": node, "cores": len(share)}
        return {"pinned": False, "numa_node": node, "why": "single rank: affinity left alone"}
    except Exception as e:                   # never fail a benchmark over an affinity hint
        return {"pinned": False, "why": f"{type(e).__name__}: {e}"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_workers(n: int, argv) -> int:
    """`python bench.py --gpus N` outside torchrun: start one worker process per GPU (rank i drives
    GPU i), same environment contract as torch.distributed.run."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


# ------------------------------------------------------------------------------ problems
def make_problem(seed: int, size=(512, 512), depth=32):
    """Synthetic denoising problem of the reference's shape: clean = 5x5 box-blur
    of U(0,1) noise, target = clip(clean + N(0,(25/255)^2)), z = get_noise(depth,'noise') ~ U(0,0.1)."""
    import torch
    from utils.common_utils import get_noise
    torch.manual_seed(seed)
    np.random.seed(seed)
    z = get_noise(depth, 'noise', size)
    clean = torch.nn.functional.avg_pool2d(torch.rand(1, 3, size[0] + 4, size[1] + 4), 5, stride=1)
    noisy = np.clip(clean.numpy() + np.random.normal(scale=25 / 255., size=clean.shape), 0, 1).astype(np.float32)
    return z, torch.from_numpy(noisy)


def build_net(config: str):
    from models import get_net
    from models.skip import skip
    if config in ("default", "sr"):        # denoising.ipynb:160-165, super-resolution.ipynb:141-148
        return get_net(32, 'skip', 'reflection', skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                       upsample_mode='bilinear'), 32
    if config == "kate":                   # inpainting.ipynb:203-209
        return skip(32, 3, num_channels_down=[128] * 5, num_channels_up=[128] * 5, num_channels_skip=[128] * 5,
                    filter_size_up=3, filter_size_down=3, upsample_mode='nearest', filter_skip_size=1,
                    need_sigmoid=True, need_bias=True, pad='reflection', act_fun='LeakyReLU'), 32
    if config == "library":                # inpainting.ipynb:222-232
        ch = [16, 32, 64, 128, 128, 128]
        return skip(1, 3, num_channels_down=ch, num_channels_up=ch, num_channels_skip=[0] * 6, filter_size_up=3,
                    filter_size_down=5, filter_skip_size=1, upsample_mode='nearest', need1x1_up=False,
                    need_sigmoid=True, need_bias=True, pad='reflection', act_fun='LeakyReLU'), 1
    if config == "snail":                  # denoising.ipynb:143-150
        return skip(3, 3, num_channels_down=[8, 16, 32, 64, 128], num_channels_up=[8, 16, 32, 64, 128],
                    num_channels_skip=[0, 0, 0, 4, 4], upsample_mode='bilinear', need_sigmoid=True, need_bias=True,
                    pad='reflection', act_fun='LeakyReLU'), 3
    raise ValueError(config)


class Fit:
    """One independent image fit: net, input, target, closure and optimiser."""

    def __init__(self, config, seed, dev, closure_kind):
        import torch
        from utils.common_utils import get_params
        from dip_optim import FusedAdam
        self.config, self.dev = config, dev
        size = CONFIGS[config]["size"]
        torch.manual_seed(seed)
        net, depth = build_net(config)
        self.net = net.to(dev)
        z, target = make_problem(seed, size, depth)
        self.z = z.to(dev)
        self.target = target.to(dev)
        self.mask = None
        self.reg_std = {"default": 1. / 30., "snail": 1. / 30., "sr": 0.03, "kate": 0.03, "library": 0.0}[config]
        if config in ("kate", "library"):
            g = torch.Generator().manual_seed(seed + 1000)
            self.mask = (torch.rand(1, 1, *size, generator=g) > 0.3).float().expand(1, 3, *size).contiguous().to(dev)
        self.down = None
        if config == "sr":
            from models.downsampler import Downsampler
            self.down = Downsampler(n_planes=3, factor=4, kernel_type='lanczos2', phase=0.5, preserve_size=True).to(dev)
            self.target = torch.nn.functional.avg_pool2d(self.target, 4)        # a 128x128 LR image
        self.loss = None
        self.avg = None
        self.closure = self._notebook_closure() if closure_kind == "notebook" else self._fused_closure()
        self.opt = FusedAdam(get_params('net', self.net, self.z), lr=float(os.environ.get("DIP_BENCH_LR", "0.01")))      # == optimize('adam', ...)

    # the notebook's closure (denoising.ipynb:204-221, super-resolution.ipynb:169-186,
    # inpainting.ipynb:295-313) minus the per-iteration host syncs (prints / plots / PSNR on the CPU)
    def _notebook_closure(self, exp_weight=0.99):
        import torch
        mse = torch.nn.MSELoss()
        saved, noise = self.z.detach().clone(), self.z.detach().clone()

        def closure():
            net_input = saved
            if self.reg_std > 0:
                net_input = saved + (noise.normal_() * self.reg_std)
            out = self.net(net_input)
            if self.config in ("default", "snail"):
                self.avg = out.detach() if self.avg is None else self.avg * exp_weight + out.detach() * (1 - exp_weight)
            if self.down is not None:
                total_loss = mse(self.down(out), self.target)
            elif self.mask is not None:
                total_loss = mse(out * self.mask, self.target * self.mask)
            else:
                total_loss = mse(out, self.target)
            total_loss.backward()
            self.loss = total_loss.detach()
            return total_loss

        return closure

    # the same arithmetic through the opt-in device helpers; replay-safe (in-place state only; the first call, which
    # initialises the EMA, is an eager warm-up iteration, never the captured one)
    def _fused_closure(self, exp_weight=0.99):
        import torch
        from utils.reg_noise import RegNoise
        from utils.loss_head import MSEHead
        reg = RegNoise(self.z, self.reg_std, seed=1234)
        head = None if self.down is not None else MSEHead(self.net, self.target, self.mask)
        mse = torch.nn.MSELoss()
        ema = self.config in ("default", "snail")
        if ema:
            self.avg = torch.zeros_like(self.target)
        self.loss = torch.zeros((), device=self.dev)
        first = [True]

        def closure():
            net_input = reg()
            if head is not None:
                total_loss, out = head(net_input)
            else:                                   # SR: the loss goes through the Lanczos down-sampler
                out = self.net(net_input)
                total_loss = mse(self.down(out), self.target)
            if ema:
                if first[0]:               # denoising.ipynb:214-215: out_avg = out on the first iteration
                    self.avg.copy_(out.detach())
                    first[0] = False
                else:
                    self.avg.mul_(exp_weight).add_(out.detach(), alpha=1 - exp_weight)
            total_loss.backward()
            self.loss.copy_(total_loss.detach())
            return total_loss

        return closure

    def step(self):
        self.opt.zero_grad()
        self.closure()
        self.opt.step()

    @property
    def engine(self):
        return self.net.__dict__["_dip_engine"]


# ------------------------------------------------------------------------------ roofline
def conv_flops(eng):
    """Algorithmic conv FLOPs per iteration, per op name: 2*Cout*Ho*Wo*Cin*k*k of the LAYER for its
    forward, its weight gradient and (where the input needs it) its data gradient."""
    fl = {}

    def dims(r, H, W, pool):
        s = 1 if pool else r.stride
        Ho, Wo = (H + 2 * r.P - r.ks) // s + 1, (W + 2 * r.P - r.ks) // s + 1
        return 2.0 * r.Cout * Ho * Wo * r.Cin * r.ks * r.ks

    for i, s in enumerate(eng.sc):
        st = s.st
        H, W = st["H"], st["W"]
        Ho, Wo = st.get("Ho", H), st.get("Wo", W)
        for attr, (h, w) in (("skip_conv", (H, W)), ("down_a", (H, W)), ("down_b", (st["d1"].H, st["d1"].W)), ("up", (Ho, Wo)),
                             ("up1", (Ho, Wo))):
            r = getattr(s, attr)
            if r is not None:
                f = dims(r, h, w, attr == "down_a" and s.pool is not None)
                for pre in ("conv_fwd:", "wgrad:", "dgrad:", "dgrad+:"):
                    fl[pre + r.name] = f
                fl["dgthin:" + r.name] = 0.0       # thin columns of a 132-column data gradient: time counted, FLOPs are in "dgrad:"
    r = eng.out_conv
    f = dims(r, eng.Hout, eng.Wout, False)
    fl["conv_fwd:out"] = fl["wgrad:out"] = fl["dgrad:out"] = f
    return fl


def profile_ops(eng, reps=3):
    """HIP-event time of every launch of one iteration, in sequence on the engine's own stream
    (torch's current stream), averaged over `reps` instrumented iterations.  Composite dispatches of
    dip_conv_igemm are split into their kernels: "#thin4" / "#dma" (132-column data gradients) and
    "#main" / "#finish" (split-K launches of the LDS-DMA kernel and their reduction)."""
    import ctypes as C
    import torch
    import dip_native as N
    lib = N.lib()
    lib.dip_conv_igemm_dma.restype, lib.dip_conv_igemm_dma.argtypes = C.c_int, [C.POINTER(N.DipConvDesc), C.c_int, C.c_void_p]
    stream = torch.cuda.current_stream(eng.device)
    sptr = stream.cuda_stream
    acc = {}
    for rep in range(reps):
        for ops in (eng.fwd_ops, eng.bwd_ops):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
            evs[0].record(stream)
            mids = {}
            for k, (fn, args, name) in enumerate(ops):
                variant = lib.dip_conv_variant(args[0]) if fn is lib.dip_conv_igemm else -1
                if variant == 3:
                    # 132-column data gradient = conv_thin4 + a 128-column launch of the dominant kernel
                    ncols = args[0]._obj.Cout - 128
                    lib.dip_conv_thin4(args[0], ncols, sptr)
                    mids[k] = ("#thin4", "#dma", torch.cuda.Event(enable_timing=True))
                    mids[k][2].record(stream)
                    lib.dip_conv_igemm_dma_cols(args[0], ncols, sptr)
                elif variant in (1, 4, 5) and args[0]._obj.ksplit > 1:
                    lib.dip_conv_igemm_dma(args[0], args[0]._obj.ksplit, sptr)
                    mids[k] = ("#main", "#finish", torch.cuda.Event(enable_timing=True))
                    mids[k][2].record(stream)
                    lib.dip_conv_splitk_finish(args[0], sptr)
                else:
                    fn(*args, sptr)
                evs[k + 1].record(stream)
            torch.cuda.synchronize()
            for k, (_, _, name) in enumerate(ops):
                acc.setdefault(name, []).append(evs[k].elapsed_time(evs[k + 1]))
                if k in mids:
                    a, b, ev = mids[k]
                    acc.setdefault(name + a, []).append(evs[k].elapsed_time(ev))
                    acc.setdefault(name + b, []).append(ev.elapsed_time(evs[k + 1]))
    return {k: float(np.mean(v)) for k, v in acc.items()}


def count_kernels(eng):
    """Kernel launches of one iteration: the engine's op lists (an op = one C-ABI call; split-K dispatches are conv + finish,
    a 132-column data gradient inside dip_conv_igemm is thin4 + 128 columns, a bf16-pipe weight gradient of a 132-channel
    layer has a packed-tail launch behind it) + noise, counter, weight packing (+ the bf16 planes), layout, loss reduction, Adam
    tick + step."""
    import dip_native as N
    lib = N.lib()
    n = 0
    for ops in (eng.fwd_ops, eng.bwd_ops):
        for fn, args, name in ops:
            n += 1
            if fn is lib.dip_conv_igemm:
                d = args[0]._obj
                v = lib.dip_conv_variant(args[0])
                if d.ksplit > 1:
                    n += 1
                if v == 3:
                    n += 1
            elif fn is lib.dip_conv_wgrad:
                d = args[0]._obj
                if lib.dip_wgrad_bf3_eligible(args[0]) and 1 <= (d.Cin & 31) <= 4:
                    n += 1
    return n + 7 + (1 if getattr(eng, "bf3", False) else 0)


DOMINANT = "conv_igemm_dma_kernel<3,128,*>"
WGRAD = "conv_wgrad_kernel<3,*,9,1>"


def dominant_ops(eng, fl):
    """Launch-list entries that run the dominant kernel, as {timing key: (algorithmic flops,
    compulsory bytes)}: 3x3 convolutions (forward and data gradient) that dip_conv_igemm dispatches
    to the LDS-DMA implicit-GEMM kernel with a 128-column tile -- dip_conv_variant == 1 (timing key
    "#main" when split-K) and the 128-column part ("#dma") of the 132-column data gradients
    (variant 3).  FLOPs are the LAYER's algorithmic FLOPs (scaled by the share of the columns this
    kernel computes), not the MACs the launch executes on its padded / dilated domain."""
    import dip_native as N
    lib = N.lib()
    out = {}
    for ops in (eng.fwd_ops, eng.bwd_ops):
        for fn, args, name in ops:
            if name.partition(":")[0] not in ("conv_fwd", "dgrad", "dgrad+"):
                continue
            d = args[0]._obj
            if d.ks != 3 or d.Cout < 128 or fn in (lib.dip_conv_small, lib.dip_conv_dgrad_ring):   # (conv_small_kernel)
                continue
            v = lib.dip_conv_variant(args[0])
            if v not in (1, 3, 7):
                continue
            # with the bf16-pipe kernel switched on, IT is the dominant kernel (the >= 256-tile layers); what is left on
            # the fp32 LDS-DMA kernel (128 x 128 layers) is reported under roofline_conv3x3_all
            on_bf3 = bool(lib.dip_conv_bf3_eligible(args[0]))
            if bool(lib.dip_conv_bf3_terms()) != on_bf3:
                continue
            cols = d.Cout if v in (1, 7) else 128
            flops = fl[name] * cols / d.Cout
            # compulsory traffic of the launch: input and packed weights read once, output written once
            nbytes = 4.0 * (d.Hin * d.Win * d.Cin + 9 * d.Cin * cols + d.Hout * d.Wout * cols)
            if fn is lib.dip_conv_igemm_dma_cols:            # the engine launches the 128-column part itself
                key = name
            else:
                key = name + "#dma" if v == 3 else (name + "#main" if d.ksplit > 1 else name)
            out[key] = (flops, nbytes)
    return out


def roofline_thin(eng, per_op_ms, fl):
    """The thin-layer kernels of round 6 (conv_thin_kernel: dip_conv_variant == 8; wgrad_thin_kernel: dip_wgrad_thin_eligible)
    as a group: the layers' algorithmic FLOPs / their per-launch HIP-event time against the fp32 MFMA peak (they run
    v_mfma_f32_16x16x4_f32).  None when the net has no such layer (the default / kate nets)."""
    import dip_native as N
    lib = N.lib()
    groups = {"conv_thin_kernel": [0.0, 0.0, 0], "wgrad_thin_kernel": [0.0, 0.0, 0]}
    for ops in (eng.fwd_ops, eng.bwd_ops):
        for fn, args, name in ops:
            kind = name.partition(":")[0]
            ms = per_op_ms.get(name)
            if ms is None:
                continue
            if kind in ("conv_fwd", "dgrad") and fn is lib.dip_conv_igemm and lib.dip_conv_variant(args[0]) == 8:
                g = groups["conv_thin_kernel"]
            elif kind == "wgrad" and lib.dip_wgrad_thin_eligible(args[0]):
                g = groups["wgrad_thin_kernel"]
            else:
                continue
            g[0] += fl.get(name, 0.0); g[1] += ms; g[2] += 1
    if not any(g[2] for g in groups.values()):
        return None
    out = {"bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "traffic": None}
    tf, tm = 0.0, 0.0
    for k, (f, ms, n) in groups.items():
        if n:
            out[k] = {"launches_per_step": n, "ms_per_step": round(ms, 3), "algorithmic_gflop_per_step": round(f / 1e9, 2),
                      "achieved": round(f / (ms * 1e-3) / 1e12, 2), "frac": round(f / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
            tf += f; tm += ms
    out["achieved"] = round(tf / (tm * 1e-3) / 1e12, 2)
    out["frac"] = round(out["achieved"] / PEAK_FP32_MFMA_TFLOPS, 4)
    return out


def hbm_ops(eng):
    """Compulsory HBM bytes of every memory-bound launch of one iteration under THIS engine's fusion plan, as
    {op name: bytes}: each operand tensor of the launch read once, each result written once (4 bytes per float,
    channel strides as stored).  The MFMA convolutions are not in here (roofline / roofline_conv3x3_all); the thin
    1x1 convs (<= 4 output channels) and the thin data-gradient columns are -- they stream a tensor for a few FLOP
    per byte.  SURVEY.md 8(d) asks for this figure next to the 9.80 GB/iter of the unfused reference graph."""
    from dip_native import round_up
    B = {}
    F4 = 4.0
    n_arena = eng.n_arena
    z = eng.H * eng.W
    B["noise_axpy"] = F4 * 2 * z * eng.Cimg
    B["nchw_to_nhwc"] = F4 * z * (eng.Cimg + round_up(eng.Cimg, 4))
    B["pack_weights"] = F4 * (sum(r.Cout * r.Cin * r.ks * r.ks for r in eng.convs) + eng.packed.numel())
    B["adam"] = F4 * 7 * n_arena
    oc = eng.out_conv
    B["loss_head_fwd"] = F4 * z * (eng.last_act.Cs + 2 * oc.Cout)
    B["loss_head_bwd"] = F4 * z * (2 * oc.Cout + round_up(oc.Cout, 4))

    def conv_io(r, x, pixels_out):
        return F4 * (x.H * x.W * x.Cs + pixels_out * round_up(r.Cout, 4))

    for i, s in enumerate(eng.sc):
        st = s.st
        H, W, xin = st["H"], st["W"], st["xin"]
        acts = {"skip_bn": st.get("s_act"), "down_a_bn": st["d1"], "down_b_bn": st["d2"], "cat_bn": st["cat_act"],
                "up_bn": st["u"], "up1_bn": st.get("u1")}
        for key, a in acts.items():
            if a is None:
                continue
            bn = a.bn
            t = F4 * a.H * a.W * a.Cs
            B[f"bnb_stats:{bn.name}"] = 2 * t                    # g (the ring of a padded g: < 2 %) + y
            B[f"bnb_apply:{bn.name}"] = 3 * t                    # g + y -> dy  (in place after upb_stats: dz + y -> dz)
            B[f"bnb_one:{bn.name}"] = 3 * t                      # one-launch form: g + y -> dy (the second pass re-reads from L2)
        if s.ns:
            B[f"conv_fwd:{s.skip_conv.name}"] = conv_io(s.skip_conv, xin, H * W)
            B[f"dgrad+:{s.skip_conv.name}"] = F4 * H * W * (round_up(s.ns, 4) + 2 * xin.Cs)
            B[f"wgrad:{s.skip_conv.name}"] = F4 * H * W * (xin.Cs + round_up(s.ns, 4))
        deep, cat = st["deep"], st["cat_act"]
        B[f"upcat:{s.cat_bn.name}"] = F4 * (H * W * (round_up(s.ns, 4) if s.ns else 0) + deep.H * deep.W * deep.Cs + H * W * cat.Cs)
        B[f"upb_stats:{deep.bn.name}"] = F4 * (H * W * deep.C + 2 * deep.H * deep.W * deep.Cs)
        B[f"upb_one:{deep.bn.name}"] = F4 * (H * W * deep.C + 2 * deep.H * deep.W * deep.Cs)
        if s.up.Cin > 128 and s.up.Cin <= 132:                   # thin columns of the 132-column data gradient
            B[f"dgthin:{s.up.name}"] = F4 * H * W * (round_up(s.up.Cout, 4) + 4)
    B[f"wgrad:{oc.name}"] = F4 * z * (eng.last_act.Cs + round_up(oc.Cout, 4))
    return B


def roofline_hbm(eng, per_op_ms):
    """The memory-bound launches as a group: compulsory bytes of the fusion plan (hbm_ops) / their HIP-event time,
    against 8 TB/s (spec) and the 6.29 TB/s copy ceiling (MI355X_MICROARCH.md), next to the traffic the PMC passes
    measured for the same kernels (profiles/r03_pmc_traffic.json, taken on another box in another run)."""
    B = hbm_ops(eng)
    tot_b = tot_ms = 0.0
    timed = {}
    for k, b in B.items():
        ms = per_op_ms.get(k)
        if ms is None:                    # not a launch of this plan (statistics fused elsewhere), or issued outside the
            continue                      # engine's op lists (noise, Adam, loss head, layout, weight packing)
        timed[k] = (b, ms)
        tot_b += b
        tot_ms += ms
    # launches that move data but have no compulsory count above (finalisations, slab / split-K reductions): their
    # time counts against the group, their bytes are reported separately (they exist only because of the plan)
    extra_ms = sum(ms for k, ms in per_op_ms.items() if k.startswith(("bn_fin:", "bnb_fin:", "wgred:")) or k.endswith("#finish"))
    ach = tot_b / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    untimed = sum(b for k, b in B.items() if k not in timed and ":" not in k)
    pmc = None
    try:
        pth = [q for q in (os.path.join(ROOT, "profiles", f"{r}_pmc_traffic.json") for r in ("r06", "r05", "r04", "r03")) if os.path.exists(q)][0]
        with open(pth) as f:
            pmc = json.load(f).get("memory_bound_group")
    except (OSError, ValueError):
        pass
    return {"bound": "hbm", "kernels": "BatchNorm backward, up-sample + concat and its adjoint, thin 1x1 convs / weight gradients, "
                                       "thin data-gradient columns (per-launch HIP events, single stream)",
            "achieved": round(ach, 3), "peak": 8.0, "unit": "TB/s", "frac": round(ach / 8.0, 4),
            "frac_of_copy_ceiling_6.29": round(ach / 6.29, 4),
            "compulsory_gb_per_step": round(tot_b / 1e9, 3), "ms_per_step": round(tot_ms, 3), "launches_per_step": len(timed),
            "compulsory_gb_of_launches_outside_the_op_lists": round(untimed / 1e9, 3),
            "finalise_and_reduction_launches_ms_per_step_not_included": round(extra_ms, 3),
            "unfused_reference_graph_gb_per_step": 9.80 if (eng.H, eng.W) == (512, 512) else None,
            "traffic": pmc["bytes_per_step"] if pmc else None,
            "traffic_source": (pmc.get("source") if pmc else None)}


def conv3x3_all(eng, per_op_ms, fl):
    """roofline_conv3x3_all: EVERY launch that does 3x3-conv work -- forward, data and weight gradient, stride 1 and 2,
    the thin columns, the N = 160 variant, and the split-K / slab reductions behind them -- against the algorithmic
    FLOPs of the 3x3 layers (north_star: ">= 60 % of the fp32 MFMA roofline on the 3x3 conv layers")."""
    f = ms = 0.0
    n = 0
    names3 = {r.name for r in eng.convs if r.ks == 3}
    for k, t in per_op_ms.items():
        base = k.split("#")[0]
        kind, _, lname = base.partition(":")
        if lname not in names3 or kind not in ("conv_fwd", "dgrad", "dgrad+", "dgthin", "wgrad", "wgred"):
            continue
        if "#" in k and k.split("#")[1] in ("thin4", "dma", "main", "finish"):
            continue                      # the parts of a composite dispatch: `base` already carries the whole time
        ms += t
        n += 1
        if kind in ("conv_fwd", "dgrad", "dgrad+", "wgrad"):
            f += fl.get(base, 0.0)
    ach = f / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {"bound": "mfma", "kernels": "every launch of the 3x3 layers: LDS-DMA / register-staged implicit GEMM (stride 1, "
                                        "stride-2 modes), conv_thin4, conv_wgrad + slab reductions, split-K reductions",
            "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "algorithmic_gflop_per_step": round(f / 1e9, 2),
            "ms_per_step_serial": round(ms, 3), "launches_per_step": n, "traffic": None}


class PowerSampler:
    """Side thread: the GPU's hwmon power1_average / power1_input (PPT) and freq1_input (sclk) at ~20 Hz from before the
    warm-up to the end of the timed region -- the pool has two speed classes of boxes that differ in what they draw / clock
    under the same code.  summary(t0, t1): mean / max over the timed window, plus the raw trace (ms relative to t0)."""

    def __init__(self, dev_index=0, period=0.05):
        import glob
        import threading
        self.paths = []
        card = gpu_sysfs_dir(dev_index)
        if card is not None:
            for hw in glob.glob(card + "/hwmon/hwmon*"):
                pw = next((hw + "/" + f for f in ("power1_average", "power1_input") if os.path.exists(hw + "/" + f)), None)
                fq = hw + "/freq1_input" if os.path.exists(hw + "/freq1_input") else None
                if pw:
                    self.paths.append((pw, fq))
        self.period, self.samples = period, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        for pw, fq in self.paths[:1]:
            try:
                t = time.perf_counter()
                with open(pw) as f:
                    w = int(f.read()) / 1e6
                mhz = None
                if fq:
                    with open(fq) as f:
                        mhz = int(f.read()) / 1e6
                self.samples.append((t, w, mhz))
            except (OSError, ValueError):
                pass

    def _run(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(self.period)

    def __enter__(self):
        if self.paths:
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.paths:
            self._th.join(timeout=1.0)
            self._read()

    def summary(self, t0, t1):
        if not self.samples:
            return None
        win = [x for x in self.samples if t0 <= x[0] <= t1] or self.samples[-1:]
        out = {"sensor": os.path.basename(self.paths[0][0]) + " (PPT)", "samples_in_timed_region": len(win),
               "power_w_mean": round(float(np.mean([x[1] for x in win])), 1), "power_w_max": round(float(np.max([x[1] for x in win])), 1),
               "power_w_max_since_warmup": round(float(np.max([x[1] for x in self.samples])), 1)}
        mh = [x[2] for x in win if x[2] is not None]
        if mh:
            out["sclk_mhz_mean"], out["sclk_mhz_min"] = round(float(np.mean(mh))), round(float(np.min(mh)))
        k = max(1, len(self.samples) // 40)
        out["trace_ms_w_mhz"] = [[round(1e3 * (x[0] - t0)), round(x[1]), None if x[2] is None else round(x[2])] for x in self.samples[::k]]
        return out


def device_info(dev_index=0):
    """Clocks / power cap / partition mode of the GPU this line was measured on (the pool has two speed classes of
    boxes): sysfs + rocm-smi, best effort."""
    import glob
    info = {}

    def rd(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    c = gpu_sysfs_dir(dev_index)
    if c is not None:
        info["pci"] = os.path.basename(c)
        for key, fn in (("sclk_levels", "pp_dpm_sclk"), ("mclk_levels", "pp_dpm_mclk"), ("fclk_levels", "pp_dpm_fclk"),
                        ("perf_level", "power_dpm_force_performance_level"), ("compute_partition", "current_compute_partition"),
                        ("memory_partition", "current_memory_partition"), ("vbios", "vbios_version"),
                        ("numa_node", "numa_node")):
            v = rd(f"{c}/{fn}")
            if v is not None:
                info[key] = " | ".join(v.split("\n")) if "\n" in v else v.replace("\n", " | ")
        for hw in glob.glob(c + "/hwmon/hwmon*"):
            cap, avg = rd(hw + "/power1_cap"), rd(hw + "/power1_average") or rd(hw + "/power1_input")
            if cap:
                info["power_cap_w"] = int(cap) / 1e6
            if avg:
                info["power_now_w"] = int(avg) / 1e6
    try:
        import torch
        p = torch.cuda.get_device_properties(dev_index)
        info["name"], info["cus"] = p.name, p.multi_processor_count
        info["clock_rate_mhz"] = getattr(p, "clock_rate", 0) / 1e3 or None
    except Exception:
        pass
    return info


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r0N_pmc_traffic.json, produced by tools/pmc_traffic.py); None when absent."""
    import dip_native as N
    want = "conv_bf3_kernel<*>" if N.lib().dip_conv_bf3_terms() else DOMINANT
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")) as f:
                t = json.load(f)
            if t.get("kernel") == want:
                t["round"] = rnd
                return t
        except (OSError, ValueError):
            pass
    return None


def ubench_ceiling(terms):
    """What the bare MFMA loops of tools/ubench reach on this part (mfma_peak.hip: v_mfma_f32_32x32x2_f32; bf16x9.hip: the
    9-product bf16 inner loop, fp32-equivalent TFLOP/s), read from the committed output of the round's evidence call
    (profiles/r0N_ubench_ceilings.json) -- not a constant in this file."""
    for rnd in ("r06", "r05"):
        pth = os.path.join(ROOT, "profiles", f"{rnd}_ubench_ceilings.json")
        try:
            with open(pth) as f:
                u = json.load(f)
            v = u["mfma_f32_tflops"] if not terms else u["bf16x9_fp32_equivalent_tflops"] * 9.0 / terms
            return {"tflops": round(v, 1), "source": os.path.relpath(pth, ROOT),
                    "stale": os.environ.get("DIP_BENCH_PMC_SAME_CALL") != "1"}
        except (OSError, ValueError, KeyError):
            continue
    return None


def roofline(eng, per_op_ms, with_pmc=True):
    """roofline (dominant conv kernel) and roofline_wgrad (3x3 weight-gradient kernel)."""
    fl = conv_flops(eng)
    dom = dominant_ops(eng, fl)
    tot_f = sum(f for f, _ in dom.values())
    alg_bytes = sum(b for _, b in dom.values())
    tot_ms = sum(per_op_ms[k] for k in dom)
    n = len(dom)
    ach = tot_f / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    fin_ms = sum(ms for k, ms in per_op_ms.items() if k.endswith("#finish"))
    big = per_op_ms.get("conv_fwd:s0.up")
    # every MFMA conv launch (forward, data and weight gradient, all kernels) for the whole-path figure
    all_f = sum(f for k, f in fl.items() if k in per_op_ms)
    all_ms = sum(ms for k, ms in per_op_ms.items() if k in fl)          # (the "#..." parts are not in fl)
    pmc = pmc_traffic() if with_pmc else None     # the PMC passes were taken on the default workload only
    import dip_native as N
    terms = N.lib().dip_conv_bf3_terms()
    if terms:
        # fp32 operands split exactly into three bf16 terms; `terms` cross products per fp32 product on the bf16 pipe:
        # the fp32-equivalent ceiling of that scheme is the bf16 dense peak / terms
        kname = (f"conv_bf3_kernel<{terms},*> (3x3 stride-1 forward + data-gradient launches with >= 256 tiles: fp32 operands "
                 f"as three exact bf16 terms, {terms} partial products per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 "
                 "accumulation)")
        peak = PEAK_BF16_MFMA_TFLOPS / terms
    else:
        kname, peak = DOMINANT + " (3x3 stride-1 forward + 3x3 data-gradient launches)", PEAK_FP32_MFMA_TFLOPS
    rl = {"bound": "mfma", "kernel": kname,
          "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
          "frac": round(ach / peak, 4),
          "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
          # DIP_BENCH_PMC_SAME_CALL=1 is set by the round's evidence script (tools/gpu_round6_final.sh), which takes the counter
          # passes and this line in ONE gpurun call on ONE box; any other run reads what is committed: stale
          "traffic_stale": (os.environ.get("DIP_BENCH_PMC_SAME_CALL") != "1") if pmc else None,
          "traffic_source": (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes committed under profiles/ ({pmc.get('round', 'r02')}): "
                             + ("the same gpurun call and box as this line" if os.environ.get("DIP_BENCH_PMC_SAME_CALL") == "1"
                                else "STALE: another box, another run than this line")) if pmc else None,
          "launches_per_step": n, "avg_launch_us": round(1e3 * tot_ms / max(n, 1), 1),
          "algorithmic_gflop_per_launch": round(tot_f / 1e9 / max(n, 1), 2),
          "algorithmic_bytes_per_launch": round(alg_bytes / max(n, 1)),
          "splitk_finish_ms_per_step_not_included": round(fin_ms, 3),
          "measured_mfma_ceiling_tflops": ubench_ceiling(terms),
          "all_conv_launches": {"gflop_per_step": round(all_f / 1e9, 1), "ms_per_step_serial": round(all_ms, 3),
                                "tflops": round(all_f / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else None}}
    try:        # matrix-pipe utilisation from the committed counter pass (tools/pmc_mfma.py; another run than this line)
        pmf = [q for q in (os.path.join(ROOT, "profiles", f"{r}_pmc_mfma.json") for r in ("r06", "r05", "r04")) if os.path.exists(q)][0]
        with open(pmf) as f:
            pm = json.load(f)
        key = "conv_bf3_kernel" if terms else "conv_igemm_dma_kernel<3, 128"
        us = [(v["launches"], v["utilisation"], v.get("clock_ghz") or 2.4) for k, v in pm["kernels"].items()
              if k.startswith(key) and not v.get("short_dispatch")]
        if us:
            nl = sum(n for n, _, _ in us)
            util, ghz = sum(n * u for n, u, _ in us) / nl, sum(n * g for n, _, g in us) / nl
            rl["mfma_util_pmc"] = round(util, 4)
            rl["mfma_util_pmc_stale"] = os.environ.get("DIP_BENCH_PMC_SAME_CALL") != "1"
            rl["clock_ghz_pmc"] = round(ghz, 3)
            # frac prices the kernel against the peak at the nominal 2.4 GHz; the counters count CYCLES: the two meet at
            # utilisation x (clock the kernel ran at) / 2.4 (every executed MFMA of these launches is algorithmic work)
            rl["frac_from_pmc"] = round(util * ghz / 2.4, 4)
            rl["mfma_util_pmc_source"] = (os.path.relpath(pmf, ROOT).replace("_pmc_mfma.json", "_rocprofv3_pmc_MFMA.txt") + " (a counter pass of the same code; same box and call as this "
                                          "line when tools/gpu_round5_final.sh produced both): utilisation = SQ_VALU_MFMA_BUSY_CYCLES / "
                                          "(128 x GRBM_GUI_ACTIVE), clock = GRBM_GUI_ACTIVE / (8 x duration), units calibrated on the "
                                          "pure-MFMA loops of tools/ubench in the same call; frac_from_pmc = utilisation x clock / 2.4 GHz "
                                          f"is the counter-derived value of `frac`: the matrix pipes of this kernel are busy {100 * util:.0f} % "
                                          f"of the cycles, and the cycles come at {ghz:.2f} GHz (power management under bf16 MFMA load; the "
                                          "bare bf16 MFMA loop runs at 2.06 GHz)")
    except (OSError, ValueError, KeyError):
        pass
    if terms:
        rl["peak_is"] = (f"bf16 dense MFMA peak {PEAK_BF16_MFMA_TFLOPS:.0f} TFLOP/s / {terms} products per fp32 product = fp32-equivalent "
                         "ceiling of the scheme; `achieved` counts the layer's algorithmic fp32 FLOPs")
        rl["executed_bf16_tflops"] = round(ach * terms, 1)
        rl["frac_of_fp32_mfma_peak_157.3"] = round(ach / PEAK_FP32_MFMA_TFLOPS, 4)
    if big and "conv_fwd:s0.up" in fl:
        rl["largest_layer"] = {"name": "s0.up forward", "gflop": round(fl["conv_fwd:s0.up"] / 1e9, 2),
                               "us": round(1e3 * big, 1), "tflops": round(fl["conv_fwd:s0.up"] / (big * 1e-3) / 1e12, 2)}
    # 3x3 weight gradients (conv_wgrad_kernel<3,S,9,1>; the slab reduction is a separate kernel)
    wk = {}
    for fn, args, name in eng.bwd_ops:
        if name.startswith("wgrad:") and args[0]._obj.ks == 3:
            wk[name] = fl[name]
    w_f, w_ms = sum(wk.values()), sum(per_op_ms[k] for k in wk)
    w_ach = w_f / (w_ms * 1e-3) / 1e12 if w_ms > 0 else 0.0
    bigw = per_op_ms.get("wgrad:s0.up")
    rw = {"bound": "mfma", "kernel": WGRAD + " (all 3x3 weight-gradient launches)", "achieved": round(w_ach, 2),
          "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(w_ach / PEAK_FP32_MFMA_TFLOPS, 4),
          "traffic": None, "launches_per_step": len(wk), "ms_per_step": round(w_ms, 3),
          "reduce_ms_per_step_not_included": round(sum(ms for k, ms in per_op_ms.items() if k.startswith("wgred:")), 3)}
    # the big layers' weight gradients run wgrad_bf3_kernel (bf16 pipe) when it is switched on: split the set
    lib3 = N.lib()
    w3 = {name for fn, args, name in eng.bwd_ops if name in wk and lib3.dip_wgrad_bf3_eligible(args[0])}
    if w3:
        f3, ms3 = sum(wk[k] for k in w3), sum(per_op_ms[k] for k in w3)
        a3 = f3 / (ms3 * 1e-3) / 1e12 if ms3 > 0 else 0.0
        rw["kernel"] = (f"wgrad_bf3_kernel<{terms},*> on the {len(w3)} layers with >= 256 x 256 outputs (bf16 pipe, incl. the fp32 packed-tail "
                        f"launch of the 132-channel layers) + {WGRAD} on the other {len(wk) - len(w3)}: `achieved` / `frac` are the whole set "
                        "against the fp32 MFMA peak")
        rw["bf16_pipe_launches"] = {"launches_per_step": len(w3), "ms_per_step": round(ms3, 3), "achieved": round(a3, 2),
                                    "peak": round(PEAK_BF16_MFMA_TFLOPS / terms, 1), "frac": round(a3 / (PEAK_BF16_MFMA_TFLOPS / terms), 4),
                                    "frac_of_fp32_mfma_peak_157.3": round(a3 / PEAK_FP32_MFMA_TFLOPS, 4)}
    if bigw and "wgrad:s0.up" in fl:
        rw["largest_layer"] = {"name": "s0.up weight gradient", "gflop": round(fl["wgrad:s0.up"] / 1e9, 2),
                               "us": round(1e3 * bigw, 1), "tflops": round(fl["wgrad:s0.up"] / (bigw * 1e-3) / 1e12, 2)}
    return rl, rw


# ------------------------------------------------------------------------------ CPU baseline
def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seed=0, timed=5, warm=2):
    """The CPU oracle on this box's host cores: same net, same 512x512 workload, same closure; `warm` warm-up + `timed`
    timed Adam iterations (2 + >= 5), median -- at the best thread count of a short sweep taken in
    this call (one timed iteration each at 8 / 16 / 32 threads after one warm-up) and recorded in the line."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dip_oracle as O
    # torch-CPU conv scaling collapses on many-core hosts: on the 256-thread MI355X host a sweep
    # (tools/cpu_sweep.py, 256x256) gave 3.47 / 2.38 / 1.25 / 0.59 / 0.011 it/s at 16 / 32 / 64 /
    # 128 / 256 threads, so the baseline sweeps 8 / 16 / 32 threads and uses the best, not all of them.
    ncpu = os.cpu_count() or 1
    torch.manual_seed(seed)
    net, _ = build_net("default")
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in O.param_shapes(O.default_spec())}
    onet = O.OracleNet(O.default_spec(), sd)
    z, target = make_problem(seed)
    mse = torch.nn.MSELoss()
    saved, noise = z.clone(), z.clone()
    st = {"avg": None}

    def closure():
        out = onet(saved + noise.normal_() * (1. / 30.))
        st["avg"] = out.detach() if st["avg"] is None else st["avg"] * 0.99 + out.detach() * 0.01
        loss = mse(out, target)
        loss.backward()
        return loss

    opt = torch.optim.Adam(onet.params, lr=0.01)

    def one():
        t0 = time.time()
        opt.zero_grad()
        closure()
        opt.step()
        return time.time() - t0

    sweep = {}
    for th in sorted({min(ncpu, t) for t in (8, 16, 32)}):
        torch.set_num_threads(th)
        one()                                   # (thread pool / oneDNN primitive warm-up at this setting)
        sweep[th] = one()
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    for _ in range(warm):
        one()
    tt = [one() for _ in range(timed)]
    return {"value": round(1.0 / float(np.median(tt)), 4), "unit": "it/s", "cores": cores, "kind": "port",
            "note": "the unmodified reference (/root/reference) does not exist on the GPU box: this is oracle/dip_oracle.py, "
                    "its restatement on torch.nn.functional, verified bitwise against the real reference in the build container",
            "host": f"{host_cpu_model()} ({ncpu} hardware threads)",
            "thread_sweep_it_s": {str(k): round(1.0 / v, 3) for k, v in sweep.items()},
            "sample": f"default skip-net 512x512, {warm} warm-up + {len(tt)} timed Adam iterations of the CPU oracle "
                      f"(torch {torch.__version__} CPU, {cores} threads = the best of the sweep), median; min/max "
                      f"{1.0 / max(tt):.3f}/{1.0 / min(tt):.3f} it/s"}


# ------------------------------------------------------------------------------ main
def timed_run(fits, steps, warmup, use_graph, barrier):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize.
    Returns (seconds, graph_used, note)."""
    import torch
    from dip_optim import GraphedIteration
    note = None
    graph = None
    with PowerSampler(torch.cuda.current_device()) as ps:
        if use_graph:
            try:
                graph = GraphedIteration.group([(f.opt, f.closure) for f in fits], warmup=min(3, max(warmup, 1)))
                rest = warmup - graph.iterations
                if rest > 0:
                    graph.run(rest)
            except Exception as e:                               # report, then time the eager loop instead
                note = f"graph capture failed: {type(e).__name__}: {e}"
                graph = None
                torch.cuda.synchronize()
        if graph is None:
            for _ in range(warmup):
                for f in fits:
                    f.step()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        if graph is not None:
            graph.run(steps)
        else:
            for _ in range(steps):
                for f in fits:
                    f.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        t = t1 - t0
    barrier()
    return t, graph is not None, note, ps.summary(t0, t1)


def host_issue(fits, n=4):
    """Host time of ISSUING one iteration (the Python thread, no synchronisation inside the window; the queues are empty at
    its start and n iterations fit into them): with the engine's command lists (dip_list_run: one foreign call per direction)
    and, for comparison, launch by launch from Python as rounds 1-5 did.  What N ranks on one host cost it per GPU."""
    import torch
    out = {}
    engs = [f.engine for f in fits]
    was = [e.use_clist for e in engs]
    try:
        for label, flag in (("command_list", True), ("python_loop", False)):
            for e in engs:
                e.use_clist, e._clists = flag, {}
            for f in fits:
                f.step()                       # (compiles the lists / creates the events)
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    for f in fits:
                        f.step()
                t = (time.perf_counter() - t0) / n
                best = t if best is None else min(best, t)
            torch.cuda.synchronize()
            out[label + "_ms_per_iteration"] = round(1e3 * best, 3)
    finally:
        for e, w in zip(engs, was):
            e.use_clist, e._clists = w, {}
    nl = count_kernels(engs[0])
    out["launches_per_iteration"] = nl
    out["us_per_launch"] = {k.replace("_ms_per_iteration", ""): round(1e3 * v / max(nl, 1), 2) for k, v in out.items() if k.endswith("_ms_per_iteration")}
    out["what"] = ("host wall time of issuing one whole iteration (closure + optimiser step) with nothing waited for: minimum over "
                   "3 windows of %d iterations; the GPU needs ms_per_step for it" % n)
    return out


def grouped_fits(config, images, dev):
    """The same fits as one dip_group.GroupedFits (ONE launch list for all of them): fresh nets with the seeds of `images`."""
    from dip_group import GroupedFits
    fits = [Fit(config, img, dev, "fused") for img in images]
    f0 = fits[0]
    if f0.down is not None:
        raise NotImplementedError("the SR closure (loss through the Downsampler) has no grouped form")
    ema = config in ("default", "snail")
    return GroupedFits([f.net for f in fits], [f.z for f in fits], [f.target for f in fits],
                       masks=None if f0.mask is None else [f.mask for f in fits], reg_noise_std=f0.reg_std,
                       seeds=[1234] * len(fits), lr=0.01, exp_weight=0.99 if ema else None, ema_init="first", device=dev)


def timed_run_grouped(g, steps, warmup, use_graph, barrier):
    """timed_run for a GroupedFits: W untimed grouped iterations (the hipGraph capture among them), then exactly K timed."""
    import torch
    with PowerSampler(torch.cuda.current_device()) as ps:
        if use_graph:
            g.capture(warmup=min(3, max(warmup, 1)))
            rest = warmup - g.iterations
            if rest > 0:
                g.run(rest)
        else:
            g.step(warmup)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        if use_graph:
            g.run(steps)
        else:
            g.step(steps)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    barrier()
    return t1 - t0, ps.summary(t0, t1)


def selftest_rank(args, rank, world, barrier):
    """DIP_BENCH_SELFTEST=1: the N-rank plumbing (spawn, rendezvous, barrier, max-over-ranks, per-rank
    gather, JSON) with a dummy CPU step instead of the GPU fit -- driven by tests/test_host.py."""
    barrier()
    t0 = time.perf_counter()
    x = 0.0
    for k in range(args.steps):
        x += float(np.sum(np.arange(20000, dtype=np.float64) * (rank + 1)))
    t = time.perf_counter() - t0 + 1e-4
    barrier()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="default", choices=sorted(CONFIGS))
    ap.add_argument("--closure", default="fused", choices=["fused", "notebook"])
    ap.add_argument("--mode", default="auto", choices=["auto", "graph", "eager"],
                    help="timed region: hipGraph replays, eager launches, or both (the faster one is reported)")
    ap.add_argument("--no-graph", action="store_true", help="same as --mode eager")
    ap.add_argument("--instances", type=int, default=1, help="independent fits per GPU, grouped into one hipGraph")
    ap.add_argument("--group", default="both", choices=["both", "native", "graphs"],
                    help="--instances B > 1: 'native' = ONE launch list for the B fits (dip_group.GroupedFits: every launch "
                         "serves all instances), 'graphs' = B hipGraphs on B streams, 'both' = time both, report the faster")
    ap.add_argument("--first-image", type=int, default=0,
                    help="index (= seed) of the first image: `--gpus 1 --first-image r` is the solo run of what rank r of a "
                         "multi-GPU run fits (per_rank_final_loss_hex must match bit for bit: tests/test_shard_gpu.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-eager-line", action="store_true")
    ap.add_argument("--dump-ops", default=None, help="write the per-launch HIP-event table (JSON) here")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_workers(args.gpus, sys.argv[1:]))        # this process only supervises the N workers
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` "
                         "or through torch.distributed.run with --nproc-per-node N")

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")                          # host-side rendezvous only; no RCCL on this path

    def barrier():
        if dist is not None:
            dist.barrier()

    if SELFTEST:
        affinity = pin_to_gpu_numa(local, world)         # (exercised on CPU: no GPU sysfs entry -> left alone)
        assert isinstance(affinity, dict) and "pinned" in affinity
        t = selftest_rank(args, rank, world, barrier)
        per_rank = gather_floats(args.steps / t)
        tmax = reduce_max_time(t)
        if rank == 0:
            print(json.dumps({"metric": "selftest", "value": world * args.steps / tmax, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "per_rank_it_s": per_rank}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    import __graft_entry__ as ge
    ge.build()
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    affinity = pin_to_gpu_numa(local, world)

    # one independent image (x --instances) per rank; image index = global fit index = seed
    n_inst = max(args.instances, 1)
    my_images = [args.first_image + i for i in shard_images(world * n_inst, rank, world)]
    fits = [Fit(args.config, img, dev, args.closure) for img in my_images]
    # Execution mode of the timed region: eager launches or hipGraph replays of the same iteration.
    # `auto` times K steps in each mode and reports the faster one as `value` (the other goes to
    # "other_mode"): at 512x512 the GPU is throughput-bound and the eager two-stream schedule wins, small
    # nets are launch-bound and the graph wins.
    modes = {"graph": [True], "eager": [False], "auto": [False, True]}["eager" if args.no_graph else args.mode]
    # per-launch HIP-event timing for the roofline objects: BEFORE any hipGraph exists in the process (event churn
    # after a captured graph had been torn down aborted 10-25 % of the small-config runs on ROCm 7.2 with glibc
    # heap-corruption errors; in this order: 0 / 60)
    per_op = None
    if rank == 0 and not args.no_roofline:
        for _ in range(max(3, min(args.warmup, 10))):
            fits[0].step()
        torch.cuda.synchronize()
        per_op = profile_ops(fits[0].engine)
    runs = []
    native_ok = n_inst > 1 and args.closure == "fused" and args.config != "sr" and args.group != "graphs"
    if not (native_ok and args.group == "native"):
        for use_graph in modes:
            t, graphed, note, power = timed_run(fits, args.steps, args.warmup, use_graph, barrier)
            runs.append({"t": reduce_max_time(t), "mine": len(fits) * args.steps / t, "graphed": graphed, "note": note,
                         "power": power, "form": "one hipGraph per fit on its own stream" if graphed else
                         "eager launch lists, one fit after the other", "loss": float(fits[0].loss.item())})
    grouped = None
    if native_ok:
        # the same fits through ONE launch list (csrc/dip_group.h): every kernel launch serves all n_inst instances
        try:
            for use_graph in modes:
                grouped = grouped_fits(args.config, my_images, dev)
                t, power = timed_run_grouped(grouped, args.steps, args.warmup, use_graph, barrier)
                runs.append({"t": reduce_max_time(t), "mine": len(fits) * args.steps / t, "graphed": use_graph, "note": None,
                             "power": power, "form": f"grouped: one launch list for the {n_inst} fits"
                             + (", replayed as ONE hipGraph" if use_graph else ", eager launches"),
                             "loss": float(grouped.losses[0].item()), "native_mask": int(grouped.lib.dip_group_native(-1))})
        except Exception as e:
            if args.group == "native" or world > 1:      # (all ranks must time the same forms: no silent per-rank divergence)
                raise
            runs.append({"t": float("inf"), "mine": 0.0, "graphed": False, "power": None, "loss": float("nan"),
                         "note": f"grouped form failed: {type(e).__name__}: {e}", "form": "grouped (failed)"})
            torch.cuda.synchronize()
    runs.sort(key=lambda r: r["t"])
    best = runs[0]
    tmax, graphed, note = best["t"], best["graphed"], best["note"]
    per_rank = gather_floats(best["mine"])
    # (the per-fit forms keep training the same fits across the timed modes: their final loss is read at the end)
    final_loss = best["loss"] if best["form"].startswith("grouped") else float(fits[0].loss.item())
    # every rank's first fit: rank r's image index is r * instances, i.e. "rank r == the solo fit with that seed" can be
    # checked bitwise across runs
    per_rank_loss = gather_floats(final_loss)
    # per-rank power / clock of the reported form's timed region and the host's issue time per iteration: what limits N
    # ranks on one host (8 x ~1.1 kW, eight launching threads) shows up rank by rank
    pw = best.get("power") or {}
    per_rank_power = gather_floats(float(pw.get("power_w_mean") or float("nan")))
    per_rank_sclk = gather_floats(float(pw.get("sclk_mhz_mean") or float("nan")))
    rank_issue = None
    if world > 1 and not best["graphed"] and not best["form"].startswith("grouped"):
        try:
            rank_issue = host_issue(fits).get("command_list_ms_per_iteration")
        except Exception:
            rank_issue = None
    per_rank_issue = gather_floats(float(rank_issue) if rank_issue is not None else float("nan")) if world > 1 else None

    if rank == 0:
        eng = fits[0].engine
        if not hasattr(eng, "fwd_ops") and grouped is not None:
            eng = grouped.eng                  # --group native without the per-launch profile: the per-fit nets never ran
        rl = rw = rh = r3 = rthin = None
        if not args.no_roofline:
            rl, rw = roofline(eng, per_op, with_pmc=(args.config == "default"))
            rh = roofline_hbm(eng, per_op)
            r3 = conv3x3_all(eng, per_op, conv_flops(eng))
            rthin = roofline_thin(eng, per_op, conv_flops(eng))
            if args.dump_ops:
                fl = conv_flops(eng)
                with open(args.dump_ops, "w") as f:
                    json.dump({k: {"ms": v, "gflop": fl.get(k, 0) / 1e9} for k, v in per_op.items()}, f, indent=1)
        eager = None
        if world == 1 and not args.no_eager_line and not (args.closure == "notebook" and not graphed):
            # the notebook's own torch closure, eager launches (what an unmodified notebook cell runs)
            nb = Fit(args.config, 0, dev, "notebook")
            k = max(10, min(args.steps, 30))
            te, _, _, _ = timed_run([nb], k, 3, False, lambda: None)
            eager = {"it_s": round(k / te, 3), "ms_per_step": round(1e3 * te / k, 3), "steps": k,
                     "closure": "notebook torch ops (normal_, MSELoss, out-of-place EMA), eager launches"}
            del nb
        # the same fit with the fp32-MFMA kernels only (DIP_CONV_BF3=0), in a process of its own (the switch is read once)
        fp32_only = None
        import dip_native as _N
        terms = _N.lib().dip_conv_bf3_terms()
        if world == 1 and terms and not args.no_eager_line and os.environ.get("DIP_BENCH_CHILD") is None:
            import subprocess
            env = dict(os.environ, DIP_CONV_BF3="0", DIP_BENCH_CHILD="1")
            cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", str(max(10, min(args.steps, 50))),
                   "--warmup", "5", "--mode", "eager", "--no-cpu-baseline", "--no-roofline", "--no-eager-line"]
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
                ln = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
                if ln:
                    o = json.loads(ln[-1])
                    fp32_only = {"it_s": o["value"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
                                 "what": "DIP_CONV_BF3=0: every convolution on v_mfma_f32_32x32x2_f32 (the round-3 arithmetic), eager launches"}
            except Exception as e:          # the headline does not depend on it
                fp32_only = {"error": str(e)[:200]}
        # a BATCH of independent fits on this GPU through ONE launch list (dip_group.GroupedFits): 4 images of
        # the headline configuration, in a process of its own; reported next to the headline, never as it
        batch4 = None
        if world == 1 and n_inst == 1 and args.config == "default" and not args.no_eager_line \
                and os.environ.get("DIP_BENCH_CHILD") is None:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--instances", "4", "--group", "native", "--mode", "eager",
                   "--steps", str(max(10, min(args.steps, 30))), "--warmup", "5", "--no-cpu-baseline", "--no-roofline",
                   "--no-eager-line"]
            try:
                r = subprocess.run(cmd, env=dict(os.environ, DIP_BENCH_CHILD="1"), capture_output=True, text=True, timeout=300)
                ln = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
                if ln:
                    o = json.loads(ln[-1])
                    batch4 = {"images_per_gpu": 4, "it_s": o["value"], "ms_per_grouped_iteration": o["ms_per_step"],
                              "steps": o["steps"], "form": o["config"].get("reported_mode"),
                              "what": "4 independent 512x512 fits of the headline configuration on this GPU, every kernel launch "
                                      "serving all four (each fit bit-identical to its solo run, tests/test_group_gpu.py); it_s = all "
                                      "four fits' iterations / s"}
                else:
                    batch4 = {"error": (r.stderr.strip().splitlines() or ["no line"])[-1][:200]}
            except Exception as e:          # the headline does not depend on it
                batch4 = {"error": str(e)[:200]}
        # BASELINE.md section 4's own protocol next to the driver's short one: 50 warm-up + 300 timed iterations of the
        # reported mode, power / clock samples of that region (the driver's 20 steps = 0.1 s end before PPT has settled)
        sustained = None
        if world == 1 and n_inst == 1 and not args.no_eager_line and os.environ.get("DIP_BENCH_CHILD") is None \
                and not best["form"].startswith("grouped"):
            ts, sg, _, spower = timed_run(fits, 300, 50, graphed, lambda: None)
            sustained = {"it_s": round(len(fits) * 300 / ts, 3), "ms_per_step": round(1e3 * ts / 300, 3), "steps": 300,
                         "warmup": 50, "hipgraph": sg, "power": spower,
                         "what": "BASELINE.md section 4 protocol: 50 warm-up + 300 timed iterations, same fit, same mode"}
        hissue = None
        if world == 1 and n_inst == 1 and not graphed and os.environ.get("DIP_BENCH_CHILD") is None:
            try:
                hissue = host_issue(fits)
            except Exception as e:          # the headline does not depend on it
                hissue = {"error": str(e)[:200]}
        cb = None if (args.no_cpu_baseline or world > 1 or args.config != "default") else cpu_baseline()
        its = world * len(fits) * args.steps / tmax
        n_launch = count_kernels(eng)
        ko = os.environ.get("DIP_KNOCKOUT") or (("learning rate " + os.environ["DIP_BENCH_LR"]) if "DIP_BENCH_LR" in os.environ else None)
        line = {
            "metric": f"KNOCK-OUT EXPERIMENT ({ko!r}: launches left out / parameters frozen, results wrong): not a measurement of the path" if ko
            else "optimisation iters/sec per image (skip-net 512x512 denoising)" if args.config == "default"
            else f"optimisation iters/sec per image ({args.config} config)",
            "value": round(its, 3), "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * tmax / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": CONFIGS[args.config]["desc"] + ": reg-noise + forward + MSE + backward + fused Adam, "
                                   f"{n_inst} independent image(s) per GPU; value = all images' iterations / s",
                       "images": world * n_inst, "closure": args.closure, "hipgraph": graphed,
                       "kernel_launches_per_iteration": n_launch, "final_loss": round(final_loss, 6),
                       "arithmetic": ("fp32 tensors, fp32 accumulation everywhere.  3x3 stride-1 layers with >= 256 tiles: each fp32 "
                                      f"operand split EXACTLY into three bf16 terms, {terms} of the 9 cross products (each exact in "
                                      "fp32) summed in fp32 on v_mfma_f32_32x32x16_bf16"
                                      + ("" if terms == 9 else " (8: all but lo x lo, which is < 2^-32 of a product = 2^-8 of the "
                                         "rounding error of one fp32 accumulation step; DIP_CONV_BF3=9 adds it back)" if terms == 8
                                         else " (6: the six largest)")
                                      + " -- per-op error vs fp64 <= the fp32 MFMA's (tests/test_bf3_gpu.py); all other layers: "
                                        "v_mfma_f32_32x32x2_f32") if terms else
                                     "fp32 everywhere (v_mfma_f32_32x32x2_f32)"},
            "per_rank_it_s": [round(v, 3) for v in per_rank],
            "per_rank_final_loss": [round(v, 6) for v in per_rank_loss],
            "per_rank_final_loss_hex": [float(v).hex() for v in per_rank_loss],
            "timed_region_power": best.get("power"),
            "roofline": rl, "roofline_wgrad": rw, "roofline_conv3x3_all": r3, "roofline_thin": rthin, "roofline_hbm": rh,
            "sustained": sustained, "host_issue": hissue, "build_id": _N.lib().dip_build_id().decode(),
            "per_rank_power_w": [None if v != v else round(v, 1) for v in per_rank_power],
            "per_rank_sclk_mhz": [None if v != v else round(v) for v in per_rank_sclk],
            "per_rank_host_issue_ms": None if per_rank_issue is None else [None if v != v else round(v, 3) for v in per_rank_issue],
            "cpu_baseline": cb, "eager_notebook": eager, "fp32_mfma_only": fp32_only, "grouped_batch_of_4": batch4,
            "device": device_info(local),
            "host_affinity_rank0": affinity,
        }
        line["config"]["reported_mode"] = ("hipGraph replays" if graphed else "eager launches (main + side + bulk HIP stream)") + \
            " of the iteration with the fused closure (RegNoise + MSEHead + in-place EMA)"
        if n_inst > 1:
            line["config"]["reported_mode"] = best["form"] + "; fused closure (RegNoise + MSEHead + in-place EMA)"
            line["config"]["grouped_native_families_mask"] = best.get("native_mask")
        if len(runs) > 1:
            line["other_mode"] = [{"form": o["form"], "hipgraph": o["graphed"],
                                   "it_s": round(world * len(fits) * args.steps / o["t"], 3) if o["t"] != float("inf") else None,
                                   "ms_per_step": round(1e3 * o["t"] / args.steps, 3) if o["t"] != float("inf") else None,
                                   "note": o.get("note")} for o in runs[1:]]
            if len(runs) == 2:                   # (the single-fit line keeps its round-3 shape: one object)
                line["other_mode"] = line["other_mode"][0]
        if note:
            line["config"]["note"] = note
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
