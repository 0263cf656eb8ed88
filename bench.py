"""Benchmark of the deep-image-prior hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one optimisation iteration (reg-noise perturbation, skip-net forward, MSE, backward,
fused Adam) of ONE 512x512 denoising fit with the default net
  get_net(32,'skip','reflection',skip_n33d=128,skip_n33u=128,skip_n11=4,num_scales=5,'bilinear')
(BASELINE.json configs[1]/[4] at the size the metric is quoted on; SURVEY.md section 8d "M1").
With N GPUs every rank optimises its own independent image (no collective on the data path:
train-mode BatchNorm forbids batching images, so images shard one-per-GPU) -> "scaling": "weak",
value = N * K / max-over-ranks time.

The JSON line also carries
  roofline     : the dominant kernel (3x3 stride-1 implicit-GEMM conv, 128-wide N block: forward
                 and data-gradient launches) -- algorithmic FLOPs / HIP-event time, vs the
                 157.3 TFLOP/s fp32 MFMA peak (MI355X_MICROARCH.md chip table);
  cpu_baseline : the CPU oracle (oracle/dip_oracle.py, a bitwise-verified restatement of the
                 reference's PyTorch-CPU path) timed on this box's host cores on the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
SIZE = 512


def shard_images(n_images: int, rank: int, world: int):
    """Static round-robin partition of independent image fits over ranks (no data exchange)."""
    return [i for i in range(n_images) if i % world == rank]


def reduce_max_time(t: float, device) -> float:
    """max over ranks of a wall-time (the only collective in the benchmark; not on the data path)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return t
    x = torch.tensor([t], dtype=torch.float64, device=device)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return float(x.item())


def make_problem(seed: int, size: int = SIZE):
    """Synthetic denoising problem of the reference's shape: clean = 5x5 box-blur
    of U(0,1) noise, target = clip(clean + N(0,(25/255)^2)), z = get_noise(32,'noise') ~ U(0,0.1)."""
    from utils.common_utils import get_noise
    torch.manual_seed(seed)
    np.random.seed(seed)
    z = get_noise(32, 'noise', (size, size))
    clean = torch.nn.functional.avg_pool2d(torch.rand(1, 3, size + 4, size + 4), 5, stride=1)
    noisy = np.clip(clean.numpy() + np.random.normal(scale=25 / 255., size=clean.shape), 0, 1).astype(np.float32)
    return z, torch.from_numpy(noisy)


def build_fit(seed: int, dev, size: int = SIZE):
    from models import get_net
    torch.manual_seed(seed)
    net = get_net(32, 'skip', 'reflection', skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                  upsample_mode='bilinear').to(dev)
    z, target = make_problem(seed, size)
    return net, z.to(dev), target.to(dev)


def make_closure(net, z, target, reg_noise_std=1. / 30., exp_weight=0.99):
    """The denoising notebook's closure (reference denoising.ipynb:204-221) minus the per-iteration
    host syncs (PSNR prints / plots / CPU parameter snapshots), which are not the hot path."""
    mse = torch.nn.MSELoss()
    st = {"saved": z.detach().clone(), "noise": z.detach().clone(), "avg": None, "loss": None, "i": 0}

    def closure():
        net_input = st["saved"] + (st["noise"].normal_() * reg_noise_std)
        out = net(net_input)
        st["avg"] = out.detach() if st["avg"] is None else st["avg"] * exp_weight + out.detach() * (1 - exp_weight)
        total_loss = mse(out, target)
        total_loss.backward()
        st["loss"] = total_loss.detach()
        st["i"] += 1
        return total_loss

    return closure, st


def conv_flops(eng):
    """Algorithmic conv FLOPs per iteration, per op name: 2*Cout*Ho*Wo*Cin*k*k for forward,
    weight-gradient and (where the input needs it) data-gradient."""
    fl = {}

    def dims(r, H, W):
        Ho, Wo = (H + 2 * r.P - r.ks) // r.stride + 1, (W + 2 * r.P - r.ks) // r.stride + 1
        return 2.0 * r.Cout * Ho * Wo * r.Cin * r.ks * r.ks

    for i, s in enumerate(eng.sc):
        st = s.st
        H, W = st["H"], st["W"]
        for attr, (h, w) in (("skip_conv", (H, W)), ("down_a", (H, W)), ("down_b", (H // 2, W // 2)), ("up", (H, W)),
                             ("up1", (H, W))):
            r = getattr(s, attr)
            if r is not None:
                f = dims(r, h, w)
                fl["conv_fwd:" + r.name] = f
                fl["wgrad:" + r.name] = f
                fl["dgrad:" + r.name] = f
                fl["dgrad+:" + r.name] = f
    r = eng.out_conv
    f = dims(r, eng.H, eng.W)
    fl["conv_fwd:out"] = fl["wgrad:out"] = fl["dgrad:out"] = f
    return fl


def profile_ops(eng, reps=3):
    """HIP-event time of every launch of one iteration, in sequence on the engine's own stream
    (torch's current stream), averaged over `reps` instrumented iterations."""
    import ctypes as C
    import dip_native as N
    lib = N.lib()
    for f in (lib.dip_conv_thin4, lib.dip_conv_igemm_dma_cols):      # internal entry points of the dispatcher
        f.restype, f.argtypes = C.c_int, [C.POINTER(N.DipConvDesc), C.c_int, C.c_void_p]
    stream = torch.cuda.current_stream(eng.device)
    sptr = stream.cuda_stream
    acc = {}
    for rep in range(reps):
        for ops in (eng.fwd_ops, eng.bwd_ops):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
            evs[0].record(stream)
            mids = {}
            for k, (fn, args, name) in enumerate(ops):
                if fn is lib.dip_conv_igemm and lib.dip_conv_variant(args[0]) == 3:
                    # 132-column data gradient = conv_thin4 + a 128-column launch of the dominant kernel
                    ncols = args[0]._obj.Cout - 128
                    lib.dip_conv_thin4(args[0], ncols, sptr)
                    mids[k] = torch.cuda.Event(enable_timing=True)
                    mids[k].record(stream)
                    lib.dip_conv_igemm_dma_cols(args[0], ncols, sptr)
                else:
                    fn(*args, sptr)
                evs[k + 1].record(stream)
            torch.cuda.synchronize()
            for k, (_, _, name) in enumerate(ops):
                acc.setdefault(name, []).append(evs[k].elapsed_time(evs[k + 1]))
                if k in mids:
                    acc.setdefault(name + "#thin4", []).append(evs[k].elapsed_time(mids[k]))
                    acc.setdefault(name + "#dma", []).append(mids[k].elapsed_time(evs[k + 1]))
    return {k: float(np.mean(v)) for k, v in acc.items()}


DOMINANT = "conv_igemm_dma_kernel<3,128,*>"


def dominant_ops(eng):
    """Launch-list entries that run the dominant kernel, as {timing key: (flops, compulsory bytes)}:
    3x3 convolutions (forward and data gradient) that dip_conv_igemm dispatches to the LDS-DMA
    implicit-GEMM kernel with a 128-column tile -- dip_conv_variant == 1, and the 128-column part
    ("#dma", timed on its own by profile_ops) of the 132-column data gradients (variant 3).  The
    N = 160 split-K variant of conv_igemm_kernel, the stride-2 forwards and conv_thin4 are other
    kernels and are not counted."""
    import dip_native as N
    lib = N.lib()
    out = {}
    for ops in (eng.fwd_ops, eng.bwd_ops):
        for fn, args, name in ops:
            if name.partition(":")[0] not in ("conv_fwd", "dgrad", "dgrad+"):
                continue
            d = args[0]._obj
            if d.ks != 3 or d.Cout < 128:
                continue
            v = lib.dip_conv_variant(args[0])
            if v not in (1, 3):
                continue
            cols = d.Cout if v == 1 else 128
            flops = 2.0 * cols * d.Hout * d.Wout * d.Cin * 9
            # compulsory traffic of the launch: input and packed weights read once, output written once
            nbytes = 4.0 * (d.Hin * d.Win * d.Cin + 9 * d.Cin * cols + d.Hout * d.Wout * cols)
            out[name if v == 1 else name + "#dma"] = (flops, nbytes)
    return out


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_traffic.json, produced by tools/pmc_traffic.py); None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            t = json.load(f)
        return t if t.get("kernel") == DOMINANT else None
    except (OSError, ValueError):
        return None


def roofline(eng, per_op_ms):
    """Dominant kernel = conv_igemm_dma_kernel<3,128,*> (all its launches of one iteration, see
    dominant_ops).  achieved = their algorithmic FLOPs (2*Cout*Ho*Wo*Cin*9 per launch)
    / their HIP-event time on the engine's stream."""
    fl = conv_flops(eng)
    dom = dominant_ops(eng)
    tot_f = sum(f for f, _ in dom.values())
    alg_bytes = sum(b for _, b in dom.values())
    tot_ms = sum(per_op_ms[k] for k in dom)
    n = len(dom)
    ach = tot_f / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    big = per_op_ms.get("conv_fwd:s0.up")
    # every MFMA conv launch (forward, data and weight gradient, all kernels) for the whole-path figure
    all_f = sum(f for k, f in fl.items() if k in per_op_ms)
    all_ms = sum(ms for k, ms in per_op_ms.items() if k in fl)          # ("#thin4"/"#dma" parts are not in fl)
    pmc = pmc_traffic()
    return {"bound": "mfma", "kernel": DOMINANT + " (3x3 stride-1 forward + 3x3 data-gradient launches)",
            "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
            "traffic": pmc["traffic_bytes_per_launch"] if pmc else None,
            "launches_per_step": n, "avg_launch_us": round(1e3 * tot_ms / max(n, 1), 1),
            "algorithmic_gflop_per_launch": round(tot_f / 1e9 / max(n, 1), 2),
            "algorithmic_bytes_per_launch": round(alg_bytes / max(n, 1)),
            "measured_mfma_ceiling_tflops": 151.9,      # tools/ubench/mfma_peak.hip on this chip (2 waves/SIMD)
            "largest_layer": {"name": "3.1 (132->128 3x3 @512^2) forward", "gflop": round(fl["conv_fwd:s0.up"] / 1e9, 2),
                              "us": round(1e3 * big, 1) if big else None,
                              "tflops": round(fl["conv_fwd:s0.up"] / (big * 1e-3) / 1e12, 2) if big else None},
            "all_conv_launches": {"gflop_per_step": round(all_f / 1e9, 1), "ms_per_step_serial": round(all_ms, 3),
                                  "tflops": round(all_f / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else None}}


def cpu_baseline(seed=0, budget_s=25.0):
    """The CPU oracle on this box's host cores: same net, same 512x512 workload, same closure;
    1 warm-up + up to 2 timed iterations (bounded: ~10 s each on 8 cores)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dip_oracle as O
    from models import get_net
    # torch-CPU conv scaling collapses on many-core hosts: on the 256-thread MI355X host a sweep
    # (tests/cpu_sweep.py, 256x256) gave 3.47 / 2.38 / 1.25 / 0.59 / 0.011 it/s at 16 / 32 / 64 /
    # 128 / 256 threads, so the baseline uses the best setting, 16 threads, not all of them.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    torch.manual_seed(seed)
    net = get_net(32, 'skip', 'reflection', skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                  upsample_mode='bilinear')
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in O.param_shapes(O.default_spec())}
    onet = O.OracleNet(O.default_spec(), sd)
    z, target = make_problem(seed)
    closure, st = make_closure(onet, z, target)
    opt = torch.optim.Adam(onet.params, lr=0.01)
    times = []
    t_start = time.time()
    for it in range(3):
        t0 = time.time()
        opt.zero_grad()
        closure()
        opt.step()
        times.append(time.time() - t0)
        if it >= 1 and time.time() - t_start > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    return {"value": round(1.0 / float(np.median(timed)), 4), "unit": "it/s", "cores": cores, "kind": "port",
            "sample": f"default skip-net 512x512, 1 warm-up + {len(timed)} timed Adam iterations of the CPU oracle "
                      f"(torch {torch.__version__} CPU, {cores} threads), median"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-ops", default=None, help="write the per-launch HIP-event table (JSON) here")
    args = ap.parse_args()

    ge.build()
    from utils.common_utils import get_params
    from dip_optim import FusedAdam

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    my_images = shard_images(world, rank, world)            # one independent fit per rank
    fits = []
    for img in my_images:
        net, z, target = build_fit(img, dev)
        closure, st = make_closure(net, z, target)
        opt = FusedAdam(get_params('net', net, z), lr=0.01)   # == optimize('adam', ...) unrolled for timing
        fits.append((net, closure, st, opt))

    def step():
        for net, closure, st, opt in fits:
            opt.zero_grad()
            closure()
            opt.step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    t = reduce_max_time(t, dev)
    final_loss = float(fits[0][2]["loss"].item())

    if rank == 0:
        eng = fits[0][0].__dict__["_dip_engine"]
        rl = None
        if not args.no_roofline:
            per_op = profile_ops(eng)
            rl = roofline(eng, per_op)
            if args.dump_ops:
                fl = conv_flops(eng)
                with open(args.dump_ops, "w") as f:
                    json.dump({k: {"ms": v, "gflop": fl.get(k, 0) / 1e9} for k, v in per_op.items()}, f, indent=1)
        cb = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()
        its = world * len(my_images) * args.steps / t
        line = {
            "metric": "optimisation iters/sec per image (skip-net 512x512 denoising)", "value": round(its, 3),
            "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * t / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "default skip-net (2 217 831 params) 512x512 denoising fit: reg-noise + forward "
                                   "+ MSE + backward + fused Adam, one independent image per GPU",
                       "images": world, "final_loss": round(final_loss, 6)},
            "roofline": rl, "cpu_baseline": cb,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
