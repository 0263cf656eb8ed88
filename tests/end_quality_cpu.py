"""CPU arm of tests/test_net_gpu.py::test_end_quality_default_net_128: the denoising notebook's
closure (denoising.ipynb:204-221 of the reference) on the CPU oracle with a given thread count.

    python tests/end_quality_cpu.py <threads> <iters> <out.json> [<size> [<perturb>]]

<perturb> = k > 0 multiplies ONE weight (element 0 of the k-th parameter tensor) by 1 + 2^-22 before the fit:
trajectories are chaotic (a 1e-7 relative change of one weight decorrelates the outputs within 5
iterations), so such arms sample the run-to-run spread of the end quality much better than thread counts do (which
leave the convolutions' summation order untouched).

Prints/writes {"psnr_gt": tail-averaged PSNR vs the clean image, "psnr_gt_sm": PSNR of the EMA
output, "loss": final loss, "sec": wall time}.  Test infrastructure only."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge  # noqa: E402

ge.add_to_path()
import dip_oracle as O  # noqa: E402

SIZE = 128
SIGMA = 25 / 255.
REG = 1. / 30.
TAIL = 20


TASKS = ("denoise", "sr", "inpaint")
SR_FACTOR = 4
REG_OF = {"denoise": REG, "sr": 0.03, "inpaint": 0.03}       # reg_noise_std of the three notebooks


def problem(size=None, task="denoise"):
    """Clean image: smooth colour gradients + one step edge.  Returns (clean, target):
       denoise  target = clip(clean + N(0, sigma^2))                                        (denoising.ipynb)
       sr       target = the x4 smaller image (4x4 box means of the clean one)               (super-resolution.ipynb: imgs['LR_np'])
       inpaint  target = (clean, mask): mask = 0 on three bars and a dozen small squares     (inpainting.ipynb: img_mask)"""
    size = size or SIZE
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[0:size, 0:size] / float(size)
    clean = np.stack([0.5 + 0.4 * np.sin(6 * xx) * np.cos(4 * yy), 0.5 + 0.4 * np.cos(5 * xx + 2 * yy),
                      0.3 + 0.5 * (xx > 0.5)]).astype(np.float32)
    if task == "denoise":
        noisy = np.clip(clean + rng.normal(scale=SIGMA, size=clean.shape), 0, 1).astype(np.float32)
        return clean, noisy
    if task == "sr":
        f = SR_FACTOR
        lr = clean.reshape(3, size // f, f, size // f, f).mean(axis=(2, 4)).astype(np.float32)
        return clean, lr
    assert task == "inpaint", task
    mask = np.ones((size, size), np.float32)
    w = max(size // 32, 2)
    mask[size // 5:size // 5 + w, :] = 0
    mask[:, size // 3:size // 3 + w] = 0
    for k in range(size):                                    # a diagonal bar
        mask[k, max(0, min(size - 1, size - 1 - k)):max(0, min(size, size - 1 - k + w))] = 0
    for _ in range(12):
        y0, x0 = rng.randint(0, size - 2 * w, size=2)
        mask[y0:y0 + 2 * w, x0:x0 + 2 * w] = 0
    return clean, (clean, np.broadcast_to(mask, clean.shape).copy())


def build(size=None, task="denoise", skip_fn=None, get_net_fn=None, get_noise_fn=None):
    """Net + z exactly as the notebooks build them (torch.manual_seed(0)): the default net for denoising and
    super-resolution (denoising.ipynb:160-173, super-resolution.ipynb:133-140), the 'kate' net for inpainting
    (inpainting.ipynb:150-164).  The constructors default to this package's; oracle/make_end_quality_golden.py passes the
    REAL reference's."""
    size = size or SIZE
    if get_net_fn is None:
        from models import get_net as get_net_fn
    if skip_fn is None:
        from models.skip import skip as skip_fn
    if get_noise_fn is None:
        from utils.common_utils import get_noise as get_noise_fn
    torch.manual_seed(0)
    if task == "inpaint":
        net = skip_fn(32, 3, num_channels_down=[128] * 5, num_channels_up=[128] * 5, num_channels_skip=[128] * 5,
                      filter_size_up=3, filter_size_down=3, upsample_mode='nearest', filter_skip_size=1,
                      need_sigmoid=True, need_bias=True, pad='reflection', act_fun='LeakyReLU')
    else:
        net = get_net_fn(32, 'skip', 'reflection', skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5,
                         upsample_mode='bilinear')
    z = get_noise_fn(32, 'noise', (size, size))
    return net, z


GRAD_NOISE = float(os.environ.get("EQ_GRAD_NOISE", "0"))     # relative gradient perturbation (tools: noise-floor experiment)
REG_SCALE = float(os.environ.get("EQ_REG_SCALE", "1"))       # bisect arms: 0 = the fit without the reg-noise path
DEVICE_NOISE = os.environ.get("EQ_DEVICE_NOISE", "0") == "1"  # reg-noise from the DEVICE generator (HIP-vs-HIP arms at 512^2: a host
#                                                               draw + copy of 33.5 MB per iteration would dominate the fit)
CURVE_EVERY = 50                                             # loss(t): mean over each window of 50 iterations


def run_fit(net_call, params_step, z, noisy, clean, iters, device, exp_weight=0.99, params=None, task="denoise", down=None):
    """The notebook closure with the reg-noise drawn from a host generator (same perturbations in
    every arm).  net_call(x) -> out; params_step(closure) runs the optimisation loop.
       denoise  loss = mse(out, noisy)                                  denoising.ipynb:204-221
       sr       loss = mse(down(out_HR), img_LR), PSNR on out_HR        super-resolution.ipynb:169-199 (tv_weight = 0); `down` =
                Downsampler(n_planes=3, factor=4, kernel_type='lanczos2', phase=0.5, preserve_size=True)
       inpaint  loss = mse(out * mask, img * mask), PSNR on the whole image (holes included)   inpainting.ipynb:295-313
    The EMA of the output (denoising.ipynb:213-217) is kept for all three as a second, less jittery read-out."""
    gen = torch.Generator().manual_seed(77)
    ngen = torch.Generator().manual_seed(78)
    zt = z.to(device)
    mask = None
    if task == "inpaint":
        img, m = noisy
        tgt, mask = torch.from_numpy(img)[None].to(device), torch.from_numpy(m)[None].to(device)
    else:
        tgt = torch.from_numpy(noisy)[None].to(device)
    mse = torch.nn.MSELoss()
    st = {"i": 0, "avg": None, "loss": None, "tail": [], "ltail": [], "curve": []}
    reg = REG_OF[task] * REG_SCALE

    dgen = torch.Generator(device=device).manual_seed(77) if (DEVICE_NOISE and str(device) != "cpu") else None
    # GPU arms: the host generator's draws (the SAME sequence as in the CPU arms) are produced a few iterations ahead by a
    # thread of their own (torch.randn releases the GIL) instead of in the closure: 3 .. 10 ms per iteration off a 36 ms one
    ahead = None
    if dgen is None and str(device) != "cpu":
        import queue
        import threading
        ahead = queue.Queue(maxsize=8)

        def produce():
            for _ in range(iters):
                ahead.put((torch.randn(z.shape, generator=gen) * reg).pin_memory())
        threading.Thread(target=produce, daemon=True).start()

    def closure():
        if dgen is not None:
            noise = torch.randn(z.shape, generator=dgen, device=device) * reg
        elif ahead is not None:
            noise = ahead.get()
        else:
            noise = torch.randn(z.shape, generator=gen) * reg
        out = net_call(zt + noise.to(device))
        st["avg"] = out.detach() if st["avg"] is None else st["avg"] * exp_weight + out.detach() * (1 - exp_weight)
        if task == "sr":
            loss = mse(down(out), tgt)
        elif task == "inpaint":
            loss = mse(out * mask, tgt * mask)
        else:
            loss = mse(out, tgt)
        loss.backward()
        if GRAD_NOISE > 0 and params is not None:     # diagnostic: a multiplicative roundoff-like error on every gradient element
            for p in params:
                if p.grad is not None:
                    p.grad.mul_(1.0 + GRAD_NOISE * torch.randn(p.grad.shape, generator=ngen).to(p.grad.device))
        st["i"] += 1
        st["loss"] = loss.detach()
        st["curve"].append(st["loss"])          # device scalars, read back once after the fit (no per-iteration sync)
        if st["i"] > iters - TAIL:              # single-iteration PSNR jitters by ~1 dB: average the tail
            st["tail"].append(O.psnr(clean, out.detach().cpu().numpy()[0]))
            st["ltail"].append(float(loss.detach().item()))
        return loss

    t0 = time.time()
    params_step(closure)
    # loss_tail: the SR / inpainting fits end at ~1e-4, where the loss of ONE iteration is reg-noise jitter (+-30 %)
    sec = time.time() - t0
    lc = torch.stack([c.reshape(()) for c in st["curve"]]).double().cpu().numpy()
    curve = [float(lc[k:k + CURVE_EVERY].mean()) for k in range(0, len(lc) - CURVE_EVERY + 1, CURVE_EVERY)]
    return {"psnr_gt": float(np.mean(st["tail"])), "psnr_gt_sm": O.psnr(clean, st["avg"].cpu().numpy()[0]),
            "loss": float(st["loss"].item()), "loss_tail": float(np.mean(st["ltail"])),
            "loss_last100": float(lc[-100:].mean()), "loss_curve": curve, "reg_scale": REG_SCALE, "sec": sec}


def perturb_one_weight(params, k):
    """k > 0: element 0 of the k-th parameter tensor *= 1 + 2^-22 (one or two ulps)."""
    if k > 0:
        with torch.no_grad():
            p = list(params)[k]
            p.view(-1)[0] *= 1.0 + 2.0 ** -22


def main():
    threads, iters, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    size = int(sys.argv[4]) if len(sys.argv) > 4 else SIZE
    perturb = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    torch.set_num_threads(threads)
    clean, noisy = problem(size)
    net, z = build(size)
    perturb_one_weight(net.parameters(), perturb)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in O.param_shapes(O.default_spec())}
    onet = O.OracleNet(O.default_spec(), sd)
    res = run_fit(onet, lambda c: O.optimize_adam(onet.params, c, 0.01, iters), z, noisy, clean, iters, "cpu",
                  params=list(onet.params))
    res["grad_noise"] = GRAD_NOISE
    res["threads"], res["perturb"] = threads, perturb
    with open(out, "w") as f:
        json.dump(res, f)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
