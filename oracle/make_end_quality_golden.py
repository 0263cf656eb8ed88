"""CPU arms of the BASELINE.json configs[1] end-quality check, produced by the REAL reference.

    python oracle/make_end_quality_golden.py [--task sr|inpaint] [--nomkldnn] [--reg0] <size> <iters> <threads>[:<perturb>[:<grad_noise>]] [...]

--task sr / inpaint (BASELINE.json configs[2] / [3]): the super-resolution closure (super-resolution.ipynb:169-199: the
loss goes through the reference's Downsampler, PSNR on the full-resolution output) and the masked closure
(inpainting.ipynb:295-313, the 'kate' net) instead of the denoising one; file tests/golden/end_quality_<task>_<size>_<iters>.json.

For every arm -- a thread count (= another summation order inside ATen's reductions, nothing else) and an
optional one-ulp perturbation of ONE weight (tests/end_quality_cpu.perturb_one_weight) and an optional relative
perturbation of EVERY gradient element at every step (grad_noise = 1e-6: what another, equally correct fp32
summation order does to a gradient; tests/end_quality_cpu.GRAD_NOISE) -- it runs the
reference's own `get_net` (models/__init__.py:8) + `get_noise` (utils/common_utils.py:127) +
`optimize('adam', ...)` (utils/common_utils.py:198-232) on the denoising notebook's closure
(denoising.ipynb:204-221: reg-noise, forward, EMA of the output, MSE, backward) at <size>^2 for
<iters> iterations on torch CPU fp32, and appends {"psnr_gt", "psnr_gt_sm", "loss", "threads", "sec"}
to tests/golden/end_quality_<size>_<iters>.json.  The problem (clean / noisy image), the reg-noise
generator and the PSNR definition are shared with the GPU arm (tests/end_quality_cpu.py), so the
GPU test only has to run the HIP fit and compare with the committed arms.

Build container only (needs /root/reference); ~12 minutes per arm at 256^2 x 1800 on 8 cores.
Test infrastructure only."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _refload  # noqa: E402


def main():
    task = "denoise"
    if sys.argv[1] == "--task":
        task = sys.argv[2]
        del sys.argv[1:3]
    nomkldnn = False
    if sys.argv[1] == "--nomkldnn":           # bisect arms: the reference with oneDNN switched off (ATen's im2col + GEMM convolutions), file ..._nomkldnn.json
        nomkldnn = True
        torch._C._set_mkldnn_enabled(False)
        del sys.argv[1]
    reg0 = False
    if sys.argv[1] == "--reg0":               # bisect arms: the same fits without the reg-noise path (file ..._reg0.json)
        reg0 = True
        del sys.argv[1]
    size, iters = int(sys.argv[1]), int(sys.argv[2])
    specs = []
    for t in sys.argv[3:] or [str(os.cpu_count())]:
        f = (t.split(":") + ["0", "0"])[:3]
        specs.append((int(f[0]), int(f[1]), float(f[2])))
    assert _refload.available(), "needs the reference checkout"
    RM = _refload.load_ref_models()
    RU = _refload.load_ref_common_utils()
    import end_quality_cpu as E     # problem(), run_fit(): shared with the GPU arm
    assert task in E.TASKS, task
    if reg0:
        E.REG_SCALE = 0.0
    path = os.path.join(ROOT, "tests", "golden", (f"end_quality_{size}_{iters}" if task == "denoise" else
                        f"end_quality_{task}_{size}_{iters}") + ("_reg0" if reg0 else "") + ("_nomkldnn" if nomkldnn else "") + ".json")
    arms = json.load(open(path))["cpu_arms"] if os.path.exists(path) else []
    for th, perturb, gnoise in specs:
        torch.set_num_threads(th)
        E.GRAD_NOISE = gnoise
        net, z = E.build(size, task, skip_fn=RM.skip, get_net_fn=RM.get_net, get_noise_fn=RU.get_noise)
        E.perturb_one_weight(net.parameters(), perturb)
        clean, noisy = E.problem(size, task)
        down = None
        if task == "sr":
            down = RM.downsampler.Downsampler(n_planes=3, factor=E.SR_FACTOR, kernel_type='lanczos2', phase=0.5, preserve_size=True)
        res = E.run_fit(net, lambda c: RU.optimize('adam', RU.get_params('net', net, z), c, 0.01, iters),
                        z, noisy, clean, iters, "cpu", params=list(net.parameters()), task=task, down=down)
        res["threads"], res["perturb"], res["grad_noise"] = th, perturb, gnoise
        print(json.dumps(res), flush=True)
        arms = json.load(open(path))["cpu_arms"] if os.path.exists(path) else arms     # (another generator may run)
        arms = [a for a in arms if (a["threads"], a.get("perturb", 0), a.get("grad_noise", 0.0)) != (th, perturb, gnoise)] + [res]
        with open(path, "w") as f:
            json.dump({"task": task, "size": size, "iters": iters, "sigma": E.SIGMA, "reg_noise_std": E.REG_OF[task], "lr": 0.01,
                       "source": "real reference (/root/reference) on torch CPU fp32, oracle/make_end_quality_golden.py",
                       "cpu_arms": sorted(arms, key=lambda a: (a.get("grad_noise", 0.0), a.get("perturb", 0), a["threads"]))},
                      f, indent=1)


if __name__ == "__main__":
    main()
