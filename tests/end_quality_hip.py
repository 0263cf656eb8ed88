"""HIP arm of the end-quality tests (tests/test_net_gpu.py): the denoising notebook's closure
(denoising.ipynb:204-221 of the reference) on the MI355X backend, in a process of its own so that the
environment switches that change the summation order of the kernels (read once per process:
DIP_TWO_STREAMS, DIP_CONV_PLAN_WGS, DIP_WGRAD_NO_SLIDE, DIP_CONV_NO_DMA ...) can differ between arms.

    python tests/end_quality_hip.py <size> <iters> <out.json> [<perturb> [<task>]]

<task> = denoise (default) | sr | inpaint: the closures of the three notebooks (tests/end_quality_cpu.run_fit).

Same problem, reg-noise generator and PSNR definition as the CPU arm (tests/end_quality_cpu.py).
Test infrastructure only."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
import end_quality_cpu as E  # noqa: E402


def main():
    size, iters, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    from utils.common_utils import get_params, optimize
    dev = torch.device("cuda:0")
    task = sys.argv[5] if len(sys.argv) > 5 else "denoise"
    clean, noisy = E.problem(size, task)
    net, z = E.build(size, task)
    E.perturb_one_weight(net.parameters(), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    net = net.to(dev)
    down = None
    if task == "sr":
        from models.downsampler import Downsampler
        down = Downsampler(n_planes=3, factor=E.SR_FACTOR, kernel_type='lanczos2', phase=0.5, preserve_size=True).to(dev)
    res = E.run_fit(net, lambda c: optimize("adam", get_params("net", net, None), c, 0.01, iters), z, noisy, clean,
                    iters, dev, task=task, down=down)
    res["task"] = task
    res["env"] = {k: v for k, v in os.environ.items() if k.startswith("DIP_")}
    with open(out, "w") as f:
        json.dump(res, f)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
