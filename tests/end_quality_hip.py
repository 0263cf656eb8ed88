"""HIP arm of the end-quality tests (tests/test_net_gpu.py): the denoising notebook's closure
(denoising.ipynb:204-221 of the reference) on the MI355X backend, in a process of its own so that the
environment switches that change the summation order of the kernels (read once per process:
DIP_TWO_STREAMS, DIP_CONV_PLAN_WGS, DIP_WGRAD_NO_SLIDE, DIP_CONV_NO_DMA ...) can differ between arms.

    python tests/end_quality_hip.py <size> <iters> <out.json> [<perturb>[,<perturb>...] [<task> [<family>]]]

<task> = denoise (default) | sr | inpaint: the closures of the three notebooks (tests/end_quality_cpu.run_fit).
<family> = hip (default: this package's net, optimize() and Downsampler) or one of the bisect families of DESIGN.md
section 4 (round 5), all on the same GPU with the same problem, reg-noise stream and read-outs:
    torch          the ORACLE's net (oracle/dip_oracle.py: plain torch.nn.functional) on the GPU through torch-ROCm's own
                   kernels with MIOpen switched off (torch.backends.cudnn.enabled = False: im2col + rocBLAS GEMMs, the
                   reference README's own advice for GPUs on which the method misbehaves) + torch.optim.Adam -- the
                   reference's arithmetic on a third, independent backend;
    hip_torchloss  this package's net and optimiser, the loss side (Downsampler / mask / MSE) in torch ops: separates
                   dip_lanczos_down_* from the net (sr only);
    hip_torchadam  this package's net, torch.optim.Adam on its parameters instead of the fused arena step.
EQ_REG_SCALE=0 in the environment runs any family without the reg-noise path.

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

# the host side of an arm is a few small torch CPU ops per iteration (reg-noise draw, scaling): on the 128-core GPU boxes the
# default intra-op pool (one thread per core) makes each of them a many-thread barrier.  The values do not depend on it (the
# normal draw is sequential in the generator, the scaling element-wise).
torch.set_num_threads(4)


def one_fit(size, iters, perturb, task, family):
    from utils.common_utils import get_params, optimize
    dev = torch.device("cuda:0")
    clean, noisy = E.problem(size, task)
    net, z = E.build(size, task)
    E.perturb_one_weight(net.parameters(), perturb)
    down = None

    def torch_down():
        """The reference's Downsampler as torch ops on the device (models/downsampler.py:44-71: ReplicationPad2d + a dense
        Conv2d with the taps on the channel diagonal, zero bias)."""
        import dip_oracle as O
        k = O.lanczos_kernel(E.SR_FACTOR, 0.5, 4 * E.SR_FACTOR + 1, 2)
        w = torch.zeros(3, 3, *k.shape)
        for c in range(3):
            w[c, c] = torch.from_numpy(k).float()
        w, b, pad = w.to(dev), torch.zeros(3, device=dev), int((k.shape[0] - E.SR_FACTOR) / 2.0)
        return lambda x: torch.nn.functional.conv2d(torch.nn.functional.pad(x, (pad,) * 4, mode="replicate"), w, b,
                                                    stride=E.SR_FACTOR)

    if family == "torch":
        import dip_oracle as O
        torch.backends.cudnn.enabled = False          # no MIOpen (it would JIT-compile ~70 kernels on a box without a kernel db)
        spec = O.default_spec() if task != "inpaint" else O.SkipSpec(32, 3, [128] * 5, [128] * 5, [128] * 5, pad="reflection",
                                                                      upsample_mode="nearest")
        sd = {k: v.detach().clone().to(dev) for k, v in net.state_dict().items() if k in O.param_shapes(spec)}
        onet = O.OracleNet(spec, sd).to(dev)
        if task == "sr":
            down = torch_down()
        res = E.run_fit(onet, lambda c: O.optimize_adam(onet.params, c, 0.01, iters), z, noisy, clean, iters, dev,
                        params=list(onet.params), task=task, down=down)
    else:
        net = net.to(dev)
        if task == "sr":
            if family == "hip_torchloss":
                down = torch_down()
            else:
                from models.downsampler import Downsampler
                down = Downsampler(n_planes=3, factor=E.SR_FACTOR, kernel_type='lanczos2', phase=0.5, preserve_size=True).to(dev)
        if family == "hip_torchadam":
            import dip_oracle as O
            step = lambda c: O.optimize_adam(list(net.parameters()), c, 0.01, iters)
        else:
            assert family in ("hip", "hip_torchloss"), family
            step = lambda c: optimize("adam", get_params("net", net, None), c, 0.01, iters)
        res = E.run_fit(net, step, z, noisy, clean, iters, dev, task=task, down=down)
    res["task"], res["family"], res["perturb"] = task, family, perturb
    res["env"] = {k: v for k, v in os.environ.items() if k.startswith("DIP_")}
    return res


def main():
    size, iters, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    # <perturb> may be a comma list: the fits run one after the other in THIS process (the environment is the same for all of
    # them; saves the interpreter / library start-up per arm) and <out.json> holds the list
    perturbs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]
    task = sys.argv[5] if len(sys.argv) > 5 else "denoise"
    family = sys.argv[6] if len(sys.argv) > 6 else "hip"
    res = []
    for p in perturbs:
        res.append(one_fit(size, iters, p, task, family))
        print(json.dumps(res[-1]), flush=True)
        torch.cuda.empty_cache()
    with open(out, "w") as f:
        json.dump(res if len(perturbs) > 1 else res[0], f)


if __name__ == "__main__":
    main()
