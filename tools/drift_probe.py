"""Diagnostic for the end-quality offset: how far do the parameters whose gradient is ANALYTICALLY
zero (conv biases in front of a train-mode BatchNorm, gamma of scale-invariant BatchNorms: tests/parity.zero_grad_keys)
travel during a fit?  Adam turns their roundoff-level gradients into full-size steps, so a sign-consistent residue
becomes a linear drift (lr per iteration) where sign-random roundoff only random-walks (lr * sqrt(iterations)).

    python tools/drift_probe.py hip|cpu [iters] [size]      -> JSON with per-class drift statistics
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import __graft_entry__ as ge  # noqa: E402

ge.add_to_path()
import dip_oracle as O  # noqa: E402
import end_quality_cpu as E  # noqa: E402
import parity as PT  # noqa: E402


def main():
    arm = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    clean, noisy = E.problem(size)
    net, z = E.build(size)
    spec = O.default_spec()
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items() if k in O.param_shapes(spec)}
    zero = PT.zero_grad_keys(spec, sd0)
    if arm == "hip":
        ge.build()
        from utils.common_utils import get_params, optimize
        dev = torch.device("cuda:0")
        net = net.to(dev)
        res = E.run_fit(net, lambda c: optimize("adam", get_params("net", net, None), c, 0.01, iters), z, noisy, clean, iters, dev)
        sd1 = {k: v.detach().cpu() for k, v in net.state_dict().items() if k in sd0}
    else:
        torch.set_num_threads(min(8, os.cpu_count() or 1))
        onet = O.OracleNet(spec, sd0)
        res = E.run_fit(onet, lambda c: O.optimize_adam(onet.params, c, 0.01, iters), z, noisy, clean, iters, "cpu")
        sd1 = {k: p.detach() for k, p in zip(onet.names, onet.params)}
    rows = {}
    for k in sd0:
        d = (sd1[k] - sd0[k]).double()
        cls = ("zero:" if k in zero else "live:") + ("bn_gamma" if (k.endswith(".weight") and sd0[k].dim() == 1) else
                                                     ("bias" if k.endswith(".bias") else "conv_w"))
        rows.setdefault(cls, []).append((k, d.abs().mean().item(), d.mean().item(), d.abs().max().item()))
    out = {"arm": arm, "iters": iters, "result": res, "lr_times_iters": 0.01 * iters, "lr_times_sqrt_iters": 0.01 * iters ** 0.5}
    for cls, r in sorted(rows.items()):
        out[cls] = {"tensors": len(r), "mean_abs_drift": float(np.mean([x[1] for x in r])),
                    "mean_signed_drift": float(np.mean([x[2] for x in r])), "max_abs_drift": float(max(x[3] for x in r)),
                    "worst": sorted(r, key=lambda x: -x[3])[:3]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
