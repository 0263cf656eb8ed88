"""dip_conv_wgrad on the bf16 matrix pipe for the layer shapes of tests/test_bf3_gpu.WGRAD_BF3_CASES in a process of its own
(the kernel form is chosen by an environment variable read once per process): writes dW / db of every case to <out.npz>.
Used by tests/test_bf3_gpu.py::test_wgrad_bf3_pingpong_is_bit_identical_to_the_round4_kernel.  Test infrastructure only.

    [DIP_WGRAD_BF3_V1=1] python tests/wgrad_bf3_probe.py <out.npz>
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
import dip_native as N  # noqa: E402
import hipops as H  # noqa: E402

CASES = [(128, 128, N.PAD_REFLECT, 256, 256, True), (132, 128, N.PAD_REFLECT, 256, 256, True),
         (48, 160, N.PAD_ZERO, 250, 280, False), (96, 128, N.PAD_REFLECT, 256, 272, True),      # (96: an odd number of 32-channel chunks)
         (100, 128, N.PAD_REFLECT, 256, 256, True)]      # (100: an odd number of chunks AND a 4-channel tail -- ADVICE r05)


def main():
    dev = torch.device("cuda:0")
    rec = {}
    for k, (Cin, Cout, pad, Hh, Ww, use_tr) in enumerate(CASES):
        g = torch.Generator().manual_seed(100 + k)
        x = torch.randn(1, Cin, Hh, Ww, generator=g)
        dy = torch.randn(1, Cout, Hh, Ww, generator=g)
        a, b = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
        tr = (a.to(dev), b.to(dev), 0.2) if use_tr else (None, None, 1.0)
        for terms in (8, 9):
            N.check(N.lib().dip_conv_bf3_set_terms(terms))
            dw, db = H.conv_wgrad(x.to(dev), dy.to(dev), 3, 1, pad, tr, nsplit="plan")
            rec[f"dw{k}_{terms}"], rec[f"db{k}_{terms}"] = dw.cpu().numpy(), db.cpu().numpy()
        N.lib().dip_conv_bf3_set_terms(-1)
    np.savez(sys.argv[1], **rec)


if __name__ == "__main__":
    main()
