"""Times the data gradient of the stride-2 3x3 convolutions (128 -> 128 channels, reflection padding) at the
default net's sizes: dip_conv_igemm on the descriptor dip_engine builds, split-K finish included.
Run once per setting:  [DIP_CONV_NO_PHASE=1 | DIP_CONV_PHASE_KSPLIT=k] python tools/dgrad2_sweep.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "deep-image-prior_amd"), os.path.join(ROOT, "tests")]
import dip_native as N  # noqa: E402
import hipops as H  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = N.lib()
    tag = "dilated" if os.environ.get("DIP_CONV_NO_PHASE") else "phase k=" + os.environ.get("DIP_CONV_PHASE_KSPLIT", "plan")
    for Hin in (512, 256, 128, 64, 32):
        Ho = Hin // 2
        Hg = Hin + 2
        w = torch.randn(128, 128, 3, 3, device=dev) / 34.0
        packed, _, do = H.pack(w)
        dy = torch.randn(Ho * Ho * 128, device=dev)
        g = torch.empty(Hg * Hg * 128, device=dev)
        ksplit, _, wsf = N.conv_plan_dil2(Hg, Hg, 128, 128, 3)
        ws = torch.empty(max(wsf, 4), device=dev)
        d = N.DipConvDesc(dy.data_ptr(), Ho, Ho, 128, 128, N.DipTransform(None, None, 1.0), packed.data_ptr() + 4 * do,
                          None, g.data_ptr(), Hg, Hg, 128, 128, 0, 3, 1, N.PAD_ZERO, 2, 2, 0, None, ksplit,
                          ws.data_ptr() if ksplit > 1 else None)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            N.check(lib.dip_conv_igemm(C.byref(d), st), "conv")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            lib.dip_conv_igemm(C.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        gf = 2.0 * 128 * 128 * 9 * Ho * Ho / 1e9
        print(f"{tag:14s} input {Hin:4d}^2: variant {lib.dip_conv_variant(C.byref(d))} ksplit {ksplit:2d}  {us:7.1f} us  "
              f"{gf / us * 1e3:6.1f} TF algorithmic", flush=True)


if __name__ == "__main__":
    main()
