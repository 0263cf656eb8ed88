"""Debug: shader-clock breakdown of the DMA conv kernel's K loop (library built with
make EXTRA=-DDIP_CLK_PROFILE).  Segments per unit: 0 loop head, 1 first MFMA block, 2 DMA issue,
3 MFMA blocks 1-3, 4 vmcnt wait, 5 in-place transform, 6 barrier; 7 = whole K loop."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conv_sweep import bench, lib
import torch
names = ["head", "mma0", "issue", "mma1-3", "vmwait", "trA", "barrier", "total"]
for (Cin, Hh, Ww, tr) in ((128, 64, 128, False), (128, 128, 256, False), (128, 512, 512, False), (128, 512, 512, True)):
    nt, us, _ = bench(Cin, 128, 3, Hh, Ww, tr, reps=3)
    torch.cuda.synchronize()
    buf = np.zeros((nt, 16), dtype=np.uint64)
    rc = lib.dip_debug_prof_read(buf.ctypes.data_as(C.c_void_p), nt)
    assert rc == 0, rc
    units = (Cin // 32) * 9
    tr_ = np.zeros((128, 8), dtype=np.uint32)
    lib.dip_debug_trace_read(tr_.ctypes.data_as(C.c_void_p))
    if nt <= 64:
        for u in range(units):
            print(f"     unit {u:2d} (tap {u % 9}): " + " ".join(f"{names[i]}={tr_[u, i]:5d}" for i in range(7)))
    b = buf.astype(np.float64)
    print(f"tiles={nt} Cin={Cin} tr={int(tr)}: {us:.1f} us/launch, {units} units; mean cycles per unit (wave 0), min..max over WGs of total")
    print("   " + "  ".join(f"{n}={b[:, i].mean() / units:7.1f}" for i, n in enumerate(names)) +
          f"   total min={b[:, 7].min():.0f} max={b[:, 7].max():.0f}")
    w0 = b[:, 8] - b[:, 8].min(); w1 = b[:, 9] - b[:, 8].min()          # 100 MHz ticks
    hw = buf[:, 10].astype(np.int64); xcc = buf[:, 11].astype(np.int64) & 15
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))
    print(f"   prologue={b[:, 12].mean():.0f} cyc  epilogue={b[:, 13].mean():.0f} cyc  WG lifetime mean={(w1 - w0).mean() / 100:.1f} us "
          f"(min {(w1 - w0).min() / 100:.1f}, max {(w1 - w0).max() / 100:.1f}); last end={w1.max() / 100:.1f} us; distinct CUs={len(set(cu.tolist()))}")
    # busy timeline: number of resident WGs over time, and per-CU co-residency
    order = np.argsort(w0)
    ends = np.sort(w1)
    for frac in (0.25, 0.5, 0.75, 0.9, 1.0):
        print(f"     {int(frac * 100):3d}% of WGs finished by {ends[int(frac * nt) - 1] / 100:7.1f} us", end="")
    print()
    if nt >= 1024:
        # for a few CUs: list (start, end) of their WGs
        for c in list(sorted(set(cu.tolist())))[:3]:
            idx = np.where(cu == c)[0]
            idx = idx[np.argsort(w0[idx])]
            print(f"     CU {c:#x}: " + " ".join(f"[{w0[i] / 100:.0f}-{w1[i] / 100:.0f}]" for i in idx))
