"""Plan sweeps of the fp32 weight-gradient kernel (kernel + slab reduction, MI355X): the two 3x3 stride-2 layers of the default
net over nsplit, the 1x1 layers over (nsplit, chan_block).  Calibrates dip_wgrad_plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wgrad_sweep as W
which = sys.argv[1] if len(sys.argv) > 1 else "s2"
if which == "s2":
    for (Cin, Hh) in ((32, 512), (128, 256)):
        for ns in (None, 128, 256, 512, 1024):
            plan = None if ns is None else (ns, 1, 1)
            try:
                p, us, rus, tf = W.bench(Cin, 128, 3, 2, Hh, Hh, plan=plan)
                print(f"Cin {Cin} in {Hh}^2 plan {p}: {us:7.1f} us + reduce {rus:5.1f} us  {tf:6.1f} TF", flush=True)
            except Exception as e:
                print("failed", Cin, Hh, ns, e)
else:
    for Hh in (512, 256):
        for cb in (4, 1):
            for ns in (None, 64, 128, 256, 512, 1024):
                plan = None if ns is None else (ns, 1, cb)
                try:
                    p, us, rus, tf = W.bench(128, 128, 1, 1, Hh, Hh, plan=plan)
                    print(f"1x1 128>128 {Hh}^2 plan {p}: {us:7.1f} us + reduce {rus:5.1f} us  {tf:6.1f} TF", flush=True)
                except Exception as e:
                    print("failed", Hh, cb, ns, e)
