import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from conv_sweep import bench
for (Cin, Hh, Ww) in ((128, 64, 128), (128, 256, 256), (128, 512, 512), (132, 512, 512)):
    nt, us, _ = bench(Cin, 128, 3, Hh, Ww, True, reps=6)
    print(f"tiles={nt} Cin={Cin}: {us:.1f} us")
