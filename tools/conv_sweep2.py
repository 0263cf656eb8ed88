import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
import dip_native as N
import hipops as H
from dip_native import round_up
from conv_sweep import bench
tag = "regstage" if os.environ.get("DIP_CONV_NO_DMA") else "dma"
for ks in (3, 1):
    for (Hh, Ww) in ((64, 128), (256, 256), (512, 512)):
        for Cin in (32, 64, 128, 256):
            nt, us, _ = bench(Cin, 128, ks, Hh, Ww, False, reps=30)
            print(f"{tag} k={ks} tiles={nt:5d} Cin={Cin:4d}: {us:8.1f} us", flush=True)
