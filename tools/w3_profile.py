"""Where the ping-pong weight-gradient kernel's cycles go: per-wave s_memtime sums of its phases (wgrad_bf3.hip, -DDIP_W3_PROFILE).

    python tools/w3_profile.py build      # here (hipcc): deep-image-prior_amd/lib/libdip_hip_prof.so (travels with gpurun)
    python tools/w3_profile.py            # on the MI355X: runs the layer shapes below through the profiling library, prints the table
"""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import __graft_entry__ as G
PROF = os.path.join(G.PKG, "lib", "libdip_hip_prof.so")


def build():
    G.build()
    objs = [o for o in glob.glob(os.path.join(G.CSRC, "build", "*.o")) if not os.path.basename(o).startswith("wgrad_bf3.")]
    assert len(objs) == len(G.SOURCES) - 1, objs
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-DDIP_W3_PROFILE"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(G.CSRC, "wgrad_bf3.hip"), "-o", "/tmp/wgrad_bf3_prof.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "/tmp/wgrad_bf3_prof.o", "-o", PROF])
    print("built", PROF)


def main():
    G.add_to_path()
    import dip_native as N
    N.LIB_PATH, N._lib = PROF, None
    import torch
    import hipops as H
    from dip_native import round_up
    lib = N.lib()
    raw = C.CDLL(PROF)
    dev = torch.device("cuda:0")
    st = H.stream(dev)
    names = ["mfma", "bar_after_mfma", "commit", "fetch_issue", "bar_after_stage", "half_periods", "kernel", "prologue"]
    for mode, (Cin, Cout, Hh, Ww) in [(m, sh) for sh in ((128, 128, 512, 512), (128, 128, 128, 128)) for m in (0, 1, 2)]:
        assert raw.dip_w3_prof_mode(mode) == 0
        print(f"--- mode {mode} ({['full kernel', 'knock-out: no MFMAs', 'knock-out: no staging after the prologue'][mode]})")
        g = torch.Generator().manual_seed(0)
        x = torch.randn(1, Cin, Hh, Ww, generator=g).to(dev)
        dy = torch.randn(1, Cout, Hh, Ww, generator=g).to(dev)
        a = (torch.rand(Cin, generator=g) + 0.5).to(dev); b = (torch.randn(Cin, generator=g) * 0.3).to(dev)
        xb, dyb = H.to_nhwc(x), H.to_nhwc(dy)
        trd, keep = H.transform(a, b, 0.2)
        CinP, CoutP = round_up(Cin, 32), round_up(Cout, 32)
        n, tg, cb = N.wgrad_plan2(Hh, Ww, Cin, Cout, 3, 1)
        partial = torch.zeros(n * 9 * CinP * CoutP, device=dev)
        bpart = torch.zeros(n * CoutP, device=dev)
        d = N.DipWgradDesc(xb.data_ptr(), Hh, Ww, round_up(Cin, 4), Cin, trd, dyb.data_ptr(), Hh, Ww, round_up(Cout, 4), Cout, 3, 1,
                           N.PAD_REFLECT, 1, partial.data_ptr(), bpart.data_ptr(), n, tg, cb)
        assert lib.dip_wgrad_bf3_eligible(C.byref(d))
        for _ in range(3):
            N.check(lib.dip_conv_wgrad(C.byref(d), st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); N.check(lib.dip_conv_wgrad(C.byref(d), st)); e1.record(); torch.cuda.synchronize()
        nwg = n * (CinP // 64)
        buf = (C.c_ulonglong * (nwg * 64))()
        assert raw.dip_w3_prof_read(buf, nwg * 64) == 0
        v = torch.tensor(list(buf), dtype=torch.float64).view(nwg, 8, 8)
        print(f"{Cin}->{Cout} @ {Hh}x{Ww}: {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented), {nwg} workgroups x 8 waves, nsplit {n}")
        hp = v[:, :, 5].mean().item()
        for grp in (0, 1):
            m = v[:, grp * 4:grp * 4 + 4].mean(dim=(0, 1))
            print(f"  group {grp}: " + "  ".join(f"{nm} {m[i].item():.0f}" for i, nm in enumerate(names)))
            print(f"           per tile: mfma {2 * m[0].item() / hp:.0f}  wait_after_mfma {2 * m[1].item() / hp:.0f}  commit {2 * m[2].item() / hp:.0f}  "
                  f"fetch {2 * m[3].item() / hp:.0f}  wait_after_stage {2 * m[4].item() / hp:.0f}  (ideal MFMA phase: 4608)")
        k = v[:, :, 6]
        print(f"  kernel cycles per wave: mean {k.mean().item():.0f} max {k.max().item():.0f}  -> clock {k.max().item() / (e0.elapsed_time(e1) * 1e3):.0f} MHz if the longest wave spans the launch")


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else main()
