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
    objs = [o for o in glob.glob(os.path.join(G.CSRC, "build", "*.o")) if not os.path.basename(o).startswith(("wgrad_bf3.", "conv_bf3."))]
    assert len(objs) == len(G.SOURCES) - 2, objs
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-DDIP_W3_PROFILE"]
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(G.CSRC, f + ".hip"), "-o", f"/tmp/{f}_prof.o"]) for f in ("wgrad_bf3", "conv_bf3")]
    assert all(p.wait() == 0 for p in ps)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "/tmp/wgrad_bf3_prof.o", "/tmp/conv_bf3_prof.o", "-o", PROF])
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
    # ---- conv_bf3_kernel: the clock it runs at, measured inside the kernel (the counter-derived column had rows above 2.4 GHz)
    for (Cin, Cout, Hh, Ww) in ((128, 128, 512, 512), (128, 128, 256, 256), (128, 128, 128, 128)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(1, Cin, Hh, Ww, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev)
        a = (torch.rand(Cin, generator=g) + 0.5).to(dev); b = (torch.randn(Cin, generator=g) * 0.3).to(dev)
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            H.conv_bf3(x, w, None, N.PAD_REFLECT, (a, b, 0.2), terms=8)
        nt = lib.dip_conv_ntiles(Hh, Ww)
        buf = (C.c_ulonglong * (nt * 2))()
        assert raw.dip_b3_prof_read(buf, nt * 2) == 0
        v = torch.tensor(list(buf), dtype=torch.float64).view(nt, 2)
        print(f"conv_bf3 {Cin}->{Cout} @ {Hh}x{Ww} ({nt} tiles): cycles per workgroup mean {v[:, 0].mean().item():.0f}; IN-KERNEL clock = s_memtime / "
              f"s_memrealtime x 100 MHz = {100.0 * (v[:, 0].sum() / v[:, 1].sum()).item():.0f} MHz (per workgroup {100.0 * (v[:, 0] / v[:, 1]).min().item():.0f} .. "
              f"{100.0 * (v[:, 0] / v[:, 1]).max().item():.0f})")
    # ---- the same probes inside the REAL iteration (30 steps of the headline fit: sustained clocks, three streams)
    import bench
    fit = bench.Fit("default", 0, dev, "fused")
    for _ in range(30):
        fit.step()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8192 * 2))()
    assert raw.dip_b3_prof_read(buf, 8192 * 2) == 0
    v = torch.tensor(list(buf), dtype=torch.float64).view(8192, 2)[:2048]
    print(f"IN THE ITERATION: conv_bf3 (last launches, 2048 workgroup records): in-kernel clock {100.0 * (v[:, 0].sum() / v[:, 1].sum()).item():.0f} MHz "
          f"(per workgroup {100.0 * (v[:, 0] / v[:, 1]).min().item():.0f} .. {100.0 * (v[:, 0] / v[:, 1]).max().item():.0f})")
    buf = (C.c_ulonglong * (256 * 64))()
    assert raw.dip_w3_prof_read(buf, 256 * 64) == 0
    v = torch.tensor(list(buf), dtype=torch.float64).view(256, 8, 8)
    print(f"IN THE ITERATION: wgrad_bf3 (last launch): in-kernel clock {100.0 * (v[:, :, 6].sum() / v[:, :, 7].sum()).item():.0f} MHz; "
          f"per tile: mfma {2 * v[:, :, 0].mean().item() / v[:, :, 5].mean().item():.0f} staging {2 * (v[:, :, 2] + v[:, :, 3]).mean().item() / v[:, :, 5].mean().item():.0f} cycles")
    del fit
    names = ["mfma", "bar_after_mfma", "commit", "fetch_issue", "bar_after_stage", "half_periods", "kernel", "kernel_realtime_ticks"]
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
        rt = v[:, :, 7]
        print(f"  kernel cycles per wave: mean {k.mean().item():.0f} max {k.max().item():.0f}  -> clock {k.max().item() / (e0.elapsed_time(e1) * 1e3):.0f} MHz if the "
              f"longest wave spans the launch; IN-KERNEL clock = s_memtime / s_memrealtime x 100 MHz = {100.0 * (k.sum() / rt.sum()).item():.0f} MHz "
              f"(per wave {100.0 * (k / rt).min().item():.0f} .. {100.0 * (k / rt).max().item():.0f})")


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else main()
