"""Normalised matrix-pipe utilisation AND the clock each kernel actually ran at, from a rocprofv3
--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass.

  python tools/pmc_mfma.py <calibration_pass_dir> <bench_pass_dir> [out.json] > profiles/r0N_rocprofv3_pmc_MFMA.txt

SQ_VALU_MFMA_BUSY_CYCLES counts the cycles a SIMD's matrix pipe is busy (64 per v_mfma_f32_32x32x2_f32, 32 per
v_mfma_f32_32x32x16_bf16: MI355X_MICROARCH.md), summed over the 1024 SIMDs; GRBM_GUI_ACTIVE counts GPU-busy cycles summed
over the 8 XCDs (plus a constant ~3e5 of counter start/stop per dispatch: the value a 3 us copy kernel reports, subtracted
here).  The units are not assumed but CALIBRATED: the calibration pass runs the pure-MFMA loops of tools/ubench/mfma_peak.hip
(fp32) and tools/ubench/bf16x9.hip (bf16, LDS-fed); their full-chip launches give MFMA_BUSY / GUI_ACTIVE = 126.5 = 0.988 x
(1024 SIMDs / 8 XCDs) and GUI_ACTIVE / (8 x duration) = 2.2-2.4 GHz, so

    utilisation(kernel) = MFMA_BUSY / (128 x GUI_ACTIVE)       matrix pipes busy per clock cycle
    clock(kernel)       = GUI_ACTIVE / (8 x duration_ns) GHz   the clock the kernel really ran at (power management)

bench.py's time-derived roofline fraction prices a kernel against the peak AT THE NOMINAL 2.4 GHz, so the two relate as
    frac_time = utilisation x clock / 2.4 x (share of the executed MFMAs that is algorithmic work)
-- the last column printed below; for conv_bf3_kernel<9> every executed MFMA is algorithmic (128-channel layers, no padded
tiles at 512^2/256^2/128^2)."""
import collections, glob, json, re, sqlite3, sys


SIMD_PER_XCD = 128          # 1024 SIMDs / 8 XCDs: checked by the calibration below
NOMINAL_GHZ = 2.4


def dispatches(d):
    """[(kernel, {counter: value}, duration_ns)] of every dispatch in every *.db under d"""
    out = []
    for db in glob.glob(d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        meta = {}
        for did, kn, cn, val, dur in cur.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
            per[did][cn] += val
            meta[did] = (re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", kn)).replace("void ", ""), dur)
        out += [(meta[k][0], v, meta[k][1]) for k, v in per.items()]
    return out


def main():
    cal = dispatches(sys.argv[1])
    run = dispatches(sys.argv[2])
    # constant per-dispatch counter overhead in GUI_ACTIVE: the smallest value any dispatch reports (a few-us kernel)
    ovh = min(v.get("GRBM_GUI_ACTIVE", 1e30) for _, v, _ in cal + run)
    gui = lambda v: max(v.get("GRBM_GUI_ACTIVE", 0.0) - ovh, 1.0)
    print(f"# per-dispatch GUI_ACTIVE overhead subtracted: {ovh:.0f} cycles (summed over 8 XCDs)")
    print("# calibration: the full-chip dispatch (largest busy ratio) of each pure-MFMA loop of tools/ubench")
    cals = {}
    for k, v, dur in cal:
        if ("mfma_loop" in k or "loop_bf16x9" in k) and dur > 1e6:
            r = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / gui(v)
            if r > cals.get(k, (0, 0))[0]:
                cals[k] = (r, gui(v) / (8.0 * dur))
    for k, (r, ghz) in sorted(cals.items()):
        print(f"#   {k:28s} MFMA_BUSY/GUI_ACTIVE {r:7.2f} = {r / SIMD_PER_XCD:.3f} x 128    clock {ghz:.3f} GHz")
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    for k, v, dur in run:
        a = agg[k]
        a[0] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[1] += gui(v); a[2] += dur; a[3] += 1
    print(f"{'kernel':56s} {'launches':>8s} {'avg_us':>8s} {'utilisation':>11s} {'clock_GHz':>9s} {'util*clock/2.4':>14s}")
    out = {"simd_per_xcd": SIMD_PER_XCD, "gui_overhead": ovh,
           "calibration": {k: {"busy_per_gui": r, "clock_ghz": g} for k, (r, g) in cals.items()}, "kernels": {}}
    for k, (busy, g, dur, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        if busy <= 0:
            continue
        util, ghz = busy / (SIMD_PER_XCD * g), g / (8.0 * dur)
        # Round 5: the clock column is only meaningful where the constant counter start / stop window is
        # small against the dispatch (>= 150 us): two 55 us rows of round 4 read 3.0 and 3.2 GHz on a 2.4 GHz part.  Short
        # rows print no clock (and their utilisation carries the same uncertainty: flagged).  Cross-check of the long rows with
        # clocks measured INSIDE the kernels (s_memtime / s_memrealtime, tools/w3_profile.py): conv_bf3 1.83 GHz, wgrad_bf3
        # 1.66 GHz isolated, ~2.05 GHz inside the iteration -- the column's 1.8-1.95 is real.
        short = dur / n < 150e3 or ghz > NOMINAL_GHZ
        print(f"{k[:56]:56s} {n:8d} {dur / n / 1e3:8.1f} {util:11.3f}{'~' if short else ' '}" +
              (f"{'n/a':>9s} {'n/a':>14s}" if short else f"{ghz:9.3f} {util * ghz / NOMINAL_GHZ:14.3f}"))
        out["kernels"][k] = {"launches": n, "avg_us": round(dur / n / 1e3, 2), "utilisation": round(util, 4),
                             "short_dispatch": bool(short), "clock_ghz": None if short else round(ghz, 3),
                             "util_x_clock_over_nominal": None if short else round(util * ghz / NOMINAL_GHZ, 4)}
    print("# ~ : dispatch shorter than 150 us (or a derived clock above the nominal 2.4 GHz): the constant counter window is not small "
          "against it; no clock is derived, the utilisation is approximate")
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
