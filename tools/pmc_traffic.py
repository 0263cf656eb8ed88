"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 counter slots, MI355X_MICROARCH.md).

  python tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> > profiles/r0N_pmc_traffic.json

Unit / gfx950 correction: rocprofv3 reports both in KiB of 64-byte requests; for 16-byte-per-lane
coalesced accesses gfx950 tallies 128-byte requests as 64 bytes (guide: "FETCH_SIZE reports exactly
1/2 of the bytes").  The factor is not assumed but calibrated in the same pass on
bn_bwd_apply_src_kernel, which streams exactly two tensors in and one out with the same access width:
its largest launches (scale 0, C = 128: 128 MiB per tensor) give known / reported."""
import collections, glob, json, re, sqlite3, sys

DOM = "conv_igemm_dma_kernel<3, 128"


def load(d):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    per = collections.OrderedDict()
    for did, kn, cn, val in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        e = per.setdefault(did, [kn.replace("(anonymous namespace)::", ""), 0.0])
        e[1] += val
    return per


def calib(per, known_bytes):
    """factor = known / reported for the scale-0, 128-channel bn_bwd_apply launches (the most common
    value among its big launches)."""
    vals = [v * 1024 for k, v in per.values() if "bn_bwd_apply_src_kernel" in k]
    vals = [v for v in vals if v > 0.9 * max(vals)]          # the scale-0 launches (C = 128 and C = 132)
    ratios = collections.Counter(round(known_bytes / v, 2) for v in vals)
    return ratios.most_common(1)[0][0], dict(ratios)


# the launches bench.py's roofline_hbm object counts (hbm_ops): BatchNorm backward, up-sample + concat and its
# adjoint, thin 1x1 convs / weight gradients, thin data-gradient columns
MEM_GROUP = ("bn_bwd_stats_kernel", "bn_bwd_apply_src_kernel", "bn_bwd_apply_kernel", "upcat_fwd_kernel",
             "upsample_bwd_stats_kernel", "thin1x1_wgrad_kernel", "conv_thin4_kernel", "conv_igemm_dma_kernel<1, 32")


def group_bytes(fetch, write, ff, wf):
    steps = sum(1 for k, _ in fetch.values() if "adam_kernel" in k)
    per = collections.OrderedDict()
    for tag, table, fac in (("fetch", fetch, ff), ("write", write, wf)):
        for k, v in table.values():
            for pat in MEM_GROUP:
                if pat in k:
                    e = per.setdefault(pat, {"fetch": 0.0, "write": 0.0, "launches": 0})
                    e[tag] += v * 1024 * fac
                    if tag == "fetch":
                        e["launches"] += 1
    tot = sum(e["fetch"] + e["write"] for e in per.values())
    return {"bytes_per_step": round(tot / max(steps, 1)), "steps_in_pass": steps,
            "per_kernel_bytes_per_step": {k: {"fetch": round(e["fetch"] / max(steps, 1)), "write": round(e["write"] / max(steps, 1)),
                                              "launches_per_step": round(e["launches"] / max(steps, 1), 1)} for k, e in per.items()},
            "source": "same passes; FETCH_SIZE x correction + WRITE_SIZE over the launches of the listed kernels"}


def main():
    global DOM
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    bf3 = any("conv_bf3_kernel" in k for k, _ in fetch.values())
    if bf3:                      # round 4: the dominant kernel is the bf16-pipe convolution
        DOM = "conv_bf3_kernel<"
    T = 512 * 512 * 128 * 4
    ff, fr = calib(fetch, 2 * T)
    wf, wr = calib(write, T)
    fv = [v * 1024 * ff for k, v in fetch.values() if DOM in k]
    wv = [v * 1024 * wf for k, v in write.values() if DOM in k]
    out = {
        "kernel": "conv_bf3_kernel<*>" if bf3 else "conv_igemm_dma_kernel<3,128,*>",
        "n_launches_fetch_pass": len(fv), "n_launches_write_pass": len(wv),
        "fetch_bytes_per_launch": round(sum(fv) / len(fv)), "write_bytes_per_launch": round(sum(wv) / len(wv)),
        "traffic_bytes_per_launch": round(sum(fv) / len(fv) + sum(wv) / len(wv)),
        "correction": {"FETCH_SIZE": ff, "WRITE_SIZE": wf,
                       "calibrated_on": "bn_bwd_apply_src_kernel, scale 0, C=128: reads 2 x 128 MiB, writes 128 MiB",
                       "observed_known_over_reported": {"FETCH_SIZE": fr, "WRITE_SIZE": wr}},
        "memory_bound_group": group_bytes(fetch, write, ff, wf),
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --steps 3 --warmup 2 "
                  "--no-cpu-baseline --no-roofline (tools/gpu_round.sh, DO_PMC=1); all launches of the kernel in the run",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
