"""rocprofv3 --pmc pass (rocpd sqlite) -> plain-text per-kernel / per-launch-shape counter summary.
  python tools/pmc_summary.py <pass_dir> > profiles/r0N_rocprofv3_pmc_<COUNTER>.txt
Counter values are summed over the XCD instances of a dispatch (rocprofv3 reports one row per instance)."""
import collections, glob, re, sqlite3, sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
per = collections.OrderedDict()
for did, kn, gs, ws, cn, val, dur in cur.execute(
        "select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, duration from counters_collection"):
    e = per.setdefault(did, [re.sub(r"\(anonymous namespace\)::|\(.*", "", kn).replace("void ", ""), gs // max(ws, 1), cn, 0.0, dur])
    e[3] += val
groups = collections.OrderedDict()
for name, wgs, cn, val, dur in per.values():
    g = groups.setdefault((name, wgs, cn), [0, 0.0, 0.0])
    g[0] += 1
    g[1] += val
    g[2] += dur
print(f"# source: {db}")
print(f"{'kernel':66s} {'wgs':>6s} {'n':>4s} {'avg_us':>9s}  counter per launch (KiB of 64-byte requests for FETCH_SIZE / WRITE_SIZE)")
for (name, wgs, cn), (n, val, dur) in sorted(groups.items(), key=lambda kv: -kv[1][2]):
    print(f"{name[:66]:66s} {wgs:6d} {n:4d} {dur / n / 1e3:9.1f}  {cn}={val / n:.4g}")
