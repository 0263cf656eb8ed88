"""Per-kernel PMC table from a rocprofv3 rocpd database (one --pmc pass):
   python tools/pmc_summary.py <dir> [name-filter]      values summed over counter instances, averaged per launch"""
import collections, glob, re, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size, workgroup_size from counters_collection")
per = collections.OrderedDict()
for did, kn, cn, val, dur, gs, wg in rows:
    kn = kn.replace("(anonymous namespace)::", "")
    short = re.sub(r"^void ", "", kn)
    short = re.sub(r"\(.*", "", short)[:64]
    if flt not in short:
        continue
    e = per.setdefault((short, gs // wg), {"ids": set(), "dur": 0.0, "c": collections.defaultdict(float)})
    if did not in e["ids"]:
        e["ids"].add(did)
        e["dur"] += dur
    e["c"][cn] += val
print(f"{'kernel':64s} {'wgs':>6s} {'n':>4s} {'avg_us':>9s}  counters per launch")
for (k, g), e in sorted(per.items(), key=lambda x: -x[1]["dur"]):
    n = len(e["ids"])
    print(f"{k:64s} {g:6d} {n:4d} {e['dur'] / n / 1e3:9.1f}  " + "  ".join(f"{c}={v / n:.4g}" for c, v in e["c"].items()))
