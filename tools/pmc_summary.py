"""Per-dispatch PMC table from a rocprofv3 rocpd database: python tools/pmc_summary.py <dir> [name-filter]"""
import collections, glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size, workgroup_size "
                   "from counters_collection order by dispatch_id")
disp = collections.OrderedDict()
for did, kn, cn, val, dur, gs, wg in rows:
    if flt not in kn:
        continue
    d = disp.setdefault(did, {"name": kn.split("(")[0][-60:], "dur": dur, "wgs": gs // wg})
    d[cn] = d.get(cn, 0) + val
names = sorted({k for d in disp.values() for k in d if k not in ("name", "dur", "wgs")})
print("did wgs dur_us " + " ".join(names))
for did, d in disp.items():
    print(did, d["wgs"], f"{d['dur']/1e3:.1f}", " ".join(f"{d.get(n, 0):.4g}" for n in names), d["name"])
