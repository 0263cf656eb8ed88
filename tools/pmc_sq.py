"""Per-kernel SQ counters of one rocprofv3 --pmc pass (8 SQ slots on gfx950):
  python tools/pmc_sq.py <pass_dir> > profiles/r0N_rocprofv3_pmc_SQ.txt
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES and
SQ_BUSY_CYCLES count cycles (MI355X_MICROARCH.md).  Reported per kernel (summed over its launches): share of the
wave cycles a wave is parked (s_waitcnt / barrier), issue-stalled, issuing; LDS bank-conflict cycles relative to the
LDS-active cycles; MFMA-busy cycles per SQ-busy cycle (the matrix pipes' duty cycle while the kernel runs)."""
import collections, glob, re, sqlite3, sys


def main():
    db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    nlaunch = collections.defaultdict(set)
    for did, kn, cn, val in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        kn = re.sub(r"\(anonymous namespace\)::", "", kn)
        kn = re.sub(r"\(.*", "", kn).replace("void ", "")
        per[kn][cn] += val
        nlaunch[kn].add(did)
    names = sorted({c for v in per.values() for c in v})
    print("# counters: " + ", ".join(names))
    print(f"{'kernel':58s} {'launches':>8s} {'parked%':>8s} {'stall%':>7s} {'issue%':>7s} {'ldsstall%':>9s} {'bankconf/ldsact%':>16s} {'mfma duty%':>10s}")
    rows = []
    for k, v in per.items():
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        rows.append((wc, k, v))
    for wc, k, v in sorted(rows, reverse=True)[:24]:
        busy = v.get("SQ_BUSY_CYCLES", 0.0)
        lds_act = v.get("SQ_ACTIVE_INST_LDS", 0.0)
        print(f"{k[:58]:58s} {len(nlaunch[k]):8d} {100 * v.get('SQ_WAIT_ANY', 0) / wc:8.1f} {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:7.1f} "
              f"{100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:7.1f} {100 * v.get('SQ_WAIT_INST_LDS', 0) / wc:9.1f} "
              f"{(100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / lds_act) if lds_act else float('nan'):16.1f} "
              f"{(100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy) if busy else float('nan'):10.1f}")


if __name__ == "__main__":
    main()
