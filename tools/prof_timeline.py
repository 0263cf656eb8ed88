"""Timeline of ONE optimisation iteration from a rocprofv3 --kernel-trace database of the two-stream run:
where the wall time goes (forward span, backward span, GPU idle, one-kernel vs overlapped time) and the launches
on the critical path.   python tools/prof_timeline.py <dir with *.db> [iteration index from the end, default 3]
"""
import glob, re, sqlite3, sys

db = glob.glob(sys.argv[1] + "/*.db")[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("# columns:", cols)
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(cur.execute(f"select name, start, end, {qcol or 0} from kernels order by start"))
short = lambda n: re.sub(r"\(anonymous namespace\)::|\(Dip.*|void ", "", n)[:58]
marks = [i for i, r in enumerate(rows) if "noise_axpy" in r[0]]
if not marks:      # a configuration without reg-noise (the 'library' inpainting net, reg_noise_std = 0): the weight repack opens an iteration
    marks = [i for i, r in enumerate(rows) if "pack_weights_kernel" in r[0]]
if len(marks) < back + 1:
    sys.exit("not enough iterations in the trace")
i0, i1 = marks[-back - 1], marks[-back]
it = rows[i0:i1]
t0 = it[0][1]
tend = max(r[2] for r in it)
print(f"# iteration: {len(it)} launches, wall {1e-3 * (tend - t0):.1f} us (start of noise_axpy to the end of its last kernel); "
      f"next iteration starts at +{1e-3 * (rows[i1][1] - t0):.1f} us")
# sweep: time with 0 / 1 / >= 2 kernels running
ev = sorted([(r[1], 1) for r in it] + [(r[2], -1) for r in it])
lvl, last, acc = 0, t0, {0: 0, 1: 0, 2: 0}
for t, d in ev:
    acc[min(lvl, 2)] += t - last
    last, lvl = t, lvl + d
print(f"# GPU idle {acc[0] / 1e3:.1f} us, exactly one kernel {acc[1] / 1e3:.1f} us, two or more {acc[2] / 1e3:.1f} us")
streams = sorted({r[3] for r in it})
for s in streams:
    rs = [r for r in it if r[3] == s]
    busy = sum(r[2] - r[1] for r in rs)
    print(f"# stream/queue {s}: {len(rs)} launches, busy {busy / 1e3:.1f} us, first +{(rs[0][1] - t0) / 1e3:.1f}, last end +{(rs[-1][2] - t0) / 1e3:.1f}")
fw_end = next((r[2] for r in it if "loss_head_fwd" in r[0] or "head_fwd" in r[0]), None)
if fw_end:
    print(f"# forward span {1e-3 * (fw_end - t0):.1f} us, rest {1e-3 * (tend - fw_end):.1f} us")
print(f"{'start_us':>9s} {'dur_us':>8s} {'gap_us':>7s} {'q':>3s}  kernel")
prev_end = {s: None for s in streams}
for n, a, b, s in it:
    gap = (a - prev_end[s]) / 1e3 if prev_end[s] is not None else 0.0
    prev_end[s] = b
    print(f"{(a - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f} {gap:7.1f} {str(s)[-3:]:>3s}  {short(n)}")
