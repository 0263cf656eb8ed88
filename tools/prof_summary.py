"""rocprofv3 --kernel-trace --stats output (rocpd sqlite) -> plain-text per-kernel summary."""
import glob, re, sqlite3, sys
db = glob.glob(sys.argv[1] + "/*.db")[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# source: {db}")
print(f"# {steps} optimisation steps profiled; total kernel time {tot/1e3:.2f} ms = {tot/steps/1e3:.3f} ms per step")
print(f"{'kernel':78s} {'calls':>6s} {'calls/step':>10s} {'total_ms':>9s} {'ms/step':>8s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'%':>6s}")
setup_only = False
for n, c, s, a, mi, ma in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    # a call count that is not a multiple of the step count = (mostly) one-time launches of the set-up, e.g. the ~400
    # __amd_rocclr_copyBuffer that move module parameters into the arenas: "calls/step" is NOT a per-iteration figure there
    mark = "" if c % steps == 0 else " ~"
    setup_only |= bool(mark)
    print(f"{n[:78]:78s} {c:6d} {c/steps:10.1f} {s/1e3:9.2f} {s/steps/1e3:8.3f} {a:9.1f} {mi:8.1f} {ma:9.1f} {100*s/tot:6.2f}{mark}")
if setup_only:
    print("# ~ : call count not a multiple of the profiled steps (set-up launches included); see tools/prof_timeline.py for "
          "the launches of one steady-state iteration")
print()
print("# per launch geometry of the MFMA kernels (grid in workgroups)")
for pat in ("conv_igemm_dma_kernel", "conv_igemm_kernel", "conv_wgrad_kernel", "conv_small_kernel", "conv_bf3_kernel", "wgrad_bf3_kernel"):
    for r in cur.execute("select name, grid_x/workgroup_x, grid_y, grid_z, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                         "from kernels where name like ? group by 1,2,3,4 order by 6 desc", ("%" + pat + "%",)):
        n = re.sub(r"\(anonymous namespace\)::|\(Dip.*", "", r[0]).replace("void ", "")
        print(f"{n:44s} grid=({r[1]},{r[2]},{r[3]}) n={r[4]:4d} avg={r[5]:9.1f}us min={r[6]:8.1f} max={r[7]:9.1f}")

print()
print("# dominant kernel of bench.py's roofline object: conv_igemm_dma_kernel<3, 128, *> (both transform variants)")
r = cur.execute("select count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels where name like '%conv_igemm_dma_kernel<3, 128,%'").fetchone()
if r[0]:
    print(f"launches={r[0]} ({r[0]/steps:.1f}/step)  total={r[1]/1e3:.2f} ms ({r[1]/steps/1e3:.3f} ms/step)  avg launch={r[2]:.1f} us")
else:
    print("(not launched in this run)")
