"""Bisect harness for the end-quality offsets: runs a list of whole fits
-- (family, task, size, iterations, one-ulp perturbation index, environment) -- on ONE MI355X, P jobs at a time (P = 1 unless the fits are tiny: round 5 measured 8 processes
sharing the GPU at 300 s per fit instead of 22), each job in a process of its own (tests/end_quality_hip.py), and appends one JSON line per fit to the output file.

    python tools/eq_families.py <out.jsonl> <P> <job> [<job> ...]
    <job> = family:task:size:iters:perturbs[:ENV=V,ENV=V]       perturbs = comma list, e.g.  hip:sr:128:600:0,1,2,4

Families: hip | torch | hip_torchloss | hip_torchadam (tests/end_quality_hip.py).  The fits are deterministic functions
of code + environment, so sharing the GPU between P processes changes wall time only.  Test infrastructure only."""
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "end_quality_hip.py")


def main():
    out, par = sys.argv[1], int(sys.argv[2])
    jobs = []
    for spec in sys.argv[3:]:
        f = spec.split(":")
        family, task, size, iters, perturbs = f[0], f[1], int(f[2]), int(f[3]), [int(x) for x in f[4].split(",")]
        env = dict(kv.split("=", 1) for kv in f[5].split(",")) if len(f) > 5 and f[5] else {}
        jobs.append((family, task, size, iters, perturbs, env))       # one process per job: its fits run one after the other
    tmp = tempfile.mkdtemp(prefix="eqfam")
    t0 = time.time()

    def one(job):
        k, (family, task, size, iters, p, env) = job
        o = os.path.join(tmp, f"{k}.json")
        r = subprocess.run([sys.executable, SCRIPT, str(size), str(iters), o, ",".join(map(str, p)), task, family],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=3000)
        if r.returncode != 0:
            fits = [{"error": r.stderr[-1500:], "perturb": p}]
        else:
            fits = json.load(open(o))
            fits = fits if isinstance(fits, list) else [fits]
        for res in fits:
            res.update(family=family, task=task, size=size, iters=iters, job_env=env, t_done=time.time() - t0)
            with open(out, "a") as fh:
                fh.write(json.dumps(res) + "\n")
        return fits

    with ThreadPoolExecutor(max_workers=par) as ex:
        res = [f for fits in ex.map(one, enumerate(jobs)) for f in fits]
    bad = [r for r in res if "error" in r]
    print(f"{len(res)} fits in {time.time() - t0:.0f} s, {len(bad)} failed")
    for r in bad[:3]:
        print(r["family"], r["task"], r["error"][-600:])


if __name__ == "__main__":
    main()
