"""Diagnose tests/test_net_gpu.py::test_ab_switch_branches_compute_the_same_gradients: run tests/switch_probe.py with
each given switch assignment alone and print the gradient differences against the default run.
    python tools/switch_diag.py DIP_CONV_NO_EXTRA=1 DIP_CONV_NO_RES1X1=1 "DIP_A=1,DIP_B=2" ...
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env_extra, out):
    env = {k: v for k, v in os.environ.items() if not k.startswith("DIP_")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_probe.py"), out], env=env,
                       capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED", env_extra, r.stdout[-1500:], r.stderr[-3000:])
        return None
    return np.load(out)


def main():
    tmp = tempfile.mkdtemp()
    base = run({}, os.path.join(tmp, "base.npz"))
    base2 = run({}, os.path.join(tmp, "base2.npz"))
    sets = [("default again", base2)]
    for i, a in enumerate(sys.argv[1:]):
        sw = dict(kv.split("=") for kv in a.split(","))
        sets.append((a, run(sw, os.path.join(tmp, f"s{i}.npz"))))
    for name, got in sets:
        if got is None:
            continue
        print(f"== {name}: loss {float(got['loss']):.9g} (default {float(base['loss']):.9g}), head {float(got['loss_head']):.9g}; "
              f"max|out diff| {np.abs(got['out'].astype(np.float64) - base['out']).max():.2e}")
        flips = {k: int(np.unpackbits(np.bitwise_xor(base[k], got[k])).sum()) for k in base.files if k.startswith("m/")}
        print("   LeakyReLU branches that differ:", {k: v for k, v in flips.items() if v} or 0)
        for pre in ("g/", "gh/"):
            rows = []
            for k in base.files:
                if not k.startswith(pre):
                    continue
                a, b = base[k].astype(np.float64), got[k].astype(np.float64)
                na = np.linalg.norm(a)
                rows.append((np.linalg.norm(a - b) / (na + 1e-300), na, k))
            rel = np.array([r[0] for r in rows])
            print(f"   {pre}: median {np.median(rel):.2e}  max {rel.max():.2e}")
            for r in sorted(rows, reverse=True)[:4]:
                print(f"        {r[0]:.2e}  |g|={r[1]:.2e}  {r[2]}")


if __name__ == "__main__":
    main()
