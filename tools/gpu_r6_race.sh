#!/bin/bash
# round 6: hunting the rare garbage weight gradient of the switch test's set {DIP_DEFER_WGRAD=-1, DIP_SIDE_MIN_PIXELS=16384}:
# the probe in a loop, poisoned allocations, with and without a second process keeping the GPU busy; each switch alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6r
python tests/switch_probe.py gpurun_out/r6r/base.npz > /dev/null 2>&1
cmp() { python - "$1" <<'PY'
import sys, numpy as np
a=np.load("gpurun_out/r6r/base.npz"); b=np.load("gpurun_out/r6r/p.npz")
bad=[]
for k in a.files:
    if k.startswith(("g/","gh/")):
        x,y=a[k].astype(np.float64),b[k].astype(np.float64)
        e=np.linalg.norm(x-y); n=np.linalg.norm(x)
        if not np.isfinite(e) or e>1e-2*n+1e-7: bad.append((k,float(e),float(n)))
print(sys.argv[1], "BAD" if bad else "ok", bad[:6])
PY
}
if [ "${LOAD:-1}" = 1 ]; then
  ( timeout 400 python bench.py --config snail --steps 200000 --warmup 5 --mode eager --no-cpu-baseline --no-roofline --no-eager-line > /dev/null 2>&1 ) &
  LP=$!
  sleep 8
fi
for i in $(seq 1 ${N:-12}); do
  for sw in "DIP_DEFER_WGRAD=-1 DIP_SIDE_MIN_PIXELS=16384" "DIP_DEFER_WGRAD=-1" "DIP_SIDE_MIN_PIXELS=16384" "DIP_X=1"; do
    env PROBE_POISON=1 $sw python tests/switch_probe.py gpurun_out/r6r/p.npz > /dev/null 2>&1
    cmp "$sw" | grep -v " ok " 
  done
done
echo "done $N rounds"
[ -n "${LP:-}" ] && kill $LP 2>/dev/null
