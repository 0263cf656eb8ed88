#!/bin/bash
# the other notebook configurations on the round-5 code (bench lines only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r5h; mkdir -p $O
for c in kate sr library snail; do timeout 120 python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/r05_bench_$c.json; done
timeout 120 python bench.py --config snail --instances 8 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/r05_bench_snail_x8.json
timeout 120 python bench.py --config library --instances 8 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 > $O/r05_bench_library_x8.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5h/*.json")):
    try:
        o = json.load(open(f)); print(f.split("/")[-1], o["value"], "it/s", o["ms_per_step"], "ms", o["config"].get("reported_mode", "")[:60])
    except Exception as e:
        print(f, "ERR", e)
PY
