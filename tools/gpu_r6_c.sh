#!/bin/bash
# round 6 call c: one-launch BatchNorm backward -- parity tests, then interleaved A/B on the default and library configurations
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6c; export TMPDIR=/tmp
O=gpurun_out/r6c
timeout 900 python -m pytest tests/test_bnone_gpu.py tests/test_small_gpu.py -x -q -m gpu > $O/test_bnone.log 2>&1; echo "tests rc=$?"; tail -5 $O/test_bnone.log
timeout 900 python -m pytest tests/test_net_gpu.py -x -q -m gpu -k "golden or iteration1 or parity or tiny" > $O/test_net.log 2>&1; echo "net tests rc=$?"; tail -5 $O/test_net.log
for rep in 1 2 3; do
 for v in base DIP_BNB_NO_ONE=1 DIP_BNB_ONE_MAX_PIXELS=4096 DIP_BNB_ONE_MAX_PIXELS=65536; do
  if [ "$v" = base ]; then envs=""; else envs="$v"; fi
  for cfg in default library; do
    line=$(env $envs timeout 300 python bench.py --config $cfg --steps 100 --warmup 20 --mode eager --no-cpu-baseline --no-roofline --no-eager-line 2>/dev/null | grep '^{"metric"' | tail -1)
    echo "$cfg $v rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernel_launches_per_iteration"])' 2>/dev/null)" | tee -a $O/ab.log
  done
 done
done
