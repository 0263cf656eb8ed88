set -u
ROOTD="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOTD"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4c3
python __graft_entry__.py build > gpurun_out/${T}_build.log 2>&1
timeout 900 python -m pytest tests/test_small_gpu.py -q -m gpu --no-header -p no:cacheprovider > gpurun_out/${T}_small.log 2>&1
echo "small rc=$?" | tee -a gpurun_out/${T}_small.log
timeout 1200 python -m pytest tests/test_net_gpu.py -q -m gpu -k "not end_quality and not ab_switch" --no-header -p no:cacheprovider > gpurun_out/${T}_net.log 2>&1
echo "net rc=$?" | tee -a gpurun_out/${T}_net.log
: > gpurun_out/${T}_ab.log
for rep in 1 2; do
for v in base DIP_CONV_NO_SMALL=1 DIP_SMALL_MAX_PIXELS=17000 DIP_TICKET_FIN=1; do
  if [ "$v" = base ]; then envs=""; else envs="${v//,/ }"; fi
  line=$(env $envs timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-eager-line 2>gpurun_out/${T}_bench_err.log | grep '^{"metric"' | tail -1)
  echo "$v rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["kernel_launches_per_iteration"], d["config"]["final_loss"], d.get("other_mode"))' 2>/dev/null)" | tee -a gpurun_out/${T}_ab.log
done
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof1 -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line > $ROOTD/gpurun_out/${T}_prof1.log 2>&1 )
python tools/prof_summary.py gpurun_out/prof1 13 > gpurun_out/${T}_prof1_summary.txt 2>> gpurun_out/${T}_prof1.log
python tools/prof_timeline.py gpurun_out/prof1 3 > gpurun_out/${T}_prof1_timeline.txt 2>> gpurun_out/${T}_prof1.log
rm -rf gpurun_out/prof1
tail -n 3 gpurun_out/${T}_small.log gpurun_out/${T}_net.log
