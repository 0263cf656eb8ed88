set -u
ROOTD="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOTD"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4c7
python __graft_entry__.py build > gpurun_out/${T}_build.log 2>&1
timeout 600 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 1500 python -m pytest tests/test_small_gpu.py tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider > gpurun_out/${T}_kern.log 2>&1; echo "kern rc=$?"
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -m gpu --no-header -p no:cacheprovider > gpurun_out/${T}_full.log 2>&1; echo "full rc=$?"
timeout 1500 python -m pytest tests/test_closure_gpu.py tests/test_monitor_gpu.py -q -m gpu --no-header -p no:cacheprovider > gpurun_out/${T}_clos.log 2>&1; echo "closure rc=$?"
timeout 1500 python -m pytest tests/test_net_gpu.py -q -m gpu -k "not end_quality" --no-header -p no:cacheprovider > gpurun_out/${T}_net.log 2>&1; echo "net rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOTD/gpurun_out/prof2 -o trace -- python $ROOTD/bench.py --steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line > $ROOTD/gpurun_out/${T}_prof2.log 2>&1 )
python tools/prof_timeline.py gpurun_out/prof2 3 > gpurun_out/${T}_prof2_timeline.txt 2>> gpurun_out/${T}_prof2.log
rm -rf gpurun_out/prof2
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/${T}_bench.json
tail -n 2 gpurun_out/${T}_kern.log gpurun_out/${T}_full.log gpurun_out/${T}_clos.log gpurun_out/${T}_net.log
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_wgrad']['frac'], d['roofline_conv3x3_all']['frac'], d['roofline_hbm']['frac'], d.get('other_mode'))"
