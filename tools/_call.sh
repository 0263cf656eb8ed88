set -u
ROOTD="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOTD"
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r4c12
python __graft_entry__.py build > gpurun_out/${T}_build.log 2>&1
timeout 600 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/${T}_smoke.log | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -x > gpurun_out/${T}_all.log 2>&1; echo "all rc=$?"
tail -n 5 gpurun_out/${T}_all.log | cut -c1-300
timeout 300 python bench.py --steps 100 --warmup 20 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/${T}_bench.json
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench.json')); print(d['value'], d['ms_per_step'], 'roof', d['roofline']['frac'], d['roofline']['achieved'], d['roofline'].get('frac_of_fp32_mfma_peak_157.3'), 'wgrad', d['roofline_wgrad']['frac'], 'all3x3', d['roofline_conv3x3_all']['frac'], 'hbm', d['roofline_hbm']['frac'], d.get('other_mode'), d.get('fp32_mfma_only'), d['cpu_baseline']['value'])"
