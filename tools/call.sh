set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dump-ops gpurun_out/ops.json --no-eager-line --mode eager 2>&1 | grep '^{"metric"' > gpurun_out/bench_tmp.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_tmp.json')); r=d['roofline']; w=d['roofline_wgrad']
print(d['value'],'it/s | dominant',r['achieved'],r['frac'],'avg',r['avg_launch_us'],'finish',r['splitk_finish_ms_per_step_not_included'],'allconv',r['all_conv_launches'],'| wgrad',w['achieved'],w['frac'],w['ms_per_step'],'red',w['reduce_ms_per_step_not_included'])
o=json.load(open('gpurun_out/ops.json'))
tot=sum(v['ms'] for k,v in o.items() if '#' not in k); print('ops total serial ms',tot)
PY
