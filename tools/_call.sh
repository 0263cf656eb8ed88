set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c18_build.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "fused" --no-header -p no:cacheprovider 2>&1 | tail -2
: > gpurun_out/ab.log
AB="DIP_NO_BNB_FUSE=1" REPS=3 bash tools/gpu_ab.sh
python bench.py --steps 30 --warmup 10 --mode eager --no-cpu-baseline --no-eager-line --dump-ops gpurun_out/ops_fused.json > /dev/null 2>&1
DIP_NO_BNB_FUSE=1 python bench.py --steps 30 --warmup 10 --mode eager --no-cpu-baseline --no-eager-line --dump-ops gpurun_out/ops_unfused.json > /dev/null 2>&1
python - <<'PY'
import json
a=json.load(open('gpurun_out/ops_fused.json')); b=json.load(open('gpurun_out/ops_unfused.json'))
tot_a=sum(v['ms'] for k,v in a.items() if '#' not in k); tot_b=sum(v['ms'] for k,v in b.items() if '#' not in k)
print("serial sum fused %.3f ms  unfused %.3f ms"%(tot_a,tot_b))
for k in ("dgrad:out","dgrad:s0.up1","dgrad:s0.up","dgthin:s0.up","dgrad:s1.up1","dgrad:s1.up","dgrad:s0.down_b","dgrad:s1.down_b","dgrad:s2.up","bnb_stats:s0.up1_bn","bnb_stats:s0.up_bn","bnb_stats:s0.cat_bn","bnb_stats:s0.down_a_bn","bnb_stats:s1.cat_bn","bnb_stats:s1.up_bn"):
    print(f"{k:26s} fused {1e3*a.get(k,{}).get('ms',0):7.1f}  unfused {1e3*b.get(k,{}).get('ms',0):7.1f}")
PY
