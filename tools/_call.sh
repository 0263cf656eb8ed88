set -u
ROOTD="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOTD"
mkdir -p gpurun_out; export TMPDIR=/tmp
python __graft_entry__.py build > gpurun_out/c31_build.log 2>&1
: > gpurun_out/c31_ab.log
for rep in 1 2; do
for v in base DIP_BN_APPLY_BLOCKS=2048 DIP_BN_APPLY_BLOCKS=4096 DIP_BN_BLOCKS=2048 DIP_BN_BLOCKS=2048,DIP_BN_APPLY_BLOCKS=4096; do
  if [ "$v" = base ]; then envs=""; else envs="${v//,/ }"; fi
  line=$(env $envs timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1)
  echo "$v rep$rep $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); h=d["roofline_hbm"]; print(d["value"], d["ms_per_step"], "hbm_ms", h["ms_per_step"], "TB/s", h["achieved"])' 2>/dev/null)" | tee -a gpurun_out/c31_ab.log
done
done
