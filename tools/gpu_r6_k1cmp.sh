cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6k1; mkdir -p $O
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
i=0
for v in "DIP_CONV_BF3_NO_1X1=1" "DIP_X=1"; do
  i=$((i+1))
  ( cd /tmp && env $v DIP_TWO_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- python $ROOTD/bench.py $B > $O/prof_bench.log 2>&1 )
  python tools/prof_summary.py $O/prof1 13 2>> $O/err.log | cut -c1-170 > $O/stats_$i.txt
  rm -rf $O/prof1
done
