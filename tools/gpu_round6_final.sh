#!/bin/bash
# Evidence set of round 6's FINAL code in ONE gpurun call on ONE box: counter passes (HBM traffic, MFMA utilisation + its
# calibration), the ubench ceilings, the bench line that carries them (DIP_BENCH_PMC_SAME_CALL=1: nothing on it is stale),
# rocprofv3 kernel stats / timelines of the same command, the other notebook configurations (solo + grouped), smoke().
#   gpurun --timeout 1500 -- 'bash tools/gpu_round6_final.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6f; export TMPDIR=/tmp
ROOTD=$(pwd); O=$ROOTD/gpurun_out/r6f; T=r06
LOG=$O/${T}_final.log; : > $LOG
run() { echo "=== $* ===" | tee -a $LOG; local t0=$SECONDS; timeout "${TMO:-120}" "$@" >> $LOG 2>&1; echo "--- rc=$? ($((SECONDS-t0)) s, t=$SECONDS) ---" | tee -a $LOG; }
P="--steps 3 --warmup 2 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
PRE="env LD_PRELOAD=$ROOTD/deep-image-prior_amd/lib/libdip_hip.so DIP_TWO_STREAMS=0"
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_$ctr -o pmc -- $PRE python $ROOTD/bench.py $P )
done
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/${T}_pmc_traffic.json 2>> $LOG
python tools/pmc_summary.py $O/pmc_FETCH_SIZE > $O/${T}_rocprofv3_pmc_FETCH_SIZE.txt 2>> $LOG
python tools/pmc_summary.py $O/pmc_WRITE_SIZE > $O/${T}_rocprofv3_pmc_WRITE_SIZE.txt 2>> $LOG
( cd /tmp && TMO=60 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/a -o pmc -- $ROOTD/tools/ubench/bin/mfma_peak )
( cd /tmp && TMO=60 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_cal/b -o pmc -- $ROOTD/tools/ubench/bin/bf16x9 )
( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_MFMA -o pmc -- $PRE python $ROOTD/bench.py $P )
python tools/pmc_mfma.py $O/pmc_cal $O/pmc_MFMA $O/${T}_pmc_mfma.json > $O/${T}_rocprofv3_pmc_MFMA.txt 2>> $LOG
TMO=120 run python tools/ubench_ceilings.py $O/${T}_ubench_ceilings.json
cp $O/${T}_pmc_traffic.json $O/${T}_pmc_mfma.json $O/${T}_ubench_ceilings.json profiles/ 2>> $LOG       # (bench.py reads them: same call, same box)
TMO=400 run env DIP_BENCH_PMC_SAME_CALL=1 python bench.py --steps 100 --warmup 10 --dump-ops $O/${T}_ops.json
grep '^{"metric"' $LOG | tail -1 > $O/${T}_bench_line.json
B="--steps 10 --warmup 3 --mode eager --no-cpu-baseline --no-roofline --no-eager-line"
( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --stats -d $O/prof1 -o trace -- env DIP_TWO_STREAMS=0 python $ROOTD/bench.py $B )
python tools/prof_summary.py $O/prof1 13 > $O/${T}_rocprofv3_kernel_stats_single_stream.txt 2>> $LOG
python tools/prof_timeline.py $O/prof1 3 > $O/${T}_timeline_single_stream.txt 2>> $LOG
( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --stats -d $O/prof2 -o trace -- python $ROOTD/bench.py $B )
python tools/prof_summary.py $O/prof2 13 > $O/${T}_rocprofv3_kernel_stats_three_streams.txt 2>> $LOG
python tools/prof_timeline.py $O/prof2 3 > $O/${T}_timeline_three_streams.txt 2>> $LOG
# the library configuration: kernel stats + timeline of the solo iteration, bench lines solo / grouped x8
( cd /tmp && TMO=100 run rocprofv3 --kernel-trace --stats -d $O/prof3 -o trace -- python $ROOTD/bench.py --config library $B )
python tools/prof_summary.py $O/prof3 13 > $O/${T}_rocprofv3_kernel_stats_library.txt 2>> $LOG
python tools/prof_timeline.py $O/prof3 3 > $O/${T}_timeline_library.txt 2>> $LOG
rm -rf $O/prof1 $O/prof2 $O/prof3 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_cal $O/pmc_MFMA
for cfg in library kate sr snail; do
  TMO=300 run python bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --dump-ops $O/${T}_ops_$cfg.json
  grep '^{"metric"' $LOG | tail -1 > $O/${T}_bench_$cfg.json
done
for cfg in library snail; do
  TMO=300 run python bench.py --config $cfg --instances 8 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-eager-line
  grep '^{"metric"' $LOG | tail -1 > $O/${T}_bench_${cfg}_x8.json
done
TMO=200 run python __graft_entry__.py smoke
python - <<'PY'
import json
o = json.load(open("gpurun_out/r6f/r06_bench_line.json"))
r = o["roofline"]
print("LINE", o["value"], "it/s", o["ms_per_step"], "ms | frac", r["frac"], "achieved", r["achieved"], "peak", r["peak"], "| pmc util", r.get("mfma_util_pmc"), "clock", r.get("clock_ghz_pmc"),
      "frac_from_pmc", r.get("frac_from_pmc"), "| traffic", r.get("traffic"), "stale", r.get("traffic_stale"), "| wgrad", o["roofline_wgrad"]["frac"], "| 3x3 all", o["roofline_conv3x3_all"]["frac"],
      "| hbm", o["roofline_hbm"]["frac"], "| cpu", o["cpu_baseline"]["value"] if o.get("cpu_baseline") else None, "| power", (o.get("timed_region_power") or {}).get("power_w_mean"),
      "| host_issue", o.get("host_issue"))
for c in ("library", "kate", "sr", "snail", "library_x8", "snail_x8"):
    try:
        b = json.load(open(f"gpurun_out/r6f/r06_bench_{c}.json"))
        print(c, b["value"], "it/s", b["ms_per_step"], "ms", b["config"].get("reported_mode", "")[:60])
    except Exception as e:
        print(c, "missing", e)
PY
grep -E "rc=" $LOG | tail -24
