set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for c in /sys/class/drm/card*/device; do [ -e $c/pp_dpm_sclk ] && { echo $c; ls $c/hwmon/hwmon*/; for f in $c/hwmon/hwmon*/{power,freq}*; do echo "$f: $(cat $f 2>/dev/null | head -1)"; done; }; done > $O/c41_hwmon.txt 2>&1
cat $O/c41_hwmon.txt | head -40
timeout 600 python tools/host_contention.py --k 150 > $O/r04_host_contention.txt 2> $O/c40_hc.err; echo "hc rc=$?"
cat $O/r04_host_contention.txt; tail -3 $O/c40_hc.err
