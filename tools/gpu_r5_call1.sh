#!/bin/bash
# Round 5, first GPU call: (1) the tests the driver never reached in round 4, (2) what round 4 prepared and never ran
# (64-column bf16-pipe tile, direct 1x1 weight gradient), (3) the end-quality bisect families.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5a
mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.log; }

python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
stamp "build rc=$?"

# (3a) the long fits first, in the background of everything else? no: GPU timing below must not be disturbed -> families run alone
timeout 900 python -m pytest tests/test_small_gpu.py tests/test_notebook_gpu.py \
    "tests/test_net_gpu.py::test_full_size_properties_512" "tests/test_net_gpu.py::test_reflected_3x3_skip_conv_next_to_down_a_128" \
    -m gpu -q -x -n 6 > $O/unrun_tests.log 2>&1
stamp "unrun tests rc=$? $(tail -n 1 $O/unrun_tests.log)"

DIP_CONV_BF3_N64=1 timeout 300 python -m pytest tests/test_bf3_gpu.py -k n64 -q > $O/n64_test.log 2>&1
stamp "n64 test rc=$? $(tail -n 1 $O/n64_test.log)"

timeout 120 tools/ubench/bin/wgrad1x1_direct > $O/wgrad1x1_direct.log 2>&1
stamp "wgrad1x1_direct rc=$?"

# A/B on one box, interleaved
AB="DIP_CONV_BF3_N64=1 DIP_CONV_BF3_N64=1,DIP_WGRAD_BF3_MIN_TILES=512" REPS=2 STEPS=100 timeout 600 tools/gpu_ab.sh > $O/ab.log 2>&1
cp gpurun_out/ab.log $O/ab_lines.log 2>/dev/null
stamp "ab done"

# (3) families.  128^2 fits are host-bound: 8 at a time
P=0,1,2,4,5,6,8,9
timeout 1500 python tools/eq_families.py $O/families.jsonl 8 \
    hip:sr:128:600:$P torch:sr:128:600:$P hip_torchloss:sr:128:600:0,1,2,4 hip_torchadam:sr:128:600:0,1,2,4 \
    hip:sr:128:600:0,1,2,4:EQ_REG_SCALE=0 torch:sr:128:600:0,1,2,4:EQ_REG_SCALE=0 \
    hip:inpaint:128:600:$P torch:inpaint:128:600:$P \
    hip:denoise:128:600:$P torch:denoise:128:600:$P > $O/families_128.log 2>&1
stamp "families 128 done: $(tail -n 1 $O/families_128.log)"
timeout 1200 python tools/eq_families.py $O/families.jsonl 4 \
    hip:denoise:256:1800:0,1,2,4,5,6 torch:denoise:256:1800:0,1,2,4,5,6 > $O/families_256.log 2>&1
stamp "families 256 done: $(tail -n 1 $O/families_256.log)"

# (4) end quality where the bf16 pipe is engaged: 512^2, 3000 iterations (denoising.ipynb:155), device reg-noise
timeout 900 python tools/eq_families.py $O/families512.jsonl 3 \
    hip:denoise:512:3000:0,1,2:EQ_DEVICE_NOISE=1 hip:denoise:512:3000:0,1,2:EQ_DEVICE_NOISE=1,DIP_CONV_BF3=0 \
    hip:denoise:512:3000:0,1,2:EQ_DEVICE_NOISE=1,DIP_CONV_BF3=9 > $O/families_512.log 2>&1
stamp "families 512 done: $(tail -n 1 $O/families_512.log)"
