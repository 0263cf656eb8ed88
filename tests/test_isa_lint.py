"""Build-time lint of the kernels whose global loads are inline asm with explicit vmcnt waits (tools/isa_inflight_check.py):
hipcc does not know that such a load's destination registers are in flight until the matching asm wait, and nothing in the
language stops it from copying or reusing them in between -- it did, in three kernels of rounds 4 - 6, each time with clean
source and wrong results on the GPU.  The machine code that ships is the one built here (the GPU box runs the prebuilt
library), so the check runs here, on the ISA of the same sources and flags.  No GPU needed; skipped without hipcc."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc here")
@pytest.mark.parametrize("src", ["conv_thin4.hip", "wgrad_tail.hip", "conv_small.hip", "conv_bf3.hip"])
def test_no_instruction_touches_a_register_with_an_asm_load_in_flight(src, tmp_path):
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "deep-image-prior_amd", "csrc"), "-S", "--cuda-device-only",
           os.path.join(ROOT, "deep-image-prior_amd", "csrc", src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_inflight_check.py"), str(out)], capture_output=True, text=True)
    last = r.stdout.strip().splitlines()[-1]
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 finding(s)" in last and not last.startswith("0 kernel"), last
