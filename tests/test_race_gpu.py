"""Determinism of the launch schedules on a real MI355X (tools/race_loop.py: the switch probe's 256x256 two-scale 128-channel
net, fused loss head, N backward passes in ONE process, every gradient compared BIT FOR BIT with the first pass).  A
cross-stream race, or a kernel that consumes a load before it has landed, shows as a mismatch in some passes.  Added in round 6
after the matrix-pipe form of conv_thin4 was found to produce wrong thin columns in 3 % of the passes when a chip-filling
weight-gradient launch ran beside it (DIP_DEFER_WGRAD=-1: the weight gradients inline, not deferred) -- one failure of the
switch test in ~25 suite runs was the only symptom."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switches", [{}, {"DIP_DEFER_WGRAD": "-1"}, {"DIP_DEFER_WGRAD": "0", "DIP_TAIL_INLINE": "0"},
                                      {"RACE_NET": "deep"}, {"RACE_NET": "deep", "DIP_DEFER_WGRAD": "-1"}],
                         ids=["default", "no_deferral", "defer0", "five_scales", "five_scales_no_deferral"])
def test_backward_passes_are_bit_identical(dev, switches):
    env = {k: v for k, v in os.environ.items() if not k.startswith(("DIP_", "RACE_"))}
    env.update(switches)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_loop.py"), "800"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("800 passes, 0 with a mismatch"), r.stdout[-3000:]
