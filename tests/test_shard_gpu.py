"""Multi-GPU path on ONE GPU: rank r of `bench.py --gpus 2` fits image r -- the same fit, bit for bit, as
the solo run `bench.py --gpus 1 --first-image r` (replicas only: no collective on the data path).  Both ranks
of the two-process run share cuda:0 here (LOCAL_RANK=0 for both); the gloo group carries the barrier / max-time / loss gather only."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--config", "snail", "--steps", "12", "--warmup", "3", "--mode", "eager", "--no-cpu-baseline", "--no-roofline", "--no-eager-line"]


def _line(stdout):
    ln = [l for l in stdout.splitlines() if l.startswith('{"metric"')]
    assert ln, stdout[-2000:]
    return json.loads(ln[-1])


@pytest.mark.gpu
def test_rank_r_of_a_two_rank_run_is_the_solo_fit_with_seed_r(dev):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    two = _line(outs[0][0])
    assert two["n_gpus"] == 2 and len(two["per_rank_final_loss_hex"]) == 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    for r in range(2):
        solo = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--first-image", str(r)] + ARGS,
                              env=env, capture_output=True, text=True, timeout=600)
        assert solo.returncode == 0, solo.stderr[-1500:]
        one = _line(solo.stdout)
        assert one["per_rank_final_loss_hex"][0] == two["per_rank_final_loss_hex"][r], (r, one["per_rank_final_loss_hex"], two["per_rank_final_loss_hex"])
    assert two["per_rank_final_loss_hex"][0] != two["per_rank_final_loss_hex"][1]        # (two different images)
