"""One-box rehearsal of the N > 1 path of bench.py (the driver runs the real N = 2 / 4 / 8 lines on a whole node):
two ranks launched exactly as the driver launches them (python -m torch.distributed.run, rendezvous on 127.0.0.1), both on
GPU 0, talking over gloo (MAGI_BENCH_REHEARSE=1: timings are meaningless, every code path of the job is the real one --
chain ids per rank, independent Philox streams, barrier + max-over-ranks timing, the single final gather)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_on_one_gpu_gathers_four_distinct_chains_in_global_order():
    env = dict(os.environ, MAGI_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--chains-per-gpu", "2", "--steps", "3",
           "--warmup", "1", "--burnin", "5", "--no-cpu-baseline", "--no-extra-configs", "--profile-slots", "64"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["chains_total"] == 4 and out["scaling"] == "weak"
    assert "configs[2]" in out["config"]["baseline_config"]
    assert out["gathered_unit_ids"] == [0, 1, 2, 3]                                   # rank 0: ids 0, 1; rank 1: ids 2, 3
    th = [tuple(t) for t in out["theta_last_per_chain"]]
    assert len(set(th)) == 4                                                          # four different chains
    assert out["value"] > 0 and out["leapfrogs_per_s"] > 0 and out["steps"] == 3
