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


def test_bench_line_carries_every_contract_field_on_one_gpu():
    """`python bench.py` (short run: 3 timed transitions, no CPU legs, no extra configs) prints ONE JSON line with the driver's contract
    fields, BASELINE.json's metric and the roofline object; `frac` is the live in-sampler figure (HIP events on every launch)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--burnin", "6", "--no-cpu-baseline", "--no-extra-configs",
           "--profile-slots", "64"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "leapfrogs_per_s", "us_per_slot_issued"):
        assert k in out, k
    assert out["metric"].startswith("HMC samples/sec") and out["unit"] == "samples/s" and out["n_gpus"] == 1 and out["steps"] == 3
    assert out["dtype"] == "f64" and out["data"] == "synthetic" and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert "workload" in out["config"] and "configs[1]" in out["config"]["baseline_config"]
    rf = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "standalone_frac", "frac_bytes_moved"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] < 1.3 and 5.0 < rf["us_per_launch"] < 40.0
    assert abs(out["value"] - out["config"]["chains_total"] * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-2 * out["value"]
