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
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "standalone_frac", "frac_bytes_moved", "frac_contract_bytes",
              "slot_frac_of_load_only", "kernel"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"].startswith("k_stream<1>")
    # `frac` is a fraction of the HBM peak for the bytes the kernel moves; the contract's algorithmic byte count is a third larger
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] <= 1.0 and 5.0 < rf["us_per_launch"] < 40.0
    assert rf["frac_contract_bytes"] > rf["frac"] and 0.2 < rf["slot_frac_of_load_only"] <= 1.0
    if rf["traffic"] is not None:          # committed PMC passes: the byte model the fraction rests on is what the counters saw
        assert abs(rf["traffic"] - rf["bytes_per_launch"]) < 0.05 * rf["bytes_per_launch"]
    assert abs(out["value"] - out["config"]["chains_total"] * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-2 * out["value"]


def _run_bench(extra, nproc):
    env = dict(os.environ, MAGI_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_alpha_sweep_config4_two_ranks_equal_one_rank_unit_for_unit():
    """BASELINE configs[3] across GPUs (magi_v2_amd/sweep.py, bench.py --config alpha-sweep): the 10 datasets x 8 chains = 80 units dealt to
    two ranks by whole datasets (rank 0: datasets 0, 2, ..; rank 1: 1, 3, ..), each rank building its datasets' matrices on its GPU, one
    gather.  All 80 unit ids arrive once, in global order, and every unit's last sample equals the ONE-rank run of the same command bit
    for bit: a unit's Philox stream is its global id and a dataset's chains share one handle whatever the number of ranks."""
    extra = ["--config", "alpha-sweep", "--steps", "3", "--warmup", "0", "--burnin", "5"]
    two = _run_bench(extra, 2)
    one = _run_bench(extra, 1)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "configs[3]" in two["config"]["baseline_config"]
    assert two["gathered_unit_ids"] == list(range(80)) == one["gathered_unit_ids"]
    assert two["per_rank_units"] == [40, 40] and one["per_rank_units"] == [80]
    assert two["theta_last_per_unit"] == one["theta_last_per_unit"]                   # bit for bit (10 decimal digits printed)
    assert len({tuple(t) for t in two["theta_last_per_unit"]}) == 80                   # 80 different chains
    assert two["value"] > 0 and two["leapfrogs_per_s"] > 0 and two["config"]["kernel"].startswith("k_stream")
