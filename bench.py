#!/usr/bin/env python3
"""Headline benchmark: HMC (NUTS) samples/sec on synthetic SEIR, N grid points x 4 components.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): SEIR-4 (S, E, I, R explicit), N = 1024
grid points (dt = 0.025, observations at even indices, noise 0.05 * range, PCG64(0)), dense
matrices, hyper-parameters at the reference's starting values, theta_init = (1, 1, 1), NUTS (max
tree depth 10) with dual averaging and the logarithmic annealing schedule -- i.e. the reference's
predict() sampler (magi_v2.py:357-396) -- one chain per GPU.  A "step" is one NUTS transition of
every chain on the rank.  Setup (data, GPU matrix build, 400 burn-in transitions that adapt the step size) is
untimed; the timed region is exactly K transitions with all inputs resident in HBM.

One rank per GPU (torchrun); chains are independent (no data-path collective); the post-burn-in
samples of all ranks are gathered once to rank 0 over RCCL after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch            # first: owns the HIP runtime the extension then shares
import torch.distributed as dist


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1024, help="N grid points")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="chains sampled on every GPU.  1 = BASELINE config 2 (the N=1 headline); 8 = config 3 "
                         "(64 independent chains over 8 GPUs): python -m torch.distributed.run ... bench.py --gpus 8 --chains-per-gpu 8")
    ap.add_argument("--burnin", type=int, default=400,
                    help="untimed burn-in transitions before warmup (80 %% of them adapt the step size, magi_v2.py:365).  400: the dual-averaging "
                         "step size has flattened by then (1.3e-3 after 40 steps, 2.1e-3 after 100, 2.5e-3 after 200, 2.8e-3 after 400 on this "
                         "workload), i.e. the timed transitions are those of an adapted sampler; with 40 (round 1) the trees of most chains are "
                         "twice as long as in steady state and the spread between chains is 1.5x (DESIGN.md section 5)")
    ap.add_argument("--band", type=int, default=-1, help="bandsize (-1 = dense)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2, help="oracle NUTS transitions for the CPU baseline")
    ap.add_argument("--seed", type=int, default=20250103)
    ap.add_argument("--replicate-chains", action="store_true",
                    help="DIAGNOSTIC ONLY: every rank samples the chain ids of rank 0 (identical per-GPU work, bit-identical "
                         "chains).  The default gives every chain of the job its own Philox stream -- independent chains, as "
                         "BASELINE config 3 asks -- so the max-over-ranks time includes the NUTS tree-size spread between chains")
    ap.add_argument("--cpu-threads", type=str, default="1,8,32,all", help="thread counts of the torch-CPU baseline leg")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # MAGI_BENCH_REHEARSE=1: all ranks share cuda:0 and talk over gloo -- a functional rehearsal of the N > 1 path on a
    # one-GPU box (timings meaningless); the real run is one rank per GPU over RCCL
    rehearse = os.environ.get("MAGI_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    red_dev = "cpu" if rehearse else "cuda"

    from magi_v2_amd import host
    from magi_v2_amd.engine import MagiEngine

    N, D, P = a.grid, 4, 3
    cpg = a.chains_per_gpu
    band = None if a.band < 0 else a.band

    # ---- setup (untimed) -----------------------------------------------------------------------
    I, X_obs, truth, theta_true = host.synthetic_seir(N, seed=0)
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    sig_pre0, th_pre0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(P), LB)

    eng = MagiEngine(dev_index)
    t0 = time.perf_counter()
    want_host = (rank == 0 and world == 1 and not a.no_cpu_baseline)
    mats = eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, bandsize=band, want_host=want_host)
    build_ms = (time.perf_counter() - t0) * 1e3
    eng.set_problem(Xi.mean(axis=0), N_ds.astype(np.float64), idx, y, beta, LB, "seir4")

    total = a.burnin + a.warmup + a.steps
    # stale_cache=0: the reference's annealed kernel reuses the previous step's cached target, which
    # was computed at the previous temperature (SURVEY section 7 quirk ii); on this synthetic grid the
    # log posterior is positive, which makes that offset reject every proposal, so the bench runs
    # the recomputing variant (same arithmetic per leapfrog, see DESIGN.md "stale cache").
    cfg = eng.default_cfg(num_results=a.warmup + a.steps, num_burnin_steps=a.burnin, stale_cache=0)
    rep = lambda v: np.repeat(np.asarray(v)[None], cpg, axis=0)
    from magi_v2_amd.shard import chain_ids_for_rank
    unit_ids = chain_ids_for_rank(rank, world, cpg * world)              # which (dataset, chain) units this rank owns
    chain_ids = list(range(cpg)) if a.replicate_chains else unit_ids     # the Philox streams they are sampled with
    eng.sampler_init(cfg, rep(Xhat), rep(sig_pre0), rep(th_pre0), seed=a.seed, chain_ids=chain_ids)
    eng.sampler_run(a.burnin)
    if a.warmup > 0:
        eng.sampler_run(a.warmup)

    # ---- timed region: exactly K transitions -------------------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lf, dev_ms = eng.sampler_run(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    lfs = torch.tensor([float(lf)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(lfs, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    lf_total = float(lfs.item())

    # ---- final sample gather over RCCL (the only collective of the job) -------------------------------
    from magi_v2_amd.shard import gather_samples
    Xs, sp, tp = eng.sampler_samples()
    flat = np.concatenate([Xs.reshape(cpg, Xs.shape[1], -1), sp, tp], axis=2)     # [chains, results, N*D + D + P]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    allsamp, gids = gather_samples(flat, unit_ids, dst=0)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t1) * 1e3 if world > 1 else 0.0
    th_all = allsamp[:, :, -P:] if rank == 0 else None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_chains = cpg * world
    value = n_chains * a.steps / elapsed
    diag = eng.sampler_diag()
    post = diag.tree_depth[:, a.burnin + a.warmup:]

    state = eng.sampler_state() if (world == 1 and not a.no_cpu_baseline) else None   # before the timing launches clobber it

    # ---- the same chains under round 1's protocol (40 untimed burn-in transitions): continuity with BENCH_r01 ----
    r1 = None
    if a.burnin != 40:
        c40 = eng.default_cfg(num_results=a.warmup + a.steps, num_burnin_steps=40, stale_cache=0)
        eng.sampler_init(c40, rep(Xhat), rep(sig_pre0), rep(th_pre0), seed=a.seed, chain_ids=chain_ids)
        eng.sampler_run(40 + a.warmup)
        torch.cuda.synchronize()
        tr0 = time.perf_counter()
        rlf, _ = eng.sampler_run(a.steps)
        torch.cuda.synchronize()
        r1s = time.perf_counter() - tr0
        r1 = {"burnin_untimed": 40, "samples_per_s": round(cpg * a.steps / r1s, 2), "leapfrogs_per_s": round(rlf / r1s, 1),
              "leapfrogs_per_sample": round(rlf / (cpg * a.steps), 1), "scope": "rank 0",
              "note": "after 40 steps dual averaging is still ramping (step size about half its adapted value): not the sampler's steady state"}

    # ---- fixed-L HMC (L = 32, step size from a 100-step dual-averaging warm-up), SURVEY 8d: reported next to NUTS ----
    hcfg = eng.default_cfg(num_results=a.steps, num_burnin_steps=100, stale_cache=0, mode=1, hmc_leapfrogs=32)
    eng.sampler_init(hcfg, rep(Xhat), rep(sig_pre0), rep(th_pre0), seed=a.seed, chain_ids=chain_ids)
    eng.sampler_run(100)
    torch.cuda.synchronize()
    th0 = time.perf_counter()
    hlf, _ = eng.sampler_run(a.steps)
    torch.cuda.synchronize()
    hmc_s = time.perf_counter() - th0
    hd = eng.sampler_diag()
    hmc = {"samples_per_s": round(cpg * a.steps / hmc_s, 2), "leapfrogs_per_s": round(hlf / hmc_s, 1), "L": 32,
           "accept_rate": round(float(hd.is_accepted[:, 100:].mean()), 3), "step_size": float(hd.step_size[0, -1]), "scope": "rank 0"}

    # ---- roofline of the dominant kernel (k_stream), HIP events on the engine's stream -----------------
    grad_ms, phase_ms = eng.time_gradient(cpg, 300)
    phase_bytes = eng.gradient_bytes(cpg)
    # "achieved" / "frac" are SURVEY 8d's contract: the ALGORITHMIC bytes of one gradient (3 D N W 8 + C 10 N D 8) over the
    # kernel's launch time.  The kernel itself streams fewer bytes: the symmetric operators FH and FK are stored as their
    # lower block triangle and FE serves both FE xc and FE^T f -- about 2 N^2 D values instead of 3 (bytes_per_launch).
    # frac_bytes_moved prices those bytes against the same peak; ceiling_GBps is a load-only pass over the same blocks with
    # the same access pattern, timed here; slot_frac is the per-leapfrog view (algorithmic bytes over the whole slot
    # [k_stream, k_point] as the sampler ran it in the timed region).
    W = N if band is None or 6 * band + 1 >= N else 2 * band + 1
    algorithmic = 3.0 * D * N * W * 8.0 + cpg * 10.0 * N * D * 8.0
    t_stream = phase_ms[4] * 1e-3
    achieved = algorithmic / t_stream / 1e9
    slot_s = elapsed / (lf_total / n_chains)                                 # seconds per leapfrog slot (one leapfrog of every chain on the GPU)
    n_tasks = phase_bytes[4] / (128 * 128 * 8.0 + 2.0 * cpg * 128 * 8.0)     # packed 128 x 128 blocks (magi_gradient_bytes)
    tiles_bytes = n_tasks * 128 * 128 * 8.0
    roofline = {"bound": "hbm", "kernel": "k_stream (single-phase block mat-vecs FH xc, FE xc, FE^T f, FK f over packed 128x128 blocks)",
                "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                "traffic": None, "algorithmic_bytes_per_launch": algorithmic, "bytes_per_launch": phase_bytes[4],
                "streamed_GBps": round(phase_bytes[4] / t_stream / 1e9, 1),
                "frac_bytes_moved": round(phase_bytes[4] / t_stream / 1e9 / 8000.0, 4),
                "ceiling_GBps": round(tiles_bytes / (phase_ms[7] * 1e-3) / 1e9, 1),
                "ceiling_note": "load-only kernel over the same packed blocks, same 16-B-per-lane pattern and the same alternating walk, "
                                "timed in this run; frac_of_ceiling = streamed_GBps / ceiling_GBps",
                "frac_of_ceiling": round((phase_bytes[4] / t_stream) / (tiles_bytes / (phase_ms[7] * 1e-3)), 4),
                "slot_frac": round(algorithmic / slot_s / 1e9 / 8000.0, 4),
                "slot_frac_bytes_moved": round((phase_bytes[4] + phase_bytes[6]) / slot_s / 1e9 / 8000.0, 4),
                "working_set": f"{phase_bytes[4] / 1e6:.1f} MB of operator blocks per launch: " +
                               ("resident in the 256 MiB Infinity Cache between launches, and -- odd slots walk the blocks backwards -- the "
                                "tail of one launch is still in the 8 x 4 MiB L2s for the head of the next: the HBM peak is the contract's "
                                "yardstick, not the physical source of the bytes (frac may exceed what HBM alone could deliver)"
                                if phase_bytes[4] < 200e6 else "larger than the 256 MiB Infinity Cache: streamed from HBM"),
                "us_per_launch": round(phase_ms[4] * 1e3, 3),
                "kernels_us": dict(zip(["phase1", "phase2", "phase3", "reduce", "stream", "leap_reduce", "point", "read_only"], [round(x * 1e3, 3) for x in phase_ms[:8]])),
                "three_phase_gradient_eval_us": round(grad_ms * 1e3, 3)}

    # measured memory-side traffic of the same kernel on the same workload, from the committed PMC passes
    # (rocprofv3 cannot run inside the timed process; profiles/README.md has the commands)
    try:
        import csv
        if N == 1024 and cpg == 1 and (band is None or 6 * band + 1 >= N):
            prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            name = next(f for f in ("r02_bench_pmc_traffic.csv", "r01_bench_pmc_traffic.csv") if os.path.exists(os.path.join(prof, f)))
            with open(os.path.join(prof, name)) as fh:
                tr = sum(float(r["bytes_per_launch_corrected"]) for r in csv.DictReader(fh) if "k_stream<1, 1>" in r["Kernel_Name"])
            if tr > 0:
                roofline["traffic"] = tr
                roofline["traffic_source"] = f"profiles/{name} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
    except (OSError, KeyError, ValueError, StopIteration):
        pass

    # ---- CPU baseline: torch-CPU fp64 restatement of magi_v2.py:308-348 (bmm + autograd, what XLA-CPU executes for
    #      the reference), timed at several thread counts on this host; samples/s = gradient evaluations/s over the leapfrogs
    #      per sample the GPU chain needed in the timed region.  Second leg: the numpy oracle continuing the SAME chain. ----
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        from oracle import magi_oracle as orc
        from oracle import torch_cpu
        import threadpoolctl
        C_inv, m, K_inv = mats
        pr = orc.Problem(I=I, mu=Xi.mean(axis=0), C_inv=orc.band_part(C_inv, band), m=orc.band_part(m, band),
                         K_inv=orc.band_part(K_inv, band), N_ds=N_ds.astype(np.float64), obs_idx=idx, y=y, beta=float(beta),
                         LB=LB, drift="seir4", P=P)
        Xc, spc, tpc, ss, bc = state
        ncpu = os.cpu_count() or 1
        threads = sorted({min(ncpu, ncpu if t == "all" else int(t)) for t in a.cpu_threads.split(",")})
        lf_per_sample = lf_total / (n_chains * a.steps)
        rates = torch_cpu.time_gradients(pr, Xc[0], spc[0], tpc[0], threads, min_evals=200, max_seconds=6.0)
        best_t = max(rates, key=rates.get)
        try:
            with open("/proc/cpuinfo") as fh:
                cpu_model = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
        except (OSError, StopIteration):
            cpu_model = "unknown"
        cpu = {"value": round(rates[best_t] / lf_per_sample, 5), "unit": "samples/s", "cores": best_t, "kind": "port",
               "sample": f">= 200 value+gradient evaluations (or 6 s) per thread count of oracle/torch_cpu.py -- torch-CPU fp64 bmm + autograd "
                         f"restatement of magi_v2.py:308-348 -- at the GPU chain's state; converted with the {lf_per_sample:.1f} leapfrogs per "
                         "sample of the timed GPU region",
               "gradient_evals_per_s": {str(t): round(r, 2) for t, r in rates.items()}, "leapfrogs_per_s": round(rates[best_t], 2),
               "host_cpus": ncpu, "cpu_model": cpu_model}
        # second leg: the numpy oracle (restated TFP NUTS) continues the same chain for a few transitions
        q = orc.pack(Xc[0], spc[0], tpc[0])
        fn_L = orc.make_fn_L(pr)
        L, gL = fn_L(q)
        k = total
        tc0 = time.perf_counter()
        n_lf = 0
        for s_ in range(a.cpu_steps):
            temp = orc.temperature(k + s_)
            res = orc.nuts_one_step(q, temp * L, temp * gL, float(ss[0]), temp, fn_L, k + s_, chain_ids[0], a.seed)
            n_lf += res.leapfrogs
            if res.is_accepted:
                q, L, gL = res.q, res.L, res.gL
        cpu_s = time.perf_counter() - tc0
        nthreads = max([p_.get("num_threads", 1) for p_ in threadpoolctl.threadpool_info()] + [1])
        cpu["numpy_nuts_leg"] = {"samples_per_s": round(a.cpu_steps / cpu_s, 5), "leapfrogs_per_s": round(n_lf / cpu_s, 2), "threads": nthreads,
                                 "sample": f"{a.cpu_steps} NUTS transitions ({n_lf} leapfrogs) of the same chain continued from the GPU state by "
                                           "oracle/magi_oracle.py (numpy + OpenBLAS; streams six matrices per gradient, batched matmul does not thread)"}

    out = {
        "metric": "HMC samples/sec (whole node) on SEIR, N grid pts x D comps",
        "value": round(value, 4), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SEIR N={N} x 4 components, {'dense' if band is None else 'band ' + str(band)}, "
                               f"NUTS(max depth 10)+dual averaging+log annealing, {cpg} chain(s)/GPU",
                   "grid": N, "components": D, "thetas": P, "chains_total": n_chains, "bandsize": band,
                   "burnin_untimed": a.burnin,
                   "burnin_note": "untimed transitions that adapt the step size before the timed region (the reference burns in 1000); "
                                  "round 1 used 40, see round1_protocol for that figure on the same chains and DESIGN.md section 5",
                   "parallelism": f"chains x{world}",
                   "stale_cache": 0, "stale_cache_note": "the reference's annealed kernel reuses the previous step's cached target (computed at the "
                                   "previous temperature); on this synthetic grid that offset rejects every proposal, so the bench recomputes at "
                                   "the current temperature -- same arithmetic per leapfrog (DESIGN.md 4.2)",
                   "chain_streams": "DIAGNOSTIC: every rank replicates rank 0's chain ids" if a.replicate_chains else
                                    "independent: Philox stream = global chain id (rank * chains_per_gpu + local index)"},
        "roofline": roofline, "cpu_baseline": cpu,
        "leapfrogs_per_s": round(lf_total / elapsed, 1), "us_per_leapfrog_slot": round(elapsed / (lf_total / n_chains) * 1e6, 2),
        "mean_tree_depth": round(float(post.mean()), 2), "device_ms": round(dev_ms, 2), "build_ms": round(build_ms, 1),
        "leapfrogs_per_sample": round(lf_total / (n_chains * a.steps), 1),
        "gather_ms": round(gather_ms, 3), "hmc_L32": hmc, "round1_protocol": r1, "theta_mean": [round(float(x), 4) for x in np.log1p(np.exp(th_all)).reshape(-1, P).mean(axis=0)],
    }
    if cpu:
        out["speedup_vs_cpu_port"] = round(value / cpu["value"], 1)          # against the BEST thread count of the torch-CPU leg
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
