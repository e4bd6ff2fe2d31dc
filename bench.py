#!/usr/bin/env python3
"""Headline benchmark: HMC (NUTS) samples/sec on synthetic SEIR, N grid points x 4 components.

    python bench.py --gpus N --steps K --warmup W

Workload of the headline `value` (BASELINE.json configs[1]): SEIR-4 (S, E, I, R explicit),
N = 1024 grid points (dt = 0.025, observations at even indices, noise 0.05 * range, PCG64(0)), dense matrices,
hyper-parameters at the reference's starting values, theta_init = (1, 1, 1), NUTS (max tree depth 10) with dual
averaging and the logarithmic annealing schedule -- i.e. the reference's predict() sampler (magi_v2.py:357-396) -- one
chain per GPU (`--chains-per-gpu 8` = BASELINE config 3's per-GPU share).  A "step" is one NUTS transition of every
chain on the rank.  Setup (data, GPU matrix build, 400 burn-in transitions that adapt the step size) is untimed; the
timed region is exactly K transitions with all inputs resident in HBM.

One rank per GPU (torchrun); chains are independent (no data-path collective); the post-burn-in samples of all ranks
are gathered once to rank 0 over RCCL after the timed region.

After the headline region a one-GPU run (N = 1) also measures, untimed-setup style, the single-GPU figures of the other
BASELINE configs -- config 1 (SEIR-4, N = 161, b = 80, one chain, 200 + 200 NUTS steps, next to the C-port CPU leg),
config 3's per-GPU share (8 chains, N = 1024), one dataset of config 4's alpha sweep (8 chains, N = 161, b = 80) and
config 5 (N = 8192 x 4: one pooled build with per-class device times, and the streaming kernel on 4.4 GB of operator
blocks) -- plus the hyper-parameter fit's ms per Adam step -- and reports them as FLAT scalars inside `roofline`
(`cfg1_*`, `mc8_*`, `cfg4_*`, `n8192_*`, `f1_*`).
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

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6    # AMD spec (the guide has no fp64 row)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1024, help="N grid points")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="chains sampled on every GPU.  1 = BASELINE config 2 (the N=1 headline; N > 1: its weak scaling, one chain per "
                         "GPU); 8 = config 3 (64 independent chains over 8 GPUs): python -m torch.distributed.run ... bench.py --gpus 8 "
                         "--chains-per-gpu 8")
    ap.add_argument("--burnin", type=int, default=400,
                    help="untimed burn-in transitions before warmup (80 %% of them adapt the step size, magi_v2.py:365).  400: the dual-averaging "
                         "step size has flattened by then (1.3e-3 after 40 steps, 2.1e-3 after 100, 2.5e-3 after 200, 2.8e-3 after 400 on this "
                         "workload), i.e. the timed transitions are those of an adapted sampler (DESIGN.md section 5)")
    ap.add_argument("--band", type=int, default=-1, help="bandsize (-1 = dense)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the config 1 / 3 / 4 / 5 legs of a one-GPU run")
    ap.add_argument("--cpu-steps", type=int, default=2, help="oracle NUTS transitions for the CPU baseline's numpy leg")
    ap.add_argument("--seed", type=int, default=20250103)
    ap.add_argument("--replicate-chains", action="store_true",
                    help="DIAGNOSTIC ONLY: every rank samples the chain ids of rank 0 (identical per-GPU work, bit-identical "
                         "chains).  The default gives every chain of the job its own Philox stream -- independent chains, as "
                         "BASELINE config 3 asks -- so the max-over-ranks time includes the NUTS tree-size spread between chains")
    ap.add_argument("--cpu-threads", type=str, default="1,8,16,32,64,all", help="thread counts of the CPU baseline legs")
    ap.add_argument("--profile-slots", type=int, default=512, help="leapfrog slots of the in-sampler kernel-duration leg")
    ap.add_argument("--config", choices=["headline", "alpha-sweep"], default="headline",
                    help="headline: BASELINE configs[1] / configs[2] (see --chains-per-gpu).  alpha-sweep: BASELINE configs[3] -- the ten alpha-sweep "
                         "datasets x 8 chains = 80 (dataset, chain) units dealt to the GPUs by whole datasets (magi_v2_amd/sweep.py), total work fixed: "
                         "python -m torch.distributed.run ... bench.py --gpus 8 --config alpha-sweep")
    return ap.parse_args()


def setup_problem(host, I, X_obs, P):
    """Host-side constants of a problem (reference: magi_v2.py:85-100, 105, 114, 277, 299-300) at the reference's STARTING hyper-parameters."""
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    sig_pre0, th_pre0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(P), LB)
    return dict(I=I, Xi=Xi, hp=hp, N_ds=N_ds.astype(np.float64), beta=float(beta), idx=idx, y=y, Xhat=Xhat, LB=LB, sig_pre0=sig_pre0,
                th_pre0=th_pre0, mu=Xi.mean(axis=0))


def timed_run(eng, cfg, pb, n_chains, chain_ids, seed, burnin, warmup, steps):
    """init + untimed burn-in / warm-up + `steps` timed transitions on one engine; host wall time around the timed run."""
    rep = lambda v: np.repeat(np.asarray(v)[None], n_chains, axis=0)
    eng.sampler_init(cfg, rep(pb["Xhat"]), rep(pb["sig_pre0"]), rep(pb["th_pre0"]), seed=seed, chain_ids=chain_ids)
    eng.sampler_run(burnin)
    if warmup > 0:
        eng.sampler_run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lf, dev_ms = eng.sampler_run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    slots, _ = eng.sampler_run_stats()
    return el, lf, dev_ms, slots


def cpu_quota():
    """CPU bandwidth limit of this process's cgroup in cores (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            q, p = float(fq.read()), float(fp.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_gradient_rates(pb, mats, band, drift, P, state, threads, max_seconds, leg="c", min_evals=400):
    """Value+gradient evaluations per second per thread count of a CPU restatement of magi_v2.py:308-348: leg "c" = oracle/logpost_c.c
    (C + OpenMP, analytic gradient, each matrix read once per product pair), leg "torch" = oracle/torch_cpu.py (bmm + autograd)."""
    from oracle import magi_oracle as orc
    C_inv, m, K_inv = mats
    pr = orc.Problem(I=pb["I"], mu=pb["mu"], C_inv=orc.band_part(C_inv, band), m=orc.band_part(m, band), K_inv=orc.band_part(K_inv, band),
                     N_ds=pb["N_ds"], obs_idx=pb["idx"], y=pb["y"], beta=pb["beta"], LB=pb["LB"], drift=drift, P=P)
    Xc, spc, tpc = state
    if leg == "c":
        from oracle import logpost_c
        return pr, logpost_c.time_gradients(pr, Xc, spc, tpc, threads, min_evals=min_evals, max_seconds=max_seconds)
    from oracle import torch_cpu
    return pr, torch_cpu.time_gradients(pr, Xc, spc, tpc, threads, min_evals=200, max_seconds=max_seconds)


def alpha_sweep_bench(a, rank, world, dev_index, red_dev):
    """BASELINE configs[3]: 10 datasets (alpha = 0.05 / 0.15, seeds 0-4; tests/golden/seir_alpha_sweep.npz holds the vignette thinning of the
    reference's data/*.csv) x 8 chains, N = 161, b = 80, reference defaults (stale cache as the reference).  Whole datasets are dealt to the
    ranks, each rank builds its datasets' matrices on its GPU (untimed set-up, one handle per dataset), the timed region is exactly K
    transitions of every chain, the samples of all 80 units are gathered once to rank 0.  Total work is fixed: "scaling": "strong"."""
    from magi_v2_amd.sweep import SweepRunner, alpha_sweep_datasets
    cpd = 8
    datasets = alpha_sweep_datasets(os.path.join(ROOT, "tests", "golden", "seir_alpha_sweep.npz"))
    if world > len(datasets):
        raise SystemExit("alpha-sweep: at most one rank per dataset")
    t0 = time.perf_counter()
    run = SweepRunner(dev_index, datasets, cpd, rank, world, bandsize=80)
    build_ms = (time.perf_counter() - t0) * 1e3
    run.init(a.seed, num_results=a.warmup + a.steps, num_burnin_steps=a.burnin)
    run.run(a.burnin)
    if a.warmup > 0:
        run.run(a.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lf = run.run(a.steps)
    torch.cuda.synchronize()
    own_s = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    lfs = torch.tensor([float(lf)], dtype=torch.float64, device=red_dev)
    mine = torch.tensor([own_s * 1e3, float(lf), float(len(run.unit_ids))], dtype=torch.float64, device=red_dev)
    allr = [mine]
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(lfs, op=dist.ReduceOp.SUM)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    elapsed, lf_total = float(tmax.item()), float(lfs.item())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    allsamp, gids = run.gather(dst=0)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t1) * 1e3 if world > 1 else 0.0
    kernel = run.engines[0].stream_kernel_name(cpd)
    names = [n for n, _ in datasets]
    run.close()
    if rank != 0:
        return
    units = len(datasets) * cpd
    P = 3
    th = np.log1p(np.exp(allsamp[:, a.warmup:, -P:]))                        # [units, steps, P]
    per_ds = th.reshape(len(datasets), cpd * a.steps, P).mean(axis=1)
    out = {
        "metric": "HMC samples/sec (whole node) on SEIR, N grid pts x D comps",
        "value": round(units * a.steps / elapsed, 4), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "reference data files (vignette thinning of data/SEIR_*alpha=*_seed=*.csv, tests/golden/seir_alpha_sweep.npz)",
        "config": {"workload": f"SEIR alpha-sweep: {len(datasets)} datasets (alpha=0.05/0.15, seeds 0-4) x {cpd} chains, N=161 x 4 components, band 80, "
                               "NUTS(max depth 10)+dual averaging+log annealing, reference defaults",
                   "baseline_config": "BASELINE configs[3] (alpha-sweep: 10 datasets x 8 chains sharded by dataset)",
                   "units_total": units, "datasets": names, "chains_per_dataset": cpd, "burnin_untimed": a.burnin,
                   "parallelism": f"datasets x{world} (whole datasets per GPU, round-robin; one handle per dataset, a rank's datasets run concurrently)",
                   "stale_cache": 1, "kernel": kernel},
        "leapfrogs_per_s": round(lf_total / elapsed, 1),
        "per_rank_ms": [round(float(x[0].item()), 2) for x in allr], "per_rank_leapfrogs": [int(x[1].item()) for x in allr],
        "per_rank_units": [int(x[2].item()) for x in allr],
        "rank_balance": round(float(np.mean([x[0].item() for x in allr]) / max(x[0].item() for x in allr)), 4),
        "gathered_unit_ids": [int(g_) for g_ in gids], "gather_ms": round(gather_ms, 3), "build_ms_rank0": round(build_ms, 1),
        "theta_mean_per_dataset": [[round(float(x), 4) for x in row] for row in per_ds],
        "theta_last_per_unit": [[round(float(x), 10) for x in th[u, -1]] for u in range(units)],
        "roofline": None, "cpu_baseline": None,
        "note": "secondary configuration of bench.py (the driver's N = 1 line is --config headline, which carries roofline and cpu_baseline; "
                "its cfg4_* scalars time one dataset of this sweep on one GPU)",
    }
    print(json.dumps(out))


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

    if a.config == "alpha-sweep":
        alpha_sweep_bench(a, rank, world, dev_index, red_dev)
        if world > 1:
            dist.destroy_process_group()
        return

    from magi_v2_amd import host
    from magi_v2_amd.engine import MagiEngine
    from magi_v2_amd.shard import chain_ids_for_rank, family_chains_for, gather_samples

    N, D, P = a.grid, 4, 3
    cpg = a.chains_per_gpu
    band = None if a.band < 0 else a.band

    # ---- setup (untimed) -----------------------------------------------------------------------
    I, X_obs, truth, theta_true = host.synthetic_seir(N, seed=0)
    pb = setup_problem(host, I, X_obs, P)
    eng = MagiEngine(dev_index)
    t0 = time.perf_counter()
    want_host = (rank == 0 and world == 1 and not a.no_cpu_baseline)
    mats = eng.build_matrices(I, pb["hp"]["phi1s"], pb["hp"]["phi2s"], 2.01, bandsize=band, want_host=want_host)
    build_ms = (time.perf_counter() - t0) * 1e3
    eng.set_problem(pb["mu"], pb["N_ds"], pb["idx"], pb["y"], pb["beta"], pb["LB"], "seir4")

    # (head room behind the timed steps: the in-sampler kernel-duration leg continues the same chains for a few hundred slots)
    head = 8
    # stale_cache=0: the reference's annealed kernel reuses the previous step's cached target, which
    # was computed at the previous temperature (SURVEY section 7 quirk ii); on this synthetic grid the
    # log posterior is positive, which makes that offset reject every proposal, so the bench runs
    # the recomputing variant (same arithmetic per leapfrog, see DESIGN.md "stale cache").
    cfg = eng.default_cfg(num_results=a.warmup + a.steps + head, num_burnin_steps=a.burnin, stale_cache=0)
    rep = lambda v: np.repeat(np.asarray(v)[None], cpg, axis=0)
    unit_ids = chain_ids_for_rank(rank, world, cpg * world)              # which (dataset, chain) units this rank owns
    chain_ids = list(range(cpg)) if a.replicate_chains else unit_ids     # the Philox streams they are sampled with
    eng.set_option("family_chains", family_chains_for(cpg * world, world))      # (one kernel family on every rank; a no-op for this even shard)
    eng.sampler_init(cfg, rep(pb["Xhat"]), rep(pb["sig_pre0"]), rep(pb["th_pre0"]), seed=a.seed, chain_ids=chain_ids)
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
    own_s = time.perf_counter() - t0                     # this rank's own K transitions (the job's time is the slowest rank's)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    slots_issued, graphs = eng.sampler_run_stats()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    lfs = torch.tensor([float(lf)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(lfs, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    lf_total = float(lfs.item())
    per_rank = [[own_s * 1e3, float(lf)]]
    if world > 1:                                         # every rank's own time and leapfrog count: NUTS chains build trees of different sizes,
        mine = torch.tensor([own_s * 1e3, float(lf)], dtype=torch.float64, device=red_dev)      # and K transitions per chain is the protocol
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x[0].item()), float(x[1].item())] for x in allr]

    # ---- final sample gather over RCCL (the only collective of the job) -------------------------------
    Xs, sp, tp = eng.sampler_samples()
    keep = a.warmup + a.steps                                              # (the head-room rows behind them were never sampled)
    flat = np.concatenate([Xs.reshape(cpg, Xs.shape[1], -1), sp, tp], axis=2)[:, :keep]     # [chains, results, N*D + D + P]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    allsamp, gids = gather_samples(flat, unit_ids, dst=0)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t1) * 1e3 if world > 1 else 0.0
    th_all = allsamp[:, a.warmup:, -P:] if rank == 0 else None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_chains = cpg * world
    value = n_chains * a.steps / elapsed
    diag = eng.sampler_diag()
    post = diag.tree_depth[:, a.burnin + a.warmup:a.burnin + a.warmup + a.steps]
    lf_rank0 = float(lf)
    state = eng.sampler_state()          # before the profiling / timing launches clobber it

    # ---- in-sampler kernel durations (HIP events attached to every launch of the streaming kernel and of k_point: the kernels'
    #      own begin / end stamps, decisions riding along, on the chains the timed region just ran) ----
    st_us, pt_us, prof_lf = eng.sampler_profile(a.profile_slots)

    # ---- fixed-L HMC (L = 32, step size from a 100-step dual-averaging warm-up), SURVEY 8d: reported next to NUTS ----
    hsteps = min(a.steps, 50)
    hcfg = eng.default_cfg(num_results=hsteps, num_burnin_steps=100, stale_cache=0, mode=1, hmc_leapfrogs=32)
    eng.sampler_init(hcfg, rep(pb["Xhat"]), rep(pb["sig_pre0"]), rep(pb["th_pre0"]), seed=a.seed, chain_ids=chain_ids)
    eng.sampler_run(100)
    torch.cuda.synchronize()
    th0 = time.perf_counter()
    hlf, _ = eng.sampler_run(hsteps)
    torch.cuda.synchronize()
    hmc_s = time.perf_counter() - th0
    hd = eng.sampler_diag()
    hmc = {"samples_per_s": round(cpg * hsteps / hmc_s, 2), "leapfrogs_per_s": round(hlf / hmc_s, 1), "L": 32,
           "accept_rate": round(float(hd.is_accepted[:, 100:].mean()), 3), "step_size": float(hd.step_size[0, -1]), "scope": "rank 0"}

    # ---- roofline of the dominant kernel (the streaming kernel) -------------------------------------------------
    # `achieved` / `frac`: the bytes the kernel MOVES per launch (magi_gradient_bytes: the packed operator blocks + what it stores;
    # the PMC passes of profiles/ count the same bytes at the memory side -- `traffic`) over the kernel's mean IN-SAMPLER launch
    # duration (st_us above; profiles/ holds the rocprofv3 kernel trace of this command, whose mean for the same kernel must
    # agree), against the 8 TB/s HBM peak.  SURVEY 8d's ALGORITHMIC byte count of one gradient evaluation (3 D N W 8 + C 10 N D 8)
    # is a third larger than what this formulation must move -- the symmetric operators FH and FK are stored as their lower block
    # triangle and FE serves both FE xc and FE^T f -- so the fraction on that count (`frac_contract_bytes`, rounds 1-3's `frac`) can
    # exceed 1 and is reported next to it, not as the kernel's fraction of the peak.  frac_of_ceiling prices the kernel against a
    # load-only pass over the same blocks timed in this run; slot_frac_of_load_only = that pass over the whole leapfrog slot
    # [stream, point] as the sampler ran it -- the number the per-leapfrog work is actually chasing.
    grad_ms, phase_ms = eng.time_gradient(cpg, 300)
    phase_bytes = eng.gradient_bytes(cpg)
    W = N if band is None or 6 * band + 1 >= N else 2 * band + 1
    algorithmic = 3.0 * D * N * W * 8.0 + cpg * 10.0 * N * D * 8.0
    t_stream = st_us * 1e-6
    t_alone = phase_ms[4] * 1e-3
    slot_s = elapsed / max(slots_issued, 1) if world == 1 else elapsed / (lf_total / n_chains)
    n_tasks = eng.gradient_bytes(1)[4] / (128 * 128 * 8.0 + 2.0 * 128 * 8.0)     # packed 128 x 128 blocks (magi_gradient_bytes, one chain: blocks + 2 x 128 partials each)
    tiles_bytes = n_tasks * 128 * 128 * 8.0
    ceiling = tiles_bytes / (phase_ms[7] * 1e-3) / 1e9
    roofline = {"bound": "hbm",
                "kernel": eng.stream_kernel_name(cpg) + " (single-phase block mat-vecs FH xc, FE xc, FE^T f, FK f over packed 128x128 blocks)",
                "achieved": round(phase_bytes[4] / t_stream / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(phase_bytes[4] / t_stream / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                "frac_basis": "bytes the kernel moves per launch (operator blocks + stores; = the PMC traffic of profiles/) / mean in-sampler launch "
                              f"duration of the kernel (HIP events on every launch over {a.profile_slots} slots of the timed chains) / HBM peak",
                "us_per_launch": round(st_us, 3), "us_per_launch_point": round(pt_us, 3),
                "algorithmic_bytes_per_launch": algorithmic, "bytes_per_launch": phase_bytes[4],
                "achieved_contract_bytes": round(algorithmic / t_stream / 1e9, 1),
                "frac_contract_bytes": round(algorithmic / t_stream / 1e9 / HBM_PEAK_GBPS, 4),
                "frac_contract_note": "SURVEY 8d's algorithmic bytes (three full N x N stacks per component) over the same time: the kernel moves 0.72 of "
                                      "them (symmetric block storage), so this ratio is not a fraction of the peak and may exceed 1",
                "streamed_GBps": round(phase_bytes[4] / t_stream / 1e9, 1),
                "frac_bytes_moved": round(phase_bytes[4] / t_stream / 1e9 / HBM_PEAK_GBPS, 4),
                "ceiling_GBps": round(ceiling, 1),
                "frac_of_ceiling": round((phase_bytes[4] / t_stream / 1e9) / ceiling, 4),
                "slot_frac_of_load_only": round(phase_ms[7] * 1e-3 / slot_s, 4),
                "slot_frac": round(algorithmic / slot_s / 1e9 / HBM_PEAK_GBPS, 4),
                "slot_frac_bytes_moved": round((phase_bytes[4] + phase_bytes[6]) / slot_s / 1e9 / HBM_PEAK_GBPS, 4),
                # the same kernel launched back to back WITHOUT its decision workgroups (round-1/2 definition)
                "standalone_us_per_launch": round(phase_ms[4] * 1e3, 3),
                "standalone_frac": round(phase_bytes[4] / t_alone / 1e9 / HBM_PEAK_GBPS, 4),
                "standalone_frac_contract_bytes": round(algorithmic / t_alone / 1e9 / HBM_PEAK_GBPS, 4),
                "standalone_point_us": round(phase_ms[6] * 1e3, 3), "read_only_us": round(phase_ms[7] * 1e3, 3),
                "three_phase_gradient_eval_us": round(grad_ms * 1e3, 3),
                "working_set_MB": round(phase_bytes[4] / 1e6, 1),
                "working_set": ("operator blocks resident in the 256 MiB Infinity Cache between launches: the HBM peak is the contract's yardstick, "
                                "not the physical source of the bytes" if phase_bytes[4] < 200e6 else "larger than the 256 MiB Infinity Cache: streamed from HBM")}

    # measured memory-side traffic of the same kernel on the same workload, from the committed PMC passes
    # (rocprofv3 cannot run inside the timed process; profiles/README.md has the commands)
    try:
        import csv
        if N == 1024 and cpg == 1 and (band is None or 6 * band + 1 >= N):
            prof = os.path.join(ROOT, "profiles")
            name = next(f for f in ("r04_bench_pmc_traffic.csv", "r03_bench_pmc_traffic.csv", "r02_bench_pmc_traffic.csv", "r01_bench_pmc_traffic.csv")
                        if os.path.exists(os.path.join(prof, f)))
            with open(os.path.join(prof, name)) as fh:
                tr = sum(float(r["bytes_per_launch_corrected"]) for r in csv.DictReader(fh) if "k_stream<1, 1>" in r["Kernel_Name"])
            if tr > 0:
                roofline["traffic"] = tr
                roofline["traffic_source"] = f"profiles/{name} (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
    except (OSError, KeyError, ValueError, StopIteration):
        pass

    # ---- CPU baseline: the C + OpenMP restatement of magi_v2.py:308-348 (oracle/logpost_c.c), timed at several thread
    #      counts on this host; samples/s = gradient evaluations/s over the leapfrogs per sample the GPU chain needed in the timed
    #      region.  Second leg: the torch-CPU restatement (bmm + autograd, what XLA-CPU executes for the reference).  Third leg: the
    #      numpy oracle continuing the SAME chain. ----
    cpu = None
    ncpu = os.cpu_count() or 1
    quota = cpu_quota()
    usable = int(min(ncpu, len(os.sched_getaffinity(0)), max(1.0, 2 * quota) if quota else ncpu))       # (OpenMP barriers spin: far past the quota they stall)
    threads = sorted({min(usable, usable if t == "all" else int(t)) for t in a.cpu_threads.split(",")})
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
    except (OSError, StopIteration):
        cpu_model = "unknown"
    lf_per_sample = lf_total / (n_chains * a.steps)
    if world == 1 and not a.no_cpu_baseline:
        from oracle import magi_oracle as orc
        import threadpoolctl
        Xc, spc, tpc, ss, bc = state
        pr, rates = cpu_gradient_rates(pb, mats, band, "seir4", P, (Xc[0], spc[0], tpc[0]), threads, 4.0, leg="c")
        best_t = max(rates, key=rates.get)
        t_threads = sorted({t for t in threads if t in (8, 32)} or {threads[-1]})
        _, t_rates = cpu_gradient_rates(pb, mats, band, "seir4", P, (Xc[0], spc[0], tpc[0]), t_threads, 4.0, leg="torch")
        cpu = {"value": round(rates[best_t] / lf_per_sample, 5), "unit": "samples/s", "cores": best_t, "kind": "port",
               "sample": f">= 400 value+gradient evaluations (or 4 s) per thread count of oracle/logpost_c.c -- C + OpenMP restatement of "
                         f"magi_v2.py:308-348 with the analytic gradient, dense matrices as the reference multiplies them, each read once per "
                         f"product pair -- at the GPU chain's state; converted with the {lf_per_sample:.1f} leapfrogs per sample of the timed "
                         "GPU region; the best thread count is reported",
               "gradient_evals_per_s": {str(t): round(r, 2) for t, r in rates.items()}, "leapfrogs_per_s": round(rates[best_t], 2),
               "torch_leg": {"gradient_evals_per_s": {str(t): round(r, 2) for t, r in t_rates.items()},
                             "samples_per_s": round(max(t_rates.values()) / lf_per_sample, 5),
                             "sample": "oracle/torch_cpu.py (fp64 bmm + autograd, the shape of what XLA-CPU executes for the reference): "
                                       ">= 200 evaluations or 4 s per thread count"},
               "host_cpus": ncpu, "host_cpu_quota": quota, "cpu_model": cpu_model}
        # second leg: the numpy oracle (restated TFP NUTS) continues the same chain for a few transitions
        q = orc.pack(Xc[0], spc[0], tpc[0])
        fn_L = orc.make_fn_L(pr)
        L, gL = fn_L(q)
        k = a.burnin + a.warmup + a.steps
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
        del pr

    # ---- the other BASELINE configs on this one GPU (flat scalars inside `roofline`: the driver keeps scalars only) ----
    extra_note = None
    if world == 1 and not a.no_extra_configs and N == 1024 and cpg == 1 and band is None:
        extra_note = extra_configs(a, eng, host, MagiEngine, pb, roofline, threads, dev_index)
    eng.close()

    out = {
        "metric": "HMC samples/sec (whole node) on SEIR, N grid pts x D comps",
        "value": round(value, 4), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SEIR N={N} x 4 components, {'dense' if band is None else 'band ' + str(band)}, "
                               f"NUTS(max depth 10)+dual averaging+log annealing, {cpg} chain(s)/GPU",
                   "baseline_config": ("BASELINE configs[1] (SEIR N=1024 x 4, 1 chain, 1 GPU)" if cpg == 1 and world == 1 else
                                       f"weak scaling of BASELINE configs[1]: one chain per GPU x {world} GPUs" if cpg == 1 else
                                       f"BASELINE configs[2] ({cpg * world} independent chains, {cpg} per GPU x {world} GPUs)"),
                   "grid": N, "components": D, "thetas": P, "chains_total": n_chains, "bandsize": band,
                   "burnin_untimed": a.burnin,
                   "burnin_note": "untimed transitions that adapt the step size before the timed region (the reference burns in 1000): DESIGN.md section 5",
                   "parallelism": f"chains x{world}",
                   "stale_cache": 0, "stale_cache_note": "the reference's annealed kernel reuses the previous step's cached target (computed at the "
                                   "previous temperature); on this synthetic grid that offset rejects every proposal, so the bench recomputes at "
                                   "the current temperature -- same arithmetic per leapfrog (DESIGN.md 4.2)",
                   "chain_streams": "DIAGNOSTIC: every rank replicates rank 0's chain ids" if a.replicate_chains else
                                    "independent: Philox stream = global chain id (rank * chains_per_gpu + local index)"},
        "roofline": roofline, "cpu_baseline": cpu,
        "leapfrogs_per_s": round(lf_total / elapsed, 1),
        "leapfrogs_per_s_per_gpu": round(lf_total / elapsed / world, 1),
        "per_rank_ms": [round(x[0], 2) for x in per_rank], "per_rank_leapfrogs": [int(x[1]) for x in per_rank],
        "rank_balance": round(float(np.mean([x[0] for x in per_rank]) / max(x[0] for x in per_rank)), 4),
        "rank_balance_note": "mean / max of the ranks' own times for their K NUTS transitions: chains build trees of different sizes (different leapfrog counts per "
                             "rank above), the job ends with the slowest chain, and there is no communication inside the timed region -- this ratio, not the "
                             "interconnect, is what an N-GPU efficiency below 1 consists of (fixed-L HMC, hmc_L32, has no such spread)",
        "us_per_leapfrog_slot": round(elapsed / (lf_total / n_chains) * 1e6, 2),
        "us_per_slot_issued": round(elapsed / max(slots_issued, 1) * 1e6, 3), "slots_issued_rank0": int(slots_issued),
        "slot_overhead_rank0": round(slots_issued / max(float(diag.leapfrogs_taken[:, a.burnin + a.warmup:a.burnin + a.warmup + a.steps].sum(axis=1).max()), 1.0), 4),
        "slot_note": "us_per_leapfrog_slot = elapsed / (leapfrogs per chain, mean); us_per_slot_issued = elapsed / kernel pairs [stream, point] "
                     "actually issued on rank 0 (device counter); slot_overhead = slots issued / leapfrogs of the busiest chain (skip / set-up slots)",
        "mean_tree_depth": round(float(post.mean()), 2), "device_ms": round(dev_ms, 2), "build_ms": round(build_ms, 1),
        "leapfrogs_per_sample": round(lf_per_sample, 1),
        "gather_ms": round(gather_ms, 3), "hmc_L32": hmc,
        "theta_mean": [round(float(x), 4) for x in np.log1p(np.exp(th_all)).reshape(-1, P).mean(axis=0)],
    }
    if n_chains <= 64:       # what the final gather delivered: global unit ids in order, and each chain's last theta (distinct chains differ)
        out["gathered_unit_ids"] = [int(g_) for g_ in gids]
        out["theta_last_per_chain"] = [[round(float(x), 10) for x in np.log1p(np.exp(th_all[c, -1]))] for c in range(n_chains)]
    if extra_note:
        out["extra_configs"] = extra_note
    if cpu:
        out["speedup_vs_cpu_port"] = round(value / cpu["value"], 1)          # against the BEST thread count of the C + OpenMP leg
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def extra_configs(a, eng, host, MagiEngine, pb2, roofline, threads, dev_index):
    """BASELINE configs 1, 3 (per-GPU share), 4 (one dataset) and 5 on the same GPU, after the headline region.  Results go into
    `roofline` as flat scalars; returns a short description of what was run."""
    D, P = 4, 3
    note = {}
    rep8 = lambda v: np.repeat(np.asarray(v)[None], 8, axis=0)

    # ---- config 3's per-GPU share: 8 independent chains at N = 1024 (global ids 0..7 = rank 0 of 8) ----
    steps3 = 200                        # (the chains of a run end at different times and the run ends with the slowest: 29 % of the chain-slots idle over 50 transitions, 16 % over 200)
    cfg3 = eng.default_cfg(num_results=steps3 + 3 + 8, num_burnin_steps=a.burnin, stale_cache=0)
    el, lf3, dev_ms, slots = timed_run(eng, cfg3, pb2, 8, list(range(8)), a.seed, a.burnin, 3, steps3)
    st_us, pt_us, _ = eng.sampler_profile(a.profile_slots)
    pbytes = eng.gradient_bytes(8)
    alg8 = 3.0 * D * 1024 * 1024 * 8.0 + 8 * 10.0 * 1024 * D * 8.0
    _, ph8 = eng.time_gradient(8, 200)
    roofline.update({
        "mc8_samples_per_s": round(8 * steps3 / el, 2), "mc8_leapfrogs_per_s": round(lf3 / el, 1),
        "mc8_us_per_slot": round(el / max(slots, 1) * 1e6, 3), "mc8_slots_issued": int(slots),
        "mc8_stream_us": round(st_us, 3), "mc8_point_us": round(pt_us, 3), "mc8_stream_us_standalone": round(ph8[4] * 1e3, 3),
        "mc8_kernel": eng.stream_kernel_name(8),
        "mc8_frac": round(pbytes[4] / (st_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
        "mc8_bytes_per_launch": pbytes[4], "mc8_point_bytes_per_launch": pbytes[6],
        "mc8_frac_contract_bytes": round(alg8 / (st_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
        "mc8_frac_bytes_moved": round(pbytes[4] / (st_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
        "mc8_frac_of_ceiling": round(ph8[7] * 1e3 / st_us, 4),
        "mc8_slot_frac_of_load_only": round(ph8[7] * 1e-3 / (el / max(slots, 1)), 4),
        "mc8_mfma_frac": round(8 * 8.0 * D * 1024 * 1024 / (st_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "mc8_timed_transitions": steps3})
    note["config3_share"] = f"8 chains (ids 0..7), N=1024 x 4 dense, {a.burnin} untimed burn-in + 3 warm-up, {steps3} timed NUTS transitions; {eng.stream_kernel_name(8)} durations from {a.profile_slots} event-timed slots"

    # ---- config 1: SEIR-4, N = 161 (81 thinned rows of data/SEIR_seed=0.csv, discretization 1), b = 80, 1 chain, 200 + 200 ----
    g3 = np.load(os.path.join(ROOT, "tests", "golden", "g3_pipeline.npz"))
    I1 = g3["seir4_I"][:, 0]
    X1 = g3["seir4_X_obs_discret"]
    pb1 = setup_problem(host, I1, X1, P)
    e1 = MagiEngine(dev_index)
    mats1 = e1.build_matrices(I1, pb1["hp"]["phi1s"], pb1["hp"]["phi2s"], 2.01, bandsize=80, want_host=not a.no_cpu_baseline)
    e1.set_problem(pb1["mu"], pb1["N_ds"], pb1["idx"], pb1["y"], pb1["beta"], pb1["LB"], "seir4")
    cfg1 = e1.default_cfg(num_results=200, num_burnin_steps=200)           # reference defaults (stale cache as the reference)
    el1, lf1, _, slots1 = timed_run(e1, cfg1, pb1, 1, [0], a.seed, 200, 0, 200)
    roofline.update({"cfg1_samples_per_s": round(200 / el1, 2), "cfg1_leapfrogs_per_s": round(lf1 / el1, 1),
                     "cfg1_us_per_slot": round(el1 / max(slots1, 1) * 1e6, 3), "cfg1_leapfrogs_per_sample": round(lf1 / 200, 1)})
    if not a.no_cpu_baseline:
        st = e1.sampler_state()
        _, r1 = cpu_gradient_rates(pb1, mats1, 80, "seir4", P, (st[0][0], st[1][0], st[2][0]), [t for t in threads if t <= 8] or [1], 3.0, min_evals=20000)
        bt = max(r1, key=r1.get)
        roofline.update({"cfg1_cpu_samples_per_s": round(r1[bt] / max(lf1 / 200, 1e-9), 3), "cfg1_cpu_cores": bt,
                         "cfg1_cpu_gradient_evals_per_s": round(r1[bt], 1),
                         "cfg1_speedup_vs_cpu_port": round((200 / el1) / (r1[bt] / max(lf1 / 200, 1e-9)), 1)})
    note["config1"] = "SEIR-4 N=161 (tests/golden/g3_pipeline.npz: the vignette thinning of data/SEIR_seed=0.csv), b=80, 1 chain, 200 burn-in + 200 timed NUTS samples, reference defaults; CPU: gradient rate of oracle/logpost_c.c (best of <= 8 threads) / leapfrogs per sample"

    # ---- config 4: one dataset of the alpha sweep (alpha = 0.15, seed 0) x 8 chains, N = 161, b = 80 ----
    sweep = np.load(os.path.join(ROOT, "tests", "golden", "seir_alpha_sweep.npz"))
    rows = sweep["alpha=0.15_seed=0"]
    I4, X4 = host.discretize(rows[:, 0], np.clip(rows[:, 1:5], 0.0, None), 1)
    pb4 = setup_problem(host, np.asarray(I4).reshape(-1), X4, P)
    e1.build_matrices(pb4["I"], pb4["hp"]["phi1s"], pb4["hp"]["phi2s"], 2.01, bandsize=80, want_host=False)
    e1.set_problem(pb4["mu"], pb4["N_ds"], pb4["idx"], pb4["y"], pb4["beta"], pb4["LB"], "seir4")
    cfg4 = e1.default_cfg(num_results=100, num_burnin_steps=200)
    el4, lf4, _, slots4 = timed_run(e1, cfg4, pb4, 8, list(range(8)), a.seed, 200, 0, 100)
    roofline.update({"cfg4_samples_per_s": round(8 * 100 / el4, 2), "cfg4_leapfrogs_per_s": round(lf4 / el4, 1),
                     "cfg4_us_per_slot": round(el4 / max(slots4, 1) * 1e6, 3)})
    note["config4_unit"] = "one dataset (alpha=0.15, seed 0; tests/golden/seir_alpha_sweep.npz) x 8 chains, N=161, b=80, 200 burn-in + 100 timed samples per chain"
    e1.close()

    # ---- config 5: N = 8192 x 4 -- one pooled build (wall time), one build with per-class device times, the streaming kernel ----
    N5 = 8192
    I5, X5, _, _ = host.synthetic_seir(N5, seed=0)
    pb5 = setup_problem(host, I5, X5, P)
    e5 = MagiEngine(dev_index)
    e5.build_matrices(I5, pb5["hp"]["phi1s"], pb5["hp"]["phi2s"], 2.01, want_host=False)          # (allocates the pooled work space)
    torch.cuda.synchronize()
    tb = time.perf_counter()
    e5.build_matrices(I5, pb5["hp"]["phi1s"], pb5["hp"]["phi2s"], 2.01, want_host=False)
    build_s = time.perf_counter() - tb
    pw = e5.build_profile()["potrf_wall"]                           # the two factorisations of THAT build as a whole (look-ahead on), device clock
    e5.set_option("build_profile", 1)
    e5.build_matrices(I5, pb5["hp"]["phi1s"], pb5["hp"]["phi2s"], 2.01, want_host=False)
    prof = e5.build_profile()
    prof.pop("potrf_wall")                                          # (in the serialised build it repeats three of the classes)
    e5.set_option("build_profile", 0)
    fl = lambda k: prof[k][0]
    ms = lambda k: prof[k][1]
    potrf_ms = ms("diag_chol_inv") + ms("potrf_panel") + ms("potrf_trailing_syrk")
    potrf_flop = 8 * N5 ** 3 / 3.0                                  # two factorisations (Kappa, K_d) x 4 components, N^3 / 3 each
    alg = 5.0 * N5 ** 3 * D                                         # chol x2, TRSM x2, SYRK, POTRI x2 per component
    issued = sum(v[0] for v in prof.values())
    prod_ms = ms("m_K_products")
    roofline.update({
        "n8192_build_s": round(build_s, 4), "n8192_build_frac": round(alg / build_s / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "n8192_potrf_ms": round(pw[1], 2), "n8192_potrf_frac": round(potrf_flop / (pw[1] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "n8192_potrf_serialised_ms": round(potrf_ms, 2), "n8192_potrf_serialised_frac": round(potrf_flop / (potrf_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "n8192_products_ms": round(prod_ms, 2), "n8192_products_frac": round(fl("m_K_products") / (prod_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "n8192_operators_frac": round(fl("single_phase_operators") / (ms("single_phase_operators") * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "n8192_trtri_frac": round(fl("trtri") / (ms("trtri") * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "n8192_TtT_frac": round(fl("TtT") / (ms("TtT") * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        "n8192_diag_ms": round(ms("diag_chol_inv"), 2), "n8192_rank_k_ms": round(ms("potrf_trailing_syrk"), 2), "n8192_panel_ms": round(ms("potrf_panel"), 2),
        "n8192_matern_ms": round(ms("matern"), 2),
        "n8192_issued_over_algorithmic": round(issued / alg, 3), "n8192_profiled_build_ms": round(sum(v[1] for v in prof.values()), 1)})
    e5.set_problem(pb5["mu"], pb5["N_ds"], pb5["idx"], pb5["y"], pb5["beta"], pb5["LB"], "seir4")
    lp5, *_ = e5.logpost_grad(pb5["Xhat"], pb5["sig_pre0"], pb5["th_pre0"], 1.0, fused=True)        # (loads a state for the timing launches)
    _, ph5 = e5.time_gradient(1, 20)
    b5 = e5.gradient_bytes(1)
    roofline.update({"n8192_stream_us": round(ph5[4] * 1e3, 1), "n8192_stream_GB": round(b5[4] / 1e9, 3),
                     "n8192_stream_frac_bytes_moved": round(b5[4] / (ph5[4] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "n8192_stream_frac": round(b5[7] / (ph5[4] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "n8192_read_only_frac": round((b5[4] - 2.0 * 128 * 8 * (b5[4] / (128 * 128 * 8.0 + 2.0 * 128 * 8.0))) / (ph5[7] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "n8192_logpost_finite": bool(np.isfinite(lp5))})
    note["config5"] = ("N=8192 x 4 dense: second pooled build timed on the host clock (n8192_build_s; 5 N^3 D algorithmic flops), a third build with "
                       "the per-class profile on (serialised: HIP events around every launch, look-ahead of the factorisations off: n8192_potrf_serialised_*; "
                       "n8192_potrf_ms / _frac are the two factorisations of the SECOND build as a whole, device clock, look-ahead on), then the streaming kernel "
                       "on the 4.4 GB of operator blocks (standalone, 20 launches: nothing is cache-resident at this size)")
    e5.close()

    # ---- f1: GP hyper-parameter fit (magi_v2.py:538-691) -- ms per Adam step; the reference's only timing is this step at |I| ~ 2191 x 4:
    #      about 9 s per iteration on its CPU host (output.log:23) ----
    ef = MagiEngine(dev_index)
    for Nf, iters in ((161, 100), (1024, 40), (2191, 20)):
        If, Xf_obs, _, _ = host.synthetic_seir(Nf, seed=0)
        Xf = host.linear_interpolate(Xf_obs)
        pri = [host.fourier_phi2_prior(Xf[:, d]) for d in range(D)]
        ini = host.hparams_initial(Xf)
        args = (If, Xf, Xf.mean(axis=0), [p_[0] for p_ in pri], [p_[1] for p_ in pri], ini["sigma_sqs"], ini["phi1s"], ini["phi2s"], ini["sigma_sqs"])
        ef.fit_hparams(*args, num_iters=3)                                                   # (allocations, graph capture)
        torch.cuda.synchronize()
        tf = time.perf_counter()
        fit = ef.fit_hparams(*args, num_iters=iters)
        torch.cuda.synchronize()
        roofline[f"f1_ms_per_adam_step_n{Nf}"] = round((time.perf_counter() - tf) / iters * 1e3, 4)
        roofline[f"f1_finite_n{Nf}"] = bool(np.isfinite(fit["phi2s"]).all())
    ef.close()
    note["f1"] = ("GP hyper-parameter fit on the GPU (all 4 components; every Adam step = Matern assembly + Cholesky + inverse + trace terms), wall ms per step incl. "
                  "the per-call set-up amortised over 100 / 40 / 20 steps; the reference logs ~9 s per step at |I| ~ 2191 (output.log:23)")
    return note


if __name__ == "__main__":
    main()
