"""Dev: slot time of the one-chain sampler at several grid sizes (dense) and of 8 chains at N = 161, b = 80 (the shape of BASELINE configs 1 / 4).
    python tools/exp_slot.py    (MAGI_HIP_LIB selects the library)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
cases = ((161, 80, 1), (161, 80, 8), (512, None, 1), (1024, None, 1))
if len(sys.argv) > 1:        # N:band:chains ...
    cases = tuple((int(a.split(":")[0]), (int(a.split(":")[1]) if a.split(":")[1] != "-" else None), int(a.split(":")[2])) for a in sys.argv[1:])
for N, band, n in cases:
    I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
    Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
    sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
    eng = MagiEngine(0)
    eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, bandsize=band, want_host=False)
    eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
    cfg = eng.default_cfg(num_results=60, num_burnin_steps=60, stale_cache=0)
    rep = lambda v: np.repeat(np.asarray(v)[None], n, axis=0)
    eng.sampler_init(cfg, rep(Xhat), rep(sp), rep(tp), seed=1, chain_ids=list(range(n)))
    eng.sampler_run(60)
    t0 = time.perf_counter(); lf, ms = eng.sampler_run(40); dt = time.perf_counter() - t0
    slots = eng.sampler_run_stats()[0]
    st, pt, _ = eng.sampler_profile(256)
    print(f"N={N} b={band} chains={n}: {dt / max(slots, 1) * 1e6:6.2f} us per slot issued, {lf / dt:9.0f} leapfrogs/s   (stream {st:.2f} us, point {pt:.2f} us in the sampler)", flush=True)
    eng.close()
