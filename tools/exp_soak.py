"""Dev: soak run of the sampler (several chains, pause/resume chunks, both modes) with consistency checks."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N, nch = 1024, 8
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
rep = lambda v: np.repeat(np.asarray(v)[None], nch, axis=0)
for mode, kw in (("nuts", {}), ("hmc", dict(mode=1, hmc_leapfrogs=32))):
    cfg = eng.default_cfg(num_results=60, num_burnin_steps=60, stale_cache=0, **kw)
    eng.sampler_init(cfg, rep(Xhat), rep(sp), rep(tp), seed=11)
    t0 = time.perf_counter(); total = 0
    for chunk in (7, 13, 100):
        lf, ms = eng.sampler_run(chunk); total += lf
    dt = time.perf_counter() - t0
    Xs, s_, t_ = eng.sampler_samples(); d = eng.sampler_diag()
    assert np.isfinite(Xs).all() and np.isfinite(s_).all() and np.isfinite(t_).all()
    assert d.leapfrogs_taken.sum() == total, (d.leapfrogs_taken.sum(), total)
    assert (eng.sampler_steps_done() == 120).all()
    thm = np.log1p(np.exp(t_)).reshape(-1, 3).mean(0)
    print(mode, "ok: %.1f s, %d leapfrogs (%.1f k/s), accept %.2f, depth %.2f, theta mean %s" %
          (dt, total, total / dt / 1e3, d.is_accepted[:, 60:].mean(), d.tree_depth[:, 60:].mean(), np.round(thm, 3)))
eng.close()
