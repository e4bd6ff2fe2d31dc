"""Dev: time line of one k_point workgroup (needs a -DMAGI_PT_STAMPS=<workgroup> build), inside a running chain."""
import sys
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = 1024
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
cfg = eng.default_cfg(num_results=60, num_burnin_steps=40, stale_cache=0)
eng.sampler_init(cfg, Xhat, sp0, tp0, seed=1)
eng.sampler_run(40)
rows = []
for rep in range(40):
    eng.sampler_run(1)
    rows.append(eng.debug_par(0)[40:48].copy().view(np.uint64).astype(np.int64))
rows = np.array(rows)
rel = (rows - rows[:, :1]) * 10.0
names = ["entry (plan + flag arrived)", "product sums in LDS", "barrier 1", "finish done", "barrier 2", "end"]
med = np.median(rel, axis=0)
print("k_point workgroup time line inside the sampler (ns from entry, median of %d):" % len(rows))
for k, v in zip(names, med[:6]):
    print("  %-30s %7.0f" % (k, v))
