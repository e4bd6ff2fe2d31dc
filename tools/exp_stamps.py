"""Dev: where does a leaf's tail time go?  Needs a -DMAGI_TAIL_STAMPS build."""
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
sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
cfg = eng.default_cfg(num_results=30, num_burnin_steps=30, stale_cache=0)
eng.sampler_init(cfg, Xhat, sp, tp, seed=1)
eng.sampler_run(30)
acc = []
for rep in range(40):
    eng.sampler_run(1) if False else None
import time
# sample stamps repeatedly while the chain runs one step at a time
rows = []
for rep in range(25):
    eng.sampler_run(1)
    p = eng.debug_par(0)[40:48]
    rows.append(np.diff(p) * 10.0)          # 100 MHz ticks -> ns
rows = np.array(rows)
print("ns per stage [ctl, uniforms+setup, pass loop, block_sum, scalar, decide, 4a, ctl store]")
print(np.median(rows, axis=0), "total", np.median(rows.sum(axis=1)))
