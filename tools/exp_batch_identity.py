"""Dev: is a chain bit-identical whatever the batch it runs in?  Small problem (every batch on the VALU kernel): chain id 7 alone, in a
pair, in batches of 3 / 5 / 8."""
import sys
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N, band = 161, 80
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, bandsize=band, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
cfg = eng.default_cfg(num_results=20, num_burnin_steps=20, step_size=2e-3, stale_cache=0)
ref = None
for ids in ([7], [7, 3], [3, 7], [1, 7, 2], [0, 1, 2, 3, 7], [7, 6, 5, 4, 3, 2, 1, 0]):
    n = len(ids); rep = lambda v: np.repeat(np.asarray(v)[None], n, axis=0)
    eng.sampler_init(cfg, rep(Xhat), rep(sp), rep(tp), seed=5, chain_ids=ids)
    eng.sampler_run(40)
    X, s, t = eng.sampler_samples()
    k = ids.index(7)
    if ref is None: ref = (X[k].copy(), t[k].copy())
    print(ids, "bit-identical to the chain run alone:", bool(np.array_equal(X[k], ref[0]) and np.array_equal(t[k], ref[1])), " max |dX|", float(np.abs(X[k] - ref[0]).max()))
eng.close()
