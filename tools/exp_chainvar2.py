"""Dev: spread of the timed region over chain ids against the LENGTH of the timed region (burn-in 400)."""
import sys, time
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
burn = 400
res = {20: [], 50: [], 100: [], 200: []}
for cid in range(8):
    cfg = eng.default_cfg(num_results=400, num_burnin_steps=burn, stale_cache=0)
    eng.sampler_init(cfg, Xhat, sp, tp, seed=20250103, chain_ids=[cid])
    eng.sampler_run(burn + 5)
    tot_t = 0.0; tot_lf = 0; done = 0
    for upto in (20, 50, 100, 200):
        t0 = time.perf_counter(); lf, ms = eng.sampler_run(upto - done); tot_t += time.perf_counter() - t0; tot_lf += lf; done = upto
        res[upto].append((tot_t * 1e3 / upto, tot_lf / upto))
    print("chain", cid, [round(res[k][-1][0], 2) for k in res], flush=True)
for k, v in res.items():
    a = np.array(v)
    print("timed %d transitions: ms/step mean %.2f max %.2f max/mean %.3f  min/mean %.3f" % (k, a[:, 0].mean(), a[:, 0].max(), a[:, 0].max() / a[:, 0].mean(), a[:, 0].min() / a[:, 0].mean()))
