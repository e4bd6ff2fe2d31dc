"""Dev: spread of the timed region over chain ids (what the N-GPU bench's max-over-ranks sees)."""
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
for burn in (40, 100, 200, 400):
    out = []
    for cid in range(8):
        cfg = eng.default_cfg(num_results=25, num_burnin_steps=burn, stale_cache=0)
        eng.sampler_init(cfg, Xhat, sp, tp, seed=20250103, chain_ids=[cid])
        eng.sampler_run(burn + 5)
        t0 = time.perf_counter(); lf, ms = eng.sampler_run(20); dt = time.perf_counter() - t0
        d = eng.sampler_diag()
        out.append((dt * 1e3 / 20, lf / 20, d.step_size[0, -1]))
    a = np.array(out)
    print("burnin", burn, "ms/step per chain id", np.round(a[:, 0], 2), "leapfrogs/step", np.round(a[:, 1]), "step size", np.round(a[:, 2] * 1e3, 3))
    print("   mean %.2f  max %.2f  -> max/mean %.2f" % (a[:, 0].mean(), a[:, 0].max(), a[:, 0].max() / a[:, 0].mean()))
