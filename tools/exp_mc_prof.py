"""Dev: a few timing launches of the streaming kernels for rocprofv3 (kernel trace / PMC).  python tools/exp_mc_prof.py N chains reps"""
import sys
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs)
hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi)
LB = host.sigma_sqs_lower_bound(Xhat)
sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(np.float64), idx, y, beta, LB, "seir4")
rep = lambda v: np.repeat(np.asarray(v)[None], n, axis=0)
eng.logpost_grad(rep(Xhat), rep(sp0), rep(tp0), 1.0, fused=True)
g, ph = eng.time_gradient(n, reps)
print("stream %.2f point %.2f read-only %.2f us" % (ph[4] * 1e3, ph[6] * 1e3, ph[7] * 1e3))
eng.close()
