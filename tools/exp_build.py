"""Dev: time the GPU matrix build (BASELINE config 5 at N=8192) and a few gradients on it."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]); D = 4
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
for rep in range(2):
    t = time.perf_counter()
    eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
    dt = time.perf_counter() - t
    print("build %d: %.3f s  (5 N^3 D = %.2f TF algorithmic -> %.2f TFLOP/s)" % (rep, dt, 5 * N**3 * D / 1e12, 5 * N**3 * D / 1e12 / dt), flush=True)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
lp = eng.logpost_grad(Xhat, sp, tp, 1.0)[0]
lpf = eng.logpost_grad(Xhat, sp, tp, 1.0, fused=True)[0]
print("logp 3-phase %.10g fused %.10g rel diff %.2e" % (lp, lpf, abs(lp - lpf) / abs(lp)))
tot, ph = eng.time_gradient(1, 20)
by = eng.gradient_bytes(1)
print("kernels us", np.round(ph * 1e3, 1), "GB/s", np.round(by / (ph * 1e-3) / 1e9, 0))
