"""Dev: kernel timings of the bench workload (N=1024 x 4, 1 chain) + slot time, for A/B runs of library variants."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
band = (int(sys.argv[3]) if len(sys.argv) > 3 else 0) or None          # 0 = dense
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
t0 = time.perf_counter()
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, bandsize=band, want_host=False)
print("build s %.2f" % (time.perf_counter() - t0))
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
cfg = eng.default_cfg(num_results=steps, num_burnin_steps=steps, stale_cache=0)
eng.sampler_init(cfg, np.repeat(Xhat[None], nch, 0), np.repeat(sp[None], nch, 0), np.repeat(tp[None], nch, 0), seed=1)
eng.sampler_run(steps)
t0 = time.perf_counter(); lf, ms = eng.sampler_run(steps); dt = time.perf_counter() - t0
print("slot us %.2f  (leapfrogs %d, %.1f samples/s)" % (ms * 1e3 / (lf / nch), lf, nch * steps / dt))
tot, ph = eng.time_gradient(nch, 300 if N <= 2048 else 20)
by = eng.gradient_bytes(nch)
print("stream bytes %.1f MB -> %.2f TB/s" % (by[4] / 1e6, by[4] / (ph[4] * 1e-3) / 1e12))
print("stream %.2f us  point %.2f us  leap_reduce %.2f us" % (ph[4] * 1e3, ph[6] * 1e3, ph[5] * 1e3))
