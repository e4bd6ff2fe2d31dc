"""Dev experiment: synthetic SEIR-4 grid (BASELINE config 2 shape) with ORACLE-built matrices
(dev only), gradient timing per phase + short sampler run."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import magi_oracle as orc
from tests.util import synthetic_seir_problem
from magi_v2_amd.engine import MagiEngine

N = int(sys.argv[1]); nch = int(sys.argv[2]); nsteps = int(sys.argv[3]); band = int(sys.argv[4]) if len(sys.argv) > 4 else -1
band = None if band < 0 else band
I, X_obs, truth, th = synthetic_seir_problem(N, seed=0)
t = time.time()
Xi = orc.linear_interpolate(X_obs)
hp = orc.hparams_initial(Xi)
C_inv, m, K_inv = orc.build_all(I.reshape(-1, 1), hp["phi1s"], hp["phi2s"], 2.01, None)
print("oracle build %.1fs" % (time.time() - t), "phi2", hp["phi2s"], flush=True)
D = 4
N_ds = (~np.isnan(X_obs)).sum(axis=0).astype(float)
idx = np.where(~np.isnan(X_obs).flatten())[0]
y = X_obs.reshape(-1)[idx]
Xhat = orc.cubic_smoother(I, Xi)
LB = orc.sigma_sqs_lower_bound(Xhat)
eng = MagiEngine(0)
eng.set_matrices(C_inv, m, K_inv, bandsize=band)
eng.set_problem(Xi.mean(axis=0), N_ds, idx, y, D * N / N_ds.sum(), LB, "seir4")
X0, s0, t0 = orc.initial_state(Xhat, hp["sigma_sqs"], np.ones(3), LB)
rep = lambda a: np.repeat(a[None], nch, 0)
lp = eng.logpost_grad(rep(X0), rep(s0), rep(t0), 1.0)[0]
print("logp", lp[0])
tot, ph = eng.time_gradient(nch, 200)
by = eng.gradient_bytes(nch)
print("gradient eval %.2f us; phases us %s" % (tot * 1e3, np.round(ph * 1e3, 2)))
print("phase GB/s", np.round(by / (ph * 1e-3) / 1e9, 1), "bytes MB", np.round(by / 1e6, 2))
cfg = eng.default_cfg(num_results=nsteps, num_burnin_steps=nsteps, stale_cache=0)
eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=1)
t = time.time(); lf, ms = eng.sampler_run(nsteps); dt = time.time() - t
print("burnin: wall %.2fs leapfrogs %d -> %.1f us/slot" % (dt, lf, 1e6 * dt / max(1, lf / nch)))
t = time.time(); lf, ms = eng.sampler_run(nsteps); dt = time.time() - t
d = eng.sampler_diag()
print("sample: wall %.2fs dev %.1f ms leapfrogs %d -> %.1f us/slot, %.2f samples/s; depth %.2f" % (dt, ms, lf, 1e6 * dt / max(1, lf / nch), nch * nsteps / dt, d.tree_depth[:, nsteps:].mean()))
np.set_printoptions(linewidth=200, precision=4)
k = min(40, 2 * nsteps)
print("step_size", d.step_size[0, :k])
print("lar", d.log_accept_ratio[0, :k])
print("lf", d.leapfrogs_taken[0, :k])
print("div", d.has_divergence[0, :k], "acc", d.is_accepted[0, :k])
print("target", d.target_log_prob[0, :k])
print("energy", d.energy[0, :k])
