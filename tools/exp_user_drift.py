"""Dev: generic-drift path (FitzHugh-Nagumo) and a partially observed SEIR through the API on the GPU."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import magi_v2
from magi_v2_amd.drift_examples import fitzhugh_nagumo, rk4
from magi_v2_amd import host
np.set_printoptions(precision=4, linewidth=160)

truth = np.array([0.2, 0.2, 3.0])
ts, X = rk4(fitzhugh_nagumo, [-1.0, 1.0], truth, 20.0, 41)
rng = np.random.default_rng(0)
X_obs = X + rng.normal(0, 0.2, X.shape)
t0 = time.time()
m = magi_v2.MAGI_v2(D_thetas=3, ts_obs=ts, X_obs=X_obs, bandsize=None, f_vec=fitzhugh_nagumo)
print("drift", m.drift.name, "resolve+jit s", round(time.time() - t0, 1))
for fit_on in ("grid", "observed"):
    t0 = time.time()
    m.initial_fit(discretization=2, hparam_iters=300, hparam_fit_on=fit_on)
    print(fit_on, "fit s", round(time.time() - t0, 1), "phi1", m.phi1s, "phi2", m.phi2s, "sig", np.sqrt(m.sigma_sqs_init), "theta_init", m.thetas_init)
    m.thetas_init = np.ones(3)
    res = m.predict(num_results=400, num_burnin_steps=400, n_chains=4, seed=1, stale_cache=False)
    th = res["thetas_samps"].reshape(-1, 3)
    print("  theta mean", th.mean(0), "sd", th.std(0), "minutes", res["minutes_elapsed"], "depth", res["kernel_results"]["tree_depth"].mean())
    Xm = res["X_samps"].mean(axis=(0, 1))
    _, Xt = rk4(fitzhugh_nagumo, [-1.0, 1.0], truth, 20.0, m.mag_I)
    print("  traj rmse", np.sqrt(((Xm - Xt) ** 2).mean(0)))

# partially observed SEIR-3: E never observed
g = np.load("tests/golden/g3_pipeline.npz")
X_obs = g["seir3_X_obs"].copy(); Etrue = X_obs[:, 0].copy(); X_obs[:, 0] = np.nan
m = magi_v2.MAGI_v2(D_thetas=3, ts_obs=g["seir3_ts_obs"], X_obs=X_obs, bandsize=None, f_vec="seir3")
t0 = time.time()
m.initial_fit(discretization=1, hparam_iters=200)
print("unobs fit s", round(time.time() - t0, 1), "theta_init", m.thetas_init, "phi2", m.phi2s, "sig", np.sqrt(m.sigma_sqs_init))
print("E init vs noisy E obs corr", np.corrcoef(m.Xhat_init[::2, 0], Etrue)[0, 1], "rmse", np.sqrt(((m.Xhat_init[::2, 0] - Etrue) ** 2).mean()))
res = m.predict(num_results=300, num_burnin_steps=300, n_chains=4, seed=2, stale_cache=False)
th = res["thetas_samps"].reshape(-1, 3)
print("  theta mean", th.mean(0), "sd", th.std(0), "depth", res["kernel_results"]["tree_depth"].mean())
Em = res["X_samps"].mean(axis=(0, 1))[::2, 0]
print("  E posterior mean vs E obs: corr", np.corrcoef(Em, Etrue)[0, 1], "rmse", np.sqrt(((Em - Etrue) ** 2).mean()), "scale", Etrue.max())
