import sys, time
import numpy as np
sys.path.insert(0, ".")
import magi_v2
from magi_v2_amd.drift_examples import fitzhugh_nagumo, rk4
np.set_printoptions(precision=4, linewidth=160)
truth = np.array([0.2, 0.2, 3.0])
ts, X = rk4(fitzhugh_nagumo, [-1.0, 1.0], truth, 20.0, 41)
rng = np.random.default_rng(0)
X_obs = X + rng.normal(0, 0.2, X.shape)
m = magi_v2.MAGI_v2(D_thetas=3, ts_obs=ts, X_obs=X_obs, bandsize=None, f_vec=fitzhugh_nagumo)
for disc, phi2 in ((1, None), (2, None), (2, [2.0, 2.0]), (3, None)):
    hp = {"sigma_sqs": [0.04, 0.04]}
    if phi2 is not None: hp["phi2s"] = phi2
    m.initial_fit(discretization=disc, hparams=hp)
    print("disc", disc, "phi1", m.phi1s, "phi2", m.phi2s, "theta_init", m.thetas_init)
    for init in (np.ones(3), ):
        m.thetas_init = init
        res = m.predict(num_results=500, num_burnin_steps=500, n_chains=4, seed=1, stale_cache=False)
        th = res["thetas_samps"].reshape(-1, 3)
        Xm = res["X_samps"].mean(axis=(0, 1))
        _, Xt = rk4(fitzhugh_nagumo, [-1.0, 1.0], truth, 20.0, m.mag_I)
        print("  theta mean", th.mean(0), "sd", th.std(0), "depth", res["kernel_results"]["tree_depth"].mean(), "traj rmse", np.sqrt(((Xm - Xt) ** 2).mean(0)),
              "sig", np.sqrt(res["sigma_sqs_samps"].reshape(-1, 2).mean(0)))
