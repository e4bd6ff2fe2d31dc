"""The one optional probe of VERDICT r2 (next #9): the reference-DEFAULT path (fitted hyper-parameters + the reference's theta
initialiser) on the NOISE-FREE `*_true` columns of the vignette's thinned rows (g3_pipeline.npz rows[:, 6:9]) -- the stored notebook
cell that loads the data carries the comment "let's try using the truth instead of _obs" (vignette.ipynb:111), so the printed
(5.831, 0.565, 1.77) may come from such a run.  4 chains x (1000 + 1000), bandsize 80, discretization 1.
    python tools/exp_recovery_true.py > profiles/r03_recovery_true_columns.json"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import magi_v2

B, R, NCH = 1000, 1000, 4
g = np.load("tests/golden/g3_pipeline.npz")
ts, Xtrue = g["seir3_ts_obs"], g["rows"][:, 6:9]
out = {"burnin": B, "results": R, "chains": NCH, "reference_printed": [5.831, 0.565, 1.77], "truth": [6.0, 0.6, 1.8], "data": "rows[:, 6:9] = E_true, I_true, R_true (no noise)", "runs": []}
for label, fit_kw, theta_init in (("default path on the true columns", dict(), None), ("default fit on the true columns, theta_init = 1", dict(theta_init_iters=0), np.ones(3))):
    m = magi_v2.MAGI_v2(3, ts, Xtrue, 80, "seir3")
    t0 = time.time()
    m.initial_fit(1, **fit_kw)
    fit_s = time.time() - t0
    th_fitted = np.array(m.thetas_init)
    if theta_init is not None:
        m.thetas_init = theta_init
    res = m.predict(R, B, n_chains=NCH, seed=123)
    th = res["thetas_samps"].reshape(NCH, R, 3)
    nb = 20
    bm = th[:, : R // nb * nb].reshape(NCH, nb, -1, 3).mean(axis=2)
    rec = {"label": label, "phi1s": m.phi1s.tolist(), "phi2s": m.phi2s.tolist(), "sigma_init": np.sqrt(m.sigma_sqs_init).tolist(),
           "thetas_init_fitted": th_fitted.tolist(), "thetas_init_used": np.asarray(m.thetas_init).tolist(),
           "theta_mean": th.reshape(-1, 3).mean(axis=0).tolist(), "theta_mcse": (bm.reshape(-1, 3).std(axis=0, ddof=1) / np.sqrt(NCH * nb)).tolist(),
           "theta_mean_per_chain": th.mean(axis=1).tolist(), "fit_s": round(fit_s, 2)}
    out["runs"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
    m.engine.close()
print(json.dumps(out, indent=1))
