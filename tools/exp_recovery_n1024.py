"""Recovered (beta, gamma, sigma) on BASELINE config 2's OWN grid: synthetic SEIR-4, N = 1024 points (dt = 0.025), observations at
the even grid indices with 5 % noise (host.synthetic_seir, PCG64(0)), dense matrices, through the drop-in API.

    python tools/exp_recovery_n1024.py [burnin results chains] > profiles/r04_recovery_n1024.json

Rows: (a) sensible hyper-parameters as DESIGN section 8 row (iv): phi2 = 0.5, noise at its true level, theta_init = 1, recomputed cache
      (b) the reference's STARTING hyper-parameters (hparam_iters = 0), recomputed cache -- what bench.py samples
      (c) the same with the reference's stale cache (magi_v2.py:855-879; the API default): every proposal is rejected on this grid
PARITY UNPINNED: the reference holds no known answer for this grid (its one stored theta-hat is vignette.ipynb:281-283, N = 161)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import magi_v2
from magi_v2_amd import host

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
NCH = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N = 1024
I, X_obs, truth, th_true = host.synthetic_seir(N, seed=0)
true_sd = 0.05 * (truth.max(axis=0) - truth.min(axis=0))
out = {"grid": N, "burnin": B, "results": R, "chains": NCH, "truth": th_true.tolist(), "runs": []}


def run(label, fit_kw, stale, seed=123):
    m = magi_v2.MAGI_v2(3, I, X_obs, None, "seir4")
    m.initial_fit(0, theta_init_iters=0, **fit_kw)
    m.thetas_init = np.ones(3)
    t0 = time.time()
    res = m.predict(R, B, n_chains=NCH, seed=seed, stale_cache=stale)
    samp_s = time.time() - t0
    th = res["thetas_samps"].reshape(NCH, R, 3)
    nb = 20
    bm = th[:, : R // nb * nb].reshape(NCH, nb, -1, 3).mean(axis=2)
    mcse = bm.reshape(-1, 3).std(axis=0, ddof=1) / np.sqrt(NCH * nb)
    kr = res["kernel_results"]
    Xm = res["X_samps"].reshape(NCH * R, N, 4).mean(axis=0)
    rec = {"label": label, "stale_cache": int(stale), "phi1s": m.phi1s.tolist(), "phi2s": m.phi2s.tolist(), "sigma_init": np.sqrt(m.sigma_sqs_init).tolist(),
           "theta_mean": th.reshape(-1, 3).mean(axis=0).tolist(), "theta_sd": th.reshape(-1, 3).std(axis=0).tolist(), "theta_mcse": mcse.tolist(),
           "theta_mean_per_chain": th.mean(axis=1).tolist(),
           "sigma_mean": np.sqrt(res["sigma_sqs_samps"].reshape(-1, 4).mean(axis=0)).tolist(), "true_noise_sd": true_sd.tolist(),
           "trajectory_rmse_vs_truth": np.sqrt(((Xm - truth) ** 2).mean(axis=0)).tolist(),
           "is_accepted": float(np.asarray(kr["is_accepted"]).mean()),
           "mean_depth": float(np.asarray(kr["tree_depth"]).mean()), "mean_leapfrogs": float(np.asarray(kr["leapfrogs_taken"]).mean()),
           "step_size_last": float(np.asarray(kr["step_size"]).reshape(NCH, -1)[0, -1]),
           "target_first": float(np.asarray(kr["target_log_prob"]).reshape(NCH, -1)[0, 0]), "target_last": float(np.asarray(kr["target_log_prob"]).reshape(NCH, -1)[0, -1]),
           "beta_temp_last": float(np.asarray(kr["beta_temp"]).reshape(-1)[-1]), "sample_s": round(samp_s, 2)}
    out["runs"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
    m.engine.close()


run("(a) phi2 = 0.5, true noise level, theta_init = 1", dict(hparams={"phi2s": [0.5] * 4, "sigma_sqs": true_sd ** 2}), False)
run("(b) the reference's starting hyper-parameters, theta_init = 1", dict(hparam_iters=0), False)
run("(c) as (b) with the reference's stale cache", dict(hparam_iters=0), True)
print(json.dumps(out, indent=1))
