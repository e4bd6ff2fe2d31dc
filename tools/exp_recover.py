"""Dev: does the pipeline recover theta on clean / noisy vignette data under various hyper-parameter choices?"""
import sys
import numpy as np
sys.path.insert(0, ".")
import magi_v2
g = np.load("tests/golden/g3_pipeline.npz")
rows = g["rows"]
ts = rows[:, 0]
for label, X, kw in [
    ("truth data, initial hparams", rows[:, 6:9].copy(), dict(hparam_iters=0)),
    ("truth data, fitted hparams", rows[:, 6:9].copy(), dict()),
    ("noisy data, initial hparams", np.clip(rows[:, 2:5], 0, None), dict(hparam_iters=0)),
    ("noisy data, fitted hparams", np.clip(rows[:, 2:5], 0, None), dict()),
    ("noisy data, phi2=0.5 sig=true", np.clip(rows[:, 2:5], 0, None), dict(hparams={"phi2s": [0.5, 0.5, 0.5], "sigma_sqs": (0.05 * np.ptp(rows[:, 6:9], axis=0)) ** 2})),
]:
    m = magi_v2.MAGI_v2(3, ts, X, 80, "seir3")
    m.initial_fit(1, **kw)
    m.thetas_init = np.ones(3)
    for stale in (True, False):
        res = m.predict(300, 300, n_chains=4, seed=5, stale_cache=stale)
        th = res["thetas_samps"].reshape(-1, 3)
        print(f"{label:34s} stale={stale!s:5s} phi2={np.round(m.phi2s,3)} sig={np.round(np.sqrt(m.sigma_sqs_init),4)} theta mean {np.round(th.mean(0),3)} sd {np.round(th.std(0),3)}", flush=True)
    m.engine.close()
