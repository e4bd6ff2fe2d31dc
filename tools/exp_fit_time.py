"""Dev: time of the GP hyper-parameter fit (magi_fit_hparams) per Adam step: slope between two iteration counts after a warm-up call."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
for N, iters in ((161, 400), (1024, 40), (2191, 10)):
    I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
    Xi = host.linear_interpolate(X_obs)
    pri = [host.fourier_phi2_prior(Xi[:, d]) for d in range(4)]
    init = host.hparams_initial(Xi)
    eng = MagiEngine(0)
    def run(n):
        t0 = time.perf_counter()
        out = eng.fit_hparams(I, Xi, Xi.mean(axis=0), [p[0] for p in pri], [p[1] for p in pri], init["sigma_sqs"], init["phi1s"], init["phi2s"], init["sigma_sqs"], num_iters=n)
        return time.perf_counter() - t0, out
    run(2)
    t1, _ = run(iters)
    t2, out = run(2 * iters)
    print("N %d: %.3f ms per Adam step (4 components), fixed cost %.1f ms" % (N, (t2 - t1) * 1e3 / iters, (2 * t1 - t2) * 1e3), out["phi2s"])
    eng.close()
