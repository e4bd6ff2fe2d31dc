"""Dev: where does a leaf's tail time go?  Needs a -DMAGI_TAIL_STAMPS build."""
import sys
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
cfg = eng.default_cfg(num_results=30, num_burnin_steps=30, stale_cache=0)
eng.sampler_init(cfg, Xhat, sp, tp, seed=1)
eng.sampler_run(30)
rows = []
for rep in range(40):
    eng.sampler_run(1)
    rows.append(eng.debug_par(0)[40:56].copy())
rows = np.array(rows)
u = rows.view(np.uint64)        # the stream's stamps are raw 64-bit counter values
d = lambda a, b: np.median((rows[:, b] - rows[:, a]) * 10.0)
# stamps inside the decision workgroup (decide.h): 0 entry (state staged), 1 uniforms issued, 3 partial sums added,
# 4 reduce done, 5 decision, 6 hot-path end
print("ns: issue of the first round %.0f | its wait %.0f | ->1 %.0f | add partials %.0f | param entries %.0f | decide %.0f | publish %.0f | total %.0f" %
      (d(8, 2), d(2, 0), d(0, 1), d(1, 3), d(3, 4), d(4, 5), d(5, 6), d(8, 6)))
# the slot after the last hot leaf: stream start [12] / latest stream workgroup end [11] (raw counters), decide entry [8] / hot end [6] are of the
# slot BEFORE; report durations only
print("stream (first wg start -> last wg end) %.0f ns" % np.median((u[:, 11].astype(np.int64) - u[:, 12].astype(np.int64)) * 10.0))
