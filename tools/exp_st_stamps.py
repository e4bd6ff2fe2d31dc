"""Dev: time line of one k_stream<1> workgroup (needs a -DMAGI_ST_STAMPS=<task index> build; -DMAGI_ST_STAMP_WAVE=w picks the wave;
tasks of component 0 at N = 1024: 0..35 FH, 36..99 FE, 100..135 FK).  Decision-free timing launches."""
import sys
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = 1024
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
rows = []
for rep in range(20):
    g, ph = eng.time_gradient(1, 3)
    rows.append(eng.debug_par(0)[40:48].copy().view(np.uint64).astype(np.int64))
rows = np.array(rows)
rel = (rows - rows[:, :1]) * 10.0
names = ["entry", "theta' written (wave 0 only)", "positions arrived", "barrier 1 (theta')", "barrier 2 (operands in LDS)", "row chunks done", "barrier 3", "end"]
med = np.median(rel, axis=0)
print("timing launches, stream %.2f us;  workgroup time line (ns from entry, median of %d):" % (ph[4] * 1e3, len(rows)))
for k, v in sorted(zip(names, med), key=lambda kv: kv[1]):
    if abs(v) < 1e6: print("  %-30s %7.0f" % (k, v))
