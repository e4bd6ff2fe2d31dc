"""Dev: time line of one k_stream_sep workgroup (needs a -DMAGI_SEP_STAMPS=<task index> build).  python tools/exp_sep_stamps.py [chains]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = 1024; n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
rows = []
for rep in range(12):
    g, ph = eng.time_gradient(n, 3)
    rows.append(eng.debug_par(0)[40:56].copy().view(np.uint64).astype(np.int64))
rows = np.array(rows)
rel = (rows - rows[:, :1]) * 10.0
names = ["entry", "task known", "operand loads issued", "ring issued", "operands in LDS", "barrier"] + ["step %d" % k for k in range(8)] + ["stores issued", "stores retired"]
med = np.median(rel, axis=0)
print("stream kernel %.2f us;  workgroup time line (ns from entry, median of %d):" % (ph[4] * 1e3, len(rows)))
for k, v in zip(names, med):
    print("  %-22s %7.0f" % (k, v))
