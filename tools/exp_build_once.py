"""Dev: matrix builds for rocprofv3 (--pmc / --kernel-trace): a warm-up build, then `reps` timed ones.  python tools/exp_build_once.py N [reps] [nola]
("nola": the Cholesky factorisations on one stream, for per-launch summaries of a trace)"""
import sys, time
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1; D = 4
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
hp = host.hparams_initial(host.linear_interpolate(X_obs))
eng = MagiEngine(0)
if len(sys.argv) > 3 and sys.argv[3] == "nola":
    eng.set_option("potrf_lookahead_min", 0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
for r in range(reps - 1):
    t = time.perf_counter()
    eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
    print("build %d: %.2f ms" % (r, (time.perf_counter() - t) * 1e3))
eng.close()
