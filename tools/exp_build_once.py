"""Dev: one matrix build (for rocprofv3 --pmc / --kernel-trace).  python tools/exp_build_once.py N"""
import sys
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]); D = 4
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
hp = host.hparams_initial(host.linear_interpolate(X_obs))
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.close()
