"""Dev: the N=161 hyper-parameter fit alone, for rocprofv3 --kernel-trace --stats."""
import sys
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 161
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs)
pri = [host.fourier_phi2_prior(Xi[:, d]) for d in range(4)]
init = host.hparams_initial(Xi)
eng = MagiEngine(0)
eng.fit_hparams(I, Xi, Xi.mean(axis=0), [p[0] for p in pri], [p[1] for p in pri], init["sigma_sqs"], init["phi1s"], init["phi2s"], init["sigma_sqs"], num_iters=200)
eng.close()
