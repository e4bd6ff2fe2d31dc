"""Dev experiment: vignette-like SEIR-3 run on the GPU, prints recovered parameters and timing."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4

nb, nr, nch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = load_g4("seir3_N161")
pr = problem_from_g4(g, None)
eng = engine_for(pr, 80)
X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(3), pr.LB)
cfg = eng.default_cfg(num_results=nr, num_burnin_steps=nb)
eng.sampler_init(cfg, np.repeat(X0[None], nch, 0), np.repeat(s0[None], nch, 0), np.repeat(t0[None], nch, 0), seed=2024)
t = time.time()
lf, ms = eng.sampler_run(nb + nr)
dt = time.time() - t
Xs, sp, tp = eng.sampler_samples()
sig, th = orc.transform_samples(sp, tp, pr.LB)
d = eng.sampler_diag()
print("wall %.2fs dev %.1fms leapfrogs %d -> %.1f us/leapfrog-slot, %.1f samples/s" % (dt, ms, lf, 1e3 * ms / (lf / nch), nch * (nb + nr) / dt))
print("theta mean per chain", th.mean(axis=1))
print("theta mean", th.reshape(-1, 3).mean(0), "sd", th.reshape(-1, 3).std(0))
print("sig2 mean", sig.reshape(-1, 3).mean(0))
print("step size end", d.step_size[:, -1], "mean depth", d.tree_depth.mean(), "div", d.has_divergence.sum())
print("accept", np.exp(np.minimum(d.log_accept_ratio[:, nb:], 0)).mean())
