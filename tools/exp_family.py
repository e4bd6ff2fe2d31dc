"""Dev: one chain under the two streaming-kernel families (VALU k_stream<1> vs the separable matrix-core kernel, MAGI_STREAM_FAMILY=mc):
per transition, the difference of the kept theta and of the leapfrog counts -- rounding-level differences amplified by the dynamics,
or a bug (a difference from the first transition on)."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4
tag = sys.argv[1] if len(sys.argv) > 1 else "sirw_N41"
g = load_g4(tag); pr = problem_from_g4(g, None)
X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
out = {}
for fam in ("valu", "mc"):
    if fam == "mc": os.environ["MAGI_STREAM_FAMILY"] = "mc"
    else: os.environ.pop("MAGI_STREAM_FAMILY", None)
    eng = engine_for(pr, None)
    cfg = eng.default_cfg(num_results=10, num_burnin_steps=0, step_size=float(os.environ.get('STEP0', '2e-3')))
    eng.sampler_init(cfg, X0, s0, t0, seed=31, chain_ids=[7])
    eng.sampler_run(10)
    out[fam] = (eng.sampler_samples(), eng.sampler_diag())
    lp = eng.logpost_grad(X0, s0, t0, 1.0, fused=True)
    lp3 = eng.logpost_grad(X0, s0, t0, 1.0)
    print(fam, "fused vs 3-phase: logp rel %.2e  gX rel %.2e  gth rel %.2e" % (abs(lp[0] - lp3[0]) / abs(lp3[0]), np.abs(lp[1] - lp3[1]).max() / np.abs(lp3[1]).max(), np.abs(lp[3] - lp3[3]).max() / np.abs(lp3[3]).max()))
    eng.close()
(a, da), (b, db) = out["valu"], out["mc"]
for k in range(10):
    print(k, "leapfrogs", da.leapfrogs_taken[0, k], db.leapfrogs_taken[0, k], "depth", da.tree_depth[0, k], db.tree_depth[0, k], "|dtheta| %.3e" % np.abs(a[2][0, k] - b[2][0, k]).max(), "|dX| %.3e" % np.abs(a[0][0, k] - b[0][0, k]).max())
