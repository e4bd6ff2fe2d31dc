"""Dev: fixed-L HMC, one transition, VALU vs separable family: at which leaf do they part?"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import magi_oracle as orc
from magi_v2_amd import engine as _e
if os.environ.get("MAGI_OLD_LIB"):        # an older library: bind only the symbols it has
    import ctypes
    _lib = ctypes.CDLL(os.path.abspath(os.environ["MAGI_HIP_LIB"]))
    for k in list(_e._SYMBOLS):
        if not hasattr(_lib, k): del _e._SYMBOLS[k]
from tests.util import engine_for, load_g4, problem_from_g4
tag = sys.argv[1] if len(sys.argv) > 1 else "sirw_N41"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = load_g4(tag); pr = problem_from_g4(g, None)
X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
rep = lambda v: np.repeat(np.asarray(v)[None], nch, axis=0)
for L in (1, 2):
    out = {}
    for fam in ("valu", "mc"):
        if fam == "mc": os.environ["MAGI_STREAM_FAMILY"] = "mc"
        else: os.environ.pop("MAGI_STREAM_FAMILY", None)
        eng = engine_for(pr, None)
        cfg = eng.default_cfg(num_results=1, num_burnin_steps=0, step_size=1e-3, mode=1, hmc_leapfrogs=L)
        eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=31, chain_ids=list(range(7, 7 + nch)))
        eng.sampler_run(1)
        out[fam] = (eng.sampler_samples(), eng.sampler_diag())
        eng.close()
    (a, da), (b, db) = out["valu"], out["mc"]
    print(tag, "chains", nch, "L", L, "accepted", da.is_accepted[0, 0], db.is_accepted[0, 0], "|dtheta| %.3e |dX| %.3e  target %.9e %.9e  lar %.6e %.6e  energy %.9e %.9e div %d %d" % (
        np.abs(a[2][0] - b[2][0]).max(), np.abs(a[0][0] - b[0][0]).max(), da.target_log_prob[0, 0], db.target_log_prob[0, 0],
        da.log_accept_ratio[0, 0], db.log_accept_ratio[0, 0], da.energy[0, 0], db.energy[0, 0], da.has_divergence[0, 0], db.has_divergence[0, 0]))
