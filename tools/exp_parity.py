"""Dev: the fused log posterior evaluated as an even and as an odd leapfrog slot, 1 / 2 / 3 states, against the three-phase path."""
import os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4
for tag in sys.argv[1:] or ["sirw_N41", "seir4_N81", "seir3_N161"]:
    g = load_g4(tag); pr = problem_from_g4(g, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.linspace(0.7, 1.9, pr.P), pr.LB)
    for n in (1, 2, 3):
        rng = np.random.default_rng(n)
        X = X0[None] + 0.01 * rng.standard_normal((n,) + X0.shape); sp = np.repeat(s0[None], n, 0); tp = np.repeat(t0[None], n, 0)
        eng = engine_for(pr, None)
        ref = eng.logpost_grad(X, sp, tp, 1.0)
        for par in (0, 1):
            os.environ["MAGI_FUSED_PARITY"] = str(par)
            f = eng.logpost_grad(X, sp, tp, 1.0, fused=True)
            print(tag, "states", n, "parity", par, "logp rel %.2e  gX %.2e  gsig %.2e  gth %.2e" % (np.abs((f[0] - ref[0]) / ref[0]).max(),
                  np.abs(f[1] - ref[1]).max() / np.abs(ref[1]).max(), np.abs(f[2] - ref[2]).max() / np.abs(ref[2]).max(), np.abs(f[3] - ref[3]).max() / np.abs(ref[3]).max()))
        eng.close()
