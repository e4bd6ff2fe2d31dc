"""Dev: one draw-for-draw configuration, GPU vs oracle per transition (where does a chain leave the oracle?)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.test_sampler_gpu import _run_both
np.set_printoptions(linewidth=200, precision=12)
(Xs, sp, tp, diag, lf), oracle = _run_both("seir3_N161", 80, 8, 4, seed=1234, stale=1)
(oX, osp, otp, info, da), trace = oracle[0]
print("depth gpu   ", diag.tree_depth[0], lf)
print("depth oracle", np.array([r.depth for _, r, _ in trace]))
ot = np.array([r.target_log_prob for _, r, _ in trace])
print("target rel diff", (diag.target_log_prob[0] - ot) / np.abs(ot))
ol = np.array([r.log_accept_ratio for _, r, _ in trace])
print("lar gpu", diag.log_accept_ratio[0])
print("lar orc", ol)
print("step gpu", diag.step_size[0])
print("step orc", np.array([s for _, _, s in trace]))
