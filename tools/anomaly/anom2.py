"""scratch: as anom.py (L = 1 only) + the parameter block and the state vectors after the run, saved per library for comparison"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4
g = load_g4("sirw_N41"); pr = problem_from_g4(g, None)
X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
eng = engine_for(pr, None)
lib = eng._lib
lib.magi_debug_ctl.restype = C.c_int; lib.magi_debug_ctl.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
lib.magi_debug_vec.restype = C.c_int; lib.magi_debug_vec.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
cfg = eng.default_cfg(num_results=1, num_burnin_steps=0, step_size=1e-3, mode=1, hmc_leapfrogs=1)
eng.sampler_init(cfg, X0[None], s0[None], t0[None], seed=31, chain_ids=[7])
eng.sampler_run(1)
d = eng.sampler_diag()
ctl = np.zeros(20); lib.magi_debug_ctl(eng._h, 0, ctl.ctypes.data_as(C.POINTER(C.c_double)))
par = eng.debug_par(0)
vecs = {}
buf = np.zeros(4096)
for slot, name in ((0, "Q0"), (1, "Q1"), (2, "P0"), (3, "P1"), (4, "PLEAF"), (5, "G"), (11, "PL"), (12, "QL"), (13, "GL"), (14, "PR"), (15, "QR"), (16, "GR"), (17, "CANDQ")):
    n = lib.magi_debug_vec(eng._h, 0, slot, buf.ctypes.data_as(C.POINTER(C.c_double)))
    vecs[name] = buf[:n].copy()
tag = os.path.basename(os.environ["MAGI_HIP_LIB"]).replace("var_", "").replace(".so", "")
np.savez(f"dump_{tag}.npz", ctl=ctl, par=par, lar=d.log_accept_ratio[0, 0], **vecs)
print(tag, "lar", d.log_accept_ratio[0, 0], "L_cur", ctl[3], "par[0:8]", par[:8], "sig2", par[24:28])
eng.close()
