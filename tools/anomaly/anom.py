"""scratch: one-chain SIRW fixed-L HMC first transition (eps 1e-3) on the VALU family for the library in MAGI_HIP_LIB; prints energies + ChainCtl"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4
tag = sys.argv[1] if len(sys.argv) > 1 else "sirw_N41"
g = load_g4(tag); pr = problem_from_g4(g, None)
X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
names = "init_energy cand_L cand_energy L_cur e_sum e_sum_sub sub_L sub_energy eps beta_k beta_cache lf_count is_accepted cand_bfac LL LR bfacL bfacR phase k".split()
for L in (1, 2):
    eng = engine_for(pr, None)
    fn = eng._lib.magi_debug_ctl; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    cfg = eng.default_cfg(num_results=1, num_burnin_steps=0, step_size=1e-3, mode=1, hmc_leapfrogs=L)
    eng.sampler_init(cfg, X0[None], s0[None], t0[None], seed=31, chain_ids=[7])
    out0 = np.zeros(20); fn(eng._h, 0, out0.ctypes.data_as(C.POINTER(C.c_double)))
    eng.sampler_run(1)
    d = eng.sampler_diag()
    out = np.zeros(20); fn(eng._h, 0, out.ctypes.data_as(C.POINTER(C.c_double)))
    print(f"{os.environ.get('MAGI_HIP_LIB')} {tag} L={L} accepted {d.is_accepted[0,0]} lar {d.log_accept_ratio[0,0]:.6e} target {d.target_log_prob[0,0]:.9e} energy {d.energy[0,0]:.9e}")
    print("   after init: " + " ".join(f"{n}={v:.9g}" for n, v in zip(names, out0)))
    print("   after run:  " + " ".join(f"{n}={v:.9g}" for n, v in zip(names, out)))
    eng.close()
