"""Per-class fp64 MFMA utilisation of the GPU matrix build (BASELINE config 5).  Run with MAGI_BUILD_PROFILE=1."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]); D = 4
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
eng = MagiEngine(0)
eng.build_matrices(I[:256], hp["phi1s"], hp["phi2s"], 2.01, want_host=False)      # warm-up (module load)
t = time.perf_counter()
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
wall = time.perf_counter() - t
prof = eng.build_profile()
wall_row = prof.pop("potrf_wall")        # the two factorisations as a whole: inside this (serialised) build it repeats diag + panel + rank-k, kept out of the sums
PEAK = 78.6   # TFLOP/s, MI355X fp64 matrix (AMD spec)
rows = []
for k, (f, ms, calls) in prof.items():
    tf = f / (ms * 1e-3) / 1e12 if ms > 0 and f > 0 else 0.0
    rows.append({"class": k, "calls": calls, "gflop": round(f / 1e9, 1), "ms": round(ms, 2), "tflops": round(tf, 2), "frac_fp64_mfma_peak": round(tf / PEAK, 3)})
tot_f = sum(p[0] for p in prof.values()); tot_ms = sum(p[1] for p in prof.values())
out = {"N": N, "D": D, "wall_s_profiled": round(wall, 3), "sum_ms": round(tot_ms, 1), "tflops_overall": round(tot_f / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else None,
       "algorithmic_5N3D_tflops": round(5.0 * N ** 3 * D / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else None, "peak_tflops": PEAK, "classes": rows,
       "potrf_wall_serialised_ms": round(wall_row[1], 2)}
# the same two factorisations in an UNPROFILED build (look-ahead on: they are not serialised there)
eng.set_option("build_profile", 0)
for _ in range(2):      # (the first one creates the handle's side streams inside the timed region)
    eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
f, ms, _ = eng.build_profile()["potrf_wall"]
out["potrf_wall_ms"] = round(ms, 2)
out["potrf_wall_frac_fp64_mfma_peak"] = round(f / (ms * 1e-3) / 1e12 / PEAK, 3)
print(json.dumps(out, indent=1))
