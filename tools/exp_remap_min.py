"""Dev: the build's GEMM classes as a function of the super-block tile order's threshold (option gemm_remap_min: launches with at least that many
8 x 8 super-blocks of tiles use the XCD-aware order).    python tools/exp_remap_min.py [N] [thresholds ...]"""
import sys, time
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ths = [int(x) for x in sys.argv[2:]] or [24, 16, 8, 4, 1]
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
hp = host.hparams_initial(host.linear_interpolate(X_obs))
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
for t in ths:
    eng.set_option("gemm_remap_min", t)
    eng.set_option("build_profile", 0)
    ts = []
    for r in range(2):
        t0 = time.perf_counter()
        eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
        ts.append((time.perf_counter() - t0) * 1e3)
    pw = eng.build_profile()["potrf_wall"][1]
    eng.set_option("build_profile", 1)
    eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
    p = eng.build_profile()
    frac = lambda k: p[k][0] / (p[k][1] * 1e-3) / 1e12 / 78.6
    print(f"gemm_remap_min {t:3d}: build {min(ts):7.2f} ms, factorisations {pw:6.2f} ms;  rank-k {p['potrf_trailing_syrk'][1]:6.2f} ms  panels {p['potrf_panel'][1]:5.2f}  trtri {p['trtri'][1]:6.2f} ms ({frac('trtri'):.3f})  "
          f"TtT {p['TtT'][1]:6.2f} ({frac('TtT'):.3f})  products {p['m_K_products'][1]:7.2f} ({frac('m_K_products'):.3f})  operators {p['single_phase_operators'][1]:7.2f}", flush=True)
eng.close()
