"""Dev: A/B of the diagonal-block kernel (k_diag_chol_inv) between two builds of the library: the dense outputs of matrix builds at several
grid sizes (full and ragged last blocks) must agree bit for bit, and the build / factorisation times side by side.
    MAGI_AB_OLD=<old library> python tools/exp_diag_ab.py"""
import os, subprocess, sys, json, time
sys.path.insert(0, ".")
import numpy as np

def run(lib):
    import hashlib
    if lib:
        os.environ["MAGI_HIP_LIB"] = lib
    from magi_v2_amd import host
    from magi_v2_amd.engine import MagiEngine
    out = {}
    for N in (96, 161, 300, 1024, 2048, 8192):
        I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
        hp = host.hparams_initial(host.linear_interpolate(X_obs))
        eng = MagiEngine(0)
        eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
        ts, pw = [], []
        for r in range(3):
            t = time.perf_counter()
            eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
            ts.append((time.perf_counter() - t) * 1e3)
            pw.append(eng.build_profile()["potrf_wall"][1] if "potrf_wall" in eng.build_profile() else 0.0)
        V = np.random.default_rng(1).standard_normal((4, N, 2))
        dig = hashlib.sha256(b"".join(eng.dense_apply(k, V).tobytes() for k in ("C_inv", "m", "K_inv"))).hexdigest()[:16]
        out[N] = (min(ts), min(pw), dig)
        eng.close()
    # the hyper-parameter fit's step (its Cholesky is the same kernel)
    return out

if len(sys.argv) > 1:
    print(json.dumps(run(sys.argv[1] if sys.argv[1] != "-" else None)))
else:
    res = {}
    for name, lib in (("new", "-"), ("old", os.environ["MAGI_AB_OLD"]), ("new2", "-"), ("old2", os.environ["MAGI_AB_OLD"])):
        res[name] = json.loads(subprocess.run([sys.executable, __file__, lib], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    for N in res["new"]:
        a, b = res["new"][N], res["old"][N]
        a2, b2 = res["new2"][N], res["old2"][N]
        print(f"N = {int(N):5d}: build {min(a[0], a2[0]):8.2f} ms new | {min(b[0], b2[0]):8.2f} ms old;  two factorisations {min(a[1], a2[1]):7.3f} | {min(b[1], b2[1]):7.3f} ms;  outputs bit-identical: {a[2] == b[2] == a2[2] == b2[2]}")
