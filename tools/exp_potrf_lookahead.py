"""Dev: A/B of the blocked Cholesky's look-ahead at BASELINE config 5 (N = 8192 x 4): wall time of a pooled build and of its two factorisations
(the "potrf_wall" row of magi_build_profile) with the rank-k updates forked to the CU-masked stream and without, and the dense outputs of both
(look-ahead only reorders launches: every tile takes the same updates in the same order, the results must be bit-identical).
    python tools/exp_potrf_lookahead.py [N] [reps] [panels per block column]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
panels = int(sys.argv[3]) if len(sys.argv) > 3 else None
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
hp = host.hparams_initial(host.linear_interpolate(X_obs))
eng = MagiEngine(0)
if panels is not None:
    eng.set_option("potrf_panels", panels)
print(f"N = {N}, {panels if panels is not None else 'the default (3)'} panels of 128 per block column")
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
sums = {}
for name, la_min in (("one stream", 0), ("look-ahead", 2048), ("one stream", 0), ("look-ahead", 2048)):
    eng.set_option("potrf_lookahead_min", la_min)
    ts, pw = [], []
    for r in range(reps):
        t = time.perf_counter()
        eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
        ts.append((time.perf_counter() - t) * 1e3)
        row = eng.build_profile()["potrf_wall"]
        pw.append(row[1])
    fl = row[0]
    print(f"{name:11s}: build {min(ts):7.2f} ms (min of {reps}; all {' '.join('%.1f' % x for x in ts)}), two factorisations {min(pw):6.2f} ms = "
          f"{fl / (min(pw) * 1e-3) / 1e12:5.1f} TFLOP/s = {fl / (min(pw) * 1e-3) / 1e12 / 78.6:.3f} of the fp64 MFMA peak", flush=True)
    # fingerprints of the three dense stacks: products with a fixed probe (identical matrices -> identical bits; the stacks are 6.4 GB)
    V = np.random.default_rng(1).standard_normal((4, N, 2))
    sums.setdefault(name, [eng.dense_apply(k, V).tobytes() for k in ("C_inv", "m", "K_inv")])
print("C^-1, m, K^-1 probes bit-identical between the two:", all(a == b for a, b in zip(sums["one stream"], sums["look-ahead"])))
eng.close()
