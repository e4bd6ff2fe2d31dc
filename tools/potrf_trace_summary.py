"""Dev: where the blocked Cholesky's rank-k updates spend their time, launch by launch (BASELINE config 5).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/exp_build_once.py 8192 2
    python tools/potrf_trace_summary.py DIR/*/*kernel_trace.csv 8192 4 [panels per block column, default 3]
(run the traced build with MAGI_POTRF_LOOKAHEAD_MIN=0: with look-ahead a trailing update is two launches on two streams)
The rank-k class (k_gemm_f64<3>) holds two kinds of launch: the TRAILING update of a block column (lower tiles of the whole trailing
matrix, K = 512) and the TALL 128-wide updates inside a block column (K = 128 / 256 / 384).  The trace tells them apart by grid size;
flops per launch follow from the launch order (csrc/build.hip: potrf)."""
import csv, sys
import numpy as np
path, N, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
NPAN_ARG = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rows = [r for r in csv.DictReader(open(path))]
t = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rk = [r for r in rows if "k_gemm_f64<3" in r["Kernel_Name"]]
diag = [t(r) for r in rows if "k_diag_chol_inv" in r["Kernel_Name"]]
pan = [t(r) for r in rows if "k_gemm_f64<2" in r["Kernel_Name"]]
nbuild = max(1, len(diag) // (2 * (N // 128)))
grid = np.array([int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)))) for r in rk])
dur = np.array([t(r) for r in rk])
# expected launch sequence of one factorisation: per block column j0 (NPAN panels): tall K = 128, 256, ...; then the trailing update
NB, NPAN = 128, NPAN_ARG
seq = []
for j0 in range(0, N, NPAN * NB):
    for c in range(1, NPAN):
        jc = j0 + c * NB
        if jc >= N: break
        M = N - jc
        seq.append(("tall", 2.0 * M * min(NB, N - jc) * c * NB * D, ((M + 127) // 128) * D))
    jn = j0 + NPAN * NB
    if jn < N:
        M = N - jn
        tiles = (M // 128) * (M // 128 + 1) // 2
        seq.append(("trailing", 2.0 * tiles * 128 * 128 * NPAN * NB * D, tiles * D))
per_fact = len(seq)
nfact = len(rk) // per_fact
print(f"{len(rk)} rank-k launches = {nfact} factorisations x {per_fact} ({nbuild} build(s)); diag {np.sum(diag) / nbuild / 1e3:.2f} ms, panels {np.sum(pan) / nbuild / 1e3:.2f} ms per build")
acc = {"tall": [0.0, 0.0, 0], "trailing": [0.0, 0.0, 0]}
detail = []
for i, d_ in enumerate(dur[: nfact * per_fact]):
    kind, fl, tiles = seq[i % per_fact]
    acc[kind][0] += d_; acc[kind][1] += fl; acc[kind][2] += 1
    if i < per_fact:
        detail.append((kind, tiles, d_, fl / (d_ * 1e-6) / 1e12))
for kind, (us, fl, n) in acc.items():
    print(f"  {kind:9s}: {n:4d} launches, {us / 1e3 / nbuild:7.2f} ms per build, {fl / nbuild / 1e9:8.1f} GFLOP per build, {fl / (us * 1e-6) / 1e12:6.1f} TFLOP/s = {fl / (us * 1e-6) / 1e12 / 78.6:.3f} of the fp64 MFMA peak")
print("  first factorisation, launch by launch (kind, tiles, us, TFLOP/s):")
for kind, tiles, d_, tf in detail:
    print(f"     {kind:9s} {tiles:6d} tiles {d_:9.1f} us {tf:6.1f}")
