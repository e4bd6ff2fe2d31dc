"""Dev: the triangular inverse (k_gemm_f64<4>) launch by launch from a rocprofv3 kernel trace of an N-point build (csrc/build.hip: trtri).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/exp_build_once.py 8192 2 nola
    python tools/trtri_trace_summary.py DIR/*/*kernel_trace.csv 8192 4"""
import csv, sys
path, N, D = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = sorted((r for r in csv.DictReader(open(path)) if "k_gemm_f64<4" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
# launch order of one trtri: levels s = 128, 256, ...: (tmp = L21 T11 [kmode 3], T21 = -T22 tmp [kmode 2]) per level
seq = []
s = 128
while s < N:
    pairs = N // (2 * s)
    # tiles of 128: tmp: M = s, N = s, k from n0 (T11 lower: k >= n) -> average k length s/2 + 64;  T21: k < m0 + 128 -> s/2 + 64
    fl = 2.0 * pairs * D * s * s * (s / 2.0 + 64)
    seq += [(s, "L21 T11", fl), (s, "T22 tmp", fl)]
    s *= 2
per = len(seq)
n = len(rows) // per
print(f"{len(rows)} launches = {n} triangular inverses x {per}")
tot_t = tot_f = 0.0
for i, (s, what, fl) in enumerate(seq):
    d = [(int(rows[k * per + i]["End_Timestamp"]) - int(rows[k * per + i]["Start_Timestamp"])) / 1e3 for k in range(n)]
    us = sorted(d)[len(d) // 2]
    tot_t += us; tot_f += fl
    print(f"  s = {s:5d}  {what:8s}  {us:9.1f} us  {fl / 1e9:8.1f} GFLOP  {fl / (us * 1e-6) / 1e12:6.1f} TFLOP/s = {fl / (us * 1e-6) / 1e12 / 78.6:.3f} of the fp64 MFMA peak")
print(f"  one inverse: {tot_t / 1e3:.2f} ms, {tot_f / 1e9:.0f} GFLOP issued (triangular k ranges), {tot_f / (tot_t * 1e-6) / 1e12 / 78.6:.3f} of the peak")
