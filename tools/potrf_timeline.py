"""Dev: the time line of one blocked Cholesky with look-ahead from a rocprofv3 kernel trace (who overlaps whom).
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/exp_build_once.py 8192 2
    python tools/potrf_timeline.py DIR/*/*kernel_trace.csv [first_n_rows]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 90
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST build in the trace: from the last k_matern* launch group on
names = [r["Kernel_Name"] for r in rows]
last_matern = max(i for i, n in enumerate(names) if "matern" in n.lower())
first = last_matern
while first > 0 and "matern" in names[first - 1].lower():
    first -= 1
sel = rows[first:]
t0 = int(sel[0]["Start_Timestamp"])
short = lambda n: ("diag" if "k_diag_chol_inv" in n else "panel" if "k_gemm_f64<2" in n else "rank-k" if "k_gemm_f64<3" in n else "trtri" if "k_gemm_f64<4" in n else n.split("(")[0][:28])
qcol = "Queue_Id" if "Queue_Id" in sel[0] else None
print(f"{'kernel':10s} {'queue':>6s} {'grid':>8s} {'start us':>10s} {'end us':>10s} {'dur us':>8s}  gap to previous end on the same queue")
last_end = {}
shown = 0
for r in sel:
    n = short(r["Kernel_Name"])
    if n not in ("diag", "panel", "rank-k"):
        if shown: break
        continue
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r[qcol] if qcol else "?"
    wg = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256))))
    gap = s - last_end.get(q, s)
    last_end[q] = e
    if shown < nshow:
        print(f"{n:10s} {q:>6s} {wg:8d} {s:10.1f} {e:10.1f} {e - s:8.1f}  {gap:8.1f}")
    shown += 1
