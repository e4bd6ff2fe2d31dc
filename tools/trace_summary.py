"""Dev: per-kernel count / mean / total of a rocprofv3 --kernel-trace csv.  python tools/trace_summary.py file.csv"""
import csv, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    a = agg[r["Kernel_Name"][:70]]
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:12]:
    print("%-72s %6d  mean %9.1f us  total %9.2f ms" % (n, c, t / c, t / 1e3))
