"""Dev: per-kernel count / mean / total of a rocprofv3 --kernel-trace csv, and the distribution of the streaming kernel's durations
(the rare decision paths ride in it).  python tools/trace_summary.py file.csv"""
import csv, collections, sys
import numpy as np
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(agg.items(), key=lambda x: -sum(x[1]))[:12]:
    print("%-72s %6d  mean %9.1f us  total %9.2f ms" % (n, len(v), sum(v) / len(v), sum(v) / 1e3))
for n, v in agg.items():
    if "k_stream" in n and len(v) > 1000:
        a = np.array(v); med = np.median(a)
        print("\n%s: durations (us)  p10 %.1f  p50 %.1f  p90 %.1f  p95 %.1f  p97 %.1f  p99 %.1f  max %.1f" %
              ((n[:50],) + tuple(np.percentile(a, q) for q in (10, 50, 90, 95, 97, 99)) + (a.max(),)))
        for lim in (1.3, 2.0, 4.0):
            sel = a > lim * med
            print("   launches above %.1f x median: %.2f %%, carrying %.2f us of the mean %.2f us (excess over the median)" %
                  (lim, 100.0 * sel.mean(), (a[sel] - med).sum() / len(a), a.mean()))
