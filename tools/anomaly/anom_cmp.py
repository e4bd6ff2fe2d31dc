import numpy as np, sys
a = np.load("dump_good.npz")
for t in sys.argv[1:]:
    b = np.load(f"dump_{t}.npz")
    print("==", t, "lar", float(b["lar"]))
    for k in a.files:
        x, y = a[k], b[k]
        if x.shape != y.shape: print(k, "shape"); continue
        bad = np.nonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))[0] if x.ndim else ([0] if x != y else [])
        if len(bad): print(f"  {k}: {len(bad)} of {x.size} differ; first idx {list(bad[:12])}  good {np.atleast_1d(x)[bad[:6]]}  this {np.atleast_1d(y)[bad[:6]]}")
