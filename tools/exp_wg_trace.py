"""Dev: where a launch of the streaming kernel / k_point spends its time ACROSS the grid (needs a -DMAGI_WG_TRACE build:
    python tools/build_variant.py wgtrace -DMAGI_WG_TRACE;  MAGI_HIP_LIB=build_variants/wgtrace.so python tools/exp_wg_trace.py [chains] [N]
Every workgroup leaves begin / end (100 MHz counter), XCC / SE / CU ids; the script prints the launch's time line, the spread of
workgroup lives by task kind, and the bytes each CU streamed."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs); hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi); LB = host.sigma_sqs_lower_bound(Xhat)
sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(float), idx, y, beta, LB, "seir4")
rep = lambda v: np.repeat(np.asarray(v)[None], n, axis=0)
eng.logpost_grad(rep(Xhat), rep(sp0), rep(tp0), 1.0, fused=True)          # states, operand mirror
fn = eng._lib.magi_debug_wg_trace
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]

# the task table as pack.hip builds it
D, nb = 4, (N + 127) // 128
tasks = [(d, k, bi, bj) for d in range(D) for k in range(3) for bi in range(nb) for bj in range(nb) if k == 2 or bj <= bi]
# (pack.hip's launch order for a three-round launch: light FH tasks on the launch positions of the CUs that host three workgroups)
C_, T_ = 256, len(tasks) + 1
if (T_ + C_ - 1) // C_ == 3 and T_ % C_ > 1:
    r_ = T_ % C_ - 1
    heavy = [t for t in tasks if t[1] != 0]; off = [t for t in tasks if t[1] == 0 and t[2] != t[3]]; diag = [t for t in tasks if t[1] == 0 and t[2] == t[3]]
    tasks = off[:r_] + heavy[:C_ - r_] + diag[:r_] + off[r_:2 * r_] + heavy[C_ - r_:] + off[2 * r_:] + diag[r_:]
sep = n >= 3 and len(tasks) * ((n + 1) // 2) > 320
if sep and len(tasks) > 256:
    st = [(d, k, bi, bj, (k != 2 and bi == bj)) for (d, k, bi, bj) in tasks if not (k == 1 and bi == bj)]
else:
    st = [(d, k, bi, bj, False) for (d, k, bi, bj) in tasks]
names = {0: "FH", 1: "FK", 2: "FE"}

for with_point in (0, 1):
    allr = []
    for r in range(reps):
        out = np.zeros((2, 4096, 4), dtype=np.uint64)
        eng._check(fn(eng._h, n, 20, with_point, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        allr.append(out.astype(np.int64))
    out = allr[-1]
    for kern, label in ((0, "stream"), (1, "point")):
        if kern == 1 and not with_point:
            continue
        tr = out[kern]
        used = tr[:, 0] > 0
        nw = int(used.sum())
        t0 = tr[used, 0].min()
        b = (tr[used, 0] - t0) * 0.01
        e = (tr[used, 1] - t0) * 0.01
        life = e - b
        hw = tr[used, 2]
        xcc = (hw >> 32) & 0xF
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
        key = xcc * 1000 + se * 100 + sh * 16 + cu
        print(f"--- {label} (with_point={with_point}), {n} chains, N={N}: {nw} workgroups, launch span {e.max():.2f} us "
              f"(spans of the {reps} traces: {', '.join('%.2f' % (((a[kern][a[kern][:,0]>0,1]).max() - (a[kern][a[kern][:,0]>0,0]).min()) * 0.01) for a in allr)})")
        print(f"    begin: min 0, median {np.median(b):.2f}, max {b.max():.2f} | end: min {e.min():.2f}, median {np.median(e):.2f}, p90 {np.percentile(e, 90):.2f}, max {e.max():.2f}"
              f" | life: median {np.median(life):.2f}, p90 {np.percentile(life, 90):.2f}, max {life.max():.2f}")
        ucu, cnt = np.unique(key, return_counts=True)
        print(f"    distinct CUs {len(ucu)} (XCCs {len(np.unique(xcc))}); workgroups per CU: " + ", ".join(f"{c}x{(cnt == c).sum()}" for c in np.unique(cnt)))
        if kern == 0 and nw == len(st):
            kinds = np.array([3 if s[4] else s[1] for s in st])
            for kk, nm in ((0, "FH"), (1, "FK"), (2, "FE"), (3, "pair FH+FK diag")):
                m = kinds == kk
                if m.any():
                    print(f"    {nm:16s} n={m.sum():4d}  life median {np.median(life[m]):6.2f}  max {life[m].max():6.2f}   end median {np.median(e[m]):6.2f}  max {e[m].max():6.2f}")
            # bytes per CU (a pair streams two blocks) against the CU's last end
            byt = np.where(kinds == 3, 2, 1) * 128.0
            cu_end = {k: e[key == k].max() for k in ucu}
            cu_kb = {k: byt[key == k].sum() for k in ucu}
            for kb in sorted(set(cu_kb.values())):
                ends = [cu_end[k] for k in ucu if cu_kb[k] == kb]
                print(f"    CUs streaming {kb:5.0f} KB: {len(ends):3d}   last end median {np.median(ends):6.2f}  max {np.max(ends):6.2f}")
            order = np.argsort(e)[-8:]
            print("    last to end: " + "; ".join(f"wg{w} {names[st[w][1]]}{'+FKdiag' if st[w][4] else ''} d{st[w][0]} ({st[w][2]},{st[w][3]}) b {b[w]:.2f} e {e[w]:.2f}" for w in order))
        if kern == 0:
            # which workgroups share a CU (dispatch pattern): index distance between co-resident workgroups
            wgs = np.nonzero(used)[0]
            dist_ = []
            for k in ucu:
                w = np.sort(wgs[key == k])
                dist_ += list(np.diff(w))
            dv, dc = np.unique(dist_, return_counts=True)
            top = np.argsort(dc)[::-1][:6]
            print("    index distance of workgroups sharing a CU: " + ", ".join(f"{dv[i]}x{dc[i]}" for i in top))
        hist, edges = np.histogram(e, bins=12)
        print("    end histogram: " + " ".join(f"{edges[i]:.1f}:{hist[i]}" for i in range(len(hist))))
eng.close()
