"""Dev: the matrix-core multi-chain streaming kernel (k_stream_mc) against the three-phase reference-order path, and its
timing.  python tools/exp_mc.py [N] [chains ...]   (MAGI_STREAM_KERNEL=valu selects the VALU kernel for A/B)"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from magi_v2_amd import host
from magi_v2_amd.engine import MagiEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chains = [int(a) for a in sys.argv[2:]] or [3, 8, 16]
band = None
I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
Xi = host.linear_interpolate(X_obs)
hp = host.hparams_initial(Xi)
N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
Xhat = host.cubic_smoother(I, Xi)
LB = host.sigma_sqs_lower_bound(Xhat)
sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
eng = MagiEngine(0)
eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, bandsize=band, want_host=False)
eng.set_problem(Xi.mean(axis=0), N_ds.astype(np.float64), idx, y, beta, LB, "seir4")
rng = np.random.default_rng(1)
print("kernel:", os.environ.get("MAGI_STREAM_KERNEL", "mfma (default)"))
for n in chains:
    X = Xhat[None] + 0.01 * rng.standard_normal((n,) + Xhat.shape)
    sp = sp0[None] + 0.1 * rng.standard_normal((n, 4))
    tp = np.log(np.expm1(th))[None] + 0.1 * rng.standard_normal((n, 3))
    a = eng.logpost_grad(X, sp, tp, 0.8)
    b = eng.logpost_grad(X, sp, tp, 0.8, fused=True)
    rel = lambda u, v: float(np.abs(u - v).max() / np.abs(u).max())
    print(f"chains {n:3d}: logp rel {np.abs((a[0]-b[0])/a[0]).max():.2e}  gX {rel(a[1], b[1]):.2e}  gsig {rel(a[2], b[2]):.2e}  gth {rel(a[3], b[3]):.2e}", flush=True)
    g, ph = eng.time_gradient(n, 200)
    print(f"            stream {ph[4]*1e3:.2f} us  point {ph[6]*1e3:.2f} us  read-only {ph[7]*1e3:.2f} us", flush=True)
for n in chains:
    cfg = eng.default_cfg(num_results=10, num_burnin_steps=30, stale_cache=0)
    rep = lambda v: np.repeat(np.asarray(v)[None], n, axis=0)
    eng.sampler_init(cfg, rep(Xhat), rep(sp0), rep(tp0), seed=7, chain_ids=list(range(n)))
    eng.sampler_run(30)
    t0 = time.perf_counter()
    lf, ms = eng.sampler_run(10)
    dt = time.perf_counter() - t0
    d = eng.sampler_diag()
    print(f"sampler {n:3d} chains: {lf/dt:9.0f} leapfrogs/s  {dt/(lf/n)*1e6:6.2f} us/slot  depth {d.tree_depth[:, 30:].mean():.2f}  finite {np.isfinite(eng.sampler_samples()[2]).all()}", flush=True)
eng.close()
