"""GPU tests at BASELINE.json's full sizes (configs 2-5), through size-independent properties -- the oracle
cannot run there in seconds:
  * N=1024 / N=4096 x 4 components: GPU build -> (i) the reference-order three-phase log posterior and the
    sampler's single-phase kernel agree, (ii) the analytic gradient matches a central finite difference of the
    log posterior along random directions, (iii) C^-1 Kappa = I on random columns.
  * config 4 (alpha sweep): the ten thinned datasets x chains run end to end through the drop-in API.
  * config 3: chains are independent of how they are batched (ids keyed Philox), at N = 256 with 16 chains and at the
    config's own size (N = 1024, 8 chains per GPU) with an oracle draw-for-draw check on the GPU-built matrices.
  * config 5 (N = 8192): the build's OUTPUTS against Matern columns evaluated by the oracle (never downloads a matrix).

Which test pins which GEMM class of csrc/build.hip (k_gemm_f64<CLS>; the XCD-aware super-block tile order switches on by itself
from N = 4096 on, `MAGI_GEMM_REMAP_MIN=1` forces it on every launch):
  <2> potrf panels, <3> potrf rank-k updates, <4> trtri, <5> T^T T, <6> W^T / m / K_d products
      -- plain tile order:  test_full_size_inverse_properties[1024 / 2048] (dense host truth: C^-1 Kappa = I, m Kappa = p_Kappa,
         K^-1 K_ref = I), tests/test_build_gpu.py (mpmath / reference golden at small N);
      -- super-block order: test_remapped_tile_order_is_bit_identical_n2048 (every output equal BIT FOR BIT to the plain-order
         build at a size where the dense host truth above applies), test_full_size_inverse_properties[2048-remap-*] (the same
         host truth with the order forced and 2 / 8 panels per block column), test_config5_build_outputs_against_matern_columns
         (N = 8192, the order as it runs in production, truth = oracle Matern columns);
  <7> single-phase operators E = Ks M, H = M^T E + Cs: the fused vs three-phase log posterior at N = 1024 .. 8192
      (test_full_size_logpost_consistency_and_gradient) and bit-identity of the fused log posterior under the forced order."""
import os

import numpy as np
import pytest

from magi_v2_amd import host
from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


def _synthetic_engine(N, band=None):
    from magi_v2_amd.engine import MagiEngine
    I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    eng = MagiEngine(0)
    eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, bandsize=band, want_host=False)
    eng.set_problem(Xi.mean(axis=0), N_ds.astype(np.float64), idx, y, beta, LB, "seir4")
    sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], th, LB)
    return eng, I, hp, Xhat, sp, tp


@pytest.mark.parametrize("N,band", [(1024, None), (1024, 80), (4096, None), (8192, None)])
def test_full_size_logpost_consistency_and_gradient(N, band):
    eng, I, hp, Xhat, sp, tp = _synthetic_engine(N, band)
    rng = np.random.default_rng(N)
    X = Xhat + 0.01 * rng.standard_normal(Xhat.shape)
    lp, gX, gs, gt = eng.logpost_grad(X, sp, tp, 0.8)
    lf, gXf, gsf, gtf = eng.logpost_grad(X, sp, tp, 0.8, fused=True)
    assert abs(lp - lf) <= 1e-9 * abs(lp)                                        # two formulations, one value
    assert np.abs(gX - gXf).max() <= 1e-9 * np.abs(gX).max()
    assert np.abs(gt - gtf).max() <= 1e-9 * np.abs(gt).max()
    # directional derivative: (L(q + h v) - L(q - h v)) / 2h = grad . v
    for trial in range(2):
        vX = rng.standard_normal(X.shape) * 1e-3
        vs = rng.standard_normal(sp.shape) * 1e-2
        vt = rng.standard_normal(tp.shape) * 1e-2
        h = 1e-4
        lp_p = eng.logpost_grad(X + h * vX, sp + h * vs, tp + h * vt, 0.8)[0]
        lp_m = eng.logpost_grad(X - h * vX, sp - h * vs, tp - h * vt, 0.8)[0]
        fd = (lp_p - lp_m) / (2 * h)
        an = (gX * vX).sum() + gs @ vs + gt @ vt
        assert abs(fd - an) <= 1e-4 * max(abs(an), abs(fd)) + 1e-6, (fd, an)     # FD truncation (h^2 term) dominates at large N
    eng.close()


def test_logpost_at_n4096_against_the_cpu_oracle():
    """The first CPU truth for the dense paths above 8 block rows: at N = 4096 x 4 (8 448 operator blocks, 1.1 GB) the three stacks the
    GPU built are downloaded once (magi_get_dense, 1.6 GB) and the oracle -- the numpy restatement of magi_v2.py:308-348, pinned to the
    golden vectors at small N -- evaluates log posterior and gradient on them: the reference-order three-phase kernels to 1e-10, the
    sampler's single-phase kernels to 1e-9, for one state (VALU streaming kernel) and for a batch of three (matrix-core kernel).
    (At N = 8192 the same comparison would move 6.4 GB and minutes of numpy: the kernels are the same code with more blocks.)"""
    from magi_v2_amd.engine import MagiEngine
    from oracle import magi_oracle as orc
    N = 4096
    I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    eng = MagiEngine(0)
    eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01, want_host=False)
    eng.set_problem(Xi.mean(axis=0), N_ds.astype(np.float64), idx, y, beta, LB, "seir4")
    C_inv, m, K_inv = eng.get_dense()
    pr = orc.Problem(I=I, mu=Xi.mean(axis=0), C_inv=C_inv, m=m, K_inv=K_inv, N_ds=N_ds.astype(np.float64), obs_idx=idx, y=y, beta=float(beta),
                     LB=LB, drift="seir4", P=3)
    sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], th, LB)
    rng = np.random.default_rng(4096)
    Xb = Xhat[None] + 0.01 * rng.standard_normal((3,) + Xhat.shape)
    spb = sp[None] + 0.1 * rng.standard_normal((3, 4))
    tpb = tp[None] + 0.1 * rng.standard_normal((3, 3))
    truth_b = [orc.logpost_grad(Xb[c], spb[c], tpb[c], 0.8, pr) for c in range(3)]
    assert eng.stream_kernel_name(1) == "k_stream<1>" and eng.stream_kernel_name(3).startswith("k_stream_sep")

    def check(got, ref, tol):
        l0, gx0, gs0, gt0 = ref
        assert abs(got[0] - l0) <= tol * abs(l0), (got[0], l0)
        assert np.abs(got[1] - gx0).max() <= tol * np.abs(gx0).max()
        assert np.abs(got[2] - gs0).max() <= tol * np.abs(gs0).max()
        assert np.abs(got[3] - gt0).max() <= tol * np.abs(gt0).max()

    check(eng.logpost_grad(Xb[0], spb[0], tpb[0], 0.8), truth_b[0], 1e-10)                # three phases, reference op order
    check(eng.logpost_grad(Xb[0], spb[0], tpb[0], 0.8, fused=True), truth_b[0], 1e-9)     # k_stream<1> + k_point
    three = eng.logpost_grad(Xb, spb, tpb, 0.8)
    fused = eng.logpost_grad(Xb, spb, tpb, 0.8, fused=True)                                # k_stream_sep + k_point (separable path)
    for c in range(3):
        check([a[c] for a in three], truth_b[c], 1e-10)
        check([a[c] for a in fused], truth_b[c], 1e-9)
    eng.close()


def test_config5_sampler_runs_at_n8192():
    """BASELINE config 5 (N = 8192 x 4, one chain): the block-streaming sampler on 33 024 operator blocks (4.4 GB) -- a few
    NUTS transitions stay finite, move the state and take leapfrogs; the fixed-L mode does the same work per leapfrog."""
    eng, I, hp, Xhat, sp, tp = _synthetic_engine(8192, None)
    sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), host.sigma_sqs_lower_bound(Xhat))
    cfg = eng.default_cfg(num_results=2, num_burnin_steps=6, stale_cache=0)
    eng.sampler_init(cfg, Xhat, sp0, tp0, seed=3)
    lf, ms = eng.sampler_run(8)
    Xs, s_, t_ = eng.sampler_samples()
    d = eng.sampler_diag()
    assert lf == d.leapfrogs_taken.sum() and lf >= 8
    assert np.isfinite(Xs).all() and np.isfinite(t_).all() and np.abs(Xs[0, -1] - Xhat).max() > 0
    eng.close()


class _env:
    """Environment switches of csrc/build.hip for the duration of a block (read by the library at every launch)."""
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items() if v is not None}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("N,remap,panels,lookahead", [(1024, None, None, None), (2048, None, None, None), (2048, 1, 2, None), (2048, 1, 8, None), (2048, None, None, 1024),
                                                      (2048, 1, 4, 1024)],
                         ids=["1024", "2048", "2048-remap-2panels", "2048-remap-8panels", "2048-lookahead", "2048-remap-4panels-lookahead"])
def test_full_size_inverse_properties(N, remap, panels, lookahead):
    """C^-1 Kappa = I, m Kappa = p_Kappa and K^-1 K_ref = I on random columns at BASELINE sizes; K_ref = Kappa_pp +
    p_Kappa Kappa^-1 p_Kappa is formed on the host from the GPU's Matern blocks (pinned against mpmath at small N).
    K^-1 is the worst-conditioned product of the build: Schur complement, second Cholesky, second inverse.
    remap / panels: the XCD-aware super-block tile order forced on every GEMM launch (it switches on by itself only above
    N = 4096), 2 / 4 / 8 panels per potrf block column instead of 3, and the factorisation's look-ahead (rank-k updates forked to the
    CU-masked side stream, on by itself from N = 4096) -- the code paths of the N = 8192 build at a size where the dense host truth is
    affordable."""
    from magi_v2_amd.engine import MagiEngine
    EPS = np.finfo(float).eps
    I = np.arange(N) * 0.025
    eng = MagiEngine(0)
    if remap is not None:
        eng.set_option("gemm_remap_min", remap)
    if panels is not None:
        eng.set_option("potrf_panels", panels)
    if lookahead is not None:
        eng.set_option("potrf_lookahead_min", lookahead)
    C_inv, m, K_inv = eng.build_matrices(I, [0.05], [0.1], 2.01)
    Kap, pK, Kpp = eng.matern_blocks(I, 0.05, 0.1, 2.01)
    cols = np.random.default_rng(0).integers(0, N, 16)
    cond = np.linalg.cond(Kap)
    R = C_inv[0] @ Kap[:, cols]
    R[cols, np.arange(16)] -= 1.0
    assert np.abs(R).max() < 100 * cond * EPS
    assert np.abs(m[0] @ Kap[:, cols] - pK[:, cols]).max() < 100 * cond * EPS * np.abs(pK).max()
    m_ref = np.linalg.solve(Kap, pK.T).T                      # p_Kappa Kappa^-1   (Kappa_p = -p_Kappa, magi_v2.py:805)
    K_ref = Kpp + m_ref @ pK
    K_ref = 0.5 * (K_ref + K_ref.T)
    condK = np.linalg.cond(K_ref)
    R = K_inv[0] @ K_ref[:, cols]
    R[cols, np.arange(16)] -= 1.0
    assert np.abs(R).max() < 200 * cond * EPS * condK ** 0.5, (np.abs(R).max(), cond, condK)
    assert np.abs(K_inv[0] - K_inv[0].T).max() == 0.0
    eng.close()


def test_remapped_tile_order_is_bit_identical_n2048():
    """The super-block tile order of k_gemm_f64 (csrc/build.hip: XCD x takes 8 x 8 super-blocks, rotated; lower-only launches
    enumerate the lower super-blocks) only changes WHICH workgroup computes a tile, never the tile's arithmetic: with the order
    forced (MAGI_GEMM_REMAP_MIN=1) every output of the build -- C^-1, m, K^-1 of two components with different hyper-parameters,
    and the single-phase operators through the fused log posterior -- equals the plain-order build bit for bit.  A dropped,
    duplicated or mis-placed tile in any GEMM class shows up here; the plain-order build itself is held to dense host truth
    by test_full_size_inverse_properties[2048]."""
    from magi_v2_amd.engine import MagiEngine
    N = 2048
    I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    sp, tp = host.softplus_inverse_inits(hp["sigma_sqs"], th, LB)
    X = Xhat + 0.01 * np.random.default_rng(5).standard_normal(Xhat.shape)

    def build(remap):
        eng = MagiEngine(0)
        if remap is not None:
            eng.set_option("gemm_remap_min", remap)
        mats = eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01)
        eng.set_problem(Xi.mean(axis=0), N_ds.astype(np.float64), idx, y, beta, LB, "seir4")
        out = mats + tuple(eng.logpost_grad(X, sp, tp, 0.8, fused=True))
        eng.close()
        return out

    plain, forced = build(None), build(1)
    for a, b, what in zip(plain, forced, ("C_inv", "m", "K_inv", "logp", "gX", "gsig", "gth")):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b), err_msg=what)
    assert np.isfinite(plain[3]) and np.abs(plain[0]).max() > 0


def test_lookahead_factorisation_is_bit_identical_n2048():
    """Look-ahead of the blocked Cholesky (csrc/build.hip: potrf -- the rank-k update right of the next block column runs on a second,
    CU-masked stream under the next column's chain) reorders launches, never a tile's arithmetic or the order of the updates a tile
    takes: with it forced on at N = 2048 (four components, different hyper-parameters; 3 and 4 panels per block column) every dense
    output equals the one-stream build bit for bit.  A missing event between the two streams would show up here as a tile that missed
    an update or took it twice; the one-stream build is held to dense host truth by test_full_size_inverse_properties[2048]."""
    from magi_v2_amd.engine import MagiEngine
    N = 2048
    I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
    hp = host.hparams_initial(host.linear_interpolate(X_obs))
    for panels in (3, 4):
        outs = []
        eng = MagiEngine(0)
        eng.set_option("potrf_panels", panels)
        for la_min in (0, 1024, 1024):                   # (twice with look-ahead: the second build reuses streams and events)
            eng.set_option("potrf_lookahead_min", la_min)
            outs.append(eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01))
            wall = eng.build_profile()["potrf_wall"]     # (flops, ms, calls) of the two factorisations as a whole
            assert wall[1] > 0.0 and wall[2] == 2 and wall[0] == pytest.approx(2 * 4 * N ** 3 / 3.0, rel=1e-12)
        eng.close()
        for later in outs[1:]:
            for a, b, what in zip(outs[0], later, ("C_inv", "m", "K_inv")):
                np.testing.assert_array_equal(a, b, err_msg=f"{what}, {panels} panels")
        assert np.abs(outs[0][2]).max() > 0


def test_config5_build_outputs_against_matern_columns():
    """BASELINE config 5's build (N = 8192; magi_v2.py:818-820, 126-128) checked on its OUTPUTS without downloading a matrix:
    16 random columns of Kappa, p_Kappa, Kappa_pp from the oracle (magi_v2.py:781-815 in their mpmath-pinned cancellation-free forms), and
    the device-resident C^-1, m, K^-1 applied to them (magi_dense_apply):
        C^-1 Kappa[:, c] = e_c,     m Kappa[:, c] = p_Kappa[:, c],     K^-1 K_ref[:, c] = e_c
    with K_ref[:, c] = Kappa_pp[:, c] + m p_Kappa[:, c]  (K = Kappa_pp - p_Kappa Kappa^-1 Kappa_p, Kappa_p = -p_Kappa, :805, :820;
    m was checked the line before).  At this size every large product runs in the super-block tile order.  Tolerances as in
    test_full_size_inverse_properties; the condition numbers are those of the same grid spacing at N = 1024 (they do not grow
    with N at fixed spacing: SURVEY section 7)."""
    from magi_v2_amd.engine import MagiEngine
    from oracle import magi_oracle as orc
    EPS = np.finfo(float).eps
    N, phi1, phi2 = 8192, 0.05, 0.1
    I = np.arange(N) * 0.025
    Ks, pKs, Kpps = orc.matern_blocks(I[:1024], phi1, phi2)
    cond = np.linalg.cond(Ks)
    Kref_s = Kpps + np.linalg.solve(Ks, pKs.T).T @ pKs
    Kref_s = 0.5 * (Kref_s + Kref_s.T)
    condK, normK = np.linalg.cond(Kref_s), np.linalg.norm(Kref_s, 2)
    eng = MagiEngine(0)
    eng.build_matrices(I, [phi1], [phi2], 2.01, want_host=False)
    cols = np.sort(np.random.default_rng(8192).choice(N, 16, replace=False))
    cols[0], cols[-1] = 0, N - 1                                   # the ragged ends of the tile grid included
    # truth columns: the cancellation-free forms (pinned to mpmath, tests/test_oracle_golden.py).  The reference's own Kappa_pp
    # expression carries ~2e-10 of its scale at small lags, which K^-1 amplifies to 1e-5 in the third identity below; its
    # columns (matern_block_columns, the literal restatement of magi_v2.py:781-815) agree with these to its own accuracy
    Kap, pK, Kpp = orc.matern_block_columns_accurate(I, cols, phi1, phi2)
    # (on this grid, t up to 205, the literal Kappa_pp loses more: its (2 v s^2 - 4 v s t + 2 v t^2) term cancels eight digits at
    #  neighbouring points -- measured 7.6e-7 of the block's scale)
    for a, b in zip(orc.matern_block_columns(I, cols, phi1, phi2), (Kap, pK, Kpp)):
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()

    def apply(which, V):                                           # [N, 16] -> [N, 16], eight columns per call
        return np.concatenate([eng.dense_apply(which, V[None, :, k:k + 8])[0] for k in (0, 8)], axis=1)

    E = np.zeros((N, 16))
    E[cols, np.arange(16)] = 1.0
    R = apply("C_inv", Kap) - E
    assert np.abs(R).max() < 100 * cond * EPS, ("C^-1 Kappa", np.abs(R).max(), cond)
    mK = apply("m", Kap)
    assert np.abs(mK - pK).max() < 100 * cond * EPS * np.abs(pK).max(), ("m Kappa", np.abs(mK - pK).max())
    K_ref = Kpp + apply("m", pK)
    R = apply("K_inv", K_ref) - E
    # K is a Schur complement: its scale (normK) is ~1e-4 of Kappa_pp's, so a rounding-level difference between two independent
    # evaluations of Kappa_pp (scipy's AMOS here, Temme / Steed on the device; both ~1e-14 of the block's scale) is that much larger
    # relative to K, and K^-1 carries it into the residual with cond(K)
    tolK = 500 * EPS * condK * np.abs(Kpp).max() / normK
    assert np.abs(R).max() < tolK, ("K^-1 K", np.abs(R).max(), tolK, cond, condK, normK)
    # transposes: C^-1 and K^-1 are symmetric by construction, so A^T v must reproduce A v to rounding of the summation order
    v = np.random.default_rng(1).standard_normal((1, N))
    for which in ("C_inv", "K_inv"):
        a, b = eng.dense_apply(which, v), eng.dense_apply(which, v, transpose=True)
        assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max(), which
    eng.close()


def test_config4_alpha_sweep_runs_through_the_api():
    """BASELINE config 4: ten datasets (alpha in {.05, .15} x seeds 0-4) x 8 chains each through the drop-in API (60 + 60 steps:
    a smoke-level length; statistical recovery is asserted at the reference's length in tests/test_fit_gpu.py).  Beyond shapes:
    the posterior-mean trajectories stay within a few noise standard deviations of the TRUE curves the files carry, the eight
    chains of a dataset are distinct, and the noisier datasets yield the wider posterior noise level."""
    import magi_v2
    sweep = np.load(os.path.join(GOLDEN, "seir_alpha_sweep.npz"))
    names = [k for k in sweep.files if k.startswith("alpha=")]
    assert len(names) == 10
    sig = {"0.05": [], "0.15": []}
    for name in names:
        rows = sweep[name]
        alpha = name.split("=")[1].split("_")[0]
        ts, X, truth = rows[:, 0], np.clip(rows[:, 1:5], 0.0, None), rows[:, 5:9]
        model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=ts, X_obs=X, bandsize=80, f_vec="seir4")
        model.initial_fit(discretization=1, hparam_iters=0)
        model.thetas_init = np.ones(3)
        res = model.predict(num_results=60, num_burnin_steps=60, n_chains=8, seed=1)
        assert res["X_samps"].shape == (8, 60, 161, 4) and np.isfinite(res["thetas_samps"]).all() and np.isfinite(res["X_samps"]).all()
        assert len({tuple(np.round(c, 12)) for c in res["thetas_samps"][:, -1]}) == 8                 # eight different chains
        noise_sd = float(alpha) * np.ptp(truth, axis=0)
        Xm = res["X_samps"].mean(axis=(0, 1))[::2]                                                    # posterior mean on the observation times
        assert np.all(np.abs(Xm - truth).max(axis=0) < 4.0 * noise_sd + 0.02), (name, np.abs(Xm - truth).max(axis=0), noise_sd)
        sig[alpha].append(np.sqrt(res["sigma_sqs_samps"].mean(axis=(0, 1))))
        model.engine.close()
    assert np.all(np.mean(sig["0.15"], axis=0) > np.mean(sig["0.05"], axis=0))


def test_config3_many_chains_sharded_equals_batched():
    """64 chains over 8 GPUs = 8 per GPU: chains 8..15 (rank 1's block) give the same samples when run
    as their own batch as when they run among 16 chains -- placement independence."""
    from magi_v2_amd.shard import chain_ids_for_rank
    eng, I, hp, Xhat, sp, tp = _synthetic_engine(256, None)
    cfg = eng.default_cfg(num_results=3, num_burnin_steps=5, stale_cache=0)
    rep = lambda v, n: np.repeat(np.asarray(v)[None], n, axis=0)
    ids_all = list(range(16))
    eng.sampler_init(cfg, rep(Xhat, 16), rep(sp, 16), rep(tp, 16), seed=4, chain_ids=ids_all)
    eng.sampler_run(8)
    _, _, tp_all = eng.sampler_samples()
    ids_r1 = chain_ids_for_rank(1, 8, 64)
    assert ids_r1 == list(range(8, 16))
    eng.sampler_init(cfg, rep(Xhat, 8), rep(sp, 8), rep(tp, 8), seed=4, chain_ids=ids_r1)
    eng.sampler_run(8)
    _, _, tp_r1 = eng.sampler_samples()
    np.testing.assert_array_equal(tp_all[8:16], tp_r1)
    eng.close()


def test_config3_at_size_n1024_eight_chains_per_gpu():
    """BASELINE config 3 at its size: N = 1024 x 4 components, 8 independent chains on one GPU (rank r of 8 owns global chain
    ids 8r .. 8r+7; here rank 1's block, i.e. ids that are not 0..7).
      * the batch of 8 equals the same ids run as batches of 4, 4 and 3 bit for bit, and -- for two of them -- one at a time
        (the one-chain kernel sums in another order: same tree sizes, states to 1e-8): a chain's samples do not depend on
        what shares the GPU with it (placement independence); with the option "family_chains" set to the job's largest per-GPU
        share the one- and two-chain batches are bit-identical too (one kernel family on every rank of an uneven shard);
      * chain 8 equals the CPU oracle draw for draw (tree depths, leapfrog counts, flags exact; states to 1e-8) on the
        GPU-BUILT matrices pulled to the host -- the oracle comparison the small-N sampler tests make, at config 2/3's size."""
    from magi_v2_amd.engine import MagiEngine
    from magi_v2_amd.shard import chain_ids_for_rank
    from oracle import magi_oracle as orc
    N, P = 1024, 3
    I, X_obs, truth, th = host.synthetic_seir(N, seed=0)
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(P), LB)
    eng = MagiEngine(0)
    C_inv, m, K_inv = eng.build_matrices(I, hp["phi1s"], hp["phi2s"], 2.01)
    eng.set_problem(Xi.mean(axis=0), N_ds.astype(np.float64), idx, y, beta, LB, "seir4")
    ids = chain_ids_for_rank(1, 8, 64)
    assert ids == list(range(8, 16))
    # (from TFP's initial step 0.1 every early proposal on this grid diverges at its first leaf and the chains never move:
    #  start at the step size the adaptation settles at, so that the transitions build trees and accept states)
    # 8 transitions (6 burn-in, of which 4 adapt the step size, + 2 kept), trees capped at depth 8 so that the oracle's share
    # (one numpy gradient per leapfrog at N = 1024) stays within a minute
    burnin, results, seed, step0, depth = 6, 2, 77, 1e-3, 8
    cfg = eng.default_cfg(num_results=results, num_burnin_steps=burnin, stale_cache=0, step_size=step0, max_tree_depth=depth)
    rep = lambda v, n: np.repeat(np.asarray(v)[None], n, axis=0)

    def run(chain_ids):
        n = len(chain_ids)
        eng.sampler_init(cfg, rep(Xhat, n), rep(sp0, n), rep(tp0, n), seed=seed, chain_ids=chain_ids)
        lf, _ = eng.sampler_run(burnin + results)
        return eng.sampler_samples(), eng.sampler_diag(), lf

    (X8, s8, t8), d8, lf8 = run(ids)
    assert lf8 == d8.leapfrogs_taken.sum() and np.isfinite(X8).all()
    assert d8.leapfrogs_taken.max() >= 31 and d8.is_accepted.sum() >= 16 and not np.array_equal(t8[0], t8[1])     # trees grow, states move, chains differ
    for part in (ids[:4], ids[4:], ids[5:8], [ids[2]], [ids[7]]):
        (Xp, sp_, tp_), dp, _ = run(part)
        sel = [ids.index(c) for c in part]
        np.testing.assert_array_equal(dp.leapfrogs_taken, d8.leapfrogs_taken[sel])
        np.testing.assert_array_equal(dp.tree_depth, d8.tree_depth[sel])
        if len(part) >= 3:          # batches of three or more chains run the same (matrix-core) kernel: bit for bit
            np.testing.assert_array_equal(Xp, X8[sel])
            np.testing.assert_array_equal(tp_, t8[sel])
        else:                       # one or two chains run the one-chain kernels: another summation order, same chain to rounding
            np.testing.assert_allclose(Xp, X8[sel], rtol=0, atol=1e-8 * np.abs(X8).max())
            np.testing.assert_allclose(tp_, t8[sel], rtol=1e-7, atol=1e-9)
    # An uneven shard (say 20 chains over 8 GPUs: 3, 3, 3, 3, 2, 2, 2, 2) would put some chains on the matrix-core kernel and others on the
    # VALU kernel.  With the job's largest per-GPU share handed to every handle (shard.family_chains_for -> option "family_chains") ONE family
    # serves every batch: one chain and two chains now equal their rows of the batch of 8 bit for bit as well.
    from magi_v2_amd.shard import family_chains_for
    assert family_chains_for(64, 8) == 8 and family_chains_for(20, 8) == 3 and family_chains_for(3, 8) == 1
    eng.set_option("family_chains", family_chains_for(20, 8))
    assert eng.stream_kernel_name(2).startswith("k_stream_sep")
    for part in ([ids[2]], ids[6:8]):
        (Xp, sp_, tp_), dp, _ = run(part)
        sel = [ids.index(c) for c in part]
        np.testing.assert_array_equal(Xp, X8[sel])
        np.testing.assert_array_equal(tp_, t8[sel])
        np.testing.assert_array_equal(dp.leapfrogs_taken, d8.leapfrogs_taken[sel])
    eng.set_option("family_chains", 0)
    assert eng.stream_kernel_name(2).startswith("k_stream<2>")
    eng.close()

    pr = orc.Problem(I=I, mu=Xi.mean(axis=0), C_inv=C_inv, m=m, K_inv=K_inv, N_ds=N_ds.astype(np.float64), obs_idx=idx, y=y,
                     beta=float(beta), LB=LB, drift="seir4", P=P)
    trace = []
    oX, osp, otp, info, _ = orc.sample_chain(pr, Xhat, hp["sigma_sqs"], np.ones(P), results, burnin, seed=seed, chain=ids[0],
                                             step_size=step0, stale_cache=False, trace=trace, max_tree_depth=depth)
    np.testing.assert_array_equal(d8.tree_depth[0], [r.depth for _, r, _ in trace])
    np.testing.assert_array_equal(d8.leapfrogs_taken[0], [r.leapfrogs for _, r, _ in trace])
    np.testing.assert_array_equal(d8.is_accepted[0], [int(r.is_accepted) for _, r, _ in trace])
    np.testing.assert_array_equal(d8.has_divergence[0], [int(r.has_divergence) for _, r, _ in trace])
    np.testing.assert_allclose(d8.step_size[0], [s for _, _, s in trace], rtol=1e-9)
    np.testing.assert_allclose(d8.target_log_prob[0], [r.target_log_prob for _, r, _ in trace], rtol=1e-8)
    np.testing.assert_allclose(X8[0], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
    np.testing.assert_allclose(t8[0], otp, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(s8[0], osp, rtol=1e-7, atol=1e-9)
