"""Generic ODEs on the GPU (callers magi_v2.py:155, 206, 335): a traced f_vec runs through the
kernels compiled for it and matches the oracle, whose Jacobians come from complex-step differentiation of the
same callable (independent of the sympy tracing)."""
import numpy as np
import pytest

from magi_v2_amd import drift, host
from magi_v2_amd.drift_examples import EXAMPLES, fitzhugh_nagumo, rk4
from magi_v2_amd.engine import MagiEngine
from oracle import magi_oracle as orc
from tests.test_drift_cpu import complex_step_jacobians

pytestmark = pytest.mark.gpu


def oracle_drift(f_vec):
    def fn(X, th):
        J, T = complex_step_jacobians(f_vec, np.asarray(X, dtype=np.float64), np.asarray(th, dtype=np.float64))
        return np.asarray(f_vec(None, X, th), dtype=np.float64), J, T
    return fn


def make_problem(name, N=41, band=None, seed=0):
    f_vec, D, P = EXAMPLES[name]
    truth = {"fhn": np.array([0.2, 0.2, 3.0]), "lotka_volterra": np.array([1.5, 1.0, 3.0, 1.0]),
             "ptrans": np.array([0.07, 0.6, 0.05, 0.3, 0.017, 0.3]),
             "competition7": np.array([1.5, 1.0, 3.0, 1.0, 0.1, 0.05, 0.2])}[name]
    x0 = {"fhn": [-1.0, 1.0], "lotka_volterra": [1.0, 1.5], "ptrans": [1.0, 0.0, 1.0, 0.0, 0.0], "competition7": [1.0, 1.5]}[name]
    I, X = rk4(f_vec, x0, truth, {"fhn": 20.0, "lotka_volterra": 8.0, "ptrans": 100.0, "competition7": 8.0}[name], N)
    rng = np.random.default_rng(seed)
    X_obs = X + rng.normal(0, 0.1, X.shape)
    X_obs[1::2] = np.nan                                                  # observations on every other grid point
    d = drift.resolve(f_vec, D, P)
    eng = MagiEngine(0, drift=d)
    Xi = host.linear_interpolate(X_obs)
    hp = host.hparams_initial(Xi)
    C_inv, m, K_inv = eng.build_matrices(I, hp["phi1s"], np.full(D, 1.5), 2.01, bandsize=None)
    N_ds, beta, idx, y = host.observation_bookkeeping(X_obs, X_obs)
    Xhat = host.cubic_smoother(I, Xi)
    LB = host.sigma_sqs_lower_bound(Xhat)
    orc.DRIFTS[name] = (oracle_drift(f_vec), D, P)
    pr = orc.Problem(I=I, mu=Xi.mean(axis=0), C_inv=orc.band_part(C_inv, band), m=orc.band_part(m, band), K_inv=orc.band_part(K_inv, band),
                     N_ds=N_ds.astype(np.float64), obs_idx=idx, y=y, beta=float(beta), LB=LB, drift=name, P=P)
    eng.set_matrices(C_inv, m, K_inv, bandsize=band)
    eng.set_problem(pr.mu, pr.N_ds, idx, y, beta, LB, d)
    return eng, pr, Xhat, hp, truth


@pytest.mark.parametrize("name,band", [("fhn", None), ("lotka_volterra", None), ("fhn", 6), ("ptrans", None), ("competition7", None)])
def test_user_drift_log_posterior_and_gradient_match_oracle(name, band):
    eng, pr, Xhat, hp, truth = make_problem(name, band=band)
    rng = np.random.default_rng(5)
    D, P = Xhat.shape[1], len(truth)
    for rep in range(2):
        X = Xhat + rng.normal(0, 0.05, Xhat.shape)
        sp, tp = rng.normal(-3, 0.5, D), rng.normal(0.3, 0.4, P)
        for temp in (1.0, 0.1316):
            L, gX, gs, gt = orc.logpost_grad(X, sp, tp, temp, pr)
            for fused in (False, True):
                out = eng.logpost_grad(X, sp, tp, temp, fused=fused)
                assert abs(out[0] - L) <= 1e-9 * abs(L), (name, fused)
                scale = np.abs(gX).max()
                np.testing.assert_allclose(out[1], gX, rtol=0, atol=1e-9 * scale)
                np.testing.assert_allclose(out[2], gs, rtol=1e-8, atol=1e-9 * scale)
                np.testing.assert_allclose(out[3], gt, rtol=1e-8, atol=1e-9 * scale)
    eng.close()


def test_user_drift_chain_matches_oracle_draw_for_draw():
    eng, pr, Xhat, hp, truth = make_problem("fhn")
    LB = pr.LB
    sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(3), LB)
    cfg = eng.default_cfg(num_results=4, num_burnin_steps=8, stale_cache=0)
    eng.sampler_init(cfg, Xhat, sp0, tp0, seed=77)
    eng.sampler_run(12)
    Xs, sp, tp = eng.sampler_samples()
    diag = eng.sampler_diag()
    trace = []
    oX, osp, otp, info, _ = orc.sample_chain(pr, Xhat, hp["sigma_sqs"], np.ones(3), 4, 8, seed=77, stale_cache=False, trace=trace)
    np.testing.assert_array_equal(diag.leapfrogs_taken[0], [r.leapfrogs for _, r, _ in trace])
    np.testing.assert_array_equal(diag.tree_depth[0], [r.depth for _, r, _ in trace])
    np.testing.assert_allclose(tp[0], otp, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Xs[0], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
    eng.close()


def test_user_drift_through_the_api_and_posterior_is_centred_on_the_truth():
    """The reference's call sequence with a callable that matches no compiled-in drift.  With the annealing off and
    the chain started at the truth, the posterior mean of theta must stay there (a wrong drift Jacobian would not)."""
    import magi_v2
    truth = np.array([0.2, 0.2, 3.0])
    ts, X = rk4(fitzhugh_nagumo, [-1.0, 1.0], truth, 20.0, 41)
    X_obs = X + np.random.default_rng(0).normal(0, 0.1, X.shape)
    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=ts, X_obs=X_obs, bandsize=None, f_vec=fitzhugh_nagumo)
    assert not model.drift.is_builtin
    model.initial_fit(discretization=2, hparams={"phi2s": [2.0, 2.0], "sigma_sqs": [0.01, 0.01]}, theta_init_iters=200)
    assert model.thetas_init.shape == (3,) and np.isfinite(model.thetas_init).all()
    model.thetas_init = truth.copy()
    res = model.predict(num_results=200, num_burnin_steps=200, n_chains=2, seed=4, stale_cache=False, anneal=False)
    assert res["X_samps"].shape == (2, 200, 161, 2) and np.isfinite(res["X_samps"]).all()
    th = res["thetas_samps"].reshape(-1, 3).mean(axis=0)
    assert np.all(np.abs(th - truth) < np.array([0.1, 0.25, 0.4])), th


def test_tiny_grid_runs_and_matches_oracle():
    """Smallest sensible problem (N = 11 grid points, one operator block, one point workgroup): gradient and a short chain."""
    eng, pr, Xhat, hp, truth = make_problem("lotka_volterra", N=11)
    rng = np.random.default_rng(6)
    X = Xhat + rng.normal(0, 0.02, Xhat.shape)
    sp, tp = rng.normal(-3, 0.3, 2), rng.normal(0.3, 0.3, 4)
    L, gX, gs, gt = orc.logpost_grad(X, sp, tp, 1.0, pr)
    for fused in (False, True):
        out = eng.logpost_grad(X, sp, tp, 1.0, fused=fused)
        assert abs(out[0] - L) <= 1e-9 * abs(L)
        np.testing.assert_allclose(out[1], gX, rtol=0, atol=1e-9 * np.abs(gX).max())
    sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.ones(4), pr.LB)
    cfg = eng.default_cfg(num_results=3, num_burnin_steps=5, stale_cache=0)
    eng.sampler_init(cfg, Xhat, sp0, tp0, seed=5)
    eng.sampler_run(8)
    _, _, tp_s = eng.sampler_samples()
    trace = []
    _, _, otp, _, _ = orc.sample_chain(pr, Xhat, hp["sigma_sqs"], np.ones(4), 3, 5, seed=5, stale_cache=False, trace=trace)
    np.testing.assert_array_equal(eng.sampler_diag().leapfrogs_taken[0], [r.leapfrogs for _, r, _ in trace])
    np.testing.assert_allclose(tp_s[0], otp, rtol=1e-7, atol=1e-9)
    eng.close()


def test_five_component_system_chain_matches_oracle():
    """More than four components (the protein-transduction benchmark, D = 5, P = 6): the specialised library is built with
    eight component lanes per grid point; a NUTS chain still equals the oracle draw for draw."""
    eng, pr, Xhat, hp, truth = make_problem("ptrans")
    sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], np.full(6, 0.2), pr.LB)
    cfg = eng.default_cfg(num_results=3, num_burnin_steps=7, stale_cache=0)
    eng.sampler_init(cfg, Xhat, sp0, tp0, seed=21)
    eng.sampler_run(10)
    Xs, sp, tp = eng.sampler_samples()
    trace = []
    oX, osp, otp, info, _ = orc.sample_chain(pr, Xhat, hp["sigma_sqs"], np.full(6, 0.2), 3, 7, seed=21, stale_cache=False, trace=trace)
    np.testing.assert_array_equal(eng.sampler_diag().leapfrogs_taken[0], [r.leapfrogs for _, r, _ in trace])
    np.testing.assert_allclose(tp[0], otp, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(sp[0], osp, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Xs[0], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
    eng.close()


def test_five_component_system_through_the_api():
    """MAGI_v2(...) -> initial_fit -> predict with the 5-component protein-transduction f_vec (one component never observed)."""
    import magi_v2
    from magi_v2_amd.drift_examples import protein_transduction
    truth = np.array([0.07, 0.6, 0.05, 0.3, 0.017, 0.3])
    ts, X = rk4(protein_transduction, [1.0, 0.0, 1.0, 0.0, 0.0], truth, 100.0, 26)
    X_obs = X + np.random.default_rng(1).normal(0, 0.01, X.shape)
    X_obs[:, 1] = np.nan                                        # the degraded-signal component is not observed
    model = magi_v2.MAGI_v2(D_thetas=6, ts_obs=ts, X_obs=X_obs, bandsize=None, f_vec=protein_transduction)
    model.initial_fit(discretization=1, hparam_iters=20, theta_init_iters=500)
    assert model.C_d_invs.shape == (5, 51, 51) and np.isfinite(model.thetas_init).all() and np.isfinite(model.Xhat_init).all()
    res = model.predict(num_results=20, num_burnin_steps=20, seed=3, stale_cache=False)
    assert res["X_samps"].shape == (20, 51, 5) and res["thetas_samps"].shape == (20, 6)
    assert np.isfinite(res["X_samps"]).all() and (res["thetas_samps"] > 0).all()


def test_seven_parameter_system_chain_matches_oracle():
    """More than six parameters (P = 7): the specialised library is built with wider parameter blocks.  The chain starts at
    the generating parameters with a small step so that it samples (from theta = 1 with TFP's step 0.1 this quadratic system
    overflows on both sides, which would only compare divergence handling)."""
    eng, pr, Xhat, hp, truth = make_problem("competition7")
    sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], truth, pr.LB)
    cfg = eng.default_cfg(num_results=3, num_burnin_steps=6, stale_cache=0, step_size=2e-3)
    eng.sampler_init(cfg, Xhat, sp0, tp0, seed=31)
    eng.sampler_run(9)
    Xs, sp, tp = eng.sampler_samples()
    d = eng.sampler_diag()
    trace = []
    with np.errstate(all="raise"):                                         # no overflow / NaN anywhere on the oracle side
        oX, osp, otp, info, _ = orc.sample_chain(pr, Xhat, hp["sigma_sqs"], truth, 3, 6, seed=31, step_size=2e-3, stale_cache=False, trace=trace)
    assert np.isfinite(Xs).all() and np.isfinite(tp).all()
    # (dual averaging overshoots twice while it adapts -- two divergent one-leaf trees -- then the chain samples: trees of 31 .. 255 leaves)
    assert d.is_accepted[0].sum() >= 6 and d.leapfrogs_taken[0].max() >= 100 and d.has_divergence[0].sum() <= 2
    np.testing.assert_array_equal(d.leapfrogs_taken[0], [r.leapfrogs for _, r, _ in trace])
    np.testing.assert_array_equal(d.is_accepted[0], [int(r.is_accepted) for _, r, _ in trace])
    np.testing.assert_array_equal(d.has_divergence[0], [int(r.has_divergence) for _, r, _ in trace])
    lar = np.array([r.log_accept_ratio for _, r, _ in trace])
    fin = np.isfinite(lar)
    np.testing.assert_array_equal(np.isfinite(d.log_accept_ratio[0]), fin)
    np.testing.assert_allclose(d.log_accept_ratio[0][fin], lar[fin], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(tp[0], otp, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Xs[0], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
    eng.close()


@pytest.mark.parametrize("name", ["ptrans", "competition7", "fhn"])
def test_user_drift_batched_states_on_the_matrix_core_kernel(name, monkeypatch):
    """Three or more states per call on the matrix-core streaming kernel (k_stream_mc / k_stream_sep; these fixtures are small, so MAGI_STREAM_FAMILY=mc selects it); for a traced drift with 5 components /
    7 parameters that is the library built with the wider lane groups.  Batched fused == batched three-phase == per-state oracle."""
    monkeypatch.setenv("MAGI_STREAM_FAMILY", "mc")
    eng, pr, Xhat, hp, truth = make_problem(name)
    rng = np.random.default_rng(11)
    D, P, n = Xhat.shape[1], len(truth), 5
    X = Xhat[None] + rng.normal(0, 0.05, (n,) + Xhat.shape)
    sp, tp = rng.normal(-3, 0.5, (n, D)), np.log(np.expm1(truth))[None] + rng.normal(0, 0.2, (n, P))
    a = eng.logpost_grad(X, sp, tp, 0.7)
    b = eng.logpost_grad(X, sp, tp, 0.7, fused=True)
    for c in range(n):
        L, gX, gs, gt = orc.logpost_grad(X[c], sp[c], tp[c], 0.7, pr)
        for out in (a, b):
            assert abs(out[0][c] - L) <= 1e-9 * abs(L)
            np.testing.assert_allclose(out[1][c], gX, rtol=0, atol=1e-9 * np.abs(gX).max())
            np.testing.assert_allclose(out[3][c], gt, rtol=1e-8, atol=1e-9 * np.abs(gX).max())
    eng.close()


@pytest.mark.parametrize("name,chains,theta0", [("fhn", 1, 1.0), ("fhn", 3, 1.0), ("lotka_volterra", 1, 1.0), ("lotka_volterra", 4, 1.0),
                                                  ("ptrans", 1, 0.2), ("ptrans", 3, 0.2), ("competition7", 3, 1.0)])
def test_traced_drift_deep_trees_match_oracle_in_every_kernel_family(name, chains, theta0, stream_family):
    """Every traced drift is its own library, i.e. its own instantiations of the streaming kernels and of the decisions inlined into
    them: one chain (VALU kernel) and a batch (k_stream_sep for the separable FitzHugh-Nagumo -- three basis functions, two planes --,
    Lotka-Volterra and the seven-parameter competition system; k_stream_mc for the protein-transduction system, whose V x / (K + x)
    term is not separable) against the oracle on transitions that build trees (first step 2e-3: a rejected one-leapfrog transition,
    what TFP's 0.1 produces early on, hides whatever the leapfrog computed)."""
    eng, pr, Xhat, hp, truth = make_problem(name)
    P = len(truth)
    th0 = np.full(P, theta0)
    sp0, tp0 = host.softplus_inverse_inits(hp["sigma_sqs"], th0, pr.LB)
    burnin, results, step0, depth = 4, 2, 2e-3, 6
    cfg = eng.default_cfg(num_results=results, num_burnin_steps=burnin, stale_cache=0, step_size=step0, max_tree_depth=depth)
    rep = lambda v: np.repeat(np.asarray(v)[None], chains, axis=0)
    ids = list(range(40, 40 + chains))
    eng.sampler_init(cfg, rep(Xhat), rep(sp0), rep(tp0), seed=515, chain_ids=ids)
    lf, _ = eng.sampler_run(burnin + results)
    Xs, sp, tp = eng.sampler_samples()
    d = eng.sampler_diag()
    eng.close()
    assert lf == d.leapfrogs_taken.sum() and d.leapfrogs_taken.max() >= 15 and d.is_accepted.sum() >= chains
    for i in sorted({0, chains - 1}):
        trace = []
        oX, osp, otp, info, _ = orc.sample_chain(pr, Xhat, hp["sigma_sqs"], th0, results, burnin, seed=515, chain=ids[i], step_size=step0,
                                                 stale_cache=False, trace=trace, max_tree_depth=depth)
        np.testing.assert_array_equal(d.leapfrogs_taken[i], [r.leapfrogs for _, r, _ in trace])
        np.testing.assert_array_equal(d.is_accepted[i], [int(r.is_accepted) for _, r, _ in trace])
        np.testing.assert_allclose(d.target_log_prob[i], [r.target_log_prob for _, r, _ in trace], rtol=1e-8)
        np.testing.assert_allclose(tp[i], otp, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(Xs[i], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
