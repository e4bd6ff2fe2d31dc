"""CPU tests of SURVEY 8 row f2 -- the theta initialiser (magi_v2.py:133-179) and the joint (X_unobs, theta) initialisation
of completely unobserved components (magi_v2.py:182-249):

  * the oracle's restatement of ``theta_objective`` (incl. the reference's reshape, :155-156) against an op-for-op torch
    transcription differentiated by autograd (what tf.GradientTape does at :164-166) and against central differences;
  * the product's host-side initialisers (``MAGI_v2._fit_thetas_init`` -- both its exact-quadratic branch for drifts that
    are linear in theta and its general branch -- and ``MAGI_v2._fit_unobserved``) against the oracle's Adam loops.

The reference's own optimiser is tf_keras Adam, which is not installable here: the update rule is restated from its
documented defaults (parity unpinned, as for every TF-side piece)."""
import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import load_g4


def _vignette_pieces():
    g = load_g4("seir3_N161")
    return g, g["Xhat_init"], g["mu"], g["m"], g["K_inv"]


def test_theta_objective_matches_torch_transcription_and_finite_differences():
    import torch
    from oracle.torch_cpu import TORCH_DRIFTS
    g, Xhat, mu, m, K_inv = _vignette_pieces()
    N, D = Xhat.shape
    rng = np.random.default_rng(3)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64)
    for trial in range(3):
        th = rng.uniform(0.3, 6.0, 3)
        val, grad = orc.theta_init_objective(th, Xhat, mu, m, K_inv, "seir3")
        # magi_v2.py:139-158, line by line
        tht = torch.tensor(th, dtype=torch.float64, requires_grad=True)
        X_cent = torch.reshape(t(Xhat) - t(mu), (N, 1, D))
        m_prod = t(m) @ X_cent.permute(2, 0, 1)
        f_vals = torch.reshape(TORCH_DRIFTS["seir3"](None, t(Xhat), tht), (D, N, 1))
        toNorm = f_vals - m_prod
        loss = torch.sum(toNorm.permute(0, 2, 1) @ (t(K_inv) @ toNorm))
        loss.backward()
        assert abs(val - loss.item()) <= 1e-12 * abs(val)
        np.testing.assert_allclose(grad, tht.grad.numpy(), rtol=1e-10)
        for p in range(3):
            h = 1e-5 * th[p]
            e = np.zeros(3); e[p] = h
            fd = (orc.theta_init_objective(th + e, Xhat, mu, m, K_inv, "seir3")[0] - orc.theta_init_objective(th - e, Xhat, mu, m, K_inv, "seir3")[0]) / (2 * h)
            assert abs(fd - grad[p]) <= 1e-6 * abs(grad[p]) + 1e-6 * np.abs(grad).max()
    # the reshape is NOT the transpose the log posterior uses (magi_v2.py:335): the two objectives differ
    f = orc.DRIFTS["seir3"][0](Xhat, np.ones(3))[0]
    assert not np.allclose(np.reshape(f, (D, N)), f.T)


def _model(f_vec, D, P, Xhat, mu, m, K_inv, I):
    import magi_v2
    X_obs = np.array(Xhat)
    model = magi_v2.MAGI_v2(D_thetas=P, ts_obs=I, X_obs=X_obs, bandsize=None, f_vec=f_vec)
    model.I, model.mag_I = I.reshape(-1, 1), len(I)
    model.Xhat_init, model.mu_ds = np.array(Xhat), np.array(mu)
    model.m_ds, model.K_d_invs = np.array(m), np.array(K_inv)
    return model


def test_fit_thetas_init_linear_branch_equals_oracle():
    g, Xhat, mu, m, K_inv = _vignette_pieces()
    model = _model("seir3", 3, 3, Xhat, mu, m, K_inv, g["I"])
    for iters in (1, 200):
        got = model._fit_thetas_init(iters)
        want, losses = orc.fit_thetas_init(Xhat, mu, m, K_inv, "seir3", 3, num_iters=iters)
        np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10)
    assert losses[-1] < losses[0]


def test_fit_thetas_init_general_branch_equals_oracle():
    """A drift that is not linear in theta (FitzHugh-Nagumo: c and 1/c) takes the traced-Jacobian branch; the oracle
    differentiates the same callable by complex steps."""
    from magi_v2_amd.drift_examples import fitzhugh_nagumo, rk4
    from tests.test_drift_cpu import complex_step_jacobians
    I, X = rk4(fitzhugh_nagumo, [-1.0, 1.0], np.array([0.2, 0.2, 3.0]), 20.0, 41)
    C_inv, m, K_inv = orc.build_all(I, [1.0, 0.5], [1.5, 1.5], 2.01)
    mu = X.mean(axis=0)

    def fn(Xa, th):
        J, T = complex_step_jacobians(fitzhugh_nagumo, np.asarray(Xa, dtype=np.float64), np.asarray(th, dtype=np.float64))
        return np.asarray(fitzhugh_nagumo(None, Xa, th), dtype=np.float64), J, T
    orc.DRIFTS["fhn_cpu"] = (fn, 2, 3)
    try:
        model = _model(fitzhugh_nagumo, 2, 3, X, mu, m, K_inv, I)
        got = model._fit_thetas_init(200)
        want, losses = orc.fit_thetas_init(X, mu, m, K_inv, "fhn_cpu", 3, num_iters=200)
    finally:
        del orc.DRIFTS["fhn_cpu"]
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10)
    assert np.abs(want - 1.0).max() > 0.5 and losses[-1] < losses[0]           # 200 steps of lr .01 moved every entry


def test_fit_unobserved_equals_oracle():
    """SEIR-3 with E never observed (magi_v2.py:45-50, 182-249)."""
    import magi_v2
    from magi_v2_amd import host
    g3 = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "g3_pipeline.npz"))
    ts, X = g3["seir3_ts_obs"], g3["seir3_X_obs"].copy()
    X[:, 0] = np.nan
    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=ts, X_obs=X, bandsize=None, f_vec="seir3")
    model.I, model.X_obs_discret = host.discretize(ts, X, 1)
    model.mag_I = model.I.shape[0]
    model.X_interp_obs = host.linear_interpolate(model.X_obs_discret[:, model.observed_indicators])
    Xu, th = model._fit_unobserved(150, seed=4)
    Xs = orc.cubic_smoother(model.I, model.X_interp_obs)
    oXu, oth, losses = orc.fit_unobserved(model.I, model.X_interp_obs, Xs, model.proper_order, model.unobserved_components, "seir3", 3, seed=4, num_iters=150)
    np.testing.assert_allclose(Xu, oXu, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(th, oth, rtol=1e-8, atol=1e-10)
    assert losses[-1] < losses[0]
    # the oracle's written-out gradient against central differences of the reference-op-order loss
    rng = np.random.default_rng(0)
    v, gX, gth = orc.gradient_matching_loss_and_grads(model.I, Xs, oXu, oth, model.proper_order, model.unobserved_components, "seir3")
    assert v == pytest.approx(orc.gradient_matching_loss(model.I, Xs, oXu, oth, model.proper_order, "seir3"), rel=1e-13)
    dX, dth = rng.standard_normal(oXu.shape), rng.standard_normal(3)
    h = 1e-6
    fd = (orc.gradient_matching_loss(model.I, Xs, oXu + h * dX, oth + h * dth, model.proper_order, "seir3") -
          orc.gradient_matching_loss(model.I, Xs, oXu - h * dX, oth - h * dth, model.proper_order, "seir3")) / (2 * h)
    assert abs(fd - ((gX * dX).sum() + gth @ dth)) <= 1e-6 * abs(fd) + 1e-8
