"""Partially observed systems end to end (magi_v2.py:45-50, 182-268): a component that is
never observed is initialised by gradient matching, gets its own hyper-parameters and matrices, and is sampled
with N_d = 0 observations."""
import os

import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fitted():
    import magi_v2
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    X_obs = g["seir3_X_obs"].copy()
    E_noisy = X_obs[:, 0].copy()
    X_obs[:, 0] = np.nan                                    # E is never observed
    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=g["seir3_ts_obs"], X_obs=X_obs, bandsize=None, f_vec="seir3")
    model.initial_fit(discretization=1, hparam_iters=100, theta_init_iters=4000)
    return model, E_noisy


def test_initial_fit_fills_the_unobserved_component(fitted):
    model, E_noisy = fitted
    assert list(model.unobserved_components) == [0] and list(model.observed_components) == [1, 2]
    assert list(model.proper_order) == [2, 0, 1] and model.N_ds.tolist() == [0, 81, 81]
    for a in (model.phi1s, model.phi2s, model.sigma_sqs_init, model.mu_ds, model.thetas_init, model.Xhat_init):
        assert np.isfinite(a).all()
    assert model.C_d_invs.shape == (3, 161, 161) and np.abs(model.C_d_invs[0]).max() > 0
    # gradient matching recovers the SHAPE of E (its scale is confounded with sigma, which multiplies it)
    assert np.corrcoef(model.Xhat_init[::2, 0], E_noisy)[0, 1] > 0.85


def test_log_posterior_with_an_unobserved_component_matches_oracle(fitted):
    model, _ = fitted
    LB = orc.sigma_sqs_lower_bound(model.Xhat_init)
    pr = orc.Problem(I=model.I[:, 0], mu=model.mu_ds, C_inv=model.C_d_invs, m=model.m_ds, K_inv=model.K_d_invs,
                     N_ds=model.N_ds.astype(float), obs_idx=model.not_nan_idxs, y=model.y_tau_ds_observed, beta=float(model.beta),
                     LB=LB, drift="seir3", P=3)
    eng = model.engine
    model._sync_matrices()
    eng.set_problem(model.mu_ds, model.N_ds.astype(float), model.not_nan_idxs, model.y_tau_ds_observed, float(model.beta), LB, model.drift)
    rng = np.random.default_rng(8)
    X = model.Xhat_init + rng.normal(0, 0.003, model.Xhat_init.shape)
    sp, tp = rng.normal(-6, 0.5, 3), rng.normal(0.5, 0.3, 3)
    L, gX, gs, gt = orc.logpost_grad(X, sp, tp, 1.0, pr)
    for fused in (False, True):
        out = eng.logpost_grad(X, sp, tp, 1.0, fused=fused)
        assert abs(out[0] - L) <= 1e-9 * abs(L)
        np.testing.assert_allclose(out[1], gX, rtol=0, atol=1e-9 * np.abs(gX).max())
        np.testing.assert_allclose(out[2], gs, rtol=1e-7, atol=1e-9 * np.abs(gX).max())
        np.testing.assert_allclose(out[3], gt, rtol=1e-7, atol=1e-9 * np.abs(gX).max())


def test_predict_samples_the_unobserved_component(fitted):
    model, E_noisy = fitted
    res = model.predict(num_results=150, num_burnin_steps=150, n_chains=2, seed=2, stale_cache=False)
    assert res["X_samps"].shape == (2, 150, 161, 3) and np.isfinite(res["X_samps"]).all()
    assert (res["sigma_sqs_samps"] > 0).all() and (res["thetas_samps"] > 0).all()
    E_post = res["X_samps"].mean(axis=(0, 1))[::2, 0]
    assert np.corrcoef(E_post, E_noisy)[0, 1] > 0.9
    # the flux sigma * E (what dI/dt sees) is identified even though neither factor is
    sig = res["thetas_samps"].reshape(-1, 3)[:, 2].mean()
    flux, flux_true = sig * E_post.mean(), 1.8 * E_noisy.mean()
    assert 0.5 * flux_true < flux < 2.0 * flux_true, (flux, flux_true)
