"""GPU test of the drop-in Python surface (magi_v2.MAGI_v2): the vignette's call sequence
(vignette.ipynb cells 5-8; test_magi_script.py:72-81) end to end on the thinned SEIR rows."""
import os

import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


def vignette_f_vec(t, X, thetas):
    S = 1.0 - np.reshape(np.sum(X, axis=1), (-1, 1))
    return np.concatenate([(thetas[0] * S * X[:, 1:2]) - (thetas[2] * X[:, 0:1]),
                           (thetas[2] * X[:, 0:1]) - (thetas[1] * X[:, 1:2]),
                           (thetas[1] * X[:, 1:2])], axis=1)


@pytest.fixture(scope="module")
def fitted():
    import magi_v2
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=g["seir3_ts_obs"], X_obs=g["seir3_X_obs"], bandsize=80, f_vec=vignette_f_vec)
    # hparam_iters=0: the reference's starting hyper-parameters, so the matrices are comparable with the oracle's
    # well-conditioned build; the fitted default is covered by tests/test_fit_gpu.py
    model.initial_fit(discretization=1, verbose=False, hparam_iters=0)
    return model, g


def test_initial_fit_fills_the_reference_attributes(fitted):
    model, g = fitted
    assert model.mag_I == 161 and model.I.shape == (161, 1)
    np.testing.assert_array_equal(model.I, g["seir3_I"])
    np.testing.assert_allclose(model.Xhat_init, g["seir3_Xhat_smoothed"], atol=1e-14)
    assert model.C_d_invs.shape == model.m_ds.shape == model.K_d_invs.shape == (3, 161, 161)
    assert model.beta == pytest.approx(3 * 161 / 243)
    hp = orc.hparams_initial(g["seir3_X_interp"])
    np.testing.assert_array_equal(model.phi1s, hp["phi1s"])
    # matrices: GPU build vs the oracle's restatement of the reference build (conditioning-limited)
    C_inv, m, K_inv = orc.build_all(model.I, model.phi1s, model.phi2s, 2.01, bandsize=80)
    for d in range(3):
        assert np.abs(model.m_ds[d] - m[d]).max() < 1e-6 * np.abs(m[d]).max()
        assert np.abs(model.C_d_invs[d] - C_inv[d]).max() < 1e-6 * np.abs(C_inv[d]).max()
    # band mask applied to the host copies exactly as tf.linalg.band_part does (magi_v2.py:271-274)
    i = np.arange(161)
    far = np.abs(i[:, None] - i[None, :]) > 80
    assert (model.C_d_invs[:, far] == 0).all() and (model.m_ds[:, far] == 0).all()
    # theta init (magi_v2.py:133-179): 10 000 Adam steps on the t2-only objective with the reference's reshape, on the
    # UNbanded matrices and the interpolated (not yet smoothed) grid -- against the oracle's restatement run on the oracle's
    # own build (the two builds agree to ~1e-6, the converged optimum follows continuously; exact agreement on identical
    # matrices is tests/test_theta_init_cpu.py)
    Cd, md, Kd = orc.build_all(model.I, model.phi1s, model.phi2s, 2.01, bandsize=None)
    Xi = g["seir3_X_interp"]
    want, losses = orc.fit_thetas_init(Xi, Xi.mean(axis=0), md, Kd, "seir3", 3, num_iters=10000)
    assert model.thetas_init.shape == (3,)
    np.testing.assert_allclose(model.thetas_init, want, rtol=2e-3, atol=2e-3)


def test_predict_returns_the_reference_results_dictionary(fitted):
    model, g = fitted
    res = model.predict(num_results=30, num_burnin_steps=30, seed=7)
    for key in ("phi1s", "phi2s", "Xhat_init", "sigma_sqs_init", "thetas_init", "I", "X_samps", "sigma_sqs_samps",
                "thetas_samps", "kernel_results", "sample_results", "minutes_elapsed"):
        assert key in res, key                                                   # magi_v2.py:412-422
    assert res["X_samps"].shape == (30, 161, 3)
    assert res["sigma_sqs_samps"].shape == (30, 3) and res["thetas_samps"].shape == (30, 3)
    assert (res["thetas_samps"] > 0).all()
    LB = orc.sigma_sqs_lower_bound(model.Xhat_init)
    assert (res["sigma_sqs_samps"] > LB).all()
    assert res["kernel_results"]["leapfrogs_taken"].shape == (30,)
    # same seed -> same chain; several chains -> leading axis, chain 0 unchanged
    res2 = model.predict(num_results=30, num_burnin_steps=30, seed=7, n_chains=3)
    assert res2["X_samps"].shape == (3, 30, 161, 3)
    # (three or more chains per GPU run the matrix-core streaming kernel, one or two the one-chain kernel: another summation
    #  order, the same chain to rounding -- bit-identity holds between batches of the same kernel family, tests/test_fullsize_gpu.py)
    np.testing.assert_allclose(res2["thetas_samps"][0], res["thetas_samps"], rtol=1e-6)
    np.testing.assert_array_equal(res2["kernel_results"]["leapfrogs_taken"][0], res["kernel_results"]["leapfrogs_taken"])
    assert not np.allclose(res2["thetas_samps"][1], res["thetas_samps"])


def test_predict_matches_oracle_chain_through_the_api(fitted):
    model, g = fitted
    res = model.predict(num_results=3, num_burnin_steps=5, seed=99)
    LB = orc.sigma_sqs_lower_bound(model.Xhat_init)
    N_ds, beta, idx, y = model.N_ds.astype(float), model.beta, model.not_nan_idxs, model.y_tau_ds_observed
    pr = orc.Problem(I=model.I[:, 0], mu=model.mu_ds, C_inv=model.C_d_invs, m=model.m_ds, K_inv=model.K_d_invs, N_ds=N_ds,
                     obs_idx=idx, y=y, beta=float(beta), LB=LB, drift="seir3", P=3)
    trace = []
    oX, osp, otp, info, _ = orc.sample_chain(pr, model.Xhat_init, model.sigma_sqs_init, model.thetas_init, 3, 5, seed=99, trace=trace)
    np.testing.assert_array_equal(res["kernel_results"]["leapfrogs_taken"], [r.leapfrogs for _, r, _ in trace][5:])
    _, oth = orc.transform_samples(osp, otp, LB)
    np.testing.assert_allclose(res["thetas_samps"], oth, rtol=1e-6)


def test_user_overwritten_matrices_and_nan_asserts(fitted):
    model, g = fitted
    keep = model.K_d_invs.copy()
    model.K_d_invs = keep * 1.5                    # the reference lets users overwrite these (magi_v2.py:77-80)
    r1 = model.predict(num_results=2, num_burnin_steps=2, seed=3)
    model.K_d_invs = keep
    r2 = model.predict(num_results=2, num_burnin_steps=2, seed=3)
    assert not np.allclose(r1["kernel_results"]["target_log_prob"], r2["kernel_results"]["target_log_prob"])
    bad = model.thetas_init.copy()
    model.thetas_init = np.array([np.nan, 1.0, 1.0])
    with pytest.raises(AssertionError, match="thetas_init"):
        model.predict(num_results=2, num_burnin_steps=2)
    model.thetas_init = bad


def test_update_kernel_matrices_for_forecasting(fitted):
    model, g = fitted
    I_new = np.concatenate([model.I[:, 0], model.I[-1, 0] + 0.025 * np.arange(1, 21)])      # 20 forecast points
    old_beta = model.beta
    model.update_kernel_matrices(I_new, model.phi1s, model.phi2s)
    assert model.mag_I == 181 and model.C_d_invs.shape == (3, 181, 181)
    assert model.beta == pytest.approx(3 * 181 / 243) and model.beta != old_beta
    Kap, _, _ = orc.matern_blocks(model.I, model.phi1s[1], model.phi2s[1], 2.01)
    dense = model.engine.build_matrices(model.I, model.phi1s[1:2], model.phi2s[1:2], 2.01)[0][0]
    assert np.abs(dense @ Kap - np.eye(181)).max() < 1e-6


def test_matrices_stay_on_the_device_until_read_and_edits_are_seen():
    """The Eqn.-6 matrices are built into the handle and packed there; C_d_invs / m_ds / K_d_invs are host mirrors that are
    downloaded on first read (band-masked, as the reference's attributes are after initial_fit) and uploaded again only
    when the caller assigned one or edited a downloaded copy in place (magi_v2.py:77-80)."""
    import magi_v2
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=g["seir3_ts_obs"], X_obs=g["seir3_X_obs"], bandsize=80, f_vec="seir3")
    model.initial_fit(discretization=1, hparam_iters=0)
    assert all(m is None for m in model._host_mats)                          # nothing crossed PCIe
    base = model.predict(num_results=6, num_burnin_steps=6, seed=11)
    assert all(m is None for m in model._host_mats)
    # the N x N products of the theta initialiser ran on the device copies: same numbers as numpy on a download
    Cd, md, Kd = model.engine.get_dense(None)
    V = np.random.default_rng(0).standard_normal((3, 161, 2))
    np.testing.assert_allclose(model.engine.dense_apply("m", V), np.einsum("dij,djp->dip", md, V), rtol=1e-12, atol=1e-12 * np.abs(md).max())
    np.testing.assert_allclose(model.engine.dense_apply("K_inv", V[:, :, 0], transpose=True), np.einsum("dji,dj->di", Kd, V[:, :, 0]), rtol=1e-12,
                               atol=1e-12 * np.abs(Kd).max())
    # first read downloads the masked matrices
    C = model.C_d_invs
    i = np.arange(161)
    assert C.shape == (3, 161, 161) and (C[:, np.abs(i[:, None] - i[None, :]) > 80] == 0).all()
    np.testing.assert_array_equal(C[:, np.abs(i[:, None] - i[None, :]) <= 80], Cd[:, np.abs(i[:, None] - i[None, :]) <= 80])
    same = model.predict(num_results=6, num_burnin_steps=6, seed=11)        # reading alone changes nothing
    np.testing.assert_array_equal(same["thetas_samps"], base["thetas_samps"])
    # an in-place edit of the downloaded copy is detected (full-content hash) and uploaded
    model.C_d_invs[0] *= 1.5
    edited = model.predict(num_results=6, num_burnin_steps=6, seed=11)
    assert not np.array_equal(edited["X_samps"], base["X_samps"])
    # assigning the original back restores the original chain
    model.C_d_invs = C / np.array([1.5, 1.0, 1.0])[:, None, None]
    back = model.predict(num_results=6, num_burnin_steps=6, seed=11)
    np.testing.assert_allclose(back["thetas_samps"], base["thetas_samps"], rtol=1e-9)
    model.engine.close()
