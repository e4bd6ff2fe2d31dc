"""GPU tests of the hyper-parameter fit (reference magi_v2.py:538-691): the device evaluation of
the GP marginal likelihood and its gradient against the oracle's numpy restatement, the Adam trajectory, and the
vignette end to end with FITTED hyper-parameters (the reference's run recovered theta = (5.831, 0.565, 1.77))."""
import os

import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    return g["seir3_I"][:, 0], g["seir3_X_interp"], g


def test_device_resident_loop_equals_host_loop(data, monkeypatch):
    """The fit replays one captured graph per Adam step with the scalar tail (likelihood, priors, Adam) in a one-thread
    kernel; MAGI_FIT_HOST_LOOP=1 runs the same kernels with that tail on the host.  Same arithmetic up to libm."""
    from magi_v2_amd.engine import MagiEngine
    I, X, g = data
    I, X = I[:161], X[:161]
    pri = [orc.fourier_phi2_prior(X[:, d]) for d in range(3)]
    init = orc.hparams_initial(X)
    outs = []
    for host_loop in (False, True):
        if host_loop:
            monkeypatch.setenv("MAGI_FIT_HOST_LOOP", "1")
        eng = MagiEngine(0)
        outs.append(eng.fit_hparams(I, X, X.mean(axis=0), [p[0] for p in pri], [p[1] for p in pri], init["sigma_sqs"],
                                    init["phi1s"], init["phi2s"], init["sigma_sqs"], num_iters=60, want_trace=True))
        eng.close()
    for k in ("phi1s", "phi2s", "sigma_sqs"):
        np.testing.assert_allclose(outs[0][k], outs[1][k], rtol=1e-9)
    np.testing.assert_allclose(outs[0]["loss"], outs[1]["loss"], rtol=1e-11)


def test_single_adam_step_equals_oracle(data):
    """One iteration isolates the device marginal-likelihood gradient: after a single Adam step every variable
    moves by lr in the direction -sign(grad), so compare a 1-step and a 25-step trajectory with the oracle."""
    from magi_v2_amd.engine import MagiEngine
    I, X, g = data
    sub = slice(0, 81)                      # N = 81 keeps the numpy oracle fast
    I, X = I[sub], X[sub]
    eng = MagiEngine(0)
    pri = [orc.fourier_phi2_prior(X[:, d]) for d in range(3)]
    init = orc.hparams_initial(X)
    for iters in (1, 25):
        got = eng.fit_hparams(I, X, X.mean(axis=0), [p[0] for p in pri], [p[1] for p in pri], init["sigma_sqs"],
                              init["phi1s"], init["phi2s"], init["sigma_sqs"], num_iters=iters, want_trace=True)
        trace = []
        want = orc.fit_kernel_hparams(I, X, num_iters=iters, trace=trace)
        for k in ("phi1s", "phi2s", "sigma_sqs"):
            np.testing.assert_allclose(got[k], want[k], rtol=1e-6)
        np.testing.assert_allclose(got["loss"], trace, rtol=1e-8)
    eng.close()


def test_fit_increases_the_marginal_likelihood(data):
    from magi_v2_amd.engine import MagiEngine
    I, X, g = data
    eng = MagiEngine(0)
    pri = [orc.fourier_phi2_prior(X[:, d]) for d in range(3)]
    init = orc.hparams_initial(X)
    out = eng.fit_hparams(I, X, X.mean(axis=0), [p[0] for p in pri], [p[1] for p in pri], init["sigma_sqs"],
                          init["phi1s"], init["phi2s"], init["sigma_sqs"], num_iters=300, want_trace=True)
    assert out["loss"][-1] < out["loss"][0] - 10.0
    assert (out["phi1s"] > 0).all() and (out["phi2s"] > 0).all() and (out["sigma_sqs"] > 0).all()
    # (the fit runs on the linearly interpolated grid, as the reference does, so sigma^2 is NOT expected to
    #  approach the true observation noise: half of the points are exact interpolants)
    eng.close()


def _vignette_run(fit_kw, theta_init, stale_cache=True, burnin=1000, results=1000, chains=4, seed=123):
    """vignette.ipynb cells 5-11 through the drop-in API at the reference's own length (cell 8: 1000 + 1000 NUTS steps,
    bandsize 80, discretization 1) on the vignette rows; returns (model, theta mean [3], per-chain means [chains, 3])."""
    import magi_v2
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=g["seir3_ts_obs"], X_obs=g["seir3_X_obs"], bandsize=80, f_vec="seir3")
    model.initial_fit(discretization=1, **fit_kw)
    if theta_init is not None:
        model.thetas_init = np.asarray(theta_init, dtype=np.float64)
    res = model.predict(num_results=results, num_burnin_steps=burnin, n_chains=chains, seed=seed, stale_cache=stale_cache)
    th = res["thetas_samps"].reshape(chains, results, 3)
    assert np.isfinite(res["X_samps"]).all() and np.isfinite(th).all()
    kr = res["kernel_results"]
    assert np.asarray(kr["has_divergence"]).mean() < 0.05
    return model, th.reshape(-1, 3).mean(axis=0), th.mean(axis=1)


# Recovered parameters at the reference's length, 4 chains x (1000 + 1000), seed 123 (profiles/r02_recovery_full_length.json;
# MC standard errors from batch means).  The reference's stored notebook output for this data is (5.831, 0.565, 1.77) for the
# truth (6, 0.6, 1.8) (vignette.ipynb cell 11) -- from an unseeded run whose hyper-parameters are not recorded anywhere.
REFERENCE_PRINTED = np.array([5.831, 0.565, 1.77])


def test_recovery_at_reference_length_phi2_half_true_noise():
    """phi2 = 0.5 and the true noise level, theta_init = 1: measured (5.805 +- 0.031, 0.554 +- 0.002, 1.749 +- 0.005) -- the
    hot path (build, log posterior, NUTS, annealing) lands where the reference's notebook did."""
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    true_sd = 0.05 * np.ptp(g["rows"][:, 6:9], axis=0)
    model, th, per_chain = _vignette_run(dict(hparams={"phi2s": [0.5, 0.5, 0.5], "sigma_sqs": true_sd ** 2}, theta_init_iters=0), np.ones(3))
    print("theta", th, "per chain", per_chain)
    np.testing.assert_allclose(th, [5.805, 0.554, 1.749], atol=0, rtol=0.03)            # the recorded run (>= 5 MC standard errors)
    assert np.all(np.abs(th - REFERENCE_PRINTED) < [0.2, 0.025, 0.06]), th              # and the reference's printed values
    model.engine.close()


def test_recovery_at_reference_length_starting_hyperparameters():
    """hparam_iters = 0 (the reference's STARTING hyper-parameters: phi2 = (0.375, 0.23, 0.109), sigma^2 = (0.1 std)^2),
    theta_init = 1: measured (4.167 +- 0.057, 0.235 +- 0.016, 1.192 +- 0.022) with the reference's stale-cache quirk and
    (4.221, 0.256, 1.217) without -- NOT the notebook's values: recovery depends on the hyper-parameters, not on the sampler."""
    model, th, per_chain = _vignette_run(dict(hparam_iters=0, theta_init_iters=0), np.ones(3))
    print("theta", th, "per chain", per_chain)
    assert np.all(np.abs(th - [4.167, 0.235, 1.192]) < [0.35, 0.09, 0.13]), th           # ~6 MC standard errors of the recorded run
    assert np.all(np.abs(per_chain - th) < [0.5, 0.15, 0.2])                             # the four chains agree with each other
    model.engine.close()


def test_recovery_on_config2s_own_grid_and_the_stale_cache_finding():
    """BASELINE config 2's grid (synthetic SEIR-4, N = 1024, observations at the even indices, 5 % noise, dense matrices) through the
    drop-in API.  PARITY UNPINNED: the reference holds no known answer for this grid (its one stored theta-hat is the N = 161 vignette).
    (a) sensible hyper-parameters as DESIGN section 8 row (iv) -- phi2 = 0.5, noise at its true level, theta_init = 1, cache recomputed at
        the current temperature: the truth (6, 0.6, 1.8) is recovered.  Recorded at 4 x (1000 + 1000), profiles/r04_recovery_n1024.json:
        (6.081 +- 0.025, 0.6012 +- 0.0007, 1.7649 +- 0.0015), posterior sd (0.29, 0.018, 0.045); here 4 x (600 + 400).
    (b) [recorded only, same file] the reference's STARTING hyper-parameters give (1.14, 0.04, 0.28): what bench.py's chains sample is the
        posterior of short length scales phi2 ~ 0.06-0.12, not a stuck chain -- recovery is a property of the hyper-parameters.
    (c) the reference's stale cache (magi_v2.py:855-879, the API default) on this grid: the log posterior is POSITIVE at the start (6642), so
        the cached target of the previous, hotter temperature offsets every energy difference by (beta_k - beta_{k-1}) L << 0 and EVERY
        proposal is rejected -- the chain never leaves theta = (1, 1, 1) and dual averaging drives the step size to zero.  This is why
        bench.py runs stale_cache = 0; asserted here instead of being prose."""
    import magi_v2
    from magi_v2_amd import host
    I, X_obs, truth, th_true = host.synthetic_seir(1024, seed=0)
    true_sd = 0.05 * (truth.max(axis=0) - truth.min(axis=0))
    m = magi_v2.MAGI_v2(3, I, X_obs, None, "seir4")
    m.initial_fit(0, theta_init_iters=0, hparams={"phi2s": [0.5] * 4, "sigma_sqs": true_sd ** 2})
    m.thetas_init = np.ones(3)
    res = m.predict(400, 600, n_chains=4, seed=123, stale_cache=False)
    th = res["thetas_samps"].reshape(-1, 3).mean(axis=0)
    per_chain = res["thetas_samps"].reshape(4, 400, 3).mean(axis=1)
    print("config-2 grid, sensible hyper-parameters: theta", th, "per chain", per_chain)
    assert np.all(np.abs(th - th_true) < [0.30, 0.02, 0.10]), th                       # ~1 posterior sd of each entry around the truth
    assert np.all(np.abs(per_chain - th) < [0.30, 0.02, 0.06])                          # the four chains agree
    assert np.asarray(res["kernel_results"]["is_accepted"]).mean() > 0.9
    Xm = res["X_samps"].reshape(-1, 1024, 4).mean(axis=0)
    assert np.all(np.sqrt(((Xm - truth) ** 2).mean(axis=0)) < 0.6 * true_sd)           # the inferred trajectory is closer to the truth than one noisy observation
    # (c) the reference-faithful sampler mode on the same problem at the reference's starting hyper-parameters
    m.initial_fit(0, theta_init_iters=0, hparam_iters=0)
    m.thetas_init = np.ones(3)
    res = m.predict(30, 30, n_chains=1, seed=123)                                       # stale_cache = True: the API default = the reference
    kr = res["kernel_results"]
    assert np.asarray(kr["is_accepted"]).sum() == 0 and np.all(res["thetas_samps"] == res["thetas_samps"][0])
    assert np.asarray(kr["target_log_prob"])[0] > 0.0 and np.asarray(kr["step_size"])[-1] < 0.05
    np.testing.assert_allclose(res["thetas_samps"][0], np.ones(3), rtol=1e-12)
    res0 = m.predict(30, 30, n_chains=1, seed=123, stale_cache=False)                   # recomputed cache: the same chain moves
    assert np.asarray(res0["kernel_results"]["is_accepted"]).mean() > 0.5
    m.engine.close()


@pytest.mark.xfail(strict=True, reason="reference-default path (hyper-parameters fitted on the interpolated grid + the reference's theta "
                   "initialiser): measured theta = (0.24, 0.02, 0.11) at 4 x (1000 + 1000), against the notebook's (5.831, 0.565, 1.77). "
                   "The restated fit finds phi2 = (0.79, 0.16, 0.09) with near-zero noise (its objective there is 3692 against 3329 at "
                   "phi2 = 0.5 + true noise) and the initialiser's reshape (magi_v2.py:155-156) returns negative theta, which the -5.0 "
                   "fallback (:379-380) turns into theta ~ 0.007.  TFP / tf_keras are not installable: f1 / f2 / sampler parity unpinned.")
def test_default_path_recovers_the_notebook_values():
    model, th, per_chain = _vignette_run(dict(), None)
    print("theta", th, "phi2", model.phi2s, "thetas_init", model.thetas_init)
    assert np.all(np.abs(th - REFERENCE_PRINTED) < [0.6, 0.06, 0.2]), th


def test_vignette_end_to_end_with_fitted_hyperparameters():
    """vignette.ipynb cells 5-8 through the drop-in API (bandsize 80, discretization 1): the reference-faithful defaults (fit on
    the interpolated grid, the reference's theta initialiser) run and stay finite, and the documented deviation
    ``hparam_fit_on="observed"`` lands on the right noise level."""
    import magi_v2
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    model = magi_v2.MAGI_v2(D_thetas=3, ts_obs=g["seir3_ts_obs"], X_obs=g["seir3_X_obs"], bandsize=80, f_vec="seir3")
    model.initial_fit(discretization=1)                                   # reference-faithful path
    assert np.all(model.phi2s > 0.01) and np.all(model.phi2s < 5.0) and np.isfinite(model.thetas_init).all()
    res = model.predict(num_results=20, num_burnin_steps=20, seed=1)
    assert np.isfinite(res["X_samps"]).all() and np.isfinite(res["thetas_samps"]).all()

    model.initial_fit(discretization=1, hparam_fit_on="observed", hparam_iters=300)      # documented deviation: runs, sane noise level
    true_sd = 0.05 * np.ptp(g["rows"][:, 6:9], axis=0)
    assert np.all(np.abs(np.sqrt(model.sigma_sqs_init) / true_sd - 1.0) < 1.0)          # right order of magnitude

    # (parameter recovery at the reference's length: the three tests above)
    model.engine.close()
