"""SURVEY 8 row f2 on the GPU: the theta initialiser of initial_fit (magi_v2.py:133-179) as a device-resident Adam loop
(csrc/thetainit.hip: one captured graph per step, the host waits once) against the oracle's restatement -- the same objective with the
reference's reshape (:155-156), differentiated there with complex-step Jacobians of the same callable, the same tf_keras-default Adam.
PARITY UNPINNED against tf_keras itself (not installable); the oracle's objective is pinned to a torch transcription in
tests/test_theta_init_cpu.py."""
import time

import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4

pytestmark = pytest.mark.gpu


def test_device_theta_initialiser_equals_oracle_on_the_vignette_matrices():
    g = load_g4("seir3_N161")
    pr = problem_from_g4(g, None)
    Xhat, mu = g["Xhat_init"], g["mu"]
    eng = engine_for(pr, None)
    for iters in (1, 200):
        got, losses = eng.theta_init("seir3", Xhat, mu, iters, want_trace=True)
        want, olosses = orc.fit_thetas_init(Xhat, mu, g["m"], g["K_inv"], "seir3", 3, num_iters=iters)
        np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(losses, olosses[:iters], rtol=1e-9)
    assert losses[-1] < losses[0]
    # the band approximation comes AFTER the initialiser (magi_v2.py:271-274): a banded packing does not change it
    eng_b = engine_for(pr, 20)
    np.testing.assert_array_equal(eng_b.theta_init("seir3", Xhat, mu, 50), eng.theta_init("seir3", Xhat, mu, 50))
    eng.close(); eng_b.close()


@pytest.mark.parametrize("name", ["ptrans", "fhn"])
def test_device_theta_initialiser_equals_oracle_for_traced_drifts(name):
    """The general (not linear-in-theta) branch for a traced f_vec -- the five-component protein-transduction system with its
    V x / (K + x) term, and FitzHugh-Nagumo (c and 1 / c) -- in the library compiled for that drift."""
    from tests.test_user_drift_gpu import make_problem
    eng, pr, Xhat, hp, truth = make_problem(name)
    P = len(truth)
    got, losses = eng.theta_init(eng.user_drift, Xhat, pr.mu, 200, want_trace=True)
    want, olosses = orc.fit_thetas_init(Xhat, pr.mu, pr.m, pr.K_inv, name, P, num_iters=200)
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(losses, olosses[:200], rtol=1e-8)
    assert np.abs(want - 1.0).max() > 0.2 and losses[-1] < losses[0]
    t0 = time.perf_counter()
    full = eng.theta_init(eng.user_drift, Xhat, pr.mu, 10000)          # the reference's length (magi_v2.py:161-176)
    wall = time.perf_counter() - t0
    print(f"theta initialiser, {name}, 10 000 Adam steps on the device: {wall:.3f} s")
    assert np.isfinite(full).all() and wall < 20.0
    eng.close()


def test_initial_fit_of_a_traced_drift_runs_its_theta_initialiser_on_the_device():
    """Through the drop-in class: MAGI_v2(f_vec = FitzHugh-Nagumo).initial_fit takes the general branch, which stays on the GPU while the
    matrices are device-resident, and lands where the oracle's loop on the downloaded matrices lands."""
    import magi_v2
    from magi_v2_amd.drift_examples import fitzhugh_nagumo, rk4
    from tests.test_user_drift_gpu import oracle_drift
    I, X = rk4(fitzhugh_nagumo, [-1.0, 1.0], np.array([0.2, 0.2, 3.0]), 20.0, 41)
    Xo = X + np.random.default_rng(1).normal(0, 0.05, X.shape)
    m = magi_v2.MAGI_v2(3, I, Xo, None, fitzhugh_nagumo)
    m.initial_fit(0, hparams={"phi1s": [1.0, 0.5], "phi2s": [1.5, 1.5]}, theta_init_iters=150)
    orc.DRIFTS["fhn_api"] = (oracle_drift(fitzhugh_nagumo), 2, 3)
    try:
        Xi = m.X_interp_obs                                               # what the initialiser saw: the interpolated grid BEFORE smoothing (:112-113, 277)
        want, _ = orc.fit_thetas_init(Xi, m.mu_ds, np.asarray(m.m_ds), np.asarray(m.K_d_invs), "fhn_api", 3, num_iters=150)
    finally:
        del orc.DRIFTS["fhn_api"]
    np.testing.assert_allclose(m.thetas_init, want, rtol=1e-7, atol=1e-9)
    m.engine.close()
