"""CPU tests of the product's host logic (magi_v2_amd/host.py, api bookkeeping, sharding) against
the reference-generated golden vectors.  No GPU, no oracle import needed for the product side."""
import os

import numpy as np
import pytest

from magi_v2_amd import host
from oracle import magi_oracle as orc
from tests.util import GOLDEN


def test_host_helpers_match_reference_golden():
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    for name in ("seir3", "seir4"):
        I, Xd = host.discretize(g[f"{name}_ts_obs"], g[f"{name}_X_obs"], 1)
        np.testing.assert_array_equal(I, g[f"{name}_I"])
        np.testing.assert_array_equal(Xd, g[f"{name}_X_obs_discret"])
        Xi = host.linear_interpolate(Xd)
        np.testing.assert_array_equal(Xi, g[f"{name}_X_interp"])
        np.testing.assert_allclose(host.cubic_smoother(I, Xi), g[f"{name}_Xhat_smoothed"], rtol=0, atol=1e-14)
        I2, Xd2 = host.discretize(g[f"{name}_ts_obs"], g[f"{name}_partial_X_obs"], 2)
        np.testing.assert_array_equal(I2, g[f"{name}_partial_I"])
        np.testing.assert_array_equal(host.linear_interpolate(Xd2), g[f"{name}_partial_X_interp"])


def test_discretize_rejects_length_mismatch():
    with pytest.raises(AssertionError):
        host.discretize(np.arange(4.0), np.zeros((5, 2)), 1)


def test_short_series_skip_smoothing():
    X = np.arange(18.0).reshape(9, 2)
    assert host.cubic_smoother(np.arange(9.0), X) is X            # magi_v2.py:699-700


def test_hparams_and_boundary_transforms_match_oracle():
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    Xi = g["seir4_X_interp"]
    a, b = host.hparams_initial(Xi), orc.hparams_initial(Xi)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    Xs = g["seir4_Xhat_smoothed"]
    LB = host.sigma_sqs_lower_bound(Xs)
    np.testing.assert_array_equal(LB, orc.sigma_sqs_lower_bound(Xs))
    sig = np.array([1e-3, LB[1] * 0.5, 2e-3, 5e-4])
    th = np.array([1.0, 0.0, -2.0])
    sp, tp = host.softplus_inverse_inits(sig, th, LB)
    _, sp0, tp0 = orc.initial_state(Xs, sig, th, LB)
    np.testing.assert_array_equal(sp, sp0)
    np.testing.assert_array_equal(tp, tp0)
    assert sp[1] == -5.0 and tp[1] == -5.0 and tp[2] == -5.0
    s2, t2 = host.transform_samples(sp[None], tp[None], LB)
    o2, p2 = orc.transform_samples(sp[None], tp[None], LB)
    np.testing.assert_array_equal(s2, o2)
    np.testing.assert_array_equal(t2, p2)


def test_observation_bookkeeping_vignette():
    g = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))
    N_ds, beta, idx, y = host.observation_bookkeeping(g["seir3_X_obs"], g["seir3_X_obs_discret"])
    assert list(N_ds) == [81, 81, 81]
    assert beta == pytest.approx(3 * 161 / 243, abs=1e-15)
    assert len(idx) == 243 and np.all(np.diff(idx) > 0)
    np.testing.assert_array_equal(y, g["seir3_X_obs"].reshape(-1))


def test_numpy_drifts_match_oracle_drifts():
    rng = np.random.default_rng(0)
    for name, (fn, D, P) in orc.DRIFTS.items():
        X = rng.uniform(0, 1, (9, D))
        th = rng.uniform(0.1, 3, P)
        np.testing.assert_allclose(host.NUMPY_DRIFTS[name](None, X, th), fn(X, th)[0], rtol=1e-14, atol=1e-16)


def test_resolve_drift_accepts_reference_style_callables():
    def vignette_f_vec(t, X, thetas):          # vignette.ipynb cell 3 with tf.* spelled in numpy
        S = 1.0 - np.reshape(np.sum(X, axis=1), (-1, 1))
        return np.concatenate([(thetas[0] * S * X[:, 1:2]) - (thetas[2] * X[:, 0:1]),
                               (thetas[2] * X[:, 0:1]) - (thetas[1] * X[:, 1:2]),
                               (thetas[1] * X[:, 1:2])], axis=1)

    assert host.resolve_drift(vignette_f_vec, 3, 3) == "seir3"
    assert host.resolve_drift("sirw", 4, 5) == "sirw"
    with pytest.raises(ValueError):
        host.resolve_drift("sirw", 4, 3)          # the reference's own script passes D_thetas=3 for 5 thetas
    assert host.resolve_drift(lambda t, X, th: X * th[0], 3, 3).startswith("user_")     # traced (tests/test_drift_cpu.py)
    with pytest.raises(NotImplementedError):
        host.resolve_drift(lambda t, X, th: X.no_such_numpy_method(), 3, 3)


def test_synthetic_generator_reproduces_reference_truth_columns():
    """The *_true columns of data/SEIR_seed=0.csv (thinned rows are in the fixture) are the RK4
    solution the synthetic-grid generator integrates (consistent to 3e-7)."""
    rows = np.load(os.path.join(GOLDEN, "g3_pipeline.npz"))["rows"]          # t = 0, .05, ..., 4
    I, X_obs, truth, th = host.synthetic_seir(161, seed=0)
    np.testing.assert_allclose(I[::2], rows[:, 0], atol=1e-12)
    np.testing.assert_allclose(truth[::2], rows[:, 5:9], atol=5e-7)
    assert np.isnan(X_obs[1::2]).all() and not np.isnan(X_obs[::2]).any()
    assert (X_obs[::2] >= 0).all()
    I2, X2, _, _ = host.synthetic_seir(161, seed=0)
    np.testing.assert_array_equal(X_obs, X2)                                # seeded, reproducible


def test_api_constructor_bookkeeping_without_gpu():
    from magi_v2_amd import MAGI_v2
    X = np.random.default_rng(1).uniform(0, 1, (12, 4))
    X[::3, 2] = np.nan
    m = MAGI_v2(D_thetas=3, ts_obs=np.arange(12.0), X_obs=X, bandsize=5, f_vec="seir4")
    assert (m.N, m.D, m.BANDSIZE) == (12, 4, 5)
    assert list(m.N_ds) == [12, 12, 8, 12]
    assert list(m.observed_components) == [0, 1, 2, 3] and m.D_unobserved == 0
    assert np.isnan(m.phi1s).all() and m.C_d_invs is None
    Xu = X.copy(); Xu[:, 1] = np.nan
    m2 = MAGI_v2(3, np.arange(12.0), Xu, None, "seir4")
    assert list(m2.unobserved_components) == [1] and list(m2.proper_order) == [0, 3, 1, 2]
    import magi_v2                                               # reference module name
    assert magi_v2.MAGI_v2 is MAGI_v2
    assert magi_v2.logarithmic_temperature_schedule(0) == pytest.approx(1 / np.log(2))
    assert magi_v2.logarithmic_temperature_schedule(10 ** 7) == 0.1


def test_chain_sharding_partitions():
    from magi_v2_amd.shard import chain_ids_for_rank, shard_units
    for world in (1, 2, 3, 8):
        got = sum((chain_ids_for_rank(r, world, 64) for r in range(world)), [])
        assert got == list(range(64))
    assert chain_ids_for_rank(7, 8, 64) == list(range(56, 64))
    assert [len(chain_ids_for_rank(r, 3, 10)) for r in range(3)] == [4, 3, 3]
    units = [shard_units(10, 8, r, 8) for r in range(8)]
    assert sorted(ds for u in units for ds, _ in u) == list(range(10))
    assert all(ids == [ds * 8 + c for c in range(8)] for u in units for ds, ids in u)
    assert [len(u) for u in units] == [2, 2, 1, 1, 1, 1, 1, 1]


def test_family_chains_is_the_largest_per_rank_share():
    """shard.family_chains_for: what every rank of a sharded job hands to its handle (option "family_chains") so that one streaming-kernel
    family serves all ranks -- the largest share of chain_ids_for_rank's block partition."""
    from magi_v2_amd.shard import chain_ids_for_rank, family_chains_for
    for total, world in ((64, 8), (20, 8), (3, 8), (8, 1), (17, 4), (1, 1)):
        assert family_chains_for(total, world) == max(len(chain_ids_for_rank(r, world, total)) for r in range(world))
