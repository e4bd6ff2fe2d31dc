"""GPU parity: HIP log-posterior + gradient (through the C ABI) vs the golden vectors (torch
autograd transcription of magi_v2.py:308-348) and vs the numpy oracle, on identical matrices.
Tolerance: 1e-10 relative on the value, 1e-10 of the gradient scale (north-star bar: 1e-8)."""
import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def _check(got, want, scale=None):
    want = np.asarray(want)
    s = np.abs(want).max() if scale is None else scale
    assert np.abs(np.asarray(got) - want).max() <= RTOL * max(s, 1e-300)


@pytest.mark.parametrize("tag", ["seir3_N161", "seir4_N81", "sirw_N41"])
def test_logpost_matches_golden(tag):
    g = load_g4(tag)
    pr_dense = problem_from_g4(g, None)
    engines = {}
    for r in range(len(g["rec_logp"])):
        b = int(g["rec_band"][r])
        band = None if b < 0 else b
        if b not in engines:
            engines[b] = engine_for(pr_dense, band)
        eng = engines[b]
        si = int(g["rec_state"][r])
        X, sp, tp = g["state_X"][si], g["state_sig_pre"][si], g["state_th_pre"][si]
        temp = float(g["rec_temp"][r])
        lp, gX, gs, gt, terms = eng.logpost_grad(X, sp, tp, temp, want_terms=True)
        assert abs(lp - g["rec_logp"][r]) <= RTOL * abs(g["rec_logp"][r]), (tag, r, lp, g["rec_logp"][r])
        np.testing.assert_allclose(terms, g["rec_terms"][r], rtol=RTOL)
        _check(gX, g["rec_gX"][r])
        _check(gs, g["rec_gsig"][r])
        _check(gt, g["rec_gth"][r])
    for e in engines.values():
        e.close()


@pytest.mark.parametrize("n_chains", [2, 5, 9])
def test_logpost_batched_chains_match_oracle(n_chains):
    g = load_g4("seir4_N81")
    pr = problem_from_g4(g, None)
    eng = engine_for(pr, None)
    rng = np.random.default_rng(7)
    X = g["state_X"][1][None] + 0.02 * rng.standard_normal((n_chains,) + g["state_X"][1].shape)
    sp = rng.normal(-4, 1, (n_chains, pr.D))
    tp = rng.normal(0.5, 0.5, (n_chains, pr.P))
    lp, gX, gs, gt = eng.logpost_grad(X, sp, tp, 0.7)
    for c in range(n_chains):
        l0, gx0, gs0, gt0 = orc.logpost_grad(X[c], sp[c], tp[c], 0.7, pr)
        assert abs(lp[c] - l0) <= RTOL * abs(l0)
        _check(gX[c], gx0)
        _check(gs[c], gs0)
        _check(gt[c], gt0)
    eng.close()


@pytest.mark.parametrize("band", [3, 20, 40])
def test_banded_storage_matches_oracle_mask(band):
    """2b+1 < N selects true banded storage N x (2b+1); results must equal the reference's
    dense-with-zeros band_part semantics (magi_v2.py:271-274)."""
    g = load_g4("seir3_N161")
    pr_dense = problem_from_g4(g, None)
    pr_band = problem_from_g4(g, band)
    eng = engine_for(pr_dense, band)
    for si in range(3):
        X, sp, tp = g["state_X"][si], g["state_sig_pre"][si], g["state_th_pre"][si]
        lp, gX, gs, gt = eng.logpost_grad(X, sp, tp, 1.0)
        l0, gx0, gs0, gt0 = orc.logpost_grad(X, sp, tp, 1.0, pr_band)
        assert abs(lp - l0) <= RTOL * abs(l0)
        _check(gX, gx0)
        _check(gs, gs0)
        _check(gt, gt0)
    eng.close()


def test_nonsymmetric_user_matrices():
    """The reference's pinv outputs are not exactly symmetric and users may overwrite the matrices
    (magi_v2.py:77-80): the engine must reproduce x^T A x and (A + A^T) x for a general A."""
    g = load_g4("sirw_N41")
    pr = problem_from_g4(g, None)
    rng = np.random.default_rng(3)
    pr.C_inv = pr.C_inv + 1e-3 * np.abs(pr.C_inv).max() * rng.standard_normal(pr.C_inv.shape)
    pr.K_inv = pr.K_inv + 1e-3 * np.abs(pr.K_inv).max() * rng.standard_normal(pr.K_inv.shape)
    eng = engine_for(pr, None)
    X, sp, tp = g["state_X"][2], g["state_sig_pre"][2], g["state_th_pre"][2]
    lp, gX, gs, gt = eng.logpost_grad(X, sp, tp, 1.3)
    l0, gx0, gs0, gt0 = orc.logpost_grad(X, sp, tp, 1.3, pr)
    assert abs(lp - l0) <= RTOL * abs(l0)
    _check(gX, gx0)
    _check(gt, gt0)
    eng.close()


def test_error_paths():
    from magi_v2_amd.engine import MagiEngine, MagiHipError
    eng = MagiEngine(0)
    with pytest.raises(MagiHipError):          # problem before matrices
        eng.D = 3
        eng.set_problem(np.zeros(3), np.ones(3), np.zeros(1, dtype=np.int64), np.zeros(1), 1.0, np.zeros(3), "seir3")
    g = load_g4("sirw_N41")
    pr = problem_from_g4(g, None)
    eng.set_matrices(pr.C_inv, pr.m, pr.K_inv)
    with pytest.raises(MagiHipError):          # drift / shape mismatch (SEIR3 needs D=3)
        eng.set_problem(pr.mu, pr.N_ds, pr.obs_idx, pr.y, pr.beta, pr.LB, "seir3")
    eng.set_problem(pr.mu, pr.N_ds, pr.obs_idx, pr.y, pr.beta, pr.LB, "sirw")
    X = g["state_X"][0].copy()
    X[3, 1] = np.nan
    cfg = eng.default_cfg(num_results=2, num_burnin_steps=2)
    with pytest.raises(MagiHipError) as ei:    # reference asserts no NaN in the inits (magi_v2.py:289-291)
        eng.sampler_init(cfg, X, g["state_sig_pre"][0], g["state_th_pre"][0], seed=1)
    assert ei.value.code == -4
    eng.close()
