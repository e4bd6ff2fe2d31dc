"""GPU parity of the sampler's single-phase ("fused") gradient formulation against the golden
vectors and the oracle.  It computes t1 + t2 as xc^T FH xc - 2 f^T FE xc + f^T FK f, i.e. as a
difference of larger terms, so its tolerance is 1e-9 (value, relative) -- still an order below the
north star's 1e-8 bar -- where the three-phase reference-order path is held to 1e-10."""
import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["seir3_N161", "seir4_N81", "sirw_N41"])
def test_fused_logpost_matches_golden(tag):
    g = load_g4(tag)
    pr_dense = problem_from_g4(g, None)
    engines = {}
    worst = 0.0
    for r in range(len(g["rec_logp"])):
        b = int(g["rec_band"][r])
        if b not in engines:
            engines[b] = engine_for(pr_dense, None if b < 0 else b)
        si = int(g["rec_state"][r])
        X, sp, tp = g["state_X"][si], g["state_sig_pre"][si], g["state_th_pre"][si]
        temp = float(g["rec_temp"][r])
        lp, gX, gs, gt, terms = engines[b].logpost_grad(X, sp, tp, temp, want_terms=True, fused=True)
        ref = g["rec_logp"][r]
        worst = max(worst, abs(lp - ref) / abs(ref))
        assert abs(lp - ref) <= 1e-9 * abs(ref), (tag, r, lp, ref)
        t = g["rec_terms"][r]
        assert abs(terms[0] - (t[0] + t[1])) <= 1e-9 * abs(t[0] + t[1])
        for got, want in ((gX, g["rec_gX"][r]), (gs, g["rec_gsig"][r]), (gt, g["rec_gth"][r])):
            assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    for e in engines.values():
        e.close()
    print(tag, "worst relative logp error", worst)


@pytest.mark.parametrize("band,n_chains", [(3, 1), (10, 5), (None, 9)])
def test_fused_banded_and_batched(band, n_chains, stream_family):
    """band 3 / 10 at N=161 select the banded fused stacks (width 6b+1); 5 and 9 chains exercise
    the NC=4 / NC=8 variants with a ragged last group."""
    g = load_g4("seir3_N161")
    pr_dense = problem_from_g4(g, None)
    pr = problem_from_g4(g, band)
    eng = engine_for(pr_dense, band)
    rng = np.random.default_rng(11)
    X = g["state_X"][1][None] + 0.01 * rng.standard_normal((n_chains,) + g["state_X"][1].shape)
    sp = rng.normal(-4, 1, (n_chains, pr.D))
    tp = np.log(np.expm1(g["theta_true"]))[None] + 0.1 * rng.standard_normal((n_chains, pr.P))
    lp, gX, gs, gt = eng.logpost_grad(X, sp, tp, 0.9, fused=True)
    for c in range(n_chains):
        l0, gx0, gs0, gt0 = orc.logpost_grad(X[c], sp[c], tp[c], 0.9, pr)
        assert abs(lp[c] - l0) <= 1e-9 * abs(l0)
        assert np.abs(gX[c] - gx0).max() <= 1e-9 * np.abs(gx0).max()
        assert np.abs(gt[c] - gt0).max() <= 1e-9 * np.abs(gt0).max()
        assert np.abs(gs[c] - gs0).max() <= 1e-9 * np.abs(gs0).max()
    eng.close()
