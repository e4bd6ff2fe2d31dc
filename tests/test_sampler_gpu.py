"""GPU parity of the device-resident NUTS state machine against the oracle's restatement of
TFP's NUTS + dual averaging + LogAnnealedNUTS (magi_v2.py:357-396, 833-889) with the same
Philox streams.  Integer diagnostics (tree depth, leapfrog counts, flags) must agree exactly;
states agree to 1e-8 relative over the first transitions (the two sides differ only in fp64
summation order and libm last-bit rounding, which the leapfrog dynamics amplify slowly).
Sampler parity against TFP itself is UNPINNED (see oracle/magi_oracle.py)."""
import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import engine_for, load_g4, problem_from_g4

pytestmark = pytest.mark.gpu


def _run_both(tag, band, burnin, results, seed, chain_ids=(0,), stale=1, anneal=1, theta0=None):
    g = load_g4(tag)
    pr = problem_from_g4(g, band)
    pr_dense = problem_from_g4(g, None)
    eng = engine_for(pr_dense, band)
    P = pr.P
    theta0 = np.ones(P) if theta0 is None else theta0
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], theta0, pr.LB)
    n = len(chain_ids)
    cfg = eng.default_cfg(num_results=results, num_burnin_steps=burnin, stale_cache=stale, anneal=anneal)
    eng.sampler_init(cfg, np.repeat(X0[None], n, 0), np.repeat(s0[None], n, 0), np.repeat(t0[None], n, 0), seed=seed,
                     chain_ids=list(chain_ids))
    lf, ms = eng.sampler_run(burnin + results)
    Xs, sp, tp = eng.sampler_samples()
    diag = eng.sampler_diag()
    oracle = []
    for cid in chain_ids:
        trace = []
        out = orc.sample_chain(pr, g["Xhat_init"], g["sigma_sqs_init"], theta0, results, burnin, seed=seed, chain=cid,
                               stale_cache=bool(stale), anneal=bool(anneal), trace=trace)
        oracle.append((out, trace))
    eng.close()
    return (Xs, sp, tp, diag, lf), oracle


@pytest.mark.parametrize("tag,band,stale", [("seir3_N161", 80, 1), ("seir4_N81", None, 1), ("sirw_N41", 5, 0)])
def test_chain_matches_oracle_draw_for_draw(tag, band, stale):
    burnin, results = 8, 4
    (Xs, sp, tp, diag, lf), oracle = _run_both(tag, band, burnin, results, seed=1234, stale=stale)
    (oX, osp, otp, info, da), trace = oracle[0]
    depth = np.array([r.depth for _, r, _ in trace])
    leap = np.array([r.leapfrogs for _, r, _ in trace])
    np.testing.assert_array_equal(diag.tree_depth[0], depth)
    np.testing.assert_array_equal(diag.leapfrogs_taken[0], leap)
    np.testing.assert_array_equal(diag.has_divergence[0], [int(r.has_divergence) for _, r, _ in trace])
    np.testing.assert_array_equal(diag.is_accepted[0], [int(r.is_accepted) for _, r, _ in trace])
    assert lf == leap.sum()
    ss = np.array([s for _, _, s in trace])
    np.testing.assert_allclose(diag.step_size[0], ss, rtol=1e-9)
    lar = np.array([r.log_accept_ratio for _, r, _ in trace])
    fin = np.isfinite(lar)
    np.testing.assert_array_equal(np.isfinite(diag.log_accept_ratio[0]), fin)
    np.testing.assert_allclose(diag.log_accept_ratio[0][fin], lar[fin], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(diag.target_log_prob[0], [r.target_log_prob for _, r, _ in trace], rtol=1e-8)
    np.testing.assert_allclose(diag.beta_temp[0], [orc.temperature(k) for k in range(burnin + results)], rtol=1e-15)
    np.testing.assert_allclose(Xs[0], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
    np.testing.assert_allclose(sp[0], osp, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(tp[0], otp, rtol=1e-7, atol=1e-9)


def test_chains_are_independent_of_batching_and_ids():
    """Philox streams are keyed by the global chain id: a chain gives the same samples whether it
    runs alone, inside a batch, or on another GPU."""
    (Xs, sp, tp, diag, _), oracle = _run_both("sirw_N41", None, 4, 3, seed=99, chain_ids=(5, 2, 11))
    for i in range(3):
        (oX, osp, otp, info, da), trace = oracle[i]
        np.testing.assert_array_equal(diag.leapfrogs_taken[i], [r.leapfrogs for _, r, _ in trace])
        np.testing.assert_allclose(tp[i], otp, rtol=1e-7, atol=1e-9)
    assert not np.allclose(tp[0], tp[1])


def test_pause_resume_equals_single_run():
    g = load_g4("sirw_N41")
    pr = problem_from_g4(g, None)
    outs = []
    for chunks in ([10], [3, 1, 6]):
        eng = engine_for(pr, None)
        X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
        cfg = eng.default_cfg(num_results=4, num_burnin_steps=6)
        eng.sampler_init(cfg, X0, s0, t0, seed=5)
        for c in chunks:
            eng.sampler_run(c)
        assert list(eng.sampler_steps_done()) == [10]
        outs.append(eng.sampler_samples())
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_sampler_health_on_the_vignette_configuration():
    """Vignette configuration (SEIR-3, N=161, b=80; vignette.ipynb cells 5-8) on the fixture's matrices, four chains from
    theta = (1, 1, 1), 150 + 150 steps: the chains stay finite, accept at the adaptation target, do not diverge and keep X near
    the data.  WHERE theta lands is asserted at the reference's own length (1000 + 1000) in tests/test_fit_gpu.py -- at this
    length beta_temp is still ~0.2 and the mean is not meaningful; exact sampler behaviour is the draw-for-draw tests above."""
    g = load_g4("seir3_N161")
    pr = problem_from_g4(g, None)
    eng = engine_for(pr, 80)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(3), pr.LB)
    cfg = eng.default_cfg(num_results=150, num_burnin_steps=150)
    eng.sampler_init(cfg, np.repeat(X0[None], 4, 0), np.repeat(s0[None], 4, 0), np.repeat(t0[None], 4, 0), seed=2024)
    eng.sampler_run(300)
    Xs, sp, tp = eng.sampler_samples()
    d = eng.sampler_diag()
    _, th = orc.transform_samples(sp, tp, pr.LB)
    eng.close()
    assert np.isfinite(Xs).all() and np.isfinite(th).all()
    assert d.has_divergence[:, 150:].mean() < 0.2
    assert 0.6 < np.exp(np.minimum(d.log_accept_ratio[:, 150:], 0)).mean() <= 0.9          # dual averaging targets 0.75
    assert np.abs(Xs.mean(axis=(0, 1)) - g["Xhat_init"]).max() < 0.1
    assert th.reshape(-1, 3).mean(axis=0)[0] > 1.5                                           # beta has left theta_init = 1 upwards


@pytest.mark.parametrize("L", [1, 7, 32])
def test_fixed_length_hmc_matches_oracle(L):
    """MAGI_MODE_HMC (fixed-L HMC next to NUTS): L leapfrogs forward + Metropolis, same Philox
    streams, dual averaging on min(1, exp(dH)); compared with the oracle's hmc_one_step draw for draw."""
    g = load_g4("seir4_N81")
    pr = problem_from_g4(g, None)
    eng = engine_for(pr, None)
    theta0 = np.ones(pr.P)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], theta0, pr.LB)
    burnin, results = 12, 6
    cfg = eng.default_cfg(num_results=results, num_burnin_steps=burnin, mode=1, hmc_leapfrogs=L)
    eng.sampler_init(cfg, X0, s0, t0, seed=77)
    lf, _ = eng.sampler_run(burnin + results)
    assert lf == L * (burnin + results)
    Xs, sp, tp = eng.sampler_samples()
    d = eng.sampler_diag()
    trace = []
    oX, osp, otp, info, _ = orc.sample_chain(pr, g["Xhat_init"], g["sigma_sqs_init"], theta0, results, burnin, seed=77,
                                             trace=trace, hmc_leapfrogs=L)
    np.testing.assert_array_equal(d.is_accepted[0], [int(r.is_accepted) for _, r, _ in trace])
    np.testing.assert_array_equal(d.leapfrogs_taken[0], [L] * (burnin + results))
    np.testing.assert_allclose(d.step_size[0], [s for _, _, s in trace], rtol=1e-8)
    np.testing.assert_allclose(tp[0], otp, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(Xs[0], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
    eng.close()


def test_pause_resume_equals_single_run_on_the_matrix_core_kernel():
    """Five chains (the matrix-core streaming kernel, whose operand mirror is indexed by slot parity): stopping and resuming at
    transition boundaries -- every resume starts a new graph at slot parity 0 -- gives the samples of an uninterrupted run."""
    g = load_g4("sirw_N41")
    pr = problem_from_g4(g, None)
    outs = []
    for chunks in ([9], [2, 1, 4, 2]):
        eng = engine_for(pr, None)
        X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
        cfg = eng.default_cfg(num_results=4, num_burnin_steps=5)
        rep = lambda v: np.repeat(np.asarray(v)[None], 5, axis=0)
        eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=11, chain_ids=[0, 1, 2, 3, 4])
        for c in chunks:
            eng.sampler_run(c)
        assert list(eng.sampler_steps_done()) == [9] * 5
        outs.append(eng.sampler_samples())
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    assert not np.allclose(outs[0][2][0], outs[0][2][1])


def test_fixed_length_hmc_batched_matches_oracle():
    """Fixed-L HMC with four chains in one batch (matrix-core streaming kernel) against the oracle, chain by chain."""
    g = load_g4("seir4_N81")
    pr = problem_from_g4(g, None)
    eng = engine_for(pr, None)
    theta0 = np.ones(pr.P)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], theta0, pr.LB)
    burnin, results, L, ids = 8, 4, 5, [3, 0, 7, 1]
    cfg = eng.default_cfg(num_results=results, num_burnin_steps=burnin, mode=1, hmc_leapfrogs=L)
    rep = lambda v: np.repeat(np.asarray(v)[None], len(ids), axis=0)
    eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=77, chain_ids=ids)
    lf, _ = eng.sampler_run(burnin + results)
    assert lf == len(ids) * L * (burnin + results)
    Xs, sp, tp = eng.sampler_samples()
    d = eng.sampler_diag()
    eng.close()
    for i, cid in enumerate(ids):
        trace = []
        oX, osp, otp, info, _ = orc.sample_chain(pr, g["Xhat_init"], g["sigma_sqs_init"], theta0, results, burnin, seed=77, chain=cid,
                                                 trace=trace, hmc_leapfrogs=L)
        np.testing.assert_array_equal(d.is_accepted[i], [int(r.is_accepted) for _, r, _ in trace])
        np.testing.assert_allclose(d.step_size[i], [s for _, _, s in trace], rtol=1e-8)
        np.testing.assert_allclose(tp[i], otp, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(Xs[i], oX, rtol=0, atol=1e-8 * np.abs(oX).max())


def test_two_chains_match_oracle_draw_for_draw():
    """Exactly two chains per GPU run the two-chain instantiation of the VALU streaming kernel (k_stream<2, *>): both chains
    against the oracle draw for draw, and equal to the same ids run one at a time (k_stream<1, *>) to rounding."""
    (Xs, sp, tp, diag, lf), oracle = _run_both("seir4_N81", None, 6, 3, seed=321, chain_ids=(4, 9), stale=0)
    for i in range(2):
        (oX, osp, otp, info, da), trace = oracle[i]
        np.testing.assert_array_equal(diag.tree_depth[i], [r.depth for _, r, _ in trace])
        np.testing.assert_array_equal(diag.leapfrogs_taken[i], [r.leapfrogs for _, r, _ in trace])
        np.testing.assert_array_equal(diag.is_accepted[i], [int(r.is_accepted) for _, r, _ in trace])
        np.testing.assert_allclose(diag.step_size[i], [s for _, _, s in trace], rtol=1e-9)
        np.testing.assert_allclose(Xs[i], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
        np.testing.assert_allclose(tp[i], otp, rtol=1e-7, atol=1e-9)
    assert lf == diag.leapfrogs_taken.sum() and not np.allclose(tp[0], tp[1])


def test_one_kernel_family_makes_a_chain_independent_of_its_batch_size(monkeypatch):
    """MAGI_STREAM_FAMILY=mc routes every batch size through the matrix-core streaming kernel: chain 7 run alone, in a pair
    and inside a batch of five gives the same samples BIT FOR BIT (by default one or two chains run the VALU kernel, which sums
    in another order: same chain to rounding only -- what an uneven shard such as 5 chains on 2 GPUs = 3 + 2 would meet)."""
    g = load_g4("sirw_N41")
    pr = problem_from_g4(g, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)

    def run(ids):
        eng = engine_for(pr, None)
        cfg = eng.default_cfg(num_results=4, num_burnin_steps=6)
        rep = lambda v: np.repeat(np.asarray(v)[None], len(ids), axis=0)
        eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=31, chain_ids=ids)
        eng.sampler_run(10)
        out = eng.sampler_samples(), eng.sampler_diag().leapfrogs_taken
        eng.close()
        return out

    (_, _, t_def), _ = run([7])                        # default family for one chain (VALU kernel)
    monkeypatch.setenv("MAGI_STREAM_FAMILY", "mc")
    (X1, s1, t1), lf1 = run([7])
    (X2, s2, t2), lf2 = run([3, 7])
    (X5, s5, t5), lf5 = run([0, 7, 1, 2, 3])
    np.testing.assert_array_equal(t1[0], t2[1]); np.testing.assert_array_equal(X1[0], X2[1])
    np.testing.assert_array_equal(t1[0], t5[1]); np.testing.assert_array_equal(X1[0], X5[1])
    np.testing.assert_array_equal(lf1[0], lf5[1])
    np.testing.assert_allclose(t_def[0], t1[0], rtol=1e-6)          # the two families agree to rounding


def test_slot_budget_exit_reports_the_stuck_chain_and_the_handle_stays_usable(monkeypatch):
    """The pump of magi_sampler_run is bounded: a run that outlives its slot budget returns MAGI_E_STATE naming the chain's
    k / phase / depth (csrc/capi.hip).  Forced here with a one-graph budget (128 slots) on transitions that need more; the
    handle is then re-initialised and runs the same chain to completion."""
    from magi_v2_amd.engine import MagiHipError
    g = load_g4("seir4_N81")
    pr = problem_from_g4(g, None)
    eng = engine_for(pr, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
    cfg = eng.default_cfg(num_results=2, num_burnin_steps=6, step_size=1e-3)        # (a small first step: deep trees, > 128 slots)
    eng.sampler_init(cfg, X0, s0, t0, seed=1234)
    eng.set_option("slot_budget_graphs", 1)
    with pytest.raises(MagiHipError) as ei:
        eng.sampler_run(8)
    assert ei.value.code == -5 and "slot budget" in str(ei.value) and "chain 0: k=" in str(ei.value)
    with pytest.raises(MagiHipError):                  # the interrupted sampler refuses to go on
        eng.sampler_run(1)
    eng.set_option("slot_budget_graphs", 0)
    eng.sampler_init(cfg, X0, s0, t0, seed=1234)
    lf, _ = eng.sampler_run(8)
    slots, graphs = eng.sampler_run_stats()
    d = eng.sampler_diag()
    assert lf == d.leapfrogs_taken.sum() and lf > 128 and slots >= lf and graphs * 128 >= slots
    _, _, otp, _, _ = orc.sample_chain(pr, g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), 2, 6, seed=1234, step_size=1e-3)
    np.testing.assert_allclose(eng.sampler_samples()[2][0], otp, rtol=1e-7, atol=1e-9)
    eng.close()


@pytest.mark.parametrize("n_chains", [1, 4])
def test_checkpoint_resume_in_a_new_handle_continues_the_run_bit_for_bit(n_chains):
    """magi_sampler_get_checkpoint / set_checkpoint: 7 transitions, checkpoint, a NEW handle resumed from the checkpoint runs
    the remaining 5 -- its samples and diagnostics of those steps equal the uninterrupted 12-step run bit for bit (one chain:
    VALU kernel; four: matrix-core kernel).  The reference has no resume (a run is all-or-nothing, magi_v2.py:386-425)."""
    g = load_g4("sirw_N41")
    pr = problem_from_g4(g, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
    rep = lambda v: np.repeat(np.asarray(v)[None], n_chains, axis=0)
    ids = list(range(10, 10 + n_chains))
    burnin, results = 8, 4
    eng = engine_for(pr, None)
    cfg = eng.default_cfg(num_results=results, num_burnin_steps=burnin)
    eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=17, chain_ids=ids)
    eng.sampler_run(burnin + results)
    full, dfull = eng.sampler_samples(), eng.sampler_diag()
    eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=17, chain_ids=ids)
    eng.sampler_run(7)
    ck = eng.sampler_checkpoint()
    assert list(ck["scalars"][:, 0]) == [7.0] * n_chains
    eng.close()
    eng2 = engine_for(pr, None)
    eng2.sampler_resume(cfg, ck, seed=17, chain_ids=ids)
    assert list(eng2.sampler_steps_done()) == [7] * n_chains
    eng2.sampler_run(5)
    part, dpart = eng2.sampler_samples(), eng2.sampler_diag()
    eng2.close()
    for a, b in zip(full, part):                       # steps 8..11 are the four kept samples
        np.testing.assert_array_equal(a, b)
    for f in ("leapfrogs_taken", "tree_depth", "step_size", "target_log_prob", "is_accepted"):
        np.testing.assert_array_equal(getattr(dfull, f)[:, 7:], getattr(dpart, f)[:, 7:])


def test_checkpoint_of_another_run_or_with_broken_scalars_is_rejected():
    """A resume promises the uninterrupted run bit for bit, which holds only under the same cfg / seed / chain ids: the scalars carry a tag of
    those and magi_sampler_set_checkpoint refuses anything else (MAGI_E_BADARG), as it refuses NaN / fractional / out-of-range scalars
    instead of casting them."""
    from magi_v2_amd.engine import MagiHipError
    g = load_g4("seir4_N81")
    pr = problem_from_g4(g, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
    eng = engine_for(pr, None)
    cfg = eng.default_cfg(num_results=3, num_burnin_steps=5)
    eng.sampler_init(cfg, X0, s0, t0, seed=17, chain_ids=[4])
    eng.sampler_run(3)
    ck = eng.sampler_checkpoint()
    for kw in (dict(seed=18, chain_ids=[4]), dict(seed=17, chain_ids=[5])):                    # another seed / another Philox stream
        with pytest.raises(MagiHipError) as ei:
            eng.sampler_resume(cfg, ck, **kw)
        assert ei.value.code == -1 and "another sampler configuration" in str(ei.value)
    with pytest.raises(MagiHipError):                                                          # another burn-in
        eng.sampler_resume(eng.default_cfg(num_results=3, num_burnin_steps=6), ck, seed=17, chain_ids=[4])
    for col, bad in ((0, np.nan), (0, 2.5), (0, 99.0), (1, -1.0), (2, 0.0), (2, np.inf), (3, np.nan), (6, 0.0), (6, 1.5), (7, np.nan), (7, 2.5)):
        broken = dict(ck, scalars=ck["scalars"].copy())
        broken["scalars"][0, col] = bad
        with pytest.raises(MagiHipError) as ei:
            eng.sampler_resume(cfg, broken, seed=17, chain_ids=[4])
        assert ei.value.code == -1, (col, bad)
    eng.sampler_resume(cfg, ck, seed=17, chain_ids=[4])                                       # the untouched checkpoint still resumes
    eng.sampler_run(5)
    assert list(eng.sampler_steps_done()) == [8]
    eng.close()


@pytest.mark.parametrize("tag,chains,band", [("sirw_N41", 1, None), ("sirw_N41", 2, None), ("sirw_N41", 3, None), ("seir3_N161", 1, None), ("seir3_N161", 2, None),
                                             ("seir3_N161", 4, None), ("seir4_N81", 1, None), ("seir4_N81", 2, None), ("seir4_N81", 9, None),
                                             ("seir3_N161", 1, 20), ("seir3_N161", 4, 20), ("seir3_N161", 5, 80), ("sirw_N41", 3, 5)])
def test_deep_trees_match_oracle_draw_for_draw_in_every_kernel_family(tag, chains, band, stream_family):
    """Every streaming-kernel instantiation ("auto": one chain k_stream<1>, more: k_stream<2> in pairs on these small grids; "mc": k_stream_sep
    with its basis planes -- SIRW has three basis functions per component, i.e. a second plane on grid.z; nine chains: the 16-wide
    mirror) against the oracle on transitions that BUILD trees: a first step size of 2e-3 gives trees of depth 5-9 from the first
    transition on, where the reference's 0.1 makes the early transitions reject after one leapfrog -- and a rejected transition
    hides whatever its leapfrogs computed (round 3: a one-chain SIRW build with wrong energies passed every default-step test)."""
    g = load_g4(tag)
    pr = problem_from_g4(g, band)                       # (band: the reference's band_part mask; b = 20 at N = 161 keeps only the blocks that meet the 3b-wide band)
    eng = engine_for(problem_from_g4(g, None), band)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
    burnin, results, step0, depth = 4, 3, 2e-3, 7
    cfg = eng.default_cfg(num_results=results, num_burnin_steps=burnin, step_size=step0, max_tree_depth=depth, stale_cache=0)
    rep = lambda v: np.repeat(np.asarray(v)[None], chains, axis=0)
    ids = list(range(20, 20 + chains))
    eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=808, chain_ids=ids)
    lf, _ = eng.sampler_run(burnin + results)
    Xs, sp, tp = eng.sampler_samples()
    d = eng.sampler_diag()
    eng.close()
    assert lf == d.leapfrogs_taken.sum() and d.leapfrogs_taken.max() >= 31 and d.is_accepted.sum() >= chains
    for i in sorted({0, chains - 1}):
        trace = []
        oX, osp, otp, info, _ = orc.sample_chain(pr, g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), results, burnin, seed=808, chain=ids[i],
                                                 step_size=step0, stale_cache=False, trace=trace, max_tree_depth=depth)
        np.testing.assert_array_equal(d.tree_depth[i], [r.depth for _, r, _ in trace])
        np.testing.assert_array_equal(d.leapfrogs_taken[i], [r.leapfrogs for _, r, _ in trace])
        np.testing.assert_array_equal(d.is_accepted[i], [int(r.is_accepted) for _, r, _ in trace])
        np.testing.assert_allclose(d.target_log_prob[i], [r.target_log_prob for _, r, _ in trace], rtol=1e-8)
        np.testing.assert_allclose(Xs[i], oX, rtol=0, atol=1e-8 * np.abs(oX).max())
        np.testing.assert_allclose(tp[i], otp, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("tag", ["sirw_N41", "seir4_N81"])
def test_fused_log_posterior_is_the_same_in_even_and_odd_slots(tag, monkeypatch, stream_family):
    """The streaming kernels walk their blocks backwards in odd leapfrog slots and read the other halves of the plan ring and of the
    operand mirrors (MAGI_FUSED_PARITY=1 evaluates the fused log posterior that way): same values bit for bit, for one, two and five
    states (the three kernel families)."""
    g = load_g4(tag)
    pr = problem_from_g4(g, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.linspace(0.7, 1.9, pr.P), pr.LB)
    eng = engine_for(pr, None)
    for n in (1, 2, 5):
        X = X0[None] + 0.01 * np.random.default_rng(n).standard_normal((n,) + X0.shape)
        sp, tp = np.repeat(s0[None], n, 0), np.repeat(t0[None], n, 0)
        eng.set_option("fused_parity", 0)
        even = eng.logpost_grad(X, sp, tp, 1.0, fused=True)
        eng.set_option("fused_parity", 1)
        odd = eng.logpost_grad(X, sp, tp, 1.0, fused=True)
        eng.set_option("fused_parity", 0)
        ref = eng.logpost_grad(X, sp, tp, 1.0)
        for a, b in zip(even, odd):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(even[0], ref[0], rtol=1e-10)
        np.testing.assert_allclose(even[1], ref[1], rtol=0, atol=1e-10 * np.abs(ref[1]).max())
    eng.close()


@pytest.mark.parametrize("tag,chains", [("sirw_N41", 1), ("sirw_N41", 2), ("sirw_N41", 4), ("seir4_N81", 1), ("seir4_N81", 2)])
def test_fixed_length_hmc_first_transition_from_a_small_step_matches_oracle(tag, chains):
    """ONE fixed-L HMC transition (L = 8) from a step of 1e-3: its log acceptance ratio is a scalar summary of the eight leapfrogs'
    energies, and for SIRW the proposal is accepted (ratio 0).  This is the configuration in which round 3's one-chain SIRW kernel
    rejected with log_accept_ratio -75.9 while every default-step test passed (tools/exp_family_hmc.py)."""
    g = load_g4(tag)
    pr = problem_from_g4(g, None)
    eng = engine_for(pr, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
    L = 8
    cfg = eng.default_cfg(num_results=1, num_burnin_steps=0, step_size=1e-3, mode=1, hmc_leapfrogs=L)
    rep = lambda v: np.repeat(np.asarray(v)[None], chains, axis=0)
    ids = list(range(7, 7 + chains))
    eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=31, chain_ids=ids)
    eng.sampler_run(1)
    Xs, sp, tp = eng.sampler_samples()
    d = eng.sampler_diag()
    eng.close()
    for i in sorted({0, chains - 1}):
        trace = []
        oX, osp, otp, info, _ = orc.sample_chain(pr, g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), 1, 0, seed=31, chain=ids[i], step_size=1e-3,
                                                 hmc_leapfrogs=L, trace=trace)
        r = trace[0][1]
        assert np.isfinite(r.log_accept_ratio) and (tag != "sirw_N41" or r.is_accepted)
        assert int(d.is_accepted[i, 0]) == int(r.is_accepted)
        np.testing.assert_allclose(d.log_accept_ratio[i, 0], r.log_accept_ratio, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(d.target_log_prob[i, 0], r.target_log_prob, rtol=1e-9)
        np.testing.assert_allclose(tp[i], otp, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
def test_small_problem_chain_is_bit_identical_in_every_batch():
    """Small problems (operator blocks x chain pairs <= 320) keep every batch on the VALU streaming kernel, chain pairs side by side on
    grid.y: a chain's samples are then the same BIT FOR BIT whatever the size of the batch it runs in and wherever it sits in it --
    placement independence without a switch (for large problems see MAGI_STREAM_FAMILY)."""
    g = load_g4("seir4_N81")
    pr = problem_from_g4(g, None)
    eng = engine_for(pr, None)
    X0, s0, t0 = orc.initial_state(g["Xhat_init"], g["sigma_sqs_init"], np.ones(pr.P), pr.LB)
    cfg = eng.default_cfg(num_results=6, num_burnin_steps=6, step_size=2e-3, max_tree_depth=6, stale_cache=0)
    ref = None
    for ids in ([7], [7, 3], [3, 7], [1, 7, 2], [0, 1, 2, 3, 7], [7, 6, 5, 4, 3, 2, 1, 0, 8]):
        n = len(ids)
        rep = lambda v: np.repeat(np.asarray(v)[None], n, axis=0)
        eng.sampler_init(cfg, rep(X0), rep(s0), rep(t0), seed=5, chain_ids=ids)
        eng.sampler_run(12)
        X, sg, th = eng.sampler_samples()
        d = eng.sampler_diag()
        k = ids.index(7)
        if ref is None:
            ref = (X[k].copy(), sg[k].copy(), th[k].copy(), d.leapfrogs_taken[k].copy())
            assert ref[3].max() >= 31
        np.testing.assert_array_equal(X[k], ref[0])
        np.testing.assert_array_equal(sg[k], ref[1])
        np.testing.assert_array_equal(th[k], ref[2])
        np.testing.assert_array_equal(d.leapfrogs_taken[k], ref[3])
    eng.close()
