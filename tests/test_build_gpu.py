"""GPU parity of the matrix build (K1 Matern assembly, K2-K4 Cholesky / products / inverses)
through the C ABI: magi_matern_blocks and magi_build_matrices vs 40-digit mpmath truth (G2), the
reference's own _build_matrices outputs (G1) and the oracle.  Tolerances per layer follow
SURVEY 8c: Matern blocks ~1e-13 of scale against truth; m, K_d and the inverses are
conditioning-limited (~ cond(Kappa) * eps)."""
import numpy as np
import pytest

from oracle import magi_oracle as orc
from tests.util import GOLDEN, load_g4, problem_from_g4

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


@pytest.fixture(scope="module")
def eng():
    from magi_v2_amd.engine import MagiEngine
    e = MagiEngine(0)
    yield e
    e.close()


def relmax(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_matern_blocks_vs_mpmath(eng):
    g = np.load(f"{GOLDEN}/g2_mpmath.npz")
    for tag in g["cases"]:
        p1, p2, v = g[f"{tag}_phi"]
        Kap, pK, Kpp = eng.matern_blocks(g[f"{tag}_I"], p1, p2, v)
        assert relmax(Kap, g[f"{tag}_Kappa"]) < 2e-14, tag
        assert relmax(pK, g[f"{tag}_pKappa"]) < 2e-14, tag
        assert relmax(Kpp, g[f"{tag}_Kappapp"]) < 5e-14, tag
        assert np.array_equal(np.diag(Kap), np.full(len(Kap), p1))
        assert np.array_equal(np.diag(pK), np.zeros(len(Kap)))
        np.testing.assert_allclose(pK, -pK.T, rtol=0, atol=1e-300)          # Kappa_p = -p_Kappa (magi_v2.py:805)


@pytest.mark.parametrize("nu", [2.01, 2.5, 1.5, 3.2])
def test_matern_blocks_other_orders_vs_scipy(eng, nu):
    """nu = 2.01 is what every reference call passes (magi_v2.py:125); other orders exercise the
    order recurrences (n = 1, 3) and the half-integer branch of Temme's constants."""
    I = np.linspace(0, 3.0, 97) ** 1.3            # non-uniform grid
    Kap, pK, Kpp = eng.matern_blocks(I, 0.7, 0.45, nu)
    rK, rpK, rKpp = orc.matern_blocks(I.reshape(-1, 1), 0.7, 0.45, nu)
    assert relmax(Kap, rK) < 1e-13
    assert relmax(pK, rpK) < 1e-9                  # the reference formulas lose digits (see test_oracle_golden)
    assert relmax(Kpp, rKpp) < 1e-8


def test_matern_large_lag_underflows_cleanly(eng):
    I = np.arange(4096) * 0.025                    # u up to ~3000: K_nu underflows, no NaN/inf
    Kap, pK, Kpp = eng.matern_blocks(I, 0.05, 0.08, 2.01)
    assert np.isfinite(Kap).all() and np.isfinite(pK).all() and np.isfinite(Kpp).all()
    assert Kap[0, -1] == 0.0 and Kap[0, 1] > 0.0


def test_build_matrices_vs_mpmath(eng):
    g = np.load(f"{GOLDEN}/g2_mpmath.npz")
    for tag in g["cases"]:
        p1, p2, v = g[f"{tag}_phi"]
        C_inv, m, K_inv = eng.build_matrices(g[f"{tag}_I"], [p1], [p2], v)
        cond = np.linalg.cond(g[f"{tag}_Kappa"])
        condK = np.linalg.cond(g[f"{tag}_K"])
        assert relmax(C_inv[0], g[f"{tag}_Cinv"]) < 20 * cond * EPS, tag
        assert relmax(m[0], g[f"{tag}_m"]) < 20 * cond * EPS, tag
        assert relmax(K_inv[0], g[f"{tag}_Kinv"]) < 20 * max(cond, condK) * EPS * condK ** 0.5, tag


def test_build_matrices_vs_reference_golden(eng):
    """G1: (C, m, K) produced by the reference's own _build_matrices."""
    g = np.load(f"{GOLDEN}/g1_build_matrices.npz")
    for tag in g["cases"]:
        p1, p2, v = g[f"{tag}_phi"]
        I = g[f"{tag}_I"]
        C_inv, m, K_inv = eng.build_matrices(I, [p1], [p2], v)
        cond = np.linalg.cond(g[f"{tag}_C"])
        # C_inv * C = I
        R = C_inv[0] @ g[f"{tag}_C"] - np.eye(len(I))
        assert np.abs(R).max() < 50 * cond * EPS, (tag, np.abs(R).max())
        assert relmax(m[0], g[f"{tag}_m"]) < 50 * cond * EPS, tag
        # K_inv * K_ref = I up to the REFERENCE's own K error: its K_d is a Schur complement formed
        # with an SVD pinv and the lossy Kappa_pp formula, off from truth by ~1e-7 of scale at
        # N=161; K_inv (K_true + dK) - I ~ cond(K) * 1e-7
        R = K_inv[0] @ g[f"{tag}_K"] - np.eye(len(I))
        assert np.abs(R).max() < max(50 * cond * EPS, 2e-7) * np.linalg.cond(g[f"{tag}_K"]), (tag, np.abs(R).max())


@pytest.mark.parametrize("N", [20, 96, 161, 200, 250, 300, 513])
def test_build_multi_block_inverse_property(eng, N):
    """Sizes that are not multiples of the 128-wide Cholesky block; D = 2 components.  The last diagonal block has n = N mod 128 rows and
    the diagonal-block kernel works on ceil(n / 32) sub-blocks of 32 with an identity padding: 20 and 513 (one sub-block), 161 and 300 (two),
    96 and 200 (three), 250 (four, ragged) -- every count takes its own path through the kernel's schedule of factor and worker waves."""
    I = np.arange(N) * 0.025
    phi1, phi2 = np.array([0.03, 0.2]), np.array([0.3, 0.15])
    C_inv, m, K_inv = eng.build_matrices(I, phi1, phi2, 2.01)
    for d in range(2):
        Kap, pK, Kpp = eng.matern_blocks(I, phi1[d], phi2[d], 2.01)
        cond = np.linalg.cond(Kap)
        assert np.abs(C_inv[d] @ Kap - np.eye(N)).max() < 50 * cond * EPS
        assert np.abs(C_inv[d] - C_inv[d].T).max() == 0.0
        m_ref = np.linalg.solve(Kap, pK.T).T                      # p_Kappa Kappa^-1
        assert relmax(m[d], m_ref) < 50 * cond * EPS
        K_ref = Kpp - m_ref @ (-pK)
        K_ref = 0.5 * (K_ref + K_ref.T)
        condK = np.linalg.cond(K_ref)
        assert np.abs(K_inv[d] @ K_ref - np.eye(N)).max() < 200 * cond * EPS * condK ** 0.5


def test_concurrent_and_serial_component_builds_are_identical(eng, monkeypatch):
    """The D components are built on D streams / work spaces by default, one after the other on one work space with
    MAGI_BUILD_SERIAL=1 (also the path taken when HBM is short): same kernels, same operands, so the matrices must
    be bit-identical -- any difference would be a missing dependency between streams."""
    N = 300
    I = np.arange(N) * 0.025
    phi1, phi2 = np.array([0.03, 0.2, 1.1, 0.5]), np.array([0.3, 0.15, 0.4, 0.22])
    conc = eng.build_matrices(I, phi1, phi2, 2.01)
    monkeypatch.setenv("MAGI_BUILD_SERIAL", "1")
    ser = eng.build_matrices(I, phi1, phi2, 2.01)
    for a, b in zip(conc, ser):
        assert np.array_equal(a, b)


def test_built_matrices_feed_logpost_and_band(eng):
    """End to end on the device-resident matrices: build (band 80 at N=161 -> masked dense) and
    evaluate the log posterior; compare with the oracle on the ORACLE's matrices.  Agreement is
    limited by the conditioning of the build, not by the log-posterior kernels (those are held
    to 1e-10 on identical matrices in test_logpost_gpu.py)."""
    g = load_g4("seir3_N161")
    pr = problem_from_g4(g, 80)
    eng.build_matrices(g["I"], g["phi1s"], g["phi2s"], 2.01, bandsize=80, want_host=False)
    eng.set_problem(pr.mu, pr.N_ds, pr.obs_idx, pr.y, pr.beta, pr.LB, pr.drift)
    for si in range(3):
        X, sp, tp = g["state_X"][si], g["state_sig_pre"][si], g["state_th_pre"][si]
        lp, gX, gs, gt, terms = eng.logpost_grad(X, sp, tp, 1.0, want_terms=True)
        t_ref = orc.logpost_terms(X, sp, tp, pr)[:4]
        np.testing.assert_allclose(terms, t_ref, rtol=2e-5)
        l0, gx0, gs0, gt0 = orc.logpost_grad(X, sp, tp, 1.0, pr)
        assert abs(lp - l0) < 2e-5 * abs(l0)
        assert relmax(gt, gt0) < 1e-4


def test_build_rejects_bad_arguments(eng):
    from magi_v2_amd.engine import MagiHipError
    I = np.arange(32) * 0.1
    with pytest.raises(MagiHipError):
        eng.build_matrices(I, [0.1], [-0.3], 2.01)
    with pytest.raises(MagiHipError):
        eng.build_matrices(I, [0.1], [0.3], 0.9)           # nu <= 1: derivative process undefined
    with pytest.raises(MagiHipError) as ei:                 # duplicate grid points -> singular Kappa
        eng.build_matrices(np.zeros(16), [0.1], [0.3], 2.01)
    assert ei.value.code == -3
