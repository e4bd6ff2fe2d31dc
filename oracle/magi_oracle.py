"""CPU oracle for the MAGI hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a plain numpy (fp64) restatement of the reference's algorithm for the
north-star path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package ``magi_v2_amd`` never does.

Citations are relative to the reference tree (``/root/reference``):

* Matern(nu) kernel matrices ............ ``magi_v2.py:774-823`` (``_build_matrices``)
* inverse + band sites .................. ``magi_v2.py:126-128, 271-274``
* tempered log posterior ................ ``magi_v2.py:308-348`` (``unnormalized_log_prob``)
* state init / result transforms ........ ``magi_v2.py:299-300, 374-383, 417-419``
* temperature schedule + annealed kernel  ``magi_v2.py:833-835, 852-879``
* sampler wiring ........................ ``magi_v2.py:360-371, 386-396``
* host helpers .......................... ``magi_v2.py:475-498, 509-527, 552-556, 631-639, 695-770``
* theta / unobserved-component initialisers ``magi_v2.py:133-179, 182-249``
* drifts ................................ ``vignette.ipynb`` cell 3, ``test_magi_script.py:19-45``

Pinning status
--------------
* ``build_matrices`` / helpers: PINNED -- checked against outputs of the reference's own
  TF-free functions run in the build container (``tests/golden/make_golden.py`` ->
  ``tests/golden/*.npz``) and against 40-digit mpmath truth.
* ``logpost``: pinned against an op-for-op torch transcription + autograd (fixture G4); that transcription is also a module
  of its own, ``oracle/torch_cpu.py`` (the CPU baseline ``bench.py`` times), checked against G4 as well.
* ``theta_init_objective`` / ``fit_thetas_init`` / ``fit_unobserved`` (``magi_v2.py:133-249``): restated line by line, checked
  against an independent torch transcription + autograd and central differences; the optimiser is tf_keras Adam restated from
  its documented defaults -- **parity unpinned** against tf_keras itself.
* ``nuts_*`` / ``dual_averaging_*`` / ``sample_chain``: **PARITY UNPINNED**.  The arithmetic
  lives in tensorflow-probability 0.24.0 (``requirements.txt:8``), which is not in the
  reference tree and not installable here; the reference calls ``sample_chain`` without a
  seed (``magi_v2.py:389-395``) and ships no test/golden vector for it.  The functions
  below restate TFP's published NUTS (iterative tree doubling, multinomial sampling,
  generalised U-turn, ``max_tree_depth=10``, ``max_energy_diff=1000``) and dual averaging
  (Nesterov; ``exploration_shrinkage=0.05``, ``step_count_smoothing=10``,
  ``decay_rate=0.75``, ``shrinkage_target=10*step_size``) from memory of that code, with a
  Philox4x32-10 counter RNG of our own so the HIP sampler can be compared draw-for-draw.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Tuple

import numpy as np
from scipy.interpolate import splev, splrep
from scipy.special import gamma, kvp

# --------------------------------------------------------------------------------------
# Host helpers (magi_v2.py:475-527, 552-556, 631-639, 695-770)
# --------------------------------------------------------------------------------------


def discretize(ts_obs: np.ndarray, X_obs: np.ndarray, discretization: int):
    """``_discretize`` (magi_v2.py:475-498): insert 2^k-1 grid points between observations."""
    ts_obs = np.asarray(ts_obs, dtype=np.float64).flatten()
    assert ts_obs.shape[0] == X_obs.shape[0]
    N, D = X_obs.shape
    step = 2 ** discretization
    N_discret = step * (N - 1) + 1
    I = np.full((N_discret,), np.nan)
    X_obs_discret = np.full((N_discret, D), np.nan)
    I[::step] = ts_obs
    indices = np.arange(len(I))
    I = np.interp(x=indices, xp=indices[~np.isnan(I)], fp=I[~np.isnan(I)])
    X_obs_discret[::step] = X_obs
    return I.reshape(-1, 1), X_obs_discret


def linear_interpolate(X_partial: np.ndarray) -> np.ndarray:
    """``_linear_interpolate`` (magi_v2.py:509-527)."""
    N_partial, D_partial = X_partial.shape
    X_interp = X_partial.copy()
    indices = np.arange(N_partial)
    for d in range(D_partial):
        nan = np.isnan(X_partial[:, d])
        if np.any(nan):
            X_interp[:, d] = np.interp(x=indices, xp=indices[~nan], fp=X_partial[~nan, d])
    return X_interp


def cubic_smoother(I: np.ndarray, X_filled: np.ndarray) -> np.ndarray:
    """``cv_cubic_smoother`` (magi_v2.py:695-770).

    The reference runs a 5-fold CV over knot counts and then ignores its result: the final
    fit uses the loop variable ``knot_num`` left over from the CV loop (= ``len(I)//10``,
    magi_v2.py:747-757).  Only that effective behaviour is restated.
    """
    I = np.asarray(I, dtype=np.float64).flatten()
    if I.shape[0] < 10:
        return X_filled
    knot_num = I.shape[0] // 10
    if knot_num == 0:
        knots = np.array([])
    else:
        knots = np.linspace(start=I[0], stop=I[-1], num=knot_num + 2)[1:-1]
    cols = []
    for i in range(X_filled.shape[1]):
        tck = splrep(I, X_filled[:, i], t=knots, s=0)
        cols.append(splev(I, tck))
    return np.stack(cols, axis=1)


def fourier_phi2_prior(x: np.ndarray) -> Tuple[float, float]:
    """FFT-informed prior mean/sd for phi2 (magi_v2.py:552-556)."""
    z = np.fft.fft(x)
    zmod = np.abs(z)
    zmod_eff = zmod[1:(len(zmod) - 1) // 2 + 1]
    zsq = zmod_eff ** 2
    idxs = np.linspace(1, len(zmod_eff), len(zmod_eff))
    freq = np.sum(idxs * zsq) / np.sum(zsq)
    mu_phi2 = 0.5 / freq
    return mu_phi2, (1 - mu_phi2) / 3


def hparams_initial(X_filled: np.ndarray) -> Dict[str, np.ndarray]:
    """The reference's *initial* hyper-parameter values (magi_v2.py:631-639), i.e. the state
    before the TFP Adam fit (which is SURVEY section 8 row f1, not on the hot path)."""
    phi1 = X_filled.std(axis=0) ** 2
    phi2 = np.array([fourier_phi2_prior(X_filled[:, d])[0] for d in range(X_filled.shape[1])])
    sig = (X_filled.std(axis=0) * 0.1) ** 2
    return {"phi1s": phi1, "phi2s": phi2, "sigma_sqs": sig}


# --------------------------------------------------------------------------------------
# GP hyper-parameter fit (magi_v2.py:538-691) -- PARITY UNPINNED (TFP GaussianProcess /
# GeneralizedMatern / TruncatedNormal / tf_keras Adam are not in the reference tree).  Restated:
# objective D * sum_d [GP marginal_d + priors_d] (the [D, D] broadcast of log_prob summed by
# tape.gradient, :604-608, 649-665), GP covariance phi1 R_nu(phi2) + (sigma^2 + 1e-6) I with the
# constant mean of :559, softplus-reparameterised variables (:631-642), Adam(lr=.01) x 1000 (:654-678).
# --------------------------------------------------------------------------------------


def gradient_matching_loss(I, X_obs_smoothed, X_unobs, thetas, proper_order, drift):
    """magi_v2.py:196-216: objective of the joint (X_unobs, theta) initialisation of completely unobserved
    components -- squared mismatch between the drift and centred finite differences on the interior grid."""
    X_full = np.concatenate([X_obs_smoothed, X_unobs], axis=1)[:, proper_order]          # tf.gather(..., axis=1)
    f_vals = DRIFTS[drift][0](X_full, thetas)[0]
    f_diff = (X_full[2:, :] - X_full[:-2, :]) / (2 * (I[1, 0] - I[0, 0]))
    return float(np.sum((f_vals[1:-1] - f_diff) ** 2))


def gp_marginal_and_grad(I, x, mu, phi1, phi2, sig2, nu=2.01, jitter=1e-6):
    """log N(x; mu, S) and d/d(phi1, phi2, sig2), S = Kappa(phi1, phi2) + (sig2 + jitter) I."""
    I = np.asarray(I, dtype=np.float64).reshape(-1)
    N = I.shape[0]
    Kap, pK, _ = matern_blocks(I.reshape(-1, 1), phi1, phi2, nu)
    S = Kap + (sig2 + jitter) * np.eye(N)
    L = np.linalg.cholesky(S)
    r = x - mu
    Sinv = np.linalg.inv(S)
    a = Sinv @ r
    ll = -0.5 * r @ a - np.sum(np.log(np.diag(L))) - 0.5 * N * np.log(2.0 * np.pi)
    W = np.outer(a, a) - Sinv
    dK2 = -pK * (I[:, None] - I[None, :]) / phi2
    return ll, np.array([0.5 * np.sum(W * Kap) / phi1, 0.5 * np.sum(W * dK2), 0.5 * np.trace(W)])


def fit_kernel_hparams(I, X_filled, num_iters=1000, lr=0.01, nu=2.01, jitter=1e-6, init=None, trace=None):
    """``_fit_kernel_hparams`` (magi_v2.py:538-691) restated in numpy."""
    N, D = X_filled.shape
    mu = X_filled.mean(axis=0)
    pri = [fourier_phi2_prior(X_filled[:, d]) for d in range(D)]
    mu_phi2, sd_phi2 = np.array([p[0] for p in pri]), np.array([p[1] for p in pri])
    hp = hparams_initial(X_filled) if init is None else init
    sig_loc = (X_filled.std(axis=0) * 0.1) ** 2
    sp_inv = lambda y: np.log(np.expm1(y))
    sp = lambda x: np.log1p(np.exp(x))
    sg = lambda x: 1.0 / (1.0 + np.exp(-x))
    raw = np.concatenate([sp_inv(hp["phi1s"]), sp_inv(hp["phi2s"]), sp_inv(hp["sigma_sqs"])])
    m, v = np.zeros_like(raw), np.zeros_like(raw)
    sD = np.sqrt(D)
    for t in range(1, num_iters + 1):
        grad = np.zeros_like(raw)
        loss = 0.0
        for d in range(D):
            p1, p2, s2 = sp(raw[d]), sp(raw[D + d]), sp(raw[2 * D + d])
            ll, g3 = gp_marginal_and_grad(I, X_filled[:, d], mu[d], p1, p2, s2, nu, jitter)
            sc = np.array([1000.0 * sD, sd_phi2[d] * sD, 1000.0 * sD])
            z = (np.array([p1, p2, s2]) - np.array([1e-4, mu_phi2[d], sig_loc[d]])) / sc
            loss -= D * (ll - 0.5 * np.sum(z * z))
            gtot = g3 - z / sc
            for k in range(3):
                grad[k * D + d] = -D * gtot[k] * sg(raw[k * D + d])
        if trace is not None:
            trace.append(loss)
        a = lr * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        m = 0.9 * m + 0.1 * grad
        v = 0.999 * v + 0.001 * grad * grad
        raw = raw - a * m / (np.sqrt(v) + 1e-7)
    return {"phi1s": sp(raw[:D]), "phi2s": sp(raw[D:2 * D]), "sigma_sqs": sp(raw[2 * D:])}


# --------------------------------------------------------------------------------------
# Kernel matrices (magi_v2.py:774-823, 126-128, 271-274)
# --------------------------------------------------------------------------------------


def matern_blocks(I: np.ndarray, phi1: float, phi2: float, v: float = 2.01):
    """Kappa, p_Kappa, Kappa_pp exactly as the reference forms them (magi_v2.py:781-815)."""
    I = np.asarray(I, dtype=np.float64)
    s = np.tile(A=I.reshape(-1, 1), reps=I.shape[0])
    t = s.T
    l = np.abs(s - t)
    u = np.sqrt(2 * v) * l / phi2
    np.fill_diagonal(a=u, val=np.nan)
    with np.errstate(invalid="ignore", divide="ignore"):
        Bv0, Bv1, Bv2 = kvp(v=v, z=u, n=0), kvp(v=v, z=u, n=1), kvp(v=v, z=u, n=2)

        Kappa = (phi1 / gamma(v)) * (2 ** (1 - (v / 2))) * ((np.sqrt(v) / phi2) ** v)
        Kappa = Kappa * Bv0
        Kappa *= (l ** v)
        np.fill_diagonal(Kappa, val=phi1)

        p_Kappa = (2 ** (1 - (v / 2)))
        p_Kappa = p_Kappa * phi1 * ((u / np.sqrt(2)) ** v)
        p_Kappa *= ((u * phi2 * Bv1) + (v * phi2 * Bv0))
        p_Kappa /= (phi2 * (s - t) * gamma(v))
        np.fill_diagonal(p_Kappa, val=0.0)

        Kappa_pp = 2 * np.sqrt(2) * (v ** 1.5) * phi2 * l * Bv1
        Kappa_pp += (((v ** 2) * (phi2 ** 2)) - (v * (phi2 ** 2))) * Bv0
        Kappa_pp += ((2 * v * (s ** 2)) - (4 * v * s * t) + (2 * v * (t ** 2))) * Bv2
        Kappa_pp *= (-1.0 * (2 ** (1 - (v / 2))) * phi1 * ((u / np.sqrt(2)) ** v))
        Kappa_pp /= ((phi2 ** 2) * (l ** 2) * gamma(v))
        np.fill_diagonal(Kappa_pp, val=v * phi1 / ((phi2 ** 2) * (v - 1)))
    return Kappa, p_Kappa, Kappa_pp


def matern_block_columns(I: np.ndarray, cols, phi1: float, phi2: float, v: float = 2.01):
    """Columns ``cols`` of Kappa, p_Kappa, Kappa_pp ([N, len(cols)] each): the expressions of ``matern_blocks``
    (magi_v2.py:781-815) evaluated on those columns only -- the truth side of the N = 8192 build test, where the full
    N x N blocks (3 x 537 MB, three AMOS sweeps) are out of a test's reach."""
    I = np.asarray(I, dtype=np.float64).reshape(-1)
    cols = np.asarray(cols, dtype=np.int64)
    s = np.repeat(I[:, None], len(cols), axis=1)
    t = np.repeat(I[cols][None, :], len(I), axis=0)
    diag = (np.arange(len(I))[:, None] == cols[None, :])
    l = np.abs(s - t)
    u = np.sqrt(2 * v) * l / phi2
    u[diag] = np.nan
    with np.errstate(invalid="ignore", divide="ignore"):
        Bv0, Bv1, Bv2 = kvp(v=v, z=u, n=0), kvp(v=v, z=u, n=1), kvp(v=v, z=u, n=2)
        Kappa = (phi1 / gamma(v)) * (2 ** (1 - (v / 2))) * ((np.sqrt(v) / phi2) ** v)
        Kappa = Kappa * Bv0
        Kappa *= (l ** v)
        Kappa[diag] = phi1
        p_Kappa = (2 ** (1 - (v / 2)))
        p_Kappa = p_Kappa * phi1 * ((u / np.sqrt(2)) ** v)
        p_Kappa *= ((u * phi2 * Bv1) + (v * phi2 * Bv0))
        p_Kappa /= (phi2 * (s - t) * gamma(v))
        p_Kappa[diag] = 0.0
        Kappa_pp = 2 * np.sqrt(2) * (v ** 1.5) * phi2 * l * Bv1
        Kappa_pp += (((v ** 2) * (phi2 ** 2)) - (v * (phi2 ** 2))) * Bv0
        Kappa_pp += ((2 * v * (s ** 2)) - (4 * v * s * t) + (2 * v * (t ** 2))) * Bv2
        Kappa_pp *= (-1.0 * (2 ** (1 - (v / 2))) * phi1 * ((u / np.sqrt(2)) ** v))
        Kappa_pp /= ((phi2 ** 2) * (l ** 2) * gamma(v))
        Kappa_pp[diag] = v * phi1 / ((phi2 ** 2) * (v - 1))
    # scipy's kvp underflows to exact zeros (and 0 * inf = NaN in the products above) beyond u ~ 700: those entries are zero
    far = u > 600.0
    for A in (Kappa, p_Kappa, Kappa_pp):
        A[far & ~np.isfinite(A)] = 0.0
    return Kappa, p_Kappa, Kappa_pp


def matern_block_columns_accurate(I: np.ndarray, cols, phi1: float, phi2: float, v: float = 2.01):
    """The same three blocks on columns ``cols`` through the cancellation-free forms (with u = c |s - t|,
    c = sqrt(2 v) / phi2, A = phi1 2^(1-v) / Gamma(v)):
        kappa = A u^v K_v(u),   d kappa / ds = -A c sgn(s - t) u^v K_{v-1}(u),   d^2 kappa / ds dt = A c^2 u^(v-1) (K_{v-1}(u) - u K_{v-2}(u))
    -- mathematically the reference's expressions (magi_v2.py:790-815, via K_v' = -K_{v-1} - (v/u) K_v), without their loss of
    digits at small lags (the reference's Kappa_pp carries ~2e-10 of its scale there, which K^-1 amplifies to 1e-5 in
    K^-1 K = I at N = 8192).  Pinned to the 40-digit mpmath fixture G2 in tests/test_oracle_golden.py; it is the arbiter
    where a check has to resolve below the reference's own rounding."""
    from scipy.special import kv
    I = np.asarray(I, dtype=np.float64).reshape(-1)
    cols = np.asarray(cols, dtype=np.int64)
    dlt = I[:, None] - I[cols][None, :]
    diag = (np.arange(len(I))[:, None] == cols[None, :])
    c = np.sqrt(2 * v) / phi2
    u = c * np.abs(dlt)
    u[diag] = 1.0                                    # (placeholder, overwritten below)
    A = phi1 * 2.0 ** (1.0 - v) / gamma(v)
    with np.errstate(invalid="ignore", over="ignore", under="ignore"):
        uv = u ** v
        Kv, Kv1, Kv2 = kv(v, u), kv(v - 1.0, u), kv(v - 2.0, u)
        Kappa = A * uv * Kv
        p_Kappa = -A * c * np.sign(dlt) * uv * Kv1
        Kappa_pp = A * c * c * (u ** (v - 1.0)) * (Kv1 - u * Kv2)
    for M in (Kappa, p_Kappa, Kappa_pp):
        M[~np.isfinite(M)] = 0.0                     # (u beyond ~700: K underflows to 0, u^v overflows nowhere near: 0 * finite)
    Kappa[diag] = phi1
    p_Kappa[diag] = 0.0
    Kappa_pp[diag] = v * phi1 / ((phi2 ** 2) * (v - 1))
    return Kappa, p_Kappa, Kappa_pp


def build_matrices(I: np.ndarray, phi1: float, phi2: float, v: float = 2.01):
    """``_build_matrices`` (magi_v2.py:774-823): returns (C_d, m_d, K_d)."""
    Kappa, p_Kappa, Kappa_pp = matern_blocks(I, phi1, phi2, v)
    Kappa_p = p_Kappa * -1
    C_d, Kappa_inv = Kappa.copy(), np.linalg.pinv(Kappa)
    m_d = p_Kappa @ Kappa_inv
    K_d = Kappa_pp - (p_Kappa @ Kappa_inv @ Kappa_p)
    return C_d, m_d, K_d


def band_part(A: np.ndarray, b: Optional[int]) -> np.ndarray:
    """``tf.linalg.band_part(A, b, b)`` (magi_v2.py:271-274): zero entries with |i-j| > b."""
    if b is None:
        return A
    n = A.shape[-1]
    i = np.arange(n)
    mask = np.abs(i[:, None] - i[None, :]) <= b
    return A * mask


def build_all(I: np.ndarray, phi1s, phi2s, v: float = 2.01, bandsize: Optional[int] = None):
    """Per-component (C_d^-1, m_d, K_d^-1) stacks as ``initial_fit`` stores them
    (magi_v2.py:117-128, 271-274).  ``tf.linalg.pinv`` is restated with numpy's pinv."""
    D = len(phi1s)
    N = np.asarray(I).reshape(-1).shape[0]
    C_inv = np.zeros((D, N, N))
    m = np.zeros((D, N, N))
    K_inv = np.zeros((D, N, N))
    for d in range(D):
        C_d, m_d, K_d = build_matrices(I, phi1s[d], phi2s[d], v)
        C_inv[d] = np.linalg.pinv(C_d)
        m[d] = m_d
        K_inv[d] = np.linalg.pinv(K_d)
    return band_part(C_inv, bandsize), band_part(m, bandsize), band_part(K_inv, bandsize)


# --------------------------------------------------------------------------------------
def adam_minimise(value_and_grad, x0, iters, lr=0.01, b1=0.9, b2=0.999, eps=1e-7):
    """tf_keras.optimizers.Adam(learning_rate=lr) at its defaults (beta_1 .9, beta_2 .999, epsilon 1e-7, no amsgrad), the
    optimiser of magi_v2.py:161, 230, 654:  m, v moments;  x -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps).
    ``x0`` / the gradients may be a single array or a list of arrays (Adam is element-wise).  Returns (x, losses)."""
    xs = [np.array(a, dtype=np.float64) for a in (x0 if isinstance(x0, (list, tuple)) else [x0])]
    ms, vs = [np.zeros_like(a) for a in xs], [np.zeros_like(a) for a in xs]
    losses = []
    for t in range(1, iters + 1):
        loss, grads = value_and_grad(*xs)
        grads = grads if isinstance(grads, (list, tuple)) else [grads]
        losses.append(loss)
        alpha = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        for k, g in enumerate(grads):
            ms[k] = b1 * ms[k] + (1.0 - b1) * g
            vs[k] = b2 * vs[k] + (1.0 - b2) * g * g
            xs[k] = xs[k] - alpha * ms[k] / (np.sqrt(vs[k]) + eps)
    return (xs if isinstance(x0, (list, tuple)) else xs[0]), np.array(losses)


def theta_init_objective(thetas, Xhat_init, mu_ds, m_ds, K_d_invs, drift):
    """``theta_objective`` of magi_v2.py:148-158 and its gradient with respect to theta.

    Operation for operation, INCLUDING the reference's ``tf.reshape`` of the [N, D] drift values to [D, N, 1]
    (magi_v2.py:155-156; the log posterior transposes instead, :335), which re-interprets the row-major buffer: entry
    (d, n) of the reshaped array is element d*N + n of the flattened [N, D] array.  ``m_ds`` / ``K_d_invs`` are the
    matrices BEFORE the band approximation (the initialiser runs at :133-179, the band is applied at :271-274) and
    ``Xhat_init`` is the interpolated grid before smoothing (:112-113, smoothing at :277).
    The reference differentiates with tf.GradientTape (:164-166); the gradient here is the same derivative written out:
    d/dtheta_p = sum_d reshape(df/dtheta_p)_d^T (K_d^-1 + K_d^-T) toNorm_d."""
    N, D = Xhat_init.shape
    X_cent = (Xhat_init - mu_ds).reshape(N, 1, D)                                   # :139-141
    m_prod = m_ds @ np.transpose(X_cent, (2, 0, 1))                                 # :142   [D, N, 1]
    f, _, T = DRIFTS[drift][0](Xhat_init, thetas)                                   # [N, D], -, [N, D, P]
    f_vals = np.reshape(f, (D, N, 1))                                               # :155-156 (reshape, not transpose)
    toNorm = f_vals - m_prod                                                        # :157
    Kt = K_d_invs @ toNorm
    val = float(np.sum(np.transpose(toNorm, (0, 2, 1)) @ Kt))                       # :158
    g = (Kt + np.transpose(K_d_invs, (0, 2, 1)) @ toNorm)[:, :, 0]                  # [D, N]
    P = T.shape[2]
    grad = np.array([np.sum(np.reshape(T[:, :, p], (D, N)) * g) for p in range(P)])
    return val, grad


def fit_thetas_init(Xhat_init, mu_ds, m_ds, K_d_invs, drift, P, num_iters=10000, lr=0.01):
    """magi_v2.py:133-179: Adam(lr=.01) x 10 000 on ``theta_objective`` from theta = 1."""
    fn = lambda th: theta_init_objective(th, Xhat_init, mu_ds, m_ds, K_d_invs, drift)
    th, losses = adam_minimise(fn, np.ones(P), num_iters, lr)
    return th, losses


def gradient_matching_loss_and_grads(I, X_obs_smoothed, X_unobs, thetas, proper_order, unobserved_components, drift):
    """``unobserved_objective`` of magi_v2.py:199-216 with its gradients with respect to (X_unobs, thetas) written out
    (the reference uses tf.GradientTape, :233-235)."""
    X_full = np.concatenate([X_obs_smoothed, X_unobs], axis=1)[:, proper_order]
    f, J, T = DRIFTS[drift][0](X_full, thetas)
    h2 = 2 * (I[1, 0] - I[0, 0])
    r = f[1:-1] - (X_full[2:, :] - X_full[:-2, :]) / h2                             # [N-2, D]
    gX = np.zeros_like(X_full)
    gX[1:-1] += 2.0 * np.einsum("nd,nde->ne", r, J[1:-1])
    gX[2:] -= 2.0 * r / h2
    gX[:-2] += 2.0 * r / h2
    gth = 2.0 * np.einsum("nd,ndp->p", r, T[1:-1])
    return float(np.sum(r ** 2)), gX[:, unobserved_components], gth


def fit_unobserved(I, X_interp_obs, X_obs_smoothed, proper_order, unobserved_components, drift, P, seed, num_iters=10000, lr=0.01):
    """magi_v2.py:182-249: joint Adam on (X_unobs, theta) from X_unobs ~ N(mean of the interpolated observed values,
    root-mean variance of their columns) (:219-226; the reference's draw is unseeded -- ``seed`` feeds PCG64 here, as in the
    product) and theta = 1 (:227)."""
    n = X_interp_obs.shape[0]
    mu0 = X_interp_obs.mean()
    sd0 = (X_interp_obs.std(axis=0) ** 2).mean() ** 0.5
    rng = np.random.Generator(np.random.PCG64(seed))
    Xu0 = rng.normal(loc=mu0, scale=sd0, size=(n, len(unobserved_components)))
    fn = lambda Xu, th: (lambda v: (v[0], [v[1], v[2]]))(gradient_matching_loss_and_grads(I, X_obs_smoothed, Xu, th, proper_order, unobserved_components, drift))
    (Xu, th), losses = adam_minimise(fn, [Xu0, np.ones(P)], num_iters, lr)
    return Xu, th, losses


# Drifts f(t, X, theta) with analytic Jacobians (vignette.ipynb cell 3,
# test_magi_script.py:19-45; SEIR-4 = the four data columns with S explicit)
# --------------------------------------------------------------------------------------


def _seir3(X, th):
    E, I_, R = X[:, 0], X[:, 1], X[:, 2]
    S = 1.0 - (E + I_ + R)
    b, g, s = th
    f = np.stack([b * S * I_ - s * E, s * E - g * I_, g * I_], axis=1)
    N = X.shape[0]
    # J[n, d, d'] = d f_d / d x_d'
    J = np.zeros((N, 3, 3))
    J[:, 0, 0] = -b * I_ - s
    J[:, 0, 1] = b * S - b * I_
    J[:, 0, 2] = -b * I_
    J[:, 1, 0] = s
    J[:, 1, 1] = -g
    J[:, 2, 1] = g
    # T[n, d, p] = d f_d / d theta_p
    T = np.zeros((N, 3, 3))
    T[:, 0, 0] = S * I_
    T[:, 0, 2] = -E
    T[:, 1, 1] = -I_
    T[:, 1, 2] = E
    T[:, 2, 1] = I_
    return f, J, T


def _seir4(X, th):
    S, E, I_, R = X[:, 0], X[:, 1], X[:, 2], X[:, 3]
    b, g, s = th
    f = np.stack([-b * S * I_, b * S * I_ - s * E, s * E - g * I_, g * I_], axis=1)
    N = X.shape[0]
    J = np.zeros((N, 4, 4))
    J[:, 0, 0] = -b * I_
    J[:, 0, 2] = -b * S
    J[:, 1, 0] = b * I_
    J[:, 1, 1] = -s
    J[:, 1, 2] = b * S
    J[:, 2, 1] = s
    J[:, 2, 2] = -g
    J[:, 3, 2] = g
    T = np.zeros((N, 4, 3))
    T[:, 0, 0] = -S * I_
    T[:, 1, 0] = S * I_
    T[:, 1, 2] = -E
    T[:, 2, 1] = -I_
    T[:, 2, 2] = E
    T[:, 3, 1] = I_
    return f, J, T


def _sirw(X, th):
    S, I_, R, W = X[:, 0], X[:, 1], X[:, 2], X[:, 3]
    beta, phi, xi, chi, kappa = th
    f = np.stack([
        -beta * S * I_ + kappa * W,
        beta * S * I_ - phi * I_,
        phi * I_ - xi * R + chi * I_ * W,
        xi * R - chi * I_ * W - kappa * W,
    ], axis=1)
    N = X.shape[0]
    J = np.zeros((N, 4, 4))
    J[:, 0, 0] = -beta * I_
    J[:, 0, 1] = -beta * S
    J[:, 0, 3] = kappa
    J[:, 1, 0] = beta * I_
    J[:, 1, 1] = beta * S - phi
    J[:, 2, 1] = phi + chi * W
    J[:, 2, 2] = -xi
    J[:, 2, 3] = chi * I_
    J[:, 3, 1] = -chi * W
    J[:, 3, 2] = xi
    J[:, 3, 3] = -chi * I_ - kappa
    T = np.zeros((N, 4, 5))
    T[:, 0, 0] = -S * I_
    T[:, 0, 4] = W
    T[:, 1, 0] = S * I_
    T[:, 1, 1] = -I_
    T[:, 2, 1] = I_
    T[:, 2, 2] = -R
    T[:, 2, 3] = I_ * W
    T[:, 3, 2] = R
    T[:, 3, 3] = -I_ * W
    T[:, 3, 4] = -W
    return f, J, T


DRIFTS: Dict[str, Tuple[Callable, int, int]] = {
    # name -> (fn, D, P)
    "seir3": (_seir3, 3, 3),
    "seir4": (_seir4, 4, 3),
    "sirw": (_sirw, 4, 5),
}
DRIFT_IDS = {"seir3": 0, "seir4": 1, "sirw": 2}


# --------------------------------------------------------------------------------------
# Log posterior + analytic gradient (magi_v2.py:308-348)
# --------------------------------------------------------------------------------------


@dataclass
class Problem:
    """Constants captured by the reference's log-posterior closure (magi_v2.py:294-300)."""
    I: np.ndarray            # [N]
    mu: np.ndarray           # [D]        magi_v2.py:114, 259
    C_inv: np.ndarray        # [D,N,N]    magi_v2.py:126
    m: np.ndarray            # [D,N,N]    magi_v2.py:127
    K_inv: np.ndarray        # [D,N,N]    magi_v2.py:128
    N_ds: np.ndarray         # [D]        magi_v2.py:53
    obs_idx: np.ndarray      # [n_obs] int64 flat row-major index into X[N,D]; magi_v2.py:96
    y: np.ndarray            # [n_obs]    magi_v2.py:100
    beta: float              # D*|I| / sum(N_ds); magi_v2.py:89
    LB: np.ndarray           # [D]        magi_v2.py:300
    drift: str               # key of DRIFTS
    P: int

    @property
    def N(self):
        return self.C_inv.shape[1]

    @property
    def D(self):
        return self.C_inv.shape[0]


def logpost_terms(X, sig_pre, th_pre, pr: Problem):
    """t1..t4 and the log-Jacobians, op-for-op after magi_v2.py:318-345."""
    sigma_sqs = np.log(1.0 + np.exp(sig_pre)) + pr.LB
    thetas = np.log(1.0 + np.exp(th_pre))
    lj_s = np.sum(sig_pre - np.log(1.0 + np.exp(sig_pre)))
    lj_t = np.sum(th_pre - np.log(1.0 + np.exp(th_pre)))
    xc = (X - pr.mu).T[:, :, None]                       # [D,N,1]
    t1 = np.sum(np.transpose(xc, (0, 2, 1)) @ pr.C_inv @ xc)
    f = DRIFTS[pr.drift][0](X, thetas)[0]                # [N,D]
    toNorm = f.T[:, :, None] - (pr.m @ xc)               # [D,N,1]
    t2 = np.sum(np.transpose(toNorm, (0, 2, 1)) @ (pr.K_inv @ toNorm))
    t3 = np.sum(pr.N_ds * np.log(2.0 * np.pi * sigma_sqs))
    X_observed = X.reshape(-1)[pr.obs_idx]
    cols = pr.obs_idx % pr.D
    t4 = np.sum(np.square(X_observed - pr.y) * (1.0 / sigma_sqs)[cols])
    return t1, t2, t3, t4, lj_s, lj_t


def logpost(X, sig_pre, th_pre, beta_temp, pr: Problem) -> float:
    """``unnormalized_log_prob`` (magi_v2.py:308-348)."""
    t1, t2, t3, t4, lj_s, lj_t = logpost_terms(X, sig_pre, th_pre, pr)
    return beta_temp * (-0.5 * (((1.0 / pr.beta) * (t1 + t2)) + (t3 + t4)) + lj_s + lj_t)


def logpost_grad(X, sig_pre, th_pre, beta_temp, pr: Problem):
    """Value and analytic gradient of ``unnormalized_log_prob``.

    The reference obtains the gradient by TF reverse-mode autodiff inside TFP's leapfrog
    (induced by magi_v2.py:360-364); this is the same derivative written out
    (SURVEY section 8 row a5).  Matrices are NOT assumed symmetric."""
    D = pr.D
    sp_s = np.log(1.0 + np.exp(sig_pre))
    sigma_sqs = sp_s + pr.LB
    thetas = np.log(1.0 + np.exp(th_pre))
    sg_s = 1.0 / (1.0 + np.exp(-sig_pre))
    sg_t = 1.0 / (1.0 + np.exp(-th_pre))
    lj_s = np.sum(sig_pre - sp_s)
    lj_t = np.sum(th_pre - np.log(1.0 + np.exp(th_pre)))

    xc = (X - pr.mu).T[:, :, None]                       # [D,N,1]
    Cx = pr.C_inv @ xc
    CTx = np.transpose(pr.C_inv, (0, 2, 1)) @ xc
    t1 = np.sum(xc * Cx)
    f, J, T = DRIFTS[pr.drift][0](X, thetas)
    r = f.T[:, :, None] - (pr.m @ xc)                    # [D,N,1]
    Kr = pr.K_inv @ r
    KTr = np.transpose(pr.K_inv, (0, 2, 1)) @ r
    t2 = np.sum(r * Kr)
    g = (Kr + KTr)[:, :, 0]                              # [D,N]
    t3 = np.sum(pr.N_ds * np.log(2.0 * np.pi * sigma_sqs))
    cols = pr.obs_idx % D
    resid = X.reshape(-1)[pr.obs_idx] - pr.y
    t4 = np.sum(np.square(resid) * (1.0 / sigma_sqs)[cols])
    logp = beta_temp * (-0.5 * (((1.0 / pr.beta) * (t1 + t2)) + (t3 + t4)) + lj_s + lj_t)

    # d(t1+t2)/dX
    mTg = (np.transpose(pr.m, (0, 2, 1)) @ g[:, :, None])[:, :, 0]       # [D,N]
    d12 = (Cx + CTx)[:, :, 0] - mTg + np.einsum("dn,nde->en", g, J)       # [D,N]
    d4 = np.zeros(X.size)
    np.add.at(d4, pr.obs_idx, 2.0 * resid / sigma_sqs[cols])
    gX = beta_temp * (-0.5 * ((1.0 / pr.beta) * d12.T + d4.reshape(X.shape)))
    SS = np.zeros(D)
    np.add.at(SS, cols, np.square(resid))
    dsig = pr.N_ds / sigma_sqs - SS / sigma_sqs ** 2
    gsig = beta_temp * (-0.5 * dsig * sg_s + (1.0 - sg_s))
    dth = np.einsum("dn,ndp->p", g, T)
    gth = beta_temp * (-0.5 * (1.0 / pr.beta) * dth * sg_t + (1.0 - sg_t))
    return logp, gX, gsig, gth


# --------------------------------------------------------------------------------------
# State init + result transforms (magi_v2.py:299-300, 374-383, 417-419)
# --------------------------------------------------------------------------------------


def sigma_sqs_lower_bound(Xhat_init: np.ndarray) -> np.ndarray:
    """magi_v2.py:299-300."""
    return (Xhat_init.std(axis=0) * 0.01) ** 2


def initial_state(Xhat_init, sigma_sqs_init, thetas_init, LB):
    """softplus-inverse initial state with the -5.0 fallback (magi_v2.py:374-383)."""
    sig_pre = np.full_like(sigma_sqs_init, -5.0)
    ok = sigma_sqs_init > LB
    sig_pre[ok] = np.log(np.exp((sigma_sqs_init - LB)[ok]) - 1.0)
    th_pre = np.full_like(thetas_init, -5.0)
    ok = thetas_init > 0.0
    th_pre[ok] = np.log(np.exp((thetas_init - 0.0)[ok]) - 1.0)
    return Xhat_init.copy(), sig_pre, th_pre


def transform_samples(sig_pre_samps, th_pre_samps, LB):
    """magi_v2.py:418-419."""
    return np.log(np.exp(sig_pre_samps) + 1.0) + LB, np.log(np.exp(th_pre_samps) + 1.0)


def temperature(step: int, min_temp: float = 0.1) -> float:
    """``logarithmic_temperature_schedule`` (magi_v2.py:833-835)."""
    return max(1.0 / math.log(step + 2.0), min_temp)


# --------------------------------------------------------------------------------------
# Philox4x32-10 counter RNG (ours; shared bit-for-bit with the HIP sampler)
# --------------------------------------------------------------------------------------

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)

STREAM_MOMENTUM, STREAM_DIRECTION, STREAM_LEAF, STREAM_MERGE, STREAM_HMC = 0, 1, 2, 3, 4


def philox4x32(c0, c1, c2, c3, key: int):
    """Vectorised Philox4x32-10; counters broadcast; returns four uint64 arrays (<2^32)."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & _MASK for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint64(key & 0xFFFFFFFF)
    k1 = np.uint64((key >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK, lo1, (hi0 ^ c3 ^ k1) & _MASK, lo0
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def _u01(hi, lo):
    """53-bit uniform in (0,1): ((hi<<32|lo) >> 11 + 0.5) * 2^-53."""
    x = ((hi << np.uint64(32)) | lo) >> np.uint64(11)
    return (x.astype(np.float64) + 0.5) * (2.0 ** -53)


def rng_uniform(index: int, step: int, chain: int, stream: int, key: int) -> float:
    r0, r1, _, _ = philox4x32(index, step, chain, stream, key)
    return float(_u01(r0, r1))


def rng_bit(index: int, step: int, chain: int, stream: int, key: int) -> bool:
    r0, _, _, _ = philox4x32(index, step, chain, stream, key)
    return bool(int(r0) & 1)


def rng_normal(n: int, step: int, chain: int, key: int) -> np.ndarray:
    """n standard normals: Box-Muller on Philox pairs; element e uses counter e>>1, lane e&1."""
    pairs = (n + 1) // 2
    r0, r1, r2, r3 = philox4x32(np.arange(pairs), step, chain, STREAM_MOMENTUM, key)
    u1, u2 = _u01(r0, r1), _u01(r2, r3)
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    z = np.empty(2 * pairs)
    z[0::2] = rad * np.cos(ang)
    z[1::2] = rad * np.sin(ang)
    return z[:n]


# --------------------------------------------------------------------------------------
# NUTS + dual averaging + annealed chain -- PARITY UNPINNED (see module docstring)
# --------------------------------------------------------------------------------------


def _logaddexp(a, b):
    if a == -np.inf and b == -np.inf:
        return -np.inf
    return float(np.logaddexp(a, b))


def _has_not_u_turn(rho, p_left, p_right) -> bool:
    return bool((np.dot(rho, p_left) > 0) and (np.dot(rho, p_right) > 0))


@dataclass
class NutsResult:
    q: np.ndarray
    L: float                 # UNtempered log posterior at q
    gL: np.ndarray           # UNtempered gradient at q
    log_accept_ratio: float
    leapfrogs: int
    depth: int
    is_accepted: bool
    reach_max_depth: bool
    has_divergence: bool
    energy: float
    target_log_prob: float   # tempered, at the temperature of this step


def nuts_one_step(q, cur_target, cur_grad, step_size, temp, fn_L, step, chain, key,
                  max_tree_depth=10, max_energy_diff=1000.0) -> NutsResult:
    """One NUTS transition, TFP-style (iterative doubling, multinomial, generalised U-turn).

    ``fn_L(q) -> (L, grad L)`` is the UNtempered log posterior; the target of this step is
    ``temp * L``.  ``cur_target/cur_grad`` are the cached (possibly one-step-stale, see
    ``sample_chain``) tempered values TFP would carry in ``previous_kernel_results``.
    Energies are TFP's ``target - 0.5 p.p`` (i.e. minus the Hamiltonian)."""
    dim = q.shape[0]
    p0 = rng_normal(dim, step, chain, key)
    init_energy = cur_target - 0.5 * np.dot(p0, p0)

    # (momentum, state, target, grad) at both ends; index 0 = left, 1 = right
    ends = [(p0, q, cur_target, cur_grad), (p0, q, cur_target, cur_grad)]
    cand = {"q": q, "L": None, "gL": None, "target": cur_target, "grad": cur_grad,
            "energy": init_energy, "weight": 0.0}
    momentum_sum = p0.copy()
    energy_diff_sum, leapfrog_count = 0.0, 0
    continue_tree, not_div, is_accepted = True, True, False
    depth, leaf_ctr = 0, 0
    mem_p = [None] * (max_tree_depth + 1)
    mem_rho = [None] * (max_tree_depth + 1)

    while depth < max_tree_depth and continue_tree:
        direction = rng_bit(depth, step, chain, STREAM_DIRECTION, key)
        eps = step_size if direction else -step_size
        p, x, tgt, grd = ends[1] if direction else ends[0]
        nsteps = 1 << depth
        sub = {"q": x, "L": None, "gL": None, "target": tgt, "grad": grd, "energy": tgt,
               "weight": -np.inf}
        cumsum = np.zeros(dim)
        e_sum, lf, cont, nd = 0.0, 0, True, not_div
        it = 0
        while it < nsteps and cont:
            # leapfrog (identity mass)
            p_half = p + 0.5 * eps * grd
            x = x + eps * p_half
            Lx, gLx = fn_L(x)
            tgt, grd = temp * Lx, temp * gLx
            p = p_half + 0.5 * eps * grd
            cumsum = cumsum + p
            lf += 1
            no_u = True
            if it % 2 == 0:
                slot = bin(it).count("1")
                mem_p[slot], mem_rho[slot] = p, cumsum
            else:
                k = 1
                while (it + 1) % (1 << k) == 0 and (1 << k) <= nsteps:
                    left = it + 1 - (1 << k)
                    slot = bin(left).count("1")
                    no_u = no_u and _has_not_u_turn(cumsum - mem_rho[slot], mem_p[slot], p)
                    k += 1
            energy = tgt - 0.5 * np.dot(p, p)
            if np.isnan(energy):
                energy = -np.inf
            ediff = energy - init_energy
            not_divergent = bool(-ediff < max_energy_diff)
            wsum = _logaddexp(sub["weight"], ediff)
            thresh = ediff - wsum
            u = math.log1p(-rng_uniform(leaf_ctr, step, chain, STREAM_LEAF, key))
            leaf_ctr += 1
            if u <= thresh:
                sub.update(q=x, L=Lx, gL=gLx, target=tgt, grad=grd, energy=energy)
            sub["weight"] = wsum
            cont_tree = not_divergent and cont
            cont = no_u and cont_tree
            nd = nd and not_divergent
            if cont_tree:
                e_sum += math.exp(min(ediff, 0.0))
            it += 1

        tree_weight = sub["weight"] if cont else -np.inf
        wsum = _logaddexp(tree_weight, cand["weight"])
        thresh = tree_weight - cand["weight"]
        u = math.log1p(-rng_uniform(depth, step, chain, STREAM_MERGE, key))
        choose = bool(u <= thresh) and cont
        if choose:
            cand.update(q=sub["q"], L=sub["L"], gL=sub["gL"], target=sub["target"],
                        grad=sub["grad"], energy=sub["energy"])
        cand["weight"] = wsum
        ends[1 if direction else 0] = (p, x, tgt, grd)
        momentum_sum = momentum_sum + cumsum
        no_u_traj = _has_not_u_turn(momentum_sum, ends[0][0], ends[1][0])
        is_accepted = is_accepted or choose
        energy_diff_sum += e_sum
        leapfrog_count += lf
        continue_tree = cont and no_u_traj
        not_div = nd
        depth += 1

    with np.errstate(divide="ignore", invalid="ignore"):
        lar = float(np.log(np.float64(energy_diff_sum) / np.float64(leapfrog_count)))
    return NutsResult(q=cand["q"], L=cand["L"], gL=cand["gL"], log_accept_ratio=lar,
                      leapfrogs=leapfrog_count, depth=depth, is_accepted=is_accepted,
                      reach_max_depth=continue_tree, has_divergence=not not_div,
                      energy=cand["energy"], target_log_prob=cand["target"])


def hmc_one_step(q, cur_target, cur_grad, step_size, temp, fn_L, step, chain, key, num_leapfrog, max_energy_diff=1000.0) -> NutsResult:
    """One fixed-length HMC transition (TFP ``HamiltonianMonteCarlo`` semantics: L leapfrogs forward,
    Metropolis test on the end state), same caching / tempering conventions as ``nuts_one_step``.
    Used for the fixed-L measurements of SURVEY 8d; shares the Philox streams with the HIP sampler."""
    dim = q.shape[0]
    p0 = rng_normal(dim, step, chain, key)
    init_energy = cur_target - 0.5 * np.dot(p0, p0)
    p, x, tgt, grd = p0, q, cur_target, cur_grad
    Lx, gLx = None, None
    for it in range(num_leapfrog):
        p_half = p + 0.5 * step_size * grd
        x = x + step_size * p_half
        Lx, gLx = fn_L(x)
        tgt, grd = temp * Lx, temp * gLx
        p = p_half + 0.5 * step_size * grd
    energy = tgt - 0.5 * np.dot(p, p)
    if np.isnan(energy):
        energy = -np.inf
    ediff = energy - init_energy
    not_divergent = bool(-ediff < max_energy_diff)
    u = math.log1p(-rng_uniform(0, step, chain, STREAM_MERGE, key))
    accept = bool(u <= ediff) and not_divergent
    lar = min(ediff, 0.0) if np.isfinite(ediff) or ediff == -np.inf else -np.inf
    if accept:
        return NutsResult(q=x, L=Lx, gL=gLx, log_accept_ratio=lar, leapfrogs=num_leapfrog, depth=1, is_accepted=True,
                          reach_max_depth=False, has_divergence=not not_divergent, energy=energy, target_log_prob=tgt)
    return NutsResult(q=q, L=None, gL=None, log_accept_ratio=lar, leapfrogs=num_leapfrog, depth=1, is_accepted=False,
                      reach_max_depth=False, has_divergence=not not_divergent, energy=init_energy, target_log_prob=cur_target)


@dataclass
class DualAveragingState:
    """TFP ``DualAveragingStepSizeAdaptationResults`` fields that evolve."""
    step_size: float                      # ``new_step_size``
    error_sum: float = 0.0
    log_averaging_step: float = 0.0
    step: int = 0
    log_shrinkage_target: float = 0.0     # log(10 * initial step size)


def dual_averaging_init(step_size: float) -> DualAveragingState:
    return DualAveragingState(step_size=step_size, log_shrinkage_target=math.log(10.0 * step_size))


def dual_averaging_update(da: DualAveragingState, log_accept_ratio: float, num_adaptation_steps: int,
                          target_accept_prob=0.75, exploration_shrinkage=0.05,
                          step_count_smoothing=10.0, decay_rate=0.75) -> DualAveragingState:
    """One ``DualAveragingStepSizeAdaptation.one_step`` update after the inner NUTS step."""
    lap = log_accept_ratio if np.isfinite(log_accept_ratio) else -np.inf
    lap = min(lap, 0.0)
    accept = math.exp(lap) if lap > -np.inf else 0.0
    prev_step = da.step
    t = float(prev_step + 1)
    new_error_sum = da.error_sum + target_accept_prob - accept
    soft_t = step_count_smoothing + t
    new_log_step = da.log_shrinkage_target - (new_error_sum * math.sqrt(t)) / (soft_t * exploration_shrinkage)
    eta = t ** (-decay_rate)
    new_log_avg = eta * new_log_step + (1.0 - eta) * da.log_averaging_step
    if prev_step < num_adaptation_steps:
        new_ss = math.exp(new_log_step)
    elif prev_step > num_adaptation_steps:
        new_ss = da.step_size
    else:
        new_ss = math.exp(new_log_avg)
    if prev_step > num_adaptation_steps:
        new_error_sum, new_log_avg = da.error_sum, da.log_averaging_step
    return DualAveragingState(step_size=new_ss, error_sum=new_error_sum, log_averaging_step=new_log_avg,
                              step=prev_step + 1, log_shrinkage_target=da.log_shrinkage_target)


def pack(X, sig_pre, th_pre):
    """Flat state vector: X component-major [D][N], then sigma_pre[D], theta_pre[P]."""
    return np.concatenate([X.T.reshape(-1), sig_pre, th_pre])


def unpack(q, N, D, P):
    X = q[: N * D].reshape(D, N).T
    return X, q[N * D: N * D + D], q[N * D + D: N * D + D + P]


def make_fn_L(pr: Problem):
    N, D, P = pr.N, pr.D, pr.P

    def fn(q):
        X, s, t = unpack(q, N, D, P)
        L, gX, gs, gt = logpost_grad(np.ascontiguousarray(X), s, t, 1.0, pr)
        return L, pack(gX, gs, gt)
    return fn


def sample_chain(pr: Problem, Xhat_init, sigma_sqs_init, thetas_init, num_results, num_burnin_steps,
                 seed: int, chain: int = 0, step_size: float = 0.1, target_accept_prob: float = 0.75,
                 max_tree_depth: int = 10, min_temp: float = 0.1, stale_cache: bool = True,
                 anneal: bool = True, trace: Optional[list] = None, hmc_leapfrogs: Optional[int] = None):
    """The reference's ``predict`` sampling loop (magi_v2.py:357-396) around ``LogAnnealedNUTS``
    (magi_v2.py:852-879): at chain step k the target is ``beta_temp(k) * L`` with
    ``beta_temp(k) = max(1/ln(k+2), min_temp)``; NUTS is wrapped in dual averaging for the first
    ``int(0.8*num_burnin_steps)`` steps.

    ``stale_cache=True`` keeps TFP's behaviour of reusing the previous step's cached
    target/gradient, which were computed at the previous temperature."""
    N, D, P = pr.N, pr.D, pr.P
    X0, s0, t0 = initial_state(Xhat_init, sigma_sqs_init, thetas_init, pr.LB)
    q = pack(X0, s0, t0)
    fn_L = make_fn_L(pr)
    L, gL = fn_L(q)
    num_adapt = int(0.8 * num_burnin_steps)
    da = dual_averaging_init(step_size)
    total = num_burnin_steps + num_results
    out_q = np.zeros((num_results, q.shape[0]))
    info = {k: [] for k in ("step_size", "log_accept_ratio", "leapfrogs", "depth", "has_divergence",
                            "reach_max_depth", "target_log_prob", "energy", "is_accepted", "beta_temp")}
    temp_prev = temperature(0, min_temp) if anneal else 1.0
    for k in range(total):
        temp = temperature(k, min_temp) if anneal else 1.0
        tc = temp_prev if stale_cache else temp
        if hmc_leapfrogs is None:
            res = nuts_one_step(q, tc * L, tc * gL, da.step_size, temp, fn_L, k, chain, seed,
                                max_tree_depth=max_tree_depth)
        else:
            res = hmc_one_step(q, tc * L, tc * gL, da.step_size, temp, fn_L, k, chain, seed, hmc_leapfrogs)
        ss_used = da.step_size
        if res.is_accepted:            # a new state was accepted somewhere in the tree
            q, L, gL = res.q, res.L, res.gL
            # cached values now correspond to `temp`
            temp_prev = temp
        else:
            # state unchanged: TFP's cache still holds the OLD tempered values
            temp_prev = tc
        da = dual_averaging_update(da, res.log_accept_ratio, num_adapt, target_accept_prob)
        if trace is not None:
            trace.append((k, res, ss_used))
        if k >= num_burnin_steps:
            i = k - num_burnin_steps
            out_q[i] = q
            info["step_size"].append(ss_used)
            info["log_accept_ratio"].append(res.log_accept_ratio)
            info["leapfrogs"].append(res.leapfrogs)
            info["depth"].append(res.depth)
            info["has_divergence"].append(res.has_divergence)
            info["reach_max_depth"].append(res.reach_max_depth)
            info["target_log_prob"].append(res.target_log_prob)
            info["energy"].append(res.energy)
            info["is_accepted"].append(res.is_accepted)
            info["beta_temp"].append(temp)
    X_samps = out_q[:, : N * D].reshape(num_results, D, N).transpose(0, 2, 1)
    sig_samps = out_q[:, N * D: N * D + D]
    th_samps = out_q[:, N * D + D:]
    return X_samps, sig_samps, th_samps, {k: np.asarray(v) for k, v in info.items()}, da
