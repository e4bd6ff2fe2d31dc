"""Host-side (numpy) pieces of the reference's pipeline that surround the GPU hot path.

These are O(N*D) preprocessing steps that run once per fit plus
the state transforms at the predict() boundary.  Each function cites the reference lines it
mirrors (relative to the reference tree).  Nothing here evaluates the log posterior, builds
kernel matrices or samples -- those exist only as HIP kernels (magi_v2_amd/csrc).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np
from scipy.interpolate import splev, splrep

# --------------------------------------------------------------------------------------------
# grid / interpolation / smoothing
# --------------------------------------------------------------------------------------------


def discretize(ts_obs: np.ndarray, X_obs: np.ndarray, discretization: int):
    """magi_v2.py:475-498 -- 2^k - 1 evenly spaced points between consecutive observations."""
    ts_obs = np.asarray(ts_obs, dtype=np.float64).flatten()
    assert ts_obs.shape[0] == X_obs.shape[0], \
        "Please make sure there are equal numbers of observations in ts_obs and X_obs."
    N, D = X_obs.shape
    stride = 2 ** discretization
    n_grid = stride * (N - 1) + 1
    I = np.full((n_grid,), np.nan)
    I[::stride] = ts_obs
    pos = np.arange(n_grid)
    known = ~np.isnan(I)
    I = np.interp(x=pos, xp=pos[known], fp=I[known]).reshape(-1, 1)
    X_grid = np.full((n_grid, D), np.nan)
    X_grid[::stride] = X_obs
    return I, X_grid


def linear_interpolate(X_partial: np.ndarray) -> np.ndarray:
    """magi_v2.py:509-527 -- per-column linear fill of NaNs (all-NaN columns stay NaN)."""
    out = X_partial.copy()
    pos = np.arange(X_partial.shape[0])
    for d in range(X_partial.shape[1]):
        missing = np.isnan(X_partial[:, d])
        if missing.any():
            out[:, d] = np.interp(x=pos, xp=pos[~missing], fp=X_partial[~missing, d])
    return out


def cubic_smoother(I: np.ndarray, X_filled: np.ndarray) -> np.ndarray:
    """magi_v2.py:695-770.  The reference cross-validates the knot count and then fits with the
    loop variable left over from that search, i.e. always len(I)//10 interior knots
    (magi_v2.py:747-757); the CV therefore has no effect on the result and is not repeated."""
    t = np.asarray(I, dtype=np.float64).flatten()
    if t.shape[0] < 10:
        return X_filled
    n_knots = t.shape[0] // 10
    knots = np.linspace(t[0], t[-1], n_knots + 2)[1:-1] if n_knots > 0 else np.array([])
    return np.stack([splev(t, splrep(t, X_filled[:, d], t=knots, s=0)) for d in range(X_filled.shape[1])], axis=1)


# --------------------------------------------------------------------------------------------
# hyper-parameter starting values
# --------------------------------------------------------------------------------------------


def fourier_phi2_prior(x: np.ndarray) -> Tuple[float, float]:
    """magi_v2.py:552-556 -- spectral-centroid prior mean / sd for the Matern length scale."""
    power = np.abs(np.fft.fft(x))
    power = power[1:(len(power) - 1) // 2 + 1] ** 2
    k = np.linspace(1, len(power), len(power))
    mean = 0.5 / (np.sum(k * power) / np.sum(power))
    return mean, (1 - mean) / 3


def hparams_initial(X_filled: np.ndarray) -> Dict[str, np.ndarray]:
    """The values the reference initialises its hyper-parameter fit with (magi_v2.py:631-639):
    phi1 = var, phi2 = Fourier prior mean, sigma^2 = (0.1 std)^2."""
    sd = X_filled.std(axis=0)
    return {"phi1s": sd ** 2,
            "phi2s": np.array([fourier_phi2_prior(X_filled[:, d])[0] for d in range(X_filled.shape[1])]),
            "sigma_sqs": (sd * 0.1) ** 2}


# --------------------------------------------------------------------------------------------
# predict() boundary
# --------------------------------------------------------------------------------------------


def sigma_sqs_lower_bound(Xhat_init: np.ndarray) -> np.ndarray:
    """magi_v2.py:299-300."""
    return (Xhat_init.std(axis=0) * 0.01) ** 2


def softplus_inverse_inits(sigma_sqs_init, thetas_init, LB):
    """magi_v2.py:374-380 (with the -5.0 fallback)."""
    sig_pre = np.full_like(sigma_sqs_init, -5.0)
    ok = sigma_sqs_init > LB
    sig_pre[ok] = np.log(np.exp((sigma_sqs_init - LB)[ok]) - 1.0)
    th_pre = np.full_like(thetas_init, -5.0)
    ok = thetas_init > 0.0
    th_pre[ok] = np.log(np.exp(thetas_init[ok]) - 1.0)
    return sig_pre, th_pre


def transform_samples(sig_pre, th_pre, LB):
    """magi_v2.py:418-419."""
    return np.log(np.exp(sig_pre) + 1.0) + LB, np.log(np.exp(th_pre) + 1.0)


def observation_bookkeeping(X_obs: np.ndarray, X_grid: np.ndarray):
    """magi_v2.py:53, 89, 96-100: N_ds, beta, flat indices and values of the non-NaN grid entries."""
    N_ds = (~np.isnan(X_obs)).sum(axis=0)
    n_grid, D = X_grid.shape
    beta = (D * n_grid) / N_ds.sum()
    idx = np.where(~np.isnan(X_grid).flatten())[0]
    return N_ds, beta, idx, X_grid.reshape(-1)[idx]


# --------------------------------------------------------------------------------------------
# drifts
# --------------------------------------------------------------------------------------------


def _np_seir3(t, X, th):
    S = 1.0 - X.sum(axis=1, keepdims=True)
    E, I_ = X[:, 0:1], X[:, 1:2]
    return np.concatenate([th[0] * S * I_ - th[2] * E, th[2] * E - th[1] * I_, th[1] * I_], axis=1)


def _np_seir4(t, X, th):
    S, E, I_ = X[:, 0:1], X[:, 1:2], X[:, 2:3]
    return np.concatenate([-th[0] * S * I_, th[0] * S * I_ - th[2] * E, th[2] * E - th[1] * I_, th[1] * I_], axis=1)


def _np_sirw(t, X, th):
    S, I_, R, W = (X[:, k:k + 1] for k in range(4))
    inf, wane, boost = th[0] * S * I_, th[4] * W, th[3] * I_ * W
    return np.concatenate([wane - inf, inf - th[1] * I_, th[1] * I_ - th[2] * R + boost, th[2] * R - boost - wane], axis=1)


NUMPY_DRIFTS: Dict[str, Callable] = {"seir3": _np_seir3, "seir4": _np_seir4, "sirw": _np_sirw}


def resolve_drift(f_vec, D: int, P: int) -> str:
    """Name of the drift ``f_vec`` resolves to (magi_v2.py:33, 73): a compiled-in one ("seir3", "seir4", "sirw")
    when the callable agrees with it numerically, else the name of the traced user drift (drift.resolve)."""
    from . import drift
    return drift.resolve(f_vec, D, P).name


# --------------------------------------------------------------------------------------------
# synthetic SEIR grids (BASELINE.json configs 2, 3, 5)
# --------------------------------------------------------------------------------------------


def synthetic_seir(N: int, seed: int = 0, dt: float = 0.025, alpha: float = 0.05):
    """SEIR-4 truth by RK4 (beta=6, gamma=.6, sigma=1.8, x0=(.99,.01,0,0) -- the generator of the
    reference's data/*.csv), uniform grid of spacing dt, observations at even grid indices with
    noise N(0, (alpha * range_d)^2) drawn from PCG64(seed), clipped at 0 as the vignette does.
    Returns (I[N], X_obs[N,4] with NaN at unobserved rows, truth[N,4], theta_true)."""
    th = np.array([6.0, 0.6, 1.8])

    def f(x):
        S, E, I_, _ = x
        return np.array([-th[0] * S * I_, th[0] * S * I_ - th[2] * E, th[2] * E - th[1] * I_, th[1] * I_])

    sub = 25
    h = dt / sub
    x = np.array([0.99, 0.01, 0.0, 0.0])
    truth = np.zeros((N, 4))
    truth[0] = x
    for i in range(1, N):
        for _ in range(sub):
            k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
            x = x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        truth[i] = x
    rng = np.random.Generator(np.random.PCG64(seed))
    span = truth.max(axis=0) - truth.min(axis=0)
    X_obs = np.full((N, 4), np.nan)
    rows = np.arange(0, N, 2)
    X_obs[rows] = truth[rows] + rng.normal(size=(len(rows), 4)) * (alpha * span)
    X_obs[X_obs < 0.0] = 0.0
    return np.arange(N) * dt, X_obs, truth, th
