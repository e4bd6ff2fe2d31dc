"""Shared helpers for the tests (oracle side only; product code never imports this)."""
import os

import numpy as np

from oracle import magi_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_g4(tag):
    return np.load(os.path.join(GOLDEN, f"g4_logpost_{tag}.npz"))


def problem_from_g4(g, band=None):
    return orc.Problem(I=g["I"], mu=g["mu"], C_inv=orc.band_part(g["C_inv"], band), m=orc.band_part(g["m"], band),
                       K_inv=orc.band_part(g["K_inv"], band), N_ds=g["N_ds"], obs_idx=g["obs_idx"], y=g["y"],
                       beta=float(g["beta"]), LB=g["LB"], drift=str(g["drift"]), P=len(g["theta_true"]))


def engine_for(pr, band=None, device=0, matrices=None):
    """A MagiEngine loaded with the oracle problem's constants (UNmasked matrices + bandsize:
    the engine applies the band mask itself, as the reference does after building)."""
    from magi_v2_amd.engine import MagiEngine
    eng = MagiEngine(device)
    C_inv, m, K_inv = matrices if matrices is not None else (pr.C_inv, pr.m, pr.K_inv)
    eng.set_matrices(C_inv, m, K_inv, bandsize=band)
    eng.set_problem(pr.mu, pr.N_ds, pr.obs_idx, pr.y, pr.beta, pr.LB, pr.drift)
    return eng


def synthetic_seir_problem(N, seed=0, dt=0.025, alpha=0.05, band=None, phi=None):
    """BASELINE configs 2/3/5: SEIR-4 truth by RK4 (beta=6, gamma=.6, sigma=1.8,
    x0=(.99,.01,0,0)), uniform grid dt, observations at even grid indices with noise
    N(0, (alpha*range_d)^2) from PCG64(seed).  Matrices come from the ORACLE build (tests only)."""
    th = np.array([6.0, 0.6, 1.8])

    def f(x):
        S, E, I, R = x
        return np.array([-th[0] * S * I, th[0] * S * I - th[2] * E, th[2] * E - th[1] * I, th[1] * I])

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
    I = np.arange(N) * dt
    rng = np.random.Generator(np.random.PCG64(seed))
    rngs = truth.max(axis=0) - truth.min(axis=0)
    X_obs = np.full((N, 4), np.nan)
    obs_rows = np.arange(0, N, 2)
    X_obs[obs_rows] = truth[obs_rows] + rng.normal(size=(len(obs_rows), 4)) * (alpha * rngs)
    X_obs[X_obs < 0.0] = 0.0
    return I, X_obs, truth, th
