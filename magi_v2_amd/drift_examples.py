"""Two classic MAGI benchmark systems written the way a user of the reference writes ``f_vec`` (numpy in place of
tf.*): used by the tests of the generic-drift path and pre-built by ``__graft_entry__.build()``."""
import numpy as np


def fitzhugh_nagumo(t, X, thetas):
    """FitzHugh-Nagumo: V' = c (V - V^3/3 + R),  R' = -(V - a + b R) / c;  theta = (a, b, c)."""
    V, R = X[:, 0:1], X[:, 1:2]
    a, b, c = thetas[0], thetas[1], thetas[2]
    return np.concatenate([c * (V - V ** 3 / 3.0 + R), -(V - a + b * R) / c], axis=1)


def lotka_volterra(t, X, thetas):
    """Predator-prey: x' = a x - b x y,  y' = d x y - c y;  theta = (a, b, c, d)."""
    x, y = X[:, 0:1], X[:, 1:2]
    return np.concatenate([thetas[0] * x - thetas[1] * x * y, thetas[3] * x * y - thetas[2] * y], axis=1)


def protein_transduction(t, X, thetas):
    """Protein signalling transduction (5 components S, dS, R, RS, Rpp; 6 parameters) -- the third benchmark of the MAGI paper."""
    S, dS, R, RS, Rpp = (X[:, k:k + 1] for k in range(5))
    k1, k2, k3, k4, V, Km = (thetas[k] for k in range(6))
    return np.concatenate([-k1 * S - k2 * S * R + k3 * RS,
                           k1 * S,
                           -k2 * S * R + k3 * RS + V * Rpp / (Km + Rpp),
                           k2 * S * R - k3 * RS - k4 * RS,
                           k4 * RS - V * Rpp / (Km + Rpp)], axis=1)


def competition_7(t, X, thetas):
    """Two-species competition with self-limitation and immigration: 7 parameters (exercises builds with more than 6)."""
    x, y = X[:, 0:1], X[:, 1:2]
    a, b, c, d, e, f, g = (thetas[k] for k in range(7))
    return np.concatenate([a * x - b * x * y - e * x ** 2, d * x * y - c * y - f * y ** 2 + g * x], axis=1)


EXAMPLES = {"fhn": (fitzhugh_nagumo, 2, 3), "lotka_volterra": (lotka_volterra, 2, 4), "ptrans": (protein_transduction, 5, 6),
            "competition7": (competition_7, 2, 7)}


def rk4(f_vec, x0, thetas, T, n, substeps=20):
    """Reference trajectory on a uniform grid of n points over [0, T] (test data only)."""
    x = np.asarray(x0, dtype=np.float64)
    out = [x.copy()]
    h = T / (n - 1) / substeps
    f = lambda v: f_vec(None, v[None], thetas)[0]
    for _ in range(n - 1):
        for _ in range(substeps):
            k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
            x = x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        out.append(x.copy())
    return np.linspace(0.0, T, n), np.array(out)
