"""Drop-in for the Python surface of the reference's ``magi_v2.MAGI_v2`` (magi_v2.py:20-462).

Same constructor, methods, public attributes and results dictionary; the hot path -- kernel
matrices, log posterior + gradient, NUTS -- runs in libmagi_hip.so on an MI355X.  There is no
CPU implementation of that path in this package: without the library or a GPU the methods raise.

What differs from the reference, deliberately:
  * ``f_vec`` is a built-in drift name or a numpy-compatible callable (TensorFlow is not a dependency).  A
    callable that equals a compiled-in drift uses the hand-written kernels; any other one is traced with
    sympy and the kernels are compiled for it (drift.py, jit.py; D <= 8, P <= 8).
  * hyper-parameters are fitted on the GPU (``magi_fit_hparams``: the reference's GP marginal
    likelihood + priors + Adam, magi_v2.py:538-691, restated -- TFP itself is not available, so this
    step is parity-unpinned); ``hparams=`` / ``hparam_iters=0`` bypass it.
  * ``predict`` takes keyword-only extras (n_chains, seed, ...); defaults reproduce the reference.
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Sequence, Union

import numpy as np

from . import drift as _drift
from . import host
from .engine import DRIFT_SHAPES, MagiEngine


def logarithmic_temperature_schedule(step, min_temp: float = 0.1):
    """magi_v2.py:833-835."""
    return np.maximum(1.0 / np.log(np.asarray(step, dtype=np.float64) + 2.0), min_temp)


def _band(A: np.ndarray, b: Optional[int]) -> np.ndarray:
    if b is None:
        return A
    n = A.shape[-1]
    i = np.arange(n)
    return A * (np.abs(i[:, None] - i[None, :]) <= b)


def _digest(a) -> bytes:
    """Full-content hash of a host matrix stack (an in-place edit anywhere is seen)."""
    import hashlib
    a = np.ascontiguousarray(a)
    return hashlib.blake2b(a.view(np.uint8).reshape(-1).data, digest_size=16).digest() + repr(a.shape).encode()


class MAGI_v2:
    """MAnifold-constrained Gaussian-process Inference on an MI355X (interface of magi_v2.py:20-73)."""

    def __init__(self, D_thetas: int, ts_obs: np.ndarray, X_obs: np.ndarray, bandsize: Union[int, None],
                 f_vec: Union[str, Callable], device: int = 0):
        self.D_thetas = D_thetas
        self.BANDSIZE = bandsize
        self.ts_obs = ts_obs
        self.X_obs = X_obs
        self.N, self.D = self.X_obs.shape

        # observed vs completely unobserved components (magi_v2.py:45-50)
        self.observed_indicators = (~np.isnan(X_obs)).mean(axis=0) > 0
        self.observed_components = np.arange(self.D)[self.observed_indicators]
        self.D_observed = len(self.observed_components)
        self.unobserved_components = np.setdiff1d(np.arange(self.D), self.observed_components)
        self.D_unobserved = len(self.unobserved_components)
        self.proper_order = np.argsort(np.concatenate([self.observed_components, self.unobserved_components]))
        self.N_ds = (~np.isnan(self.X_obs)).sum(axis=0)                       # magi_v2.py:53

        self.I, self.X_obs_discret = None, None
        self.beta, self.mag_I = None, None
        self.not_nan_idxs, self.not_nan_cols = None, None
        self.y_tau_ds_observed = None
        self.X_interp_obs, self.X_interp_unobs = None, None

        self.phi1s = np.full((self.D,), np.nan)
        self.phi2s = np.full((self.D,), np.nan)
        self.sigma_sqs_init = np.full((self.D,), np.nan)
        self.Xhat_init, self.thetas_init = None, None
        self.mu_ds = np.full((self.D,), np.nan)
        # C_d_invs / m_ds / K_d_invs (magi_v2.py:117-119): the authoritative copies live on the GPU; the attributes are
        # properties that download on first read and upload again only if the caller assigned or edited them
        self._host_mats = [None, None, None]
        self._host_digests = [None, None, None]
        self._assigned = [False, False, False]
        self._dev_valid = False          # the handle's dense stacks hold the current matrices
        self._band_applied = False       # magi_v2.py:271-274 has run (host copies are masked; the device masks when packing)
        self._packed_for = None          # bandsize the device operands were last packed with

        self.f_vec = f_vec
        self.drift = _drift.resolve(f_vec, self.D, D_thetas)       # built-in, or traced + JIT-compiled (drift.py, jit.py)
        self._device = device
        self._engine: Optional[MagiEngine] = None

    # ------------------------------------------------------------------------------------------
    @property
    def engine(self) -> MagiEngine:
        if self._engine is None:
            self._engine = MagiEngine(self._device, drift=self.drift)      # raises without libmagi_hip.so / GPU
        return self._engine

    # -- matrices: device-resident, lazily mirrored on the host ----------------------------------------------------
    def _materialise(self):
        if self._dev_valid and any(m is None for m in self._host_mats):
            got = self.engine.get_dense(self.BANDSIZE if self._band_applied else None)
            for k in range(3):
                if self._host_mats[k] is None:
                    self._host_mats[k] = got[k]
                    self._host_digests[k] = _digest(got[k])

    def _get_mat(self, k):
        self._materialise()
        return self._host_mats[k]

    def _set_mat(self, k, value):
        self._host_mats[k] = value
        self._assigned[k] = value is not None
        self._host_digests[k] = None

    C_d_invs = property(lambda self: self._get_mat(0), lambda self, v: self._set_mat(0, v))
    m_ds = property(lambda self: self._get_mat(1), lambda self, v: self._set_mat(1, v))
    K_d_invs = property(lambda self: self._get_mat(2), lambda self, v: self._set_mat(2, v))

    def _host_overrides(self) -> bool:
        """True when the caller assigned one of the three attributes or edited a downloaded copy in place."""
        return any(self._assigned) or any(m is not None and d is not None and _digest(m) != d
                                          for m, d in zip(self._host_mats, self._host_digests))

    def _build(self, comps: Sequence[int], phi1s, phi2s):
        """Eqn. 6 matrices for the listed components on the GPU (magi_v2.py:122-128, 262-268, 447-451): built into the
        handle's dense stacks, no host copy."""
        self.engine.build_dense(self.I, self.D, list(comps), phi1s, phi2s, 2.01)
        self._dev_valid = True
        self._host_mats, self._host_digests, self._assigned = [None] * 3, [None] * 3, [False] * 3
        self._packed_for = None

    def _apply_band(self):
        """magi_v2.py:271-274 / 459-462: the device applies the mask when it packs; host copies that exist are masked here."""
        self._band_applied = True
        if self.BANDSIZE is not None:
            for k in range(3):
                if self._host_mats[k] is not None:
                    keep = self._assigned[k]
                    self._host_mats[k] = _band(np.asarray(self._host_mats[k]), self.BANDSIZE)
                    self._assigned[k] = keep
                    if not keep:
                        self._host_digests[k] = _digest(self._host_mats[k])
        self._packed_for = None

    def _sync_matrices(self):
        if self._host_overrides() or not self._dev_valid:
            self._materialise()                    # (the ones the caller did not touch)
            assert all(m is not None for m in self._host_mats), "C_d_invs, m_ds and K_d_invs must be set (run initial_fit first)."
            mats = [np.asarray(m, dtype=np.float64) for m in self._host_mats]
            self.engine.set_matrices(*mats, bandsize=self.BANDSIZE)
            self._dev_valid = True
            self._assigned = [False] * 3
            self._host_digests = [_digest(m) for m in self._host_mats]
            self._packed_for = ("band", self.BANDSIZE)
        elif self._packed_for != ("band", self.BANDSIZE):
            self.engine.pack_resident(self.BANDSIZE)
            self._packed_for = ("band", self.BANDSIZE)

    def _dense_apply(self, which: str, V, transpose=False):
        """A_d V[d] (or A_d^T V[d]) with the current dense matrices: on the GPU while they are device-resident, with the
        caller's arrays once those were overwritten."""
        if self._dev_valid and not self._host_overrides():
            return self.engine.dense_apply(which, V, transpose)
        A = np.asarray({"C_inv": self.C_d_invs, "m": self.m_ds, "K_inv": self.K_d_invs}[which])
        V = np.asarray(V, dtype=np.float64)
        sub = "dji" if transpose else "dij"
        return np.einsum(f"{sub},dj->di", A, V) if V.ndim == 2 else np.einsum(f"{sub},djp->dip", A, V)

    # ------------------------------------------------------------------------------------------
    def _fit_kernel_hparams(self, I, X_filled, verbose=False, num_iters: int = 1000):
        """magi_v2.py:538-691 on the GPU (magi_fit_hparams): Adam x num_iters on the GP marginal likelihood +
        the reference's Fourier-informed priors, starting from its initial values."""
        pri = [host.fourier_phi2_prior(X_filled[:, d]) for d in range(X_filled.shape[1])]
        init = host.hparams_initial(X_filled)
        if verbose:
            print(f"Fitting hparams for {X_filled.shape[1]} components on the GPU ({num_iters} Adam steps) ...")
        return self.engine.fit_hparams(I, X_filled, X_filled.mean(axis=0), [p[0] for p in pri], [p[1] for p in pri],
                                       init["sigma_sqs"], init["phi1s"], init["phi2s"], init["sigma_sqs"], num_iters=num_iters)

    def initial_fit(self, discretization: int, verbose=False, hparams: Optional[dict] = None,
                    theta_init_iters: int = 10000, hparam_iters: int = 1000, hparam_fit_on: str = "grid",
                    init_seed: int = 0):
        """magi_v2.py:82-277.  Hyper-parameters are fitted as in the reference (1000 Adam steps on the
        linearly interpolated discretisation grid, magi_v2.py:105-106) unless ``hparams`` carries phi1s /
        phi2s / sigma_sqs for the observed components, or ``hparam_iters=0`` keeps the reference's starting
        values.  ``hparam_fit_on="observed"`` is a documented deviation: it fits on the observation times only.
        The inserted grid points are exact linear interpolants, which a GP marginal likelihood can only explain
        with a short length scale and near-zero noise; on the vignette data that optimum (phi2 ~ 0.1,
        sigma ~ 0.002) ruins the parameter recovery, while the fit on the observed rows lands at the true noise level
        (theta ~ (3.9, 0.47, 1.66) against the truth (6, 0.6, 1.8); with phi2 = 0.5 and that noise level
        (5.9, 0.56, 1.76)) -- see DESIGN.md section 8 and examples/vignette_seir.py.  Components that are never
        observed are initialised as in magi_v2.py:182-268; ``init_seed`` seeds the reference's unseeded start."""
        self.I, self.X_obs_discret = host.discretize(self.ts_obs, self.X_obs, discretization)
        self.mag_I = self.I.shape[0]
        N_ds, self.beta, idx, y = host.observation_bookkeeping(self.X_obs, self.X_obs_discret)
        self.not_nan_idxs = idx
        self.not_nan_cols = idx % self.D
        self.y_tau_ds_observed = y

        self.X_interp_obs = host.linear_interpolate(self.X_obs_discret[:, self.observed_indicators])
        if hparams is not None:
            hp = dict(host.hparams_initial(self.X_interp_obs))
            hp.update({k: np.asarray(v, dtype=np.float64) for k, v in hparams.items()})
        elif hparam_iters > 0 and hparam_fit_on == "observed":
            X_rows = host.linear_interpolate(self.X_obs[:, self.observed_indicators])
            hp = self._fit_kernel_hparams(np.asarray(self.ts_obs, dtype=np.float64), X_rows, verbose=verbose, num_iters=hparam_iters)
        elif hparam_iters > 0:
            if hparam_fit_on != "grid":
                raise ValueError("hparam_fit_on must be 'grid' (reference) or 'observed'")
            hp = self._fit_kernel_hparams(self.I, self.X_interp_obs, verbose=verbose, num_iters=hparam_iters)
        else:
            hp = dict(host.hparams_initial(self.X_interp_obs))
        self.phi1s[self.observed_indicators] = hp["phi1s"]
        self.phi2s[self.observed_indicators] = hp["phi2s"]
        self.sigma_sqs_init[self.observed_indicators] = hp["sigma_sqs"]
        self.Xhat_init = self.X_obs_discret.copy()
        self.Xhat_init[:, self.observed_indicators] = self.X_interp_obs
        self.mu_ds[self.observed_indicators] = self.X_interp_obs.mean(axis=0)

        self._band_applied = False
        self._build(self.observed_components, hp["phi1s"], hp["phi2s"])        # (components not built yet are zero matrices, magi_v2.py:117-119)

        if np.all(self.observed_indicators):
            self.thetas_init = self._fit_thetas_init(theta_init_iters)
        else:
            # magi_v2.py:182-268: (X_unobs, theta) jointly by finite-difference gradient matching with the observed
            # components fixed at their smoothed values, then hyper-parameters + matrices of the unobserved components
            self.X_interp_unobs, self.thetas_init = self._fit_unobserved(theta_init_iters, init_seed)
            if hparams is not None and "phi1s_unobs" in hparams:
                hpu = {"phi1s": np.asarray(hparams["phi1s_unobs"], dtype=np.float64),
                       "phi2s": np.asarray(hparams["phi2s_unobs"], dtype=np.float64),
                       "sigma_sqs": np.asarray(hparams["sigma_sqs_unobs"], dtype=np.float64)}
            elif hparam_iters > 0:
                hpu = self._fit_kernel_hparams(self.I, self.X_interp_unobs, verbose=verbose, num_iters=hparam_iters)
            else:
                hpu = dict(host.hparams_initial(self.X_interp_unobs))
            self.phi1s[self.unobserved_components] = hpu["phi1s"]
            self.phi2s[self.unobserved_components] = hpu["phi2s"]
            self.sigma_sqs_init[self.unobserved_components] = hpu["sigma_sqs"]
            self.Xhat_init[:, self.unobserved_components] = self.X_interp_unobs
            self.mu_ds[self.unobserved_components] = self.X_interp_unobs.mean(axis=0)
            self.engine.build_dense(self.I, self.D, list(self.unobserved_components), hpu["phi1s"], hpu["phi2s"], 2.01)
            self._packed_for = None

        self._apply_band()
        self.Xhat_init = host.cubic_smoother(self.I, self.Xhat_init)

    def _fit_thetas_init(self, iters: int) -> np.ndarray:
        """magi_v2.py:133-179: Adam(lr=.01) from theta = 1 on the t2 term, *including* the
        reference's reshape (magi_v2.py:155-156 reinterprets the [N, D] drift as [D, N] instead of
        transposing it).  Drifts that are linear in theta make the objective the quadratic
        theta^T A theta - 2 b^T theta + c, and Adam iterates on (A, b) exactly; otherwise every step evaluates
        the traced drift and its theta-Jacobian."""
        P, D, n = self.D_thetas, self.D, self.mag_I
        f_np, jac_np = self.drift.f_np, self.drift.jac_np
        Xc = (self.Xhat_init - self.mu_ds).T                                     # [D, n]
        bvec = self._dense_apply("m", Xc)                                        # m_d x_c   (the N x N products run on the GPU)
        rng = np.random.default_rng(0)
        probe = rng.uniform(0.3, 2.0, size=P)
        cols = [f_np(self.I, self.Xhat_init, np.eye(P)[p]) for p in range(P)]
        linear = (np.abs(f_np(self.I, self.Xhat_init, np.zeros(P))).max() == 0.0 and
                  np.allclose(f_np(self.I, self.Xhat_init, probe), sum(probe[p] * cols[p] for p in range(P)), rtol=1e-12, atol=1e-14))
        if linear:
            F = np.stack([c.reshape(D, n) for c in cols], axis=-1)               # [D, n, P]   (the reshape quirk)
            KF = self._dense_apply("K_inv", F)
            KTF = self._dense_apply("K_inv", F, transpose=True)
            A = np.einsum("dip,diq->pq", F, KF)
            g0 = np.einsum("dip,di->p", KF + KTF, bvec)                          # gradient offset
            grad_fn = lambda th: (A + A.T) @ th - g0
        elif self._dev_valid and not self._host_overrides():
            # the general branch stays on the GPU: drift values, theta-Jacobian, (K^-1 + K^-T) r and the Adam update of every step in one
            # captured graph, the host waits once (csrc/thetainit.hip); 10 000 steps at N = 161: a fraction of a second
            return self.engine.theta_init(self.drift, self.Xhat_init, self.mu_ds, iters)
        else:
            def grad_fn(th):
                fv = f_np(self.I, self.Xhat_init, th).reshape(D, n)              # the reshape quirk
                _, T = jac_np(self.Xhat_init, th)                                # [n, D, P]
                Tq = np.stack([T[:, :, p].reshape(D, n) for p in range(P)], axis=-1)   # same reshape as the drift
                r = fv - bvec
                g = self._dense_apply("K_inv", r) + self._dense_apply("K_inv", r, transpose=True)      # (K^-1 + K^-T) r on the GPU
                return np.einsum("dip,di->p", Tq, g)
        theta = np.ones(P)
        m = np.zeros(P); v = np.zeros(P)
        b1, b2, lr, eps = 0.9, 0.999, 0.01, 1e-7
        for t in range(1, iters + 1):
            grad = grad_fn(theta)
            m = b1 * m + (1 - b1) * grad
            v = b2 * v + (1 - b2) * grad * grad
            alpha = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            theta = theta - alpha * m / (np.sqrt(v) + eps)
        return theta

    def gradient_matching_loss_and_grads(self, X_obs_smoothed, X_unobs, thetas):
        """Objective of magi_v2.py:196-216 and its gradients w.r.t. (X_unobs, thetas): the L2 mismatch between the
        drift and the centred finite differences of the state on the interior grid points."""
        I = self.I
        X_full = np.concatenate([X_obs_smoothed, X_unobs], axis=1)[:, self.proper_order]
        f = self.drift.f_np(I, X_full, thetas)
        h2 = 2.0 * (I[1, 0] - I[0, 0])
        r = f[1:-1] - (X_full[2:] - X_full[:-2]) / h2                            # [n-2, D]
        J, T = self.drift.jac_np(X_full[1:-1], thetas)                           # [n-2, D, D], [n-2, D, P]
        gX = np.zeros_like(X_full)
        gX[1:-1] += 2.0 * np.einsum("nd,ndk->nk", r, J)
        gX[2:] -= 2.0 * r / h2
        gX[:-2] += 2.0 * r / h2
        gth = 2.0 * np.einsum("nd,ndp->p", r, T)
        return float((r ** 2).sum()), gX[:, self.unobserved_components], gth

    def _fit_unobserved(self, iters: int, seed: int):
        """magi_v2.py:182-247: Adam(lr=.01) x ``iters`` on (X_unobs, theta) jointly.  The reference draws the
        starting X_unobs from an unseeded numpy normal (magi_v2.py:223); here the generator is seeded."""
        Xs = host.cubic_smoother(self.I, self.X_interp_obs)
        mu0 = self.X_interp_obs.mean()
        sd0 = (self.X_interp_obs.std(axis=0) ** 2).mean() ** 0.5
        rng = np.random.Generator(np.random.PCG64(seed))
        Xu = rng.normal(loc=mu0, scale=sd0, size=(self.mag_I, self.D_unobserved))
        th = np.ones(self.D_thetas)
        mX, vX, mt, vt = np.zeros_like(Xu), np.zeros_like(Xu), np.zeros_like(th), np.zeros_like(th)
        b1, b2, lr, eps = 0.9, 0.999, 0.01, 1e-7
        for t in range(1, iters + 1):
            _, gX, gt = self.gradient_matching_loss_and_grads(Xs, Xu, th)
            alpha = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            mX = b1 * mX + (1 - b1) * gX; vX = b2 * vX + (1 - b2) * gX * gX
            mt = b1 * mt + (1 - b1) * gt; vt = b2 * vt + (1 - b2) * gt * gt
            Xu = Xu - alpha * mX / (np.sqrt(vX) + eps)
            th = th - alpha * mt / (np.sqrt(vt) + eps)
        return Xu, th

    # ------------------------------------------------------------------------------------------
    def predict(self, num_results: int = 1000, num_burnin_steps: int = 1000, sigma_sqs_LB=None, verbose=False, *,
                n_chains: int = 1, seed: Optional[int] = None, chain_ids: Optional[Sequence[int]] = None,
                stale_cache: bool = True, anneal: bool = True, max_tree_depth: int = 10, step_size: float = 0.1,
                family_chains: Optional[int] = None):
        """magi_v2.py:286-425.  Returns the reference's results dictionary; with n_chains > 1 every
        sample array gains a leading chain axis.  ``family_chains``: in a job sharded over GPUs, the largest per-GPU share
        (shard.family_chains_for) -- every rank then samples with the same kernel family whatever its own share."""
        assert ~np.any(np.isnan(self.Xhat_init)), "Please make sure Xhat_init does not have NaNs."
        assert ~np.any(np.isnan(self.sigma_sqs_init)), "Please make sure sigma_sqs_init does not have NaNs."
        assert ~np.any(np.isnan(self.thetas_init)), "Please make sure thetas_init does not have NaNs."

        if sigma_sqs_LB is None:
            sigma_sqs_LB = host.sigma_sqs_lower_bound(self.Xhat_init)
        sigma_sqs_LB = np.asarray(sigma_sqs_LB, dtype=np.float64)
        eng = self.engine
        self._sync_matrices()
        if family_chains is not None:
            eng.set_option("family_chains", int(family_chains))
        eng.set_problem(self.mu_ds, self.N_ds.astype(np.float64), np.asarray(self.not_nan_idxs),
                        np.asarray(self.y_tau_ds_observed), float(self.beta), sigma_sqs_LB, self.drift)
        sig_pre0, th_pre0 = host.softplus_inverse_inits(np.asarray(self.sigma_sqs_init, dtype=np.float64),
                                                        np.asarray(self.thetas_init, dtype=np.float64), sigma_sqs_LB)
        if seed is None:                 # the reference calls sample_chain unseeded (magi_v2.py:389-395)
            seed = int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).astype(np.uint64) @ np.array([1, 1 << 32], dtype=np.uint64))
        cfg = eng.default_cfg(num_results=num_results, num_burnin_steps=num_burnin_steps, stale_cache=int(stale_cache),
                              anneal=int(anneal), max_tree_depth=max_tree_depth, step_size=step_size)
        rep = lambda a: np.repeat(np.asarray(a, dtype=np.float64)[None], n_chains, axis=0)
        if verbose:
            print("Starting NUTS posterior sampling ...")
        start = time.time()
        eng.sampler_init(cfg, rep(self.Xhat_init), rep(sig_pre0), rep(th_pre0), seed=seed, chain_ids=chain_ids)
        eng.sampler_run(num_results + num_burnin_steps)
        X_samps, sig_pre, th_pre = eng.sampler_samples()
        end = time.time()
        minutes = np.round((end - start) / 60, 2)
        if verbose:
            print(f"Finished sampling in {minutes} minutes.")
        diag = eng.sampler_diag()
        sig_samps, th_samps = host.transform_samples(sig_pre, th_pre, sigma_sqs_LB)
        sq = (lambda a: a[0]) if n_chains == 1 else (lambda a: a)
        B = num_burnin_steps
        kernel_results = {k: sq(getattr(diag, k)[:, B:]) for k in
                          ("step_size", "log_accept_ratio", "leapfrogs_taken", "tree_depth", "has_divergence",
                           "reach_max_depth", "is_accepted", "target_log_prob", "energy", "beta_temp")}
        kernel_results["seed"] = seed
        return {"phi1s": self.phi1s, "phi2s": self.phi2s,
                "Xhat_init": self.Xhat_init,
                "sigma_sqs_init": self.sigma_sqs_init,
                "thetas_init": self.thetas_init,
                "I": self.I,
                "X_samps": sq(X_samps),
                "sigma_sqs_samps": sq(sig_samps),
                "thetas_samps": sq(th_samps),
                "kernel_results": kernel_results,
                "sample_results": [sq(X_samps), sq(sig_pre), sq(th_pre)],
                "minutes_elapsed": minutes}

    # ------------------------------------------------------------------------------------------
    def update_kernel_matrices(self, I_new, phi1s_new, phi2s_new):
        """magi_v2.py:433-462: new grid + hyper-parameters -> rebuild every component's matrices."""
        self.I = np.asarray(I_new, dtype=np.float64).reshape(-1, 1)
        self.phi1s, self.phi2s = np.array(phi1s_new, dtype=np.float64), np.array(phi2s_new, dtype=np.float64)
        self.mag_I = self.I.shape[0]
        self.beta = (self.D * self.mag_I) / self.N_ds.sum()
        self._band_applied = False
        self._build(range(self.D), self.phi1s, self.phi2s)
        self._apply_band()

    # reference helper names kept for callers that reach into them
    def _discretize(self, ts_obs, X_obs, discretization):
        return host.discretize(ts_obs, X_obs, discretization)

    def _linear_interpolate(self, X_partial):
        return host.linear_interpolate(X_partial)

    def cv_cubic_smoother(self, I, X_filled):
        return host.cubic_smoother(I, X_filled)
