"""ctypes binding of libmagi_hip.so (include/magi_hip.h) -- the only compute path of the package.

There is deliberately no CPU fallback: if the shared library or a GPU is missing every entry point
raises.  numpy arrays in, numpy arrays out; layouts follow the reference ([N, D] trajectories).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmagi_hip.so")

DRIFT_IDS = {"seir3": 0, "seir4": 1, "sirw": 2}
DRIFT_SHAPES = {"seir3": (3, 3), "seir4": (4, 3), "sirw": (4, 5)}   # name -> (D, P)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class MagiHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmagi_hip error {code}: {msg}")
        self.code = code


class SamplerCfg(C.Structure):
    """Mirror of ``magi_sampler_cfg``; defaults = the reference's (magi_v2.py:357-366)."""
    _fields_ = [("num_results", C.c_int32), ("num_burnin_steps", C.c_int32), ("num_adaptation_steps", C.c_int32),
                ("max_tree_depth", C.c_int32), ("mode", C.c_int32), ("hmc_leapfrogs", C.c_int32),
                ("anneal", C.c_int32), ("stale_cache", C.c_int32), ("step_size", C.c_double),
                ("target_accept_prob", C.c_double), ("max_energy_diff", C.c_double), ("min_temp", C.c_double)]


_SYMBOLS = {
    "magi_create": (C.c_void_p, [C.c_int]),
    "magi_destroy": (None, [C.c_void_p]),
    "magi_last_error": (C.c_char_p, [C.c_void_p]),
    "magi_version": (C.c_char_p, []),
    "magi_user_drift_info": (C.c_int, [_ip, _ip]),
    "magi_build_matrices": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, _dp, _dp, _dp]),
    "magi_matern_blocks": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp]),
    "magi_fit_hparams": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_double,
                                   C.c_double, _dp, _dp, _dp, _dp]),
    "magi_set_matrices": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]),
    "magi_theta_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double, _dp, _dp]),
    "magi_build_dense": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int, _ip, _dp, _dp, C.c_double]),
    "magi_pack_resident": (C.c_int, [C.c_void_p, C.c_int]),
    "magi_get_dense": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
    "magi_dense_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "magi_set_problem": (C.c_int, [C.c_void_p, _dp, _dp, _lp, _dp, C.c_int64, C.c_double, _dp, C.c_int, C.c_int]),
    "magi_logpost_grad": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]),
    "magi_logpost_grad_fused": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp, _dp]),
    "magi_sampler_cfg_default": (None, [C.POINTER(SamplerCfg)]),
    "magi_sampler_init": (C.c_int, [C.c_void_p, C.POINTER(SamplerCfg), C.c_int, _dp, _dp, _dp, C.c_uint64, _lp]),
    "magi_sampler_run": (C.c_int, [C.c_void_p, C.c_int, _lp, _dp]),
    "magi_sampler_steps_done": (C.c_int, [C.c_void_p, _lp]),
    "magi_sampler_get_samples": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "magi_sampler_get_diag": (C.c_int, [C.c_void_p, _dp, _dp, _ip, _ip, _ip, _ip, _ip, _dp, _dp, _dp]),
    "magi_sampler_get_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp]),
    "magi_sample": (C.c_int, [C.c_void_p, C.POINTER(SamplerCfg), C.c_int, _dp, _dp, _dp, C.c_uint64, _lp, _dp, _dp, _dp]),
    "magi_sampler_get_checkpoint": (C.c_int, [C.c_void_p, _dp]),
    "magi_sampler_set_checkpoint": (C.c_int, [C.c_void_p, _dp]),
    "magi_sampler_run_stats": (C.c_int, [C.c_void_p, _lp, _lp]),
    "magi_sampler_profile": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _lp]),
    "magi_time_gradient": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp]),
    "magi_gradient_bytes": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "magi_stream_kernel_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "magi_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "magi_debug_par": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "magi_build_profile": (C.c_int, [C.c_void_p, _dp, _dp, _lp]),
}

_libs = {}


def load_library(path: Optional[str] = None):
    """dlopen libmagi_hip.so (or a drift-specialised build of it, magi_v2_amd.jit) and declare every symbol of
    include/magi_hip.h.  Raises if absent."""
    p = os.path.abspath(path or os.environ.get("MAGI_HIP_LIB") or LIB_PATH)      # MAGI_HIP_LIB: A/B builds in one session
    if p in _libs:
        return _libs[p]
    if not os.path.exists(p):
        raise ImportError(f"{p} not found: build it with `python -m magi_v2_amd.build` (hipcc, gfx950); "
                          "magi_v2_amd has no CPU fallback")
    lib = C.CDLL(p)
    for name, (res, args) in _SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _libs[p] = lib
    return lib


def exported_symbols():
    return sorted(_SYMBOLS)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


@dataclass
class SamplerDiag:
    step_size: np.ndarray
    log_accept_ratio: np.ndarray
    leapfrogs_taken: np.ndarray
    tree_depth: np.ndarray
    has_divergence: np.ndarray
    reach_max_depth: np.ndarray
    is_accepted: np.ndarray
    target_log_prob: np.ndarray
    energy: np.ndarray
    beta_temp: np.ndarray


class MagiEngine:
    """One handle = one GPU.  Not thread-safe; different engines may be used from different threads."""

    def __init__(self, device_id: int = 0, drift=None):
        """``drift``: a magi_v2_amd.drift.Drift traced from a user f_vec -> the engine runs the library that
        magi_v2_amd.jit compiles for it; None / a built-in -> the base library."""
        self.user_drift = None
        if drift is not None and not getattr(drift, "is_builtin", True):
            from . import jit
            self._lib = load_library(jit.library_for(drift))
            self.user_drift = drift
        else:
            self._lib = load_library()
        self._h = self._lib.magi_create(int(device_id))
        if not self._h:
            raise MagiHipError(-2, self._lib.magi_last_error(None).decode())
        self.N = self.D = self.P = None
        self.n_chains = 0
        self._cfg = None

    # -- plumbing -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.magi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise MagiHipError(rc, self._lib.magi_last_error(self._h).decode())

    @staticmethod
    def version():
        return load_library().magi_version().decode()

    _FAMILIES = {"auto": 0, "mc": 1, "valu": 2}

    def set_option(self, name, value):
        """Tuning / test switch of this handle (include/magi_hip.h: magi_set_option); stream_family also takes "auto" / "mc" / "valu"."""
        if name == "stream_family" and isinstance(value, str):
            value = self._FAMILIES[value]
        self._check(self._lib.magi_set_option(self._h, name.encode(), int(value)))

    def stream_kernel_name(self, n_chains=1):
        buf = C.create_string_buffer(64)
        self._check(self._lib.magi_stream_kernel_name(self._h, int(n_chains), buf, 64))
        return buf.value.decode()

    # -- matrices -------------------------------------------------------------------------------
    def build_matrices(self, I, phi1s, phi2s, nu=2.01, bandsize=None, want_host=True):
        """magi_v2.py:774-823 + :126-128 + :271-274 on the GPU for all components."""
        I = _f64(np.asarray(I).reshape(-1))
        phi1s, phi2s = _f64(phi1s), _f64(phi2s)
        N, D = I.shape[0], phi1s.shape[0]
        b = -1 if bandsize is None else int(bandsize)
        outs = [np.empty((D, N, N)) for _ in range(3)] if want_host else [None, None, None]
        self._check(self._lib.magi_build_matrices(self._h, _ptr(I), N, D, _ptr(phi1s), _ptr(phi2s), float(nu), b,
                                                  *[_ptr(o) for o in outs]))
        self.N, self.D = N, D
        return tuple(outs) if want_host else None

    def build_dense(self, I, D, comps, phi1s, phi2s, nu=2.01):
        """Matrices of the listed components into the device-resident dense stacks of a D-component problem (no host copy,
        no packing): magi_v2.py:122-128 / 262-268 / 447-451."""
        I = _f64(np.asarray(I).reshape(-1))
        comps = np.ascontiguousarray(comps, dtype=np.int32)
        phi1s, phi2s = _f64(phi1s, comps.shape), _f64(phi2s, comps.shape)
        self._check(self._lib.magi_build_dense(self._h, _ptr(I), I.shape[0], int(D), comps.shape[0], comps.ctypes.data_as(_ip),
                                               _ptr(phi1s), _ptr(phi2s), float(nu)))
        self.N, self.D = I.shape[0], int(D)

    def pack_resident(self, bandsize=None):
        """Band mask (magi_v2.py:271-274) + packing of the resident dense stacks."""
        self._check(self._lib.magi_pack_resident(self._h, -1 if bandsize is None else int(bandsize)))

    def get_dense(self, bandsize=None):
        """Host copies (C_inv, m, K_inv), each [D, N, N], of the resident stacks with the band mask applied."""
        outs = [np.empty((self.D, self.N, self.N)) for _ in range(3)]
        self._check(self._lib.magi_get_dense(self._h, -1 if bandsize is None else int(bandsize), *[_ptr(o) for o in outs]))
        return tuple(outs)

    def dense_apply(self, which, V, transpose=False):
        """Y[d] = A_d V[d] (or A_d^T V[d]) for the resident stack ``which`` in ("C_inv", "m", "K_inv"); V [D, N] or [D, N, nv]."""
        V = _f64(V)
        vec = V.ndim == 2
        V3 = V[:, :, None] if vec else V
        V3 = _f64(V3, (self.D, self.N, V3.shape[2]))
        Y = np.empty_like(V3)
        self._check(self._lib.magi_dense_apply(self._h, ("C_inv", "m", "K_inv").index(which), int(bool(transpose)), V3.shape[2], _ptr(V3), _ptr(Y)))
        return Y[:, :, 0] if vec else Y

    def fit_hparams(self, I, X_filled, mu, mu_phi2, sd_phi2, sigma_sq_loc, phi1_init, phi2_init, sigma_sq_init, nu=2.01,
                    num_iters=1000, learning_rate=0.01, jitter=1e-6, want_trace=False):
        """magi_v2.py:538-691 on the GPU; returns dict(phi1s, phi2s, sigma_sqs[, loss])."""
        I = _f64(np.asarray(I).reshape(-1))
        X = _f64(X_filled)
        N, D = X.shape
        phi1, phi2, sig = _f64(phi1_init, (D,)).copy(), _f64(phi2_init, (D,)).copy(), _f64(sigma_sq_init, (D,)).copy()
        trace = np.zeros(max(num_iters, 1)) if want_trace else None
        self._check(self._lib.magi_fit_hparams(self._h, _ptr(I), N, D, _ptr(X), _ptr(_f64(mu, (D,))), _ptr(_f64(mu_phi2, (D,))),
                                               _ptr(_f64(sd_phi2, (D,))), _ptr(_f64(sigma_sq_loc, (D,))), float(nu), int(num_iters),
                                               float(learning_rate), float(jitter), _ptr(phi1), _ptr(phi2), _ptr(sig), _ptr(trace)))
        out = {"phi1s": phi1, "phi2s": phi2, "sigma_sqs": sig}
        if want_trace:
            out["loss"] = trace[:num_iters]
        return out

    def theta_init(self, drift, Xhat, mu, num_iters=10000, learning_rate=0.01, theta0=None, want_trace=False):
        """magi_v2.py:133-179 on the GPU (magi_theta_init): the whole Adam loop on the device-resident UNbanded matrices, any drift
        this library carries (``drift``: built-in name or the traced Drift the engine was created for).  Returns theta[P] (, losses)."""
        if isinstance(drift, str):
            P, drift_id = DRIFT_SHAPES[drift][1], DRIFT_IDS[drift]
        else:
            P, drift_id = drift.P, drift.device_id
        Xhat = _f64(Xhat, (self.N, self.D))
        th = np.ones(P) if theta0 is None else _f64(theta0, (P,)).copy()
        trace = np.zeros(max(num_iters, 1)) if want_trace else None
        self._check(self._lib.magi_theta_init(self._h, drift_id, P, _ptr(Xhat), _ptr(_f64(mu, (self.D,))), int(num_iters), float(learning_rate),
                                              _ptr(th), _ptr(trace)))
        return (th, trace[:num_iters]) if want_trace else th

    def matern_blocks(self, I, phi1, phi2, nu=2.01):
        I = _f64(np.asarray(I).reshape(-1))
        N = I.shape[0]
        outs = [np.empty((N, N)) for _ in range(3)]
        self._check(self._lib.magi_matern_blocks(self._h, _ptr(I), N, float(phi1), float(phi2), float(nu),
                                                 *[_ptr(o) for o in outs]))
        return tuple(outs)

    def set_matrices(self, C_inv, m, K_inv, bandsize=None):
        C_inv = _f64(C_inv)
        D, N, _ = C_inv.shape
        m, K_inv = _f64(m, (D, N, N)), _f64(K_inv, (D, N, N))
        b = -1 if bandsize is None else int(bandsize)
        self._check(self._lib.magi_set_matrices(self._h, N, D, b, _ptr(C_inv), _ptr(m), _ptr(K_inv)))
        self.N, self.D = N, D

    # -- problem --------------------------------------------------------------------------------
    def set_problem(self, mu, N_ds, obs_idx, y, beta, LB, drift):
        """``drift``: built-in name, or the Drift this engine was created for."""
        D = self.D
        if not isinstance(drift, str):
            if self.user_drift is None and not drift.is_builtin:
                raise ValueError("engine was not created for this user drift: MagiEngine(device, drift=...)")
            P, drift_id = drift.P, drift.device_id
        else:
            P, drift_id = DRIFT_SHAPES[drift][1], DRIFT_IDS[drift]
        mu, N_ds, LB = _f64(mu, (D,)), _f64(N_ds, (D,)), _f64(LB, (D,))
        obs_idx = np.ascontiguousarray(obs_idx, dtype=np.int64)
        y = _f64(y, obs_idx.shape)
        self._check(self._lib.magi_set_problem(self._h, _ptr(mu), _ptr(N_ds), obs_idx.ctypes.data_as(_lp), _ptr(y),
                                               obs_idx.shape[0], float(beta), _ptr(LB), drift_id, P))
        self.P = P

    def _states(self, X, sig_pre, th_pre):
        X = _f64(X)
        if X.ndim == 2:
            X, sig_pre, th_pre = X[None], np.asarray(sig_pre)[None], np.asarray(th_pre)[None]
        n = X.shape[0]
        X = _f64(X, (n, self.N, self.D))
        return n, X, _f64(sig_pre, (n, self.D)), _f64(th_pre, (n, self.P))

    def logpost_grad(self, X, sig_pre, th_pre, beta_temp=1.0, want_terms=False, fused=False):
        """unnormalized_log_prob (magi_v2.py:308-348) + gradient for one state or a batch of states.
        fused=True evaluates it through the sampler's single-phase kernels instead."""
        single = np.asarray(X).ndim == 2
        n, X, sp, tp = self._states(X, sig_pre, th_pre)
        logp, gX = np.empty(n), np.empty((n, self.N, self.D))
        gs, gt, terms = np.empty((n, self.D)), np.empty((n, self.P)), np.empty((n, 4))
        fn = self._lib.magi_logpost_grad_fused if fused else self._lib.magi_logpost_grad
        self._check(fn(self._h, n, _ptr(X), _ptr(sp), _ptr(tp), float(beta_temp), _ptr(logp),
                       _ptr(gX), _ptr(gs), _ptr(gt), _ptr(terms)))
        out = (logp[0], gX[0], gs[0], gt[0]) if single else (logp, gX, gs, gt)
        if want_terms:
            out = out + ((terms[0] if single else terms),)
        return out

    # -- sampler --------------------------------------------------------------------------------
    def default_cfg(self, **kw) -> SamplerCfg:
        cfg = SamplerCfg()
        self._lib.magi_sampler_cfg_default(C.byref(cfg))
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown sampler option {k}")
            setattr(cfg, k, v)
        return cfg

    def sampler_init(self, cfg: SamplerCfg, X0, sig_pre0, th_pre0, seed: int, chain_ids: Optional[Sequence[int]] = None):
        n, X, sp, tp = self._states(X0, sig_pre0, th_pre0)
        ids = None
        if chain_ids is not None:
            ids = np.ascontiguousarray(chain_ids, dtype=np.int64)
            if ids.shape != (n,):
                raise ValueError("chain_ids must have one entry per chain")
        self._check(self._lib.magi_sampler_init(self._h, C.byref(cfg), n, _ptr(X), _ptr(sp), _ptr(tp),
                                                C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                                None if ids is None else ids.ctypes.data_as(_lp)))
        self.n_chains = n
        self._cfg = cfg

    def sampler_run(self, n_steps: int):
        """Advance all chains by n_steps transitions; returns (leapfrogs taken, device ms)."""
        lf = C.c_int64(0)
        ms = C.c_double(0.0)
        self._check(self._lib.magi_sampler_run(self._h, int(n_steps), C.byref(lf), C.byref(ms)))
        return lf.value, ms.value

    def sampler_steps_done(self):
        out = np.zeros(self.n_chains, dtype=np.int64)
        self._check(self._lib.magi_sampler_steps_done(self._h, out.ctypes.data_as(_lp)))
        return out

    def sampler_samples(self):
        n, R = self.n_chains, self._cfg.num_results
        X = np.empty((n, R, self.N, self.D))
        sp, tp = np.empty((n, R, self.D)), np.empty((n, R, self.P))
        self._check(self._lib.magi_sampler_get_samples(self._h, _ptr(X), _ptr(sp), _ptr(tp)))
        return X, sp, tp

    def sampler_diag(self) -> SamplerDiag:
        n, T = self.n_chains, self._cfg.num_results + self._cfg.num_burnin_steps
        f = lambda: np.zeros((n, T))
        i = lambda: np.zeros((n, T), dtype=np.int32)
        d = SamplerDiag(f(), f(), i(), i(), i(), i(), i(), f(), f(), f())
        ip = lambda a: a.ctypes.data_as(_ip)
        self._check(self._lib.magi_sampler_get_diag(self._h, _ptr(d.step_size), _ptr(d.log_accept_ratio),
                                                    ip(d.leapfrogs_taken), ip(d.tree_depth), ip(d.has_divergence),
                                                    ip(d.reach_max_depth), ip(d.is_accepted), _ptr(d.target_log_prob),
                                                    _ptr(d.energy), _ptr(d.beta_temp)))
        return d

    def sampler_state(self):
        n = self.n_chains
        X, sp, tp = np.empty((n, self.N, self.D)), np.empty((n, self.D)), np.empty((n, self.P))
        ss, bc = np.empty(n), np.empty(n)
        self._check(self._lib.magi_sampler_get_state(self._h, _ptr(X), _ptr(sp), _ptr(tp), _ptr(ss), _ptr(bc)))
        return X, sp, tp, ss, bc

    CKPT_SCALARS = 16

    def sampler_checkpoint(self):
        """Everything needed to resume the chains at the transition boundary they stand at (between two sampler_run calls):
        dict(X, sig_pre, th_pre, scalars).  See include/magi_hip.h (magi_sampler_get_checkpoint)."""
        X, sp, tp, _, _ = self.sampler_state()
        sc = np.zeros((self.n_chains, self.CKPT_SCALARS))
        self._check(self._lib.magi_sampler_get_checkpoint(self._h, _ptr(sc)))
        return {"X": X, "sig_pre": sp, "th_pre": tp, "scalars": sc}

    def sampler_resume(self, cfg: SamplerCfg, ckpt, seed: int, chain_ids: Optional[Sequence[int]] = None):
        """sampler_init at the checkpointed states + the checkpoint's scalars: the next sampler_run continues the run."""
        self.sampler_init(cfg, ckpt["X"], ckpt["sig_pre"], ckpt["th_pre"], seed, chain_ids)
        sc = _f64(ckpt["scalars"], (self.n_chains, self.CKPT_SCALARS))
        self._check(self._lib.magi_sampler_set_checkpoint(self._h, _ptr(sc)))

    def sampler_run_stats(self):
        """(leapfrog slots issued, graph launches) of the last sampler_run."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.magi_sampler_run_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sampler_profile(self, n_slots=512):
        """Mean in-sampler device time (us) per launch of the streaming kernel and of k_point over n_slots slots launched with
        per-kernel HIP events; returns (stream_us, point_us, leapfrogs).  The sampler must be re-initialised afterwards."""
        a, b, lf = C.c_double(0.0), C.c_double(0.0), C.c_int64(0)
        self._check(self._lib.magi_sampler_profile(self._h, int(n_slots), C.byref(a), C.byref(b), C.byref(lf)))
        return a.value, b.value, lf.value

    # -- instrumentation ------------------------------------------------------------------------
    def time_gradient(self, n_chains=1, reps=50):
        total = C.c_double(0.0)
        ph = np.zeros(8)
        self._check(self._lib.magi_time_gradient(self._h, int(n_chains), int(reps), C.byref(total), _ptr(ph)))
        return total.value, ph

    BUILD_CLASSES = ("matern", "diag_chol_inv", "potrf_panel", "potrf_trailing_syrk", "trtri", "TtT", "m_K_products",
                     "single_phase_operators", "potrf_wall")

    def build_profile(self):
        """Per-class (flops, ms, calls) of the last build_matrices run with set_option("build_profile", 1); the last row, "potrf_wall", is
        filled by every build: its two Cholesky factorisations as a whole (look-ahead included when the build was not profiled)."""
        f, ms = np.zeros(16), np.zeros(16)
        calls = np.zeros(16, dtype=np.int64)
        n = self._lib.magi_build_profile(self._h, _ptr(f), _ptr(ms), calls.ctypes.data_as(_lp))
        return {self.BUILD_CLASSES[i]: (f[i], ms[i], int(calls[i])) for i in range(n)}

    def debug_par(self, chain=0):
        out = np.zeros(64)
        self._check(self._lib.magi_debug_par(self._h, int(chain), _ptr(out)))
        return out

    def gradient_bytes(self, n_chains=1):
        b = np.zeros(8)
        self._check(self._lib.magi_gradient_bytes(self._h, int(n_chains), _ptr(b)))
        return b
