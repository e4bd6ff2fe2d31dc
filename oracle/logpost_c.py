"""ctypes binding of oracle/logpost_c.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as oracle/magi_oracle.py: only tests/,
__graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module).

``logpost_grad`` has the signature of ``magi_oracle.logpost_grad`` plus a thread count; ``time_gradients`` is the timing loop of the
CPU baseline (the loop itself runs in C)."""
from __future__ import annotations

import ctypes
import os
import subprocess
import time
from typing import Dict, Sequence

import numpy as np

from .magi_oracle import DRIFT_IDS, DRIFTS, Problem

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmagi_oracle_c.so")
_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)


def library_path() -> str:
    return _LIB_PATH


def load(build_if_missing: bool = True):
    """The C oracle library; built with ``make -C oracle`` when absent and gcc is here."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        if not build_if_missing:
            raise FileNotFoundError(_LIB_PATH)
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(_LIB_PATH)
    common = [ctypes.c_int] * 4 + [_dp] * 5 + [_ip, _dp, ctypes.c_int64, ctypes.c_double, _dp, _dp, _dp, _dp, ctypes.c_double, ctypes.c_int]
    lib.magi_oracle_c_logpost_grad.argtypes = common + [_dp] * 5
    lib.magi_oracle_c_logpost_grad.restype = ctypes.c_int
    lib.magi_oracle_c_time_gradients.argtypes = [ctypes.c_int] + common + [_dp]
    lib.magi_oracle_c_time_gradients.restype = ctypes.c_int
    lib.magi_oracle_c_max_threads.restype = ctypes.c_int
    _lib = lib
    return lib


class _Args:
    """Contiguous fp64 / int64 copies of a Problem and a state, kept alive for the duration of the calls."""

    def __init__(self, pr: Problem, X, sig_pre, th_pre):
        c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self.N, self.D = int(pr.N), int(pr.D)
        self.P = int(DRIFTS[pr.drift][2])
        self.keep = [c(pr.C_inv), c(pr.m), c(pr.K_inv), c(pr.mu), c(pr.N_ds), np.ascontiguousarray(pr.obs_idx, dtype=np.int64), c(pr.y), c(pr.LB),
                     c(X), c(sig_pre), c(th_pre)]
        assert self.keep[0].shape == (self.D, self.N, self.N) and self.keep[8].shape == (self.N, self.D)
        assert self.keep[9].shape == (self.D,) and self.keep[10].shape == (self.P,)
        p = lambda a: a.ctypes.data_as(_ip if a.dtype == np.int64 else _dp)
        Ci, m, Ki, mu, Nds, idx, y, LB, Xc, sp, tp = self.keep
        self.head = (self.N, self.D, self.P, DRIFT_IDS[pr.drift], p(Ci), p(m), p(Ki), p(mu), p(Nds), p(idx), p(y), int(idx.size), float(pr.beta), p(LB),
                     p(Xc), p(sp), p(tp))


def logpost_grad(X, sig_pre, th_pre, beta_temp: float, pr: Problem, threads: int = 1):
    """(logp, (t1, t2, t3, t4), gX [N,D], gsig [D], gth [P]) from the C restatement."""
    lib = load()
    a = _Args(pr, X, sig_pre, th_pre)
    lp = ctypes.c_double()
    t4, gX, gs, gt = np.zeros(4), np.zeros((a.N, a.D)), np.zeros(a.D), np.zeros(a.P)
    rc = lib.magi_oracle_c_logpost_grad(*a.head, float(beta_temp), int(threads), ctypes.byref(lp), t4.ctypes.data_as(_dp), gX.ctypes.data_as(_dp),
                                        gs.ctypes.data_as(_dp), gt.ctypes.data_as(_dp))
    if rc:
        raise ValueError("magi_oracle_c_logpost_grad: bad argument")
    return lp.value, tuple(t4), gX, gs, gt


def max_threads() -> int:
    return int(load().magi_oracle_c_max_threads())


def time_gradients(pr: Problem, X, sig_pre, th_pre, threads: Sequence[int], min_evals: int = 200, max_seconds: float = 6.0) -> Dict[int, float]:
    """Gradient evaluations per second for each thread count: batches of evaluations (timed inside C) until ``min_evals`` of them or
    ``max_seconds`` have passed, after one warm-up batch."""
    lib = load()
    a = _Args(pr, X, sig_pre, th_pre)
    sink = ctypes.c_double()
    out = {}
    for nt in threads:
        run = lambda k: lib.magi_oracle_c_time_gradients(int(k), *a.head, 1.0, int(nt), ctypes.byref(sink))
        t0 = time.perf_counter()
        if run(1):
            raise ValueError("magi_oracle_c_time_gradients: bad argument")
        first = time.perf_counter() - t0
        if first > 0.5:          # (far more threads than the cores this process may use: OpenMP's barriers stall -- one evaluation is the measurement)
            out[int(nt)] = 1.0 / first
            continue
        run(2)
        n, batch, t0 = 0, 4, time.perf_counter()
        while True:
            run(batch)
            n += batch
            el = time.perf_counter() - t0
            if n >= min_evals or el > max_seconds:
                break
            batch = min(64, batch * 2)
        out[int(nt)] = n / el
    return out
