"""Generic-drift front-end (callers magi_v2.py:155, 206, 335): tracing, symbolic Jacobians, the
emitted kernel header and the specialised library -- everything that needs no GPU."""
import ctypes
import os

import numpy as np
import pytest

from magi_v2_amd import drift, host, jit
from magi_v2_amd.drift_examples import EXAMPLES, fitzhugh_nagumo
from magi_v2_amd.engine import exported_symbols
from oracle import magi_oracle as orc


def complex_step_jacobians(f_vec, X, th):
    """Independent of sympy: derivatives of an analytic function from Im f(x + ih) / h, exact to rounding."""
    n, D = X.shape
    P = len(th)
    h = 1e-30
    J = np.zeros((n, D, D)); T = np.zeros((n, D, P))
    for k in range(D):
        Xc = X.astype(complex); Xc[:, k] += 1j * h
        J[:, :, k] = np.imag(f_vec(None, Xc, th.astype(complex))) / h
    for p in range(P):
        tc = th.astype(complex); tc[p] += 1j * h
        T[:, :, p] = np.imag(f_vec(None, X.astype(complex), tc)) / h
    return J, T


@pytest.mark.parametrize("name", sorted(EXAMPLES))
def test_traced_drift_matches_the_callable_and_complex_step_jacobians(name):
    f_vec, D, P = EXAMPLES[name]
    d = drift.resolve(f_vec, D, P)
    assert not d.is_builtin and (d.D, d.P) == (D, P) and d.device_id == drift.USER_ID
    rng = np.random.default_rng(1)
    X, th = rng.uniform(-1.5, 1.5, (11, D)), rng.uniform(0.3, 2.5, P)
    np.testing.assert_allclose(d.f_np(None, X, th), f_vec(None, X, th), rtol=1e-14, atol=1e-15)
    J, T = d.jac_np(X, th)
    Jc, Tc = complex_step_jacobians(f_vec, X, th)
    np.testing.assert_allclose(J, Jc, rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(T, Tc, rtol=1e-13, atol=1e-14)
    assert "DriftT<MAGI_DRIFT_USER>" in d.header and f"#define MAGI_USER_D {D}" in d.header


def test_builtin_drifts_are_recognised_and_their_traced_jacobians_equal_the_oracle():
    rng = np.random.default_rng(2)
    for name, (fn, D, P) in orc.DRIFTS.items():
        d = drift.resolve(host.NUMPY_DRIFTS[name], D, P)
        assert d.is_builtin and d.name == name and d.header is None
        X, th = rng.uniform(0, 1, (9, D)), rng.uniform(0.1, 3, P)
        f, J, T = fn(X, th)
        Jd, Td = d.jac_np(X, th)
        np.testing.assert_allclose(Jd, J, rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(Td, T, rtol=1e-13, atol=1e-15)


def test_limits_and_untraceable_callables_fail_loudly():
    with pytest.raises(NotImplementedError, match="D <= 8"):
        drift.trace_drift(lambda t, X, th: X, 9, 2)
    with pytest.raises(NotImplementedError, match="explicit use of t"):
        drift.trace_drift(lambda t, X, th: X * th[0] + t, 2, 1)
    with pytest.raises(ValueError, match="shape"):
        drift.trace_drift(lambda t, X, th: X[:, 0:1] * th[0], 2, 1)


def test_specialised_library_builds_and_exports_the_full_abi():
    """hipcc cross-compiles the kernels for the traced drift (no GPU needed); the library exports every symbol of
    include/magi_hip.h and reports the user drift's shape."""
    d = drift.resolve(fitzhugh_nagumo, 2, 3)
    lib = ctypes.CDLL(jit.library_for(d))
    for sym in exported_symbols():
        assert hasattr(lib, sym), sym
    D, P = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.magi_user_drift_info(ctypes.byref(D), ctypes.byref(P)) == 1 and (D.value, P.value) == (2, 3)
    base = ctypes.CDLL(os.path.join(os.path.dirname(jit.__file__), "libmagi_hip.so"))
    assert base.magi_user_drift_info(None, None) == 0


def test_gradient_matching_objective_matches_oracle_and_finite_differences(golden_dir):
    """magi_v2.py:196-216 (initialisation of completely unobserved components): the product's loss equals the
    oracle's restatement and its analytic gradients equal central differences of that restatement."""
    from magi_v2_amd.api import MAGI_v2
    g = np.load(os.path.join(golden_dir, "g3_pipeline.npz"))
    X_obs = g["seir3_X_obs"].copy()
    X_obs[:, 0] = np.nan                                   # E never observed
    m = MAGI_v2(D_thetas=3, ts_obs=g["seir3_ts_obs"], X_obs=X_obs, bandsize=None, f_vec="seir3")
    m.I, _ = host.discretize(m.ts_obs, m.X_obs, 1)
    m.mag_I = m.I.shape[0]
    rng = np.random.default_rng(3)
    Xs = host.cubic_smoother(m.I, host.linear_interpolate(host.discretize(m.ts_obs, m.X_obs, 1)[1][:, m.observed_indicators]))
    Xu, th = rng.normal(0.05, 0.02, (m.mag_I, 1)), rng.uniform(0.5, 3, 3)
    loss, gX, gth = m.gradient_matching_loss_and_grads(Xs, Xu, th)
    ref = lambda Xu_, th_: orc.gradient_matching_loss(m.I, Xs, Xu_, th_, m.proper_order, "seir3")
    assert abs(loss - ref(Xu, th)) <= 1e-12 * abs(loss)
    for p in range(3):
        e = np.zeros(3); e[p] = 1e-6
        assert abs((ref(Xu, th + e) - ref(Xu, th - e)) / 2e-6 - gth[p]) <= 1e-6 * max(1.0, abs(gth[p]))
    for i in (0, 1, 40, m.mag_I - 2, m.mag_I - 1):
        E = np.zeros_like(Xu); E[i, 0] = 1e-6
        assert abs((ref(Xu + E, th) - ref(Xu - E, th)) / 2e-6 - gX[i, 0]) <= 1e-5 * max(1.0, abs(gX[i, 0]))


def test_transcendental_drifts_trace_through_numpy_ufuncs():
    """np.exp / np.sqrt / np.tanh on the tracing arrays (numpy calls x.exp() on object entries): a saturating-growth drift."""
    def f_vec(t, X, th):
        x, y = X[:, 0:1], X[:, 1:2]
        return np.concatenate([th[0] * x * np.exp(-th[1] * y) - np.sqrt(1.0 + x ** 2), np.tanh(th[2] * x) - y / (1.0 + y ** 2)], axis=1)
    d = drift.trace_drift(f_vec, 2, 3)
    rng = np.random.default_rng(4)
    X, th = rng.uniform(0.1, 1.5, (9, 2)), rng.uniform(0.3, 2.0, 3)
    np.testing.assert_allclose(d.f_np(None, X, th), f_vec(None, X, th), rtol=1e-14, atol=1e-15)
    J, T = d.jac_np(X, th)
    Jc, Tc = complex_step_jacobians(f_vec, X, th)
    np.testing.assert_allclose(J, Jc, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(T, Tc, rtol=1e-12, atol=1e-14)
    for tok in ("exp(", "sqrt(", "tanh("):
        assert tok in d.header
    assert os.path.exists(jit.library_for(d))              # the emitted header compiles for gfx950


def test_jit_library_for_is_safe_under_concurrent_callers(tmp_path, monkeypatch):
    """torchrun ranks / xdist workers constructing the same model all call jit.library_for for the same key.  Four
    processes race on an empty cache: exactly one compiles, all get the same finished library, no partial file is left."""
    import multiprocessing as mp
    import subprocess
    import sys
    cache = str(tmp_path / "jit")
    code = ("import os, sys; os.environ['MAGI_JIT_CACHE'] = sys.argv[1]; sys.path.insert(0, sys.argv[2]);"
            "from magi_v2_amd import drift, jit; from magi_v2_amd.drift_examples import EXAMPLES;"
            "f, D, P = EXAMPLES['lotka_volterra']; print(jit.library_for(drift.resolve(f, D, P)))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", code, cache, root], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(4)]
    outs = [p.communicate() for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-500:] for o in outs]
    libs = {o[0].decode().strip().splitlines()[-1] for o in outs}
    assert len(libs) == 1
    lib = libs.pop()
    assert os.path.exists(lib) and os.path.getsize(lib) > 100000
    left = [f for f in os.listdir(os.path.dirname(lib)) if f.startswith("build_") or f.endswith(".tmp")]
    assert left == [], left


def test_prebuilt_jit_cache_is_used_without_a_compiler_and_rebuilt_for_another_one(monkeypatch):
    """The cache key of a drift-specialised library is (drift header, library sources, extra flags) -- NOT the compiler, whose digest is
    in the library's FILE NAME: a box that receives a prebuilt jit_cache/ and has no hipcc uses what is there, MAGI_JIT_CACHE_TRUST=1
    vouches for it across compiler versions, a box with a DIFFERENT compiler builds its own file beside it (no rebuilding over each
    other), and a HIPCC that is set but cannot be run is an error, not "no compiler"."""
    from magi_v2_amd import drift, jit
    from magi_v2_amd.drift_examples import EXAMPLES
    f, D, P = EXAMPLES["lotka_volterra"]
    d = drift.resolve(f, D, P)
    lib = jit.library_for(d)                                # built (or found) by this container's compiler
    folder = os.path.dirname(lib)
    assert jit._compiler_version() != "" and os.path.basename(lib) == jit._lib_name(jit._compiler_version())
    stamp = os.path.getmtime(lib)
    assert jit.library_for(d) == lib and os.path.getmtime(lib) == stamp          # found again, not rebuilt
    # a box without any compiler: the prebuilt library is used as is
    monkeypatch.setattr(jit, "_compiler_version", lambda: "")
    monkeypatch.setattr(jit, "_no_compiler_here", lambda: True)
    assert jit.library_for(d) == lib and os.path.getmtime(lib) == stamp
    # HIPCC set but not runnable: a configuration error -- the cache is not trusted on its account
    monkeypatch.setattr(jit, "_no_compiler_here", lambda: False)
    assert jit._find_cached_library(folder) is None
    with pytest.raises(RuntimeError):
        jit.library_for(d)
    # another compiler: its own file name, so this one is not "usable" for it ...
    monkeypatch.setattr(jit, "_compiler_version", lambda: "some other hipcc 9.9")
    assert jit._lib_name("some other hipcc 9.9") != os.path.basename(lib) and jit._find_cached_library(folder) is None
    monkeypatch.setenv("MAGI_JIT_CACHE_TRUST", "1")
    assert jit._find_cached_library(folder) == lib          # ... unless the caller vouches for the cache
