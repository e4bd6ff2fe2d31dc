"""Drift front-end: turn the reference's ``f_vec`` argument (magi_v2.py:33, 73; called at :155, :206, :335)
into what the engine needs.

The reference traces an arbitrary Python callable ``f_vec(t[N,1], X[N,D], theta[P]) -> [N,D]`` into XLA and
lets autodiff produce the Jacobians.  Here the callable is traced once with sympy symbols in numpy object
arrays (slicing, arithmetic, ``np.sum / np.concatenate / np.reshape / np.stack`` and the unary ufuncs
``np.exp / log / sqrt / sin / cos / tanh`` all work on those), differentiated symbolically, and emitted twice:

* vectorised numpy evaluators of f, df/dx and df/dtheta (host-side initialisation: theta / unobserved-component
  fits, magi_v2.py:133-268), and
* a C++ header with ``DriftT<MAGI_DRIFT_USER>`` and the runtime-switch functions of ``magi_internal.h``; the
  sampler's kernels are compiled for it by ``magi_v2_amd.jit`` (hipcc, gfx950) into a specialised library.

Callables that agree with a compiled-in drift (SEIR-3 of vignette.ipynb cell 3, SEIR-4, SIRW of
test_magi_script.py:19-45) use the hand-written kernels of the base library.  Limits: D <= 8, P <= 8 (the
per-workgroup partial-sum layout; more than 4 components / 6 parameters use a build with wider lanes and blocks), autonomous systems (``t`` may be passed but must not be used), drifts
expressible with elementwise arithmetic and sympy-known functions."""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import numpy as np

MAX_D, MAX_P = 8, 8        # more than 4 components / 6 parameters: the specialised library is built with -DMAGI_MAX_D=8 / -DMAGI_MAX_P=8
BUILTIN_IDS = {"seir3": 0, "seir4": 1, "sirw": 2}
USER_ID = 3


def _sympy():
    try:
        import sympy
    except ImportError as exc:   # pragma: no cover - sympy ships with the image
        raise NotImplementedError("generic drifts need sympy for tracing; use a built-in drift name") from exc
    return sympy


@dataclass
class Drift:
    """A drift the engine can run: name, shape, device id, numpy evaluators and (user drifts) C++ source."""
    name: str
    D: int
    P: int
    device_id: int
    f_np: Callable            # (t, X[N,D], th[P]) -> [N,D]
    jac_np: Callable          # (X[N,D], th[P]) -> J[N,D,D] (d f_d / d x_k), T[N,D,P] (d f_d / d theta_p)
    header: Optional[str] = None
    exprs: List = field(default_factory=list, repr=False)

    @property
    def is_builtin(self) -> bool:
        return self.device_id != USER_ID


class _Tr:
    """Tracing scalar: a sympy expression that also answers the method calls numpy's unary ufuncs make on object arrays
    (``np.exp(X)`` calls ``x.exp()`` on every entry), so drifts with exp / log / sqrt / sin / cos / tanh trace like polynomial ones."""
    __slots__ = ("e",)

    def __init__(self, e):
        self.e = e

    @staticmethod
    def _u(o):
        return o.e if isinstance(o, _Tr) else o

    # (an ndarray operand is left to numpy, which then applies the operation entry by entry)
    def _bin(self, o, f):
        return NotImplemented if isinstance(o, np.ndarray) else _Tr(f(self.e, _Tr._u(o)))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a)
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b)
    def __rpow__(self, o): return self._bin(o, lambda a, b: b ** a)
    def __neg__(self): return _Tr(-self.e)
    def __pos__(self): return self

    def exp(self): return _Tr(_sympy().exp(self.e))
    def log(self): return _Tr(_sympy().log(self.e))
    def sqrt(self): return _Tr(_sympy().sqrt(self.e))
    def sin(self): return _Tr(_sympy().sin(self.e))
    def cos(self): return _Tr(_sympy().cos(self.e))
    def tanh(self): return _Tr(_sympy().tanh(self.e))
    def square(self): return _Tr(self.e ** 2)
    def reciprocal(self): return _Tr(1 / self.e)


def _trace(f_vec: Callable, D: int, P: int):
    sp = _sympy()
    xs = sp.symbols(f"x0:{D}", real=True)
    ths = sp.symbols(f"th0:{P}", real=True)
    tsym = sp.Symbol("t_magi", real=True)
    X = np.empty((1, D), dtype=object)
    for k in range(D):
        X[0, k] = _Tr(xs[k])
    th = np.empty((P,), dtype=object)
    for k in range(P):
        th[k] = _Tr(ths[k])
    t = np.empty((1, 1), dtype=object)
    t[0, 0] = _Tr(tsym)
    try:
        out = f_vec(t, X, th)
    except Exception as exc:
        raise NotImplementedError(
            "f_vec could not be traced: write it with numpy-compatible operations (slicing, + - * / **, np.sum, "
            "np.concatenate, np.reshape, np.stack, np.exp / log / sqrt / sin / cos / tanh) -- " + repr(exc)) from exc
    out = np.asarray(out, dtype=object)
    if out.shape != (1, D):
        raise ValueError(f"f_vec must return an array of shape [N, D]; traced shape {out.shape} for D = {D}")
    exprs = [sp.sympify(_Tr._u(out[0, d])) for d in range(D)]
    for e in exprs:
        if e.has(tsym):
            raise NotImplementedError("non-autonomous drifts (explicit use of t) are not supported")
        extra = e.free_symbols - set(xs) - set(ths)
        if extra:
            raise ValueError(f"f_vec produced unknown symbols {extra}")
    return sp, xs, ths, exprs


def _c_printer():
    sp = _sympy()
    from sympy.printing.c import C99CodePrinter

    class Printer(C99CodePrinter):
        def _print_Pow(self, expr):          # small integer powers as products (fp64 pow() is a long sequence)
            b, e = expr.as_base_exp()
            if e.is_Integer and 2 <= int(e) <= 4:
                s = self.parenthesize(b, 1000)
                return "(" + "*".join([s] * int(e)) + ")"
            if e.is_Integer and -4 <= int(e) <= -1:
                s = self.parenthesize(b, 1000)
                return "(1.0/(" + "*".join([s] * (-int(e))) + "))"
            return super()._print_Pow(expr)

        def _print_Rational(self, expr):
            return f"({int(expr.p)}.0/{int(expr.q)}.0)"

        def _print_Integer(self, expr):
            return f"{int(expr)}.0"

    return Printer()


def _emit_block(sp, printer, outputs: List[Tuple[str, object]], subs: dict, indent: str = "        ") -> str:
    """C statements computing ``lhs = expr`` for every output, with common subexpressions hoisted."""
    exprs = [e.xreplace(subs) for _, e in outputs]
    temps, reduced = sp.cse(exprs, symbols=sp.numbered_symbols("q_"), optimizations="basic")
    lines = [f"{indent}const double {printer.doprint(s)} = {printer.doprint(v)};" for s, v in temps]
    for (lhs, _), r in zip(outputs, reduced):
        lines.append(f"{indent}{lhs} {printer.doprint(r)};")
    return "\n".join(lines)


MAX_NB = 4        # basis functions per component the separable streaming path carries (more: the drift runs the general path)


def _separate(sp, xs, ths, exprs):
    """Separable form f_d(x, theta) = sum_k coef_{d,k}(theta) phi_{d,k}(x): per component a list of (coef, phi) sympy pairs, or None
    when some term mixes x and theta inseparably (e.g. V x / (K + x) with K a parameter) or a component needs more than MAX_NB
    basis functions.  Terms whose theta-dependent factors are equal up to a number share one basis function (c V - c V^3 / 3 + c R ->
    coef c, phi = V - V^3 / 3 + R).  The sampler's streaming kernel for such drifts (csrc/leap.hip, k_stream_sep) applies the
    operators to the theta-free phi's and the point phase combines the products with coef(theta)."""
    xset, tset = set(xs), set(ths)
    out = []
    for e in exprs:
        groups, order = {}, []
        for term in sp.Add.make_args(sp.expand(e)):
            if term == 0:
                continue
            coef, rest = term.as_independent(*xs, as_Add=False)
            if rest.free_symbols & tset or coef.free_symbols & xset:
                return None
            num, sym = coef.as_coeff_Mul()
            if sym not in groups:
                groups[sym] = 0
                order.append(sym)
            groups[sym] = groups[sym] + num * rest
        pairs = [(sym, sp.simplify(groups[sym])) for sym in order if sp.simplify(groups[sym]) != 0]
        if len(pairs) > MAX_NB:
            return None
        out.append(pairs)
    # numerical cross-check on random inputs: the separated form must reproduce f
    rng = np.random.default_rng(7)
    vals = {**{x: float(v) for x, v in zip(xs, rng.uniform(0.1, 0.9, len(xs)))}, **{t: float(v) for t, v in zip(ths, rng.uniform(0.3, 1.7, len(ths)))}}
    for e, pairs in zip(exprs, out):
        a = complex(sp.N(e.subs(vals)))
        b = complex(sp.N(sum((c * ph for c, ph in pairs), sp.Integer(0)).subs(vals)))
        if abs(a - b) > 1e-10 * max(1.0, abs(a)):
            return None
    return out


def _sep_members(sp, pr, xs, ths, exprs, D: int, P: int, subs: dict) -> str:
    """The DriftT<> members of the separable form (or SEP = false with inert members)."""
    pairs = _separate(sp, xs, ths, exprs)
    if pairs is None:
        return f"""    static constexpr bool SEP = false;
    static constexpr int NBMAX = 1;
    __host__ __device__ static constexpr int nbasis(int) {{ return 0; }}
    static __device__ __forceinline__ void basis(const double (&)[{D}], double (&ph)[{D}][1]) {{ for (int d = 0; d < {D}; ++d) ph[d][0] = 0.0; }}
    static __device__ __forceinline__ void coefs(const double (&)[{P}], double (&c)[{D}][1]) {{ for (int d = 0; d < {D}; ++d) c[d][0] = 0.0; }}"""
    nbmax = max(1, max(len(p_) for p_ in pairs))
    nb = " : ".join([f"d == {d} ? {len(pairs[d])}" for d in range(D - 1)] + [f"{len(pairs[D - 1])}"]) if D > 1 else f"{len(pairs[0])}"
    zero = sp.Integer(0)
    b_out = [(f"ph[{d}][{k}] =", pairs[d][k][1] if k < len(pairs[d]) else zero) for d in range(D) for k in range(nbmax)]
    c_out = [(f"c[{d}][{k}] =", pairs[d][k][0] if k < len(pairs[d]) else zero) for d in range(D) for k in range(nbmax)]
    b_body = _emit_block(sp, pr, b_out, subs)
    c_body = _emit_block(sp, pr, c_out, subs)
    return f"""    // separable form f_d = sum_k coef_(d,k)(theta) phi_(d,k)(x)  (magi_v2_amd.drift._separate)
    static constexpr bool SEP = true;
    static constexpr int NBMAX = {nbmax};
    __host__ __device__ static constexpr int nbasis(int d) {{ return {nb}; }}
    static __device__ __forceinline__ void basis(const double (&x)[{D}], double (&ph)[{D}][{nbmax}]) {{
{b_body}
    }}
    static __device__ __forceinline__ void coefs(const double (&th)[{P}], double (&c)[{D}][{nbmax}]) {{
{c_body}
    }}"""


def _header(sp, xs, ths, exprs, D: int, P: int, tag: str) -> str:
    pr = _c_printer()
    xa = {xs[k]: sp.Symbol(f"x[{k}]") for k in range(D)}
    ta = {ths[k]: sp.Symbol(f"th[{k}]") for k in range(P)}
    ga = [sp.Symbol(f"g[{d}]") for d in range(D)]
    subs = {**xa, **ta}
    J = [[sp.diff(exprs[d], xs[k]) for k in range(D)] for d in range(D)]
    T = [[sp.diff(exprs[d], ths[p]) for p in range(P)] for d in range(D)]
    cJ = [sum(ga[d] * J[d][k] for d in range(D)) for k in range(D)]          # (J^T g)_k
    cT = [sum(ga[d] * T[d][p] for d in range(D)) for p in range(P)]          # (T^T g)_p
    f_body = _emit_block(sp, pr, [(f"o[{d}] =", exprs[d]) for d in range(D)], subs)
    jt_body = _emit_block(sp, pr, [(f"c[{k}] =", cJ[k]) for k in range(D)] + [(f"t[{p}] +=", cT[p]) for p in range(P)], subs)
    sel = " : ".join([f"d == {d} ? o[{d}]" for d in range(D - 1)] + [f"o[{D - 1}]"]) if D > 1 else "o[0]"
    selc = " : ".join([f"d == {d} ? c[{d}]" for d in range(D - 1)] + [f"c[{D - 1}]"]) if D > 1 else "c[0]"
    sep_members = _sep_members(sp, pr, xs, ths, exprs, D, P, subs)
    return f"""// generated by magi_v2_amd.drift ({tag}) -- do not edit
#pragma once
#define MAGI_USER_D {D}
#define MAGI_USER_P {P}
template <> struct DriftT<MAGI_DRIFT_USER> {{
    static constexpr int D = {D}, P = {P};
    static __device__ __forceinline__ void f(const double (&x)[{D}], const double (&th)[{P}], double (&o)[{D}]) {{
{f_body}
    }}
    static __device__ __forceinline__ double f1(int d, const double (&x)[{D}], const double (&th)[{P}]) {{
        double o[{D}];
        f(x, th, o);
        return {sel};
    }}
    // c[k] = sum_d g[d] df_d/dx_k ; t[p] += sum_d g[d] df_d/dtheta_p
    static __device__ __forceinline__ void jt(const double (&x)[{D}], const double (&th)[{P}], const double (&g)[{D}], double (&c)[{D}], double (&t)[{P}]) {{
{jt_body}
    }}
{sep_members}
}};
// runtime-switch entry points of the reference-order (three-phase) kernels
__device__ __forceinline__ double user_drift_f(int d, const double* xp, const double* thp) {{
    double x[{D}], th[{P}];
    for (int k = 0; k < {D}; ++k) x[k] = xp[k];
    for (int k = 0; k < {P}; ++k) th[k] = thp[k];
    return DriftT<MAGI_DRIFT_USER>::f1(d, x, th);
}}
__device__ __forceinline__ void user_drift_jt(const double* xp, const double* thp, const double* gp, double* cout, double* tacc) {{
    double x[{D}], th[{P}], g[{D}], c[{D}], t[{P}];
    for (int k = 0; k < {D}; ++k) {{ x[k] = xp[k]; g[k] = gp[k]; }}
    for (int k = 0; k < {P}; ++k) {{ th[k] = thp[k]; t[k] = 0.0; }}
    DriftT<MAGI_DRIFT_USER>::jt(x, th, g, c, t);
    if (cout) for (int k = 0; k < {D}; ++k) cout[k] = c[k];
    if (tacc) for (int k = 0; k < {P}; ++k) tacc[k] += t[k];
}}
"""


def _numpy_evaluators(sp, xs, ths, exprs, D: int, P: int):
    J = [[sp.diff(exprs[d], xs[k]) for k in range(D)] for d in range(D)]
    T = [[sp.diff(exprs[d], ths[p]) for p in range(P)] for d in range(D)]
    args = list(xs) + list(ths)
    f_l = sp.lambdify(args, exprs, modules="numpy", cse=True)
    j_l = sp.lambdify(args, [J[d][k] for d in range(D) for k in range(D)], modules="numpy", cse=True)
    t_l = sp.lambdify(args, [T[d][p] for d in range(D) for p in range(P)], modules="numpy", cse=True)

    def cols(X, th):
        X = np.asarray(X, dtype=np.float64)
        return [X[:, k] for k in range(D)] + [float(v) for v in np.asarray(th, dtype=np.float64)]

    def f_np(t, X, th):
        a = cols(X, th)
        n = np.asarray(X).shape[0]
        return np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)) for v in f_l(*a)], axis=1)

    def jac_np(X, th):
        a = cols(X, th)
        n = np.asarray(X).shape[0]
        Jv = np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)) for v in j_l(*a)], axis=1).reshape(n, D, D)
        Tv = np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)) for v in t_l(*a)], axis=1).reshape(n, D, P)
        return Jv, Tv

    return f_np, jac_np


def trace_drift(f_vec: Callable, D: int, P: int, name: Optional[str] = None) -> Drift:
    """Trace a numpy-compatible ``f_vec`` into a user :class:`Drift` (device id MAGI_DRIFT_USER)."""
    if not (1 <= D <= MAX_D and 1 <= P <= MAX_P):
        raise NotImplementedError(f"generic drifts support D <= {MAX_D} components and P <= {MAX_P} parameters (got {D}, {P})")
    sp, xs, ths, exprs = _trace(f_vec, D, P)
    key = hashlib.sha256(("|".join(sp.srepr(e) for e in exprs) + f"|{D}|{P}").encode()).hexdigest()[:16]
    tag = name or f"user_{key}"
    f_np, jac_np = _numpy_evaluators(sp, xs, ths, exprs, D, P)
    return Drift(name=tag, D=D, P=P, device_id=USER_ID, f_np=f_np, jac_np=jac_np,
                 header=_header(sp, xs, ths, exprs, D, P, tag), exprs=exprs)


def builtin_drift(name: str) -> Drift:
    """A compiled-in drift; its numpy Jacobians come from tracing the host restatement (host.NUMPY_DRIFTS)."""
    from . import host
    from .engine import DRIFT_SHAPES
    D, P = DRIFT_SHAPES[name]
    sp, xs, ths, exprs = _trace(host.NUMPY_DRIFTS[name], D, P)
    _, jac_np = _numpy_evaluators(sp, xs, ths, exprs, D, P)
    return Drift(name=name, D=D, P=P, device_id=BUILTIN_IDS[name], f_np=host.NUMPY_DRIFTS[name], jac_np=jac_np, exprs=exprs)


def resolve(f_vec, D: int, P: int) -> Drift:
    """magi_v2.py:33: accept a built-in name, a callable that equals a built-in, or any traceable callable."""
    from . import host
    from .engine import DRIFT_SHAPES
    if isinstance(f_vec, Drift):
        return f_vec
    if isinstance(f_vec, str):
        if f_vec not in DRIFT_SHAPES:
            raise ValueError(f"unknown drift {f_vec!r}; built-ins: {sorted(DRIFT_SHAPES)}")
        if DRIFT_SHAPES[f_vec] != (D, P):
            raise ValueError(f"drift {f_vec!r} needs (D, P) = {DRIFT_SHAPES[f_vec]}, got {(D, P)}")
        return builtin_drift(f_vec)
    if not callable(f_vec):
        raise TypeError("f_vec must be a drift name or a callable")
    rng = np.random.default_rng(0)
    X = rng.uniform(0.05, 0.6, size=(7, D))
    th = rng.uniform(0.2, 2.0, size=(P,))
    t = np.linspace(0, 1, 7).reshape(-1, 1)
    try:
        got = np.asarray(f_vec(t, X, th), dtype=np.float64)
    except Exception as exc:   # e.g. a TensorFlow-only function
        raise NotImplementedError(
            "f_vec could not be evaluated on numpy arrays; pass a built-in drift name "
            f"({sorted(DRIFT_SHAPES)}) or a numpy-compatible callable") from exc
    if got.shape != (7, D):
        raise ValueError(f"f_vec must return [N, D]; got {got.shape}")
    for name, (d, p) in DRIFT_SHAPES.items():
        if (d, p) == (D, P) and np.allclose(got, host.NUMPY_DRIFTS[name](t, X, th), rtol=1e-12, atol=1e-14):
            return builtin_drift(name)
    return trace_drift(f_vec, D, P)
