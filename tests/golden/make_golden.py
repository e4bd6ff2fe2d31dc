#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

Nothing from the reference is written into this repository except DATA: inputs and the
numerical outputs of the reference's own TF-free functions.  The functions are loaded at
generation time by AST-extracting their ``FunctionDef``s from ``/root/reference/magi_v2.py``
(the module itself cannot be imported: tensorflow / tfp are not installed) and exec-ing
them in a scratch namespace with numpy/scipy/sklearn and a null ``tf.device`` stub.

Fixtures:
  G1  g1_build_matrices.npz   reference ``_build_matrices`` outputs (C, m, K) + its Kappa,
                              p_Kappa, Kappa_pp for N in {11, 41, 161}
  G2  g2_mpmath.npz           40-digit mpmath truth of Kappa/p_Kappa/Kappa_pp (+ m, K by
                              exact solve) for N in {11, 41}
  G3  g3_pipeline.npz         thinned SEIR rows + reference helper outputs
                              (_discretize, _linear_interpolate, cv_cubic_smoother)
  G3b seir_alpha_sweep.npz    thinned rows of the ten alpha-sweep CSVs (BASELINE config 4)
  G4  g4_logpost_*.npz        log-posterior known answers: op-for-op torch transcription of
                              ``unnormalized_log_prob`` + torch autograd gradients, on
                              matrices from the reference ``_build_matrices`` (dense and
                              band-masked b=80/b=20 -- the latter are SURVEY's G5)

Usage:  python tests/golden/make_golden.py
"""
import ast
import contextlib
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(OUT, "..", ".."))


def load_reference_functions():
    import scipy.special
    from scipy.interpolate import splev, splrep
    from sklearn.model_selection import KFold

    src = open(os.path.join(REF, "magi_v2.py")).read()
    tree = ast.parse(src)
    wanted = {"_build_matrices", "_discretize", "_linear_interpolate", "cv_cubic_smoother",
              "single_cv_cubic_smoother"}
    funcs = []
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "MAGI_v2":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in wanted:
                    funcs.append(item)
    mod = ast.Module(body=funcs, type_ignores=[])
    ast.fix_missing_locations(mod)

    class _TF:
        @staticmethod
        def device(_):
            return contextlib.nullcontext()

    ns = {"np": np, "kvp": scipy.special.kvp, "gamma": scipy.special.gamma, "KFold": KFold,
          "splrep": splrep, "splev": splev, "tf": _TF}
    exec(compile(mod, "<reference:magi_v2.py>", "exec"), ns)

    class Ref:
        pass

    ref = Ref()
    for name in wanted:
        setattr(Ref, name, ns[name])
    return ref


def capture_blocks(ref, I, phi1, phi2, v):
    """Run the reference ``_build_matrices`` while capturing Kappa / p_Kappa / Kappa_pp: they are
    exactly the first argument of its ``np.linalg.pinv`` call and the operands of its matmuls, so
    recover them algebraically from a traced pinv instead of touching reference code."""
    captured = {}
    real_pinv = np.linalg.pinv

    def spy(a, *args, **kw):
        captured["Kappa"] = np.array(a, copy=True)
        return real_pinv(a, *args, **kw)

    np.linalg.pinv = spy
    try:
        C, m, K = ref._build_matrices(I, phi1, phi2, v=v)
    finally:
        np.linalg.pinv = real_pinv
    return C, m, K, captured["Kappa"]


def thin_vignette(csv_path, cols, t_max=4.0, d_obs=20):
    """The vignette's thinning (vignette.ipynb cell 5): rows with t <= t_max, every k-th row."""
    import pandas as pd
    raw = pd.read_csv(csv_path).query(f"t <= {t_max}")
    obs = raw.iloc[::int((raw.index.shape[0] - 1) / (d_obs * t_max))]
    ts = obs.t.values.astype(np.float64)
    X = obs[cols].to_numpy().astype(np.float64)
    X[X < 0.0] = 0.0
    allcols = obs[["t", "S_obs", "E_obs", "I_obs", "R_obs", "S_true", "E_true", "I_true", "R_true"]].to_numpy()
    return ts, X, allcols


def g1(ref):
    out = {}
    cases = []
    for N, T in ((11, 0.25), (41, 1.0), (161, 4.0)):
        I = np.linspace(0.0, T, N).reshape(-1, 1)
        combos = [(0.03, 0.3), (0.0085, 0.375), (0.034, 0.1)] if N < 161 else [(0.03, 0.3)]
        for (p1, p2) in combos:
            C, m, K, Kappa = capture_blocks(ref, I, p1, p2, 2.01)
            tag = f"N{N}_p{p1}_{p2}"
            cases.append(tag)
            out[tag + "_I"] = I[:, 0]
            out[tag + "_phi"] = np.array([p1, p2, 2.01])
            out[tag + "_C"] = C
            out[tag + "_m"] = m
            out[tag + "_K"] = K
            out[tag + "_Kappa"] = Kappa
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "g1_build_matrices.npz"), **out)
    print("G1:", cases)


def g2():
    import mpmath as mp
    mp.mp.dps = 40
    out = {}
    cases = []
    for N, T in ((11, 0.25), (41, 1.0)):
        I = np.linspace(0.0, T, N)
        for (p1, p2) in [(0.03, 0.3), (0.034, 0.1)]:
            v = mp.mpf("2.01")
            phi1, phi2 = mp.mpf(repr(p1)), mp.mpf(repr(p2))
            c = mp.sqrt(2 * v) / phi2
            A = phi1 * mp.mpf(2) ** (1 - v) / mp.gamma(v)
            Kap = mp.zeros(N, N)
            pK = mp.zeros(N, N)
            Kpp = mp.zeros(N, N)
            for i in range(N):
                for j in range(N):
                    if i == j:
                        Kap[i, j] = phi1
                        pK[i, j] = 0
                        Kpp[i, j] = v * phi1 / (phi2 ** 2 * (v - 1))
                        continue
                    s, t = mp.mpf(float(I[i])), mp.mpf(float(I[j]))
                    u = c * abs(s - t)
                    sg = 1 if s > t else -1
                    Kap[i, j] = A * u ** v * mp.besselk(v, u)
                    pK[i, j] = -A * c * sg * u ** v * mp.besselk(v - 1, u)
                    Kpp[i, j] = A * c ** 2 * u ** (v - 1) * (mp.besselk(v - 1, u) - u * mp.besselk(v - 2, u))
            Kinv = Kap ** -1
            m = pK * Kinv
            Kd = Kpp - pK * Kinv * (-pK)
            tag = f"N{N}_p{p1}_{p2}"
            cases.append(tag)
            tonp = lambda M: np.array([[float(M[i, j]) for j in range(N)] for i in range(N)])
            out[tag + "_I"] = I
            out[tag + "_phi"] = np.array([p1, p2, 2.01])
            out[tag + "_Kappa"] = tonp(Kap)
            out[tag + "_pKappa"] = tonp(pK)
            out[tag + "_Kappapp"] = tonp(Kpp)
            out[tag + "_m"] = tonp(m)
            out[tag + "_K"] = tonp(Kd)
            out[tag + "_Cinv"] = tonp(Kinv)
            out[tag + "_Kinv"] = tonp(Kd ** -1)
            print("G2:", tag)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "g2_mpmath.npz"), **out)


def g3(ref):
    out = {}
    csv = os.path.join(REF, "data", "SEIR_seed=0.csv")
    for name, cols in (("seir3", ["E_obs", "I_obs", "R_obs"]), ("seir4", ["S_obs", "E_obs", "I_obs", "R_obs"])):
        ts, X, rows = thin_vignette(csv, cols)
        I, Xd = ref._discretize(ts, X, 1)
        Xi = ref._linear_interpolate(Xd)
        Xs = ref.cv_cubic_smoother(I, Xi)
        out[name + "_ts_obs"] = ts
        out[name + "_X_obs"] = X
        out[name + "_I"] = I
        out[name + "_X_obs_discret"] = Xd
        out[name + "_X_interp"] = Xi
        out[name + "_Xhat_smoothed"] = Xs
        out["rows"] = rows
        # a partially observed variant exercises the NaN branches of the helpers
        Xp = X.copy()
        Xp[1::3, 1] = np.nan
        I2, Xd2 = ref._discretize(ts, Xp, 2)
        out[name + "_partial_X_obs"] = Xp
        out[name + "_partial_I"] = I2
        out[name + "_partial_X_obs_discret"] = Xd2
        out[name + "_partial_X_interp"] = ref._linear_interpolate(Xd2)
    np.savez_compressed(os.path.join(OUT, "g3_pipeline.npz"), **out)
    print("G3 done")

    sweep = {}
    for alpha in ("0.05", "0.15"):
        for seed in range(5):
            p = os.path.join(REF, "data", f"SEIR_beta=6_gamma=0.6_sigma=1.8_alpha={alpha}_seed={seed}.csv")
            _, _, rows = thin_vignette(p, ["S_obs", "E_obs", "I_obs", "R_obs"])
            sweep[f"alpha={alpha}_seed={seed}"] = rows
    sweep["columns"] = np.array(["t", "S_obs", "E_obs", "I_obs", "R_obs", "S_true", "E_true", "I_true", "R_true"])
    np.savez_compressed(os.path.join(OUT, "seir_alpha_sweep.npz"), **sweep)
    print("G3b done")


# ---- G4: torch transcription of unnormalized_log_prob (magi_v2.py:308-348) ----------------------

def torch_f_seir3(t, X, thetas):
    import torch
    S = 1.0 - torch.reshape(torch.sum(X, dim=1), (-1, 1))
    return torch.cat([(thetas[0] * S * X[:, 1:2]) - (thetas[2] * X[:, 0:1]),
                      (thetas[2] * X[:, 0:1]) - (thetas[1] * X[:, 1:2]),
                      (thetas[1] * X[:, 1:2])], dim=1)


def torch_f_seir4(t, X, thetas):
    import torch
    S, E, I_, R = X[:, 0:1], X[:, 1:2], X[:, 2:3], X[:, 3:4]
    return torch.cat([-thetas[0] * S * I_, thetas[0] * S * I_ - thetas[2] * E,
                      thetas[2] * E - thetas[1] * I_, thetas[1] * I_], dim=1)


def torch_f_sirw(t, X, thetas):
    """SIRW drift of test_magi_script.py:19-45 (theta = beta, phi, xi, chi, kappa)."""
    import torch
    S, Inf, R, W = (X[:, k:k + 1] for k in range(4))
    b, ph, xi, ch, ka = (thetas[k] for k in range(5))
    infect, wane, boost = b * S * Inf, ka * W, ch * Inf * W
    return torch.cat([wane - infect, infect - ph * Inf, ph * Inf - xi * R + boost,
                      xi * R - boost - wane], dim=1)


def torch_logpost(X, sp, tp, beta_temp, c, f_vec):
    import torch
    sigma_sqs = torch.log(1.0 + torch.exp(sp)) + c["LB"]
    thetas = torch.log(1.0 + torch.exp(tp))
    lj_s = torch.sum(sp - torch.log(1.0 + torch.exp(sp)))
    lj_t = torch.sum(tp - torch.log(1.0 + torch.exp(tp)))
    N, D = X.shape
    X_cent = torch.reshape(X - c["mu"], (N, 1, D))
    t1 = torch.sum((X_cent.permute(2, 1, 0) @ c["C_inv"]) @ X_cent.permute(2, 0, 1))
    f_vals = f_vec(c["I"], X, thetas)[:, None].permute(2, 0, 1)
    toNorm = f_vals - (c["m"] @ X_cent.permute(2, 0, 1))
    t2 = torch.sum(toNorm.permute(0, 2, 1) @ (c["K_inv"] @ toNorm))
    t3 = torch.sum(c["N_ds"] * torch.log(2.0 * np.pi * sigma_sqs))
    X_observed = X.reshape(-1)[c["idx"]]
    t4 = torch.sum(torch.square(X_observed - c["y"]) * (1.0 / sigma_sqs)[c["cols"]])
    lp = beta_temp * (-0.5 * (((1.0 / c["beta"]) * (t1 + t2)) + (t3 + t4)) + lj_s + lj_t)
    return lp, (t1, t2, t3, t4)


def g4(ref):
    import torch
    from oracle import magi_oracle as orc   # host helpers only (hparams_initial); pinned by G3 tests

    torch.set_default_dtype(torch.float64)
    csv = os.path.join(REF, "data", "SEIR_seed=0.csv")
    rng = np.random.default_rng(20250103)
    configs = [
        ("seir3_N161", "seir3", ["E_obs", "I_obs", "R_obs"], torch_f_seir3, 1, 4.0, np.array([6.0, 0.6, 1.8])),
        ("seir4_N81", "seir4", ["S_obs", "E_obs", "I_obs", "R_obs"], torch_f_seir4, 0, 4.0, np.array([6.0, 0.6, 1.8])),
        ("sirw_N41", "sirw", ["S_obs", "E_obs", "I_obs", "R_obs"], torch_f_sirw, 0, 2.0, np.array([0.3, 0.1, 0.01, 0.1, 0.01])),
    ]
    for tag, drift, cols, f_vec, disc, tmax, theta in configs:
        ts, X, _ = thin_vignette(csv, cols, t_max=tmax)
        if drift == "seir4":
            # also exercise missing observations (NaNs) in the likelihood term
            X[5::7, 2] = np.nan
        I, Xd = ref._discretize(ts, X, disc)
        N, D = Xd.shape
        N_ds = (~np.isnan(X)).sum(axis=0)
        beta = (D * N) / N_ds.sum()
        idx = np.where(~np.isnan(Xd).flatten())[0]
        y = Xd.reshape(-1)[idx]
        Xi = ref._linear_interpolate(Xd)
        hp = orc.hparams_initial(Xi)
        mu = Xi.mean(axis=0)
        C_inv = np.zeros((D, N, N)); m = np.zeros((D, N, N)); K_inv = np.zeros((D, N, N))
        for d in range(D):
            C_d, m_d, K_d = ref._build_matrices(I, hp["phi1s"][d], hp["phi2s"][d], v=2.01)
            C_inv[d] = np.linalg.pinv(C_d)       # stands in for tf.linalg.pinv (magi_v2.py:126)
            m[d] = m_d
            K_inv[d] = np.linalg.pinv(K_d)       # magi_v2.py:128
        Xhat = ref.cv_cubic_smoother(I, Xi)
        LB = (Xhat.std(axis=0) * 0.01) ** 2
        out = dict(I=I[:, 0], mu=mu, C_inv=C_inv, m=m, K_inv=K_inv, N_ds=N_ds.astype(np.float64),
                   obs_idx=idx.astype(np.int64), y=y, beta=np.float64(beta), LB=LB,
                   phi1s=hp["phi1s"], phi2s=hp["phi2s"], sigma_sqs_init=hp["sigma_sqs"], Xhat_init=Xhat,
                   drift=np.array(drift), theta_true=theta)
        # states
        th_pre = np.log(np.exp(theta) - 1.0)
        states = [
            (Xhat, np.full(D, -5.0), th_pre),
            (Xhat + 0.01 * rng.standard_normal(Xhat.shape), rng.normal(-4.0, 1.0, D), th_pre + 0.1 * rng.standard_normal(len(theta))),
            (Xi + 0.05 * rng.standard_normal(Xhat.shape), rng.normal(-2.0, 1.0, D), rng.normal(0.0, 1.0, len(theta))),
        ]
        temps = [1.0 / np.log(2.0), 1.0, 0.1316]
        bands = [None, 80, 20]
        recs = {k: [] for k in ("band", "temp", "state", "logp", "terms")}
        gX, gs, gt = [], [], []
        for b in bands:
            if b is not None and b >= N - 1 and b != 80:
                continue
            Cb, mb, Kb = orc.band_part(C_inv, b), orc.band_part(m, b), orc.band_part(K_inv, b)
            c = dict(I=torch.tensor(I), mu=torch.tensor(mu), C_inv=torch.tensor(Cb), m=torch.tensor(mb),
                     K_inv=torch.tensor(Kb), N_ds=torch.tensor(N_ds.astype(np.float64)),
                     idx=torch.tensor(idx), cols=torch.tensor(idx % D), y=torch.tensor(y),
                     beta=float(beta), LB=torch.tensor(LB))
            for ti, temp in enumerate(temps):
                for si, (Xs, sp, tp) in enumerate(states):
                    Xt = torch.tensor(Xs, requires_grad=True)
                    spt = torch.tensor(sp, requires_grad=True)
                    tpt = torch.tensor(tp, requires_grad=True)
                    lp, terms = torch_logpost(Xt, spt, tpt, float(temp), c, f_vec)
                    lp.backward()
                    recs["band"].append(-1 if b is None else b)
                    recs["temp"].append(temp)
                    recs["state"].append(si)
                    recs["logp"].append(lp.item())
                    recs["terms"].append([t.item() for t in terms])
                    gX.append(Xt.grad.numpy().copy()); gs.append(spt.grad.numpy().copy()); gt.append(tpt.grad.numpy().copy())
        out.update(state_X=np.stack([s[0] for s in states]), state_sig_pre=np.stack([s[1] for s in states]),
                   state_th_pre=np.stack([s[2] for s in states]),
                   rec_band=np.array(recs["band"]), rec_temp=np.array(recs["temp"]), rec_state=np.array(recs["state"]),
                   rec_logp=np.array(recs["logp"]), rec_terms=np.array(recs["terms"]),
                   rec_gX=np.stack(gX), rec_gsig=np.stack(gs), rec_gth=np.stack(gt))
        np.savez_compressed(os.path.join(OUT, f"g4_logpost_{tag}.npz"), **out)
        print("G4:", tag, "N", N, "D", D, "records", len(recs["logp"]), "logp[0]", recs["logp"][0], "terms[1]", recs["terms"][1])


if __name__ == "__main__":
    ref = load_reference_functions()
    g1(ref)
    g2()
    g3(ref)
    g4(ref)
