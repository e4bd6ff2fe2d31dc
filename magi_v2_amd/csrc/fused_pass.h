// The pointwise / reduce half of one single-phase gradient evaluation, shared by the sampler's
// tail kernel and by the validation entry point magi_logpost_grad_fused.
//
// Inputs (per chain): V_Q (state), par (its transformed parameters) and the four mat-vec results of
// k_fused_*:  V_CX = FH xc,  V_R = FE xc,  V_ETF = FEt f,  V_KR = FK f, with
//     FH = Csym + m^T Ksym m,  FE = Ksym m,  FK = Ksym,  f = drift(X, theta),  xc = X - mu.
// Then (magi_v2.py:332-337 expanded):
//     t1 + t2          = sum  xc.(FH xc) - 2 f.(FE xc) + f.(FK f)
//     Ksym r           = FK f - FE xc                      (r = f - m xc)
//     d(t1+t2)/dx_d    = 2 (FH xc)_d - 2 (FEt f)_d + sum_d' 2 (Ksym r)_d' df_d'/dx_d
// Everything else (t3, t4, Jacobians of the softplus transforms, :318-323, :340-348) is as in the
// three-phase path.  One pass over the grid also completes the leapfrog's momentum half step,
// accumulates p.p, extends the subtree's momentum sum, writes the U-turn checkpoint and
// accumulates the first two checkpointed U-turn dot products, so a leaf costs ONE sweep over
// memory and ONE block reduction before its decisions can be taken.
#pragma once
#include "magi_internal.h"

// diagnostic build only (-DMAGI_TAIL_STAMPS): 100 MHz wall-clock stamps written to par[40..]
#ifdef MAGI_TAIL_STAMPS
#define MAGI_STAMP(par, i) do { if (threadIdx.x == 0) (par)[40 + (i)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MAGI_STAMP(par, i) do { } while (0)
#endif

struct LeafArgs {
    bool leaf = false;      // false: gradient only (bootstrap / API)
    double hs = 0.0;        // 0.5 * eps * beta_k  (signed)
    bool even = false;      // write the checkpoint of this leaf
    double* ckp_w = nullptr;
    double* ckr_w = nullptr;
    int nchk = 0;           // U-turn checks folded into the pass (0..2)
    const double *cp1 = nullptr, *cr1 = nullptr, *cp2 = nullptr, *cr2 = nullptr;
};

struct PassOut {
    double L, t12, t3, t4, pp;
    double dA1, dB1, dA2, dB2;
};

template <int DRIFT> struct FusedK { static constexpr int K = 1 + DriftT<DRIFT>::D + DriftT<DRIFT>::P + 5; };
constexpr int FP_KMAX = 1 + MAGI_MAX_D + MAGI_MAX_P + 5;   // t12, ss[D], tp[P], pp, dA1, dB1, dA2, dB2

// sh: (FP_KMAX + 1) * 16 doubles, shs: >= 8 doubles
template <int DRIFT>
__device__ __forceinline__ PassOut fused_pass(const DevProblem& pb, double* vb, double* par, double* sh, double* shs, const LeafArgs& la) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P, K = FusedK<DRIFT>::K;
    const int N = pb.N, ND = pb.ND, dimp = pb.dimp;
    const double* q = vb + (size_t)V_Q * dimp;
    const double* hx = vb + (size_t)V_CX * dimp;
    const double* ex = vb + (size_t)V_R * dimp;
    const double* kf = vb + (size_t)V_KR * dimp;
    const double* etf = vb + (size_t)V_ETF * dimp;
    double* g = vb + (size_t)V_G * dimp;
    double* p = vb + (size_t)V_P * dimp;
    double* rho = vb + (size_t)V_RHOSUB * dimp;

    // parameter entries (lanes j < D + P of wave 0): operands fetched early, used after the reduce
    const int j = threadIdx.x;
    const bool plane = j < D + P;
    double pj = 0.0, rj = 0.0, c1p = 0.0, c1r = 0.0, c2p = 0.0, c2r = 0.0;
    if (plane && la.leaf) {
        pj = p[ND + j];
        rj = rho[ND + j];
        if (la.nchk >= 1) { c1p = la.cp1[ND + j]; c1r = la.cr1[ND + j]; }
        if (la.nchk >= 2) { c2p = la.cp2[ND + j]; c2r = la.cr2[ND + j]; }
    }

    double th[P], s2[D];
#pragma unroll
    for (int k = 0; k < P; ++k) th[k] = par[PAR_TH + k];
#pragma unroll
    for (int d = 0; d < D; ++d) s2[d] = par[PAR_SIG2 + d];

    double t12 = 0.0, pp = 0.0, dA1 = 0.0, dB1 = 0.0, dA2 = 0.0, dB2 = 0.0, ss[D], tp[P];
#pragma unroll
    for (int d = 0; d < D; ++d) ss[d] = 0.0;
#pragma unroll
    for (int k = 0; k < P; ++k) tp[k] = 0.0;

    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        double x[D], f[D], g2[D], hv[D], tv[D], yv[D], jt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = q[d * N + i];
        DR::f(x, th, f);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int idx = d * N + i;
            const double e = ex[idx], k = kf[idx];
            hv[d] = hx[idx];
            tv[d] = etf[idx];
            yv[d] = pb.yobs[idx];
            g2[d] = 2.0 * (k - e);
            t12 += (x[d] - pb.mu[d]) * hv[d] + f[d] * (k - 2.0 * e);
            if (!isnan(yv[d])) { const double df = x[d] - yv[d]; ss[d] = fma(df, df, ss[d]); }
        }
        DR::jt(x, th, g2, jt, tp);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int idx = d * N + i;
            const double d12 = 2.0 * hv[d] - 2.0 * tv[d] + jt[d];
            const double d4 = isnan(yv[d]) ? 0.0 : 2.0 * (x[d] - yv[d]) / s2[d];
            const double gx = -0.5 * (pb.beta_inv * d12 + d4);
            g[idx] = gx;
            if (la.leaf) {
                const double pn = p[idx] + la.hs * gx;
                p[idx] = pn;
                const double rs = rho[idx] + pn;
                rho[idx] = rs;
                pp = fma(pn, pn, pp);
                if (la.even) { la.ckp_w[idx] = pn; la.ckr_w[idx] = rs; }
                if (la.nchk >= 1) { const double df = rs - la.cr1[idx]; dA1 = fma(df, la.cp1[idx], dA1); dB1 = fma(df, pn, dB1); }
                if (la.nchk >= 2) { const double df = rs - la.cr2[idx]; dA2 = fma(df, la.cp2[idx], dA2); dB2 = fma(df, pn, dB2); }
            }
        }
    }
    MAGI_STAMP(par, 2);
    double red[K];
    red[0] = t12;
#pragma unroll
    for (int d = 0; d < D; ++d) red[1 + d] = ss[d];
#pragma unroll
    for (int k = 0; k < P; ++k) red[1 + D + k] = tp[k];
    red[K - 5] = pp; red[K - 4] = dA1; red[K - 3] = dB1; red[K - 2] = dA2; red[K - 1] = dB2;
    block_sum<K>(red, sh);
    MAGI_STAMP(par, 3);

    if (threadIdx.x < 64) {
        double t3 = 0.0, t4 = 0.0, lj = 0.0, gj = 0.0;
        if (j < D) {
            const double sg = par[PAR_SGS + j];
            const double ssd = select_lane<K>(red, 1, D, j);
            const double nds = (j == 0) ? pb.N_ds[0] : (j == 1) ? pb.N_ds[1] : (j == 2) ? pb.N_ds[2] : pb.N_ds[3];
            const double sj = select_lane<D>(s2, 0, D, j);
            t3 = nds * par[PAR_LOG2PIS + j];
            t4 = ssd * (1.0 / sj);
            lj = par[PAR_LJS + j];
            gj = -0.5 * (nds / sj - ssd / (sj * sj)) * sg + (1.0 - sg);
        } else if (plane) {
            const double sg = par[PAR_SGT + (j - D)];
            const double tpp = select_lane<K>(red, 1 + D, P, j - D);
            lj = par[PAR_LJT + (j - D)];
            gj = -0.5 * pb.beta_inv * tpp * sg + (1.0 - sg);
        }
        double ppj = 0.0, a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0;
        if (plane) {
            g[ND + j] = gj;
            if (la.leaf) {
                const double pn = pj + la.hs * gj;
                const double rs = rj + pn;
                p[ND + j] = pn;
                rho[ND + j] = rs;
                ppj = pn * pn;
                if (la.even) { la.ckp_w[ND + j] = pn; la.ckr_w[ND + j] = rs; }
                if (la.nchk >= 1) { const double df = rs - c1r; a1 = df * c1p; b1 = df * pn; }
                if (la.nchk >= 2) { const double df = rs - c2r; a2 = df * c2p; b2 = df * pn; }
            }
        }
        // lanes 0..15 carry everything: row butterflies suffice (D + P <= 16)
        t3 = row16_sum(t3); t4 = row16_sum(t4); lj = row16_sum(lj);
        ppj = row16_sum(ppj); a1 = row16_sum(a1); b1 = row16_sum(b1); a2 = row16_sum(a2); b2 = row16_sum(b2);
        if (j == 0) {
            shs[0] = -0.5 * ((pb.beta_inv * red[0]) + (t3 + t4)) + lj;
            shs[1] = t3;
            shs[2] = t4;
            shs[3] = red[K - 5] + ppj;
            shs[4] = red[K - 4] + a1;
            shs[5] = red[K - 3] + b1;
            shs[6] = red[K - 2] + a2;
            shs[7] = red[K - 1] + b2;
        }
    }
    __syncthreads();
    PassOut o;
    o.L = shs[0]; o.t12 = red[0]; o.t3 = shs[1]; o.t4 = shs[2]; o.pp = shs[3];
    o.dA1 = shs[4]; o.dB1 = shs[5]; o.dA2 = shs[6]; o.dB2 = shs[7];
    __syncthreads();   // shs may be reused by the caller
    return o;
}
