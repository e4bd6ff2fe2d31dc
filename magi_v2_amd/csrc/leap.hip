// k_leap_*: the sampler's streaming kernel -- single-phase mat-vecs + the whole elementwise half of
// a leapfrog, on all CUs.
//
// Workgroup = LEAP_RI grid indices x D waves (wave = one component d of one grid index i).  Each wave
// streams its four operator rows (FH, FE, FE^T, FK; 16-B coalesced loads) against xc_d = X_d - mu_d
// and f_d = drift_d(X, theta), both formed on the fly from the state vector (L1/L2 hits), for up to
// NC chains that share every matrix byte.  After the row sums (DPP butterflies) the D waves of a
// grid index meet in LDS and ONE lane per (grid index, chain) finishes that grid point:
//     Ksym r = FK f - FE xc,   dL/dx_d = -1/2 (beta^-1 (2 FH xc - 2 FE^T f + J^T 2 Ksym r)_d + dt4/dx_d)
//     p_leaf = p_half + hs g,  rho_sub += p_leaf,  checkpoint,  U-turn partial dots,
//     speculative next leaf:  p_half' = p_leaf + hs g,  x' = x + eps p_half'   (other buffer)
// and leaves PART_K partial sums per workgroup for the tail.  The tail therefore moves O(#WG)
// bytes in the common case instead of ~20 state-sized vectors through one CU.
// (reference arithmetic: magi_v2.py:308-348 and the leapfrog of TFP's NoUTurnSampler)
#include "magi_internal.h"
#include "leap_reduce.h"

namespace {

constexpr int LEAP_RI = 2;

template <int DRIFT>
struct GridPoint {     // finishes component d of grid index i of one chain (one lane)
    using DR = DriftT<DRIFT>;
    static constexpr int D = DR::D, P = DR::P;

    static __device__ __forceinline__ void finish(const DevProblem& pb, const DevChains& ch, int cc, int i, int d, const double* res /* [D][4] */,
                                                  double* pk /* [PART_K] */) {
        const LeafPlan lp = ch.plan[cc];
#pragma unroll
        for (int k = 0; k < PART_K; ++k) pk[k] = 0.0;
        if (!lp.active) return;
        const int N = pb.N, dimp = pb.dimp;
        double* vb = ch.vec + vec_off(pb, cc, 0);
        const double* par = ch.par + (size_t)cc * PAR_COUNT;
        const double* q = vb + (size_t)(V_Q + lp.cur) * dimp;
        const int e = d * N + i;
        // this lane's own operands first (independent of the drift algebra)
        const double y = pb.yobs[e];
        const double* ph = vb + (size_t)(V_P + lp.cur) * dimp;
        double* rho = vb + (size_t)V_RHOSUB * dimp;
        double* ckp = vb + (size_t)V_CKP0 * dimp;
        double* ckr = vb + (size_t)V_CKRHO0 * dimp;
        double phe = 0.0, rhoe = 0.0, cpk[4] = {0.0, 0.0, 0.0, 0.0}, crk[4] = {0.0, 0.0, 0.0, 0.0};
        if (lp.leaf) {
            phe = ph[e];
            rhoe = rho[e];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < lp.nchk) { cpk[k] = ckp[(size_t)lp.chk_slot[k] * dimp + e]; crk[k] = ckr[(size_t)lp.chk_slot[k] * dimp + e]; }
        }
        double th[P], x[D], f[D], g2[D], jt[D], tp[P];
#pragma unroll
        for (int k = 0; k < P; ++k) { th[k] = par[PAR_TH + k]; tp[k] = 0.0; }
#pragma unroll
        for (int dd = 0; dd < D; ++dd) x[dd] = q[dd * N + i];
        DR::f(x, th, f);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) g2[dd] = 2.0 * (res[dd * 4 + 3] - res[dd * 4 + 1]);
        DR::jt(x, th, g2, jt, tp);
        // select this lane's component
        double xd = x[0], fd = f[0], jtd = jt[0], mud = pb.mu[0];
#pragma unroll
        for (int dd = 1; dd < D; ++dd) if (d == dd) { xd = x[dd]; fd = f[dd]; jtd = jt[dd]; mud = pb.mu[dd]; }
        const double hx = res[d * 4 + 0], ex = res[d * 4 + 1], etf = res[d * 4 + 2], kf = res[d * 4 + 3];
        pk[PK_T12] = (xd - mud) * hx + fd * (kf - 2.0 * ex);
        if (d == 0) {
#pragma unroll
            for (int k = 0; k < P; ++k) pk[PK_TP + k] = tp[k];
        }
        double d4 = 0.0;
        if (!isnan(y)) {
            const double df = xd - y;
            pk[PK_SS + d] = df * df;
            d4 = 2.0 * df / par[PAR_SIG2 + d];
        }
        const double gx = -0.5 * (pb.beta_inv * (2.0 * hx - 2.0 * etf + jtd) + d4);
        (vb + (size_t)V_G * dimp)[e] = gx;
        if (lp.leaf) {
            const double pn = phe + lp.hs * gx;
            (vb + (size_t)V_PLEAF * dimp)[e] = pn;
            const double rs = rhoe + pn;
            rho[e] = rs;
            pk[PK_PP] = pn * pn;
            if (lp.even) { ckp[(size_t)lp.ck_slot * dimp + e] = pn; ckr[(size_t)lp.ck_slot * dimp + e] = rs; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < lp.nchk) { const double df = rs - crk[k]; pk[PK_DOT + 2 * k] = df * cpk[k]; pk[PK_DOT + 2 * k + 1] = df * pn; }
            const double pnext = pn + lp.hs * gx;
            (vb + (size_t)(V_P + (lp.cur ^ 1)) * dimp)[e] = pnext;
            (vb + (size_t)(V_Q + (lp.cur ^ 1)) * dimp)[e] = xd + lp.eps * pnext;
        }
    }
};

// common tail of the dense and banded kernels: row sums are in (ah, ae, at, ak)[NC] of every lane
template <int NC, int DRIFT>
__device__ __forceinline__ void leap_epilogue(const DevProblem& pb, const DevChains& ch, int c0, int ri, int d, int lane,
                                              const double (&ah)[NC], const double (&ae)[NC], const double (&at)[NC], const double (&ak)[NC],
                                              double* res /* [RI][NC][D][4] */, double* redk /* [RI][NC][D][PART_K] */) {
    constexpr int D = DriftT<DRIFT>::D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double s0 = wave_sum(ah[c]), s1 = wave_sum(ae[c]), s2 = wave_sum(at[c]), s3 = wave_sum(ak[c]);
        if (lane == 0) {
            double* r = res + (((size_t)ri * NC + c) * D + d) * 4;
            r[0] = s0; r[1] = s1; r[2] = s2; r[3] = s3;
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < LEAP_RI * NC * D) {
        const int dd = t % D, rc = t / D;
        const int rr = rc / NC, c = rc - rr * NC;
        const int i = blockIdx.x * LEAP_RI + rr, cc = c0 + c;
        double* pk = redk + (size_t)t * PART_K;
        if (i < pb.N && cc < ch.n_chains) {
            GridPoint<DRIFT>::finish(pb, ch, cc, i, dd, res + ((size_t)rr * NC + c) * D * 4, pk);
        } else {
#pragma unroll
            for (int k = 0; k < PART_K; ++k) pk[k] = 0.0;
        }
    }
    __syncthreads();
    if (t < NC * PART_K) {
        const int c = t / PART_K, k = t - c * PART_K;
        if (c0 + c < ch.n_chains) {
            double s = 0.0;
#pragma unroll
            for (int rr = 0; rr < LEAP_RI; ++rr)
#pragma unroll
                for (int dd = 0; dd < D; ++dd) s += redk[(((size_t)rr * NC + c) * D + dd) * PART_K + k];
            ch.part[((size_t)(c0 + c) * PART_K + k) * ch.n_wg + blockIdx.x] = s;
        }
    }
}

// extra block (blockIdx.x == n_wg): the data-independent uniform draws of the leaf in flight, so the
// tail starts with them in memory instead of evaluating two fp64 log1p on its critical path
template <int NC>
__device__ __forceinline__ void leap_service(const DevChains& ch, int c0) {
    const int t = threadIdx.x;
    if (t < NC && c0 + t < ch.n_chains) {
        const LeafPlan lp = ch.plan[c0 + t];
        if (lp.active && lp.leaf) {
            double* par = ch.par + (size_t)(c0 + t) * PAR_COUNT;
            par[PAR_ULEAF] = m_log1p(-rng_uniform(lp.leaf_ctr, lp.step_k, lp.chain_id, STREAM_LEAF, lp.seed));
            par[PAR_UMERGE] = m_log1p(-rng_uniform(lp.depth, lp.step_k, lp.chain_id, STREAM_MERGE, lp.seed));
        }
    }
}

// ---- dense operators: grid (ceil(N / RI), ceil(n_chains / NC)), block 64 * D * RI -------------------
// FVEC: the drift of every chain's state has been stored by k_drift (V_F) -- used when several chains share
// the matrix stream, where re-evaluating the drift for every matrix row (N-fold redundant) would dominate
template <int NC, int DRIFT, bool FVEC>
__global__ __launch_bounds__(64 * DriftT<DRIFT>::D * LEAP_RI) void k_leap_dense(DevProblem pb, DevChains ch) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P;
    if (ch.gctl->all_done) return;
    if (blockIdx.x == ch.n_wg) { leap_service<NC>(ch, blockIdx.y * NC); return; }
    __shared__ double res[LEAP_RI * NC * D * 4];
    __shared__ double redk[LEAP_RI * NC * D * PART_K];
    const int N = pb.N, ld = pb.ldf;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ri = wave / D, d = wave - ri * D;
    const int i = blockIdx.x * LEAP_RI + ri, ii = min(i, N - 1);
    const int c0 = blockIdx.y * NC;
    const size_t ro = ((size_t)d * N + ii) * ld;
    const double2* ph = reinterpret_cast<const double2*>(pb.FH + ro);
    const double2* pe = reinterpret_cast<const double2*>(pb.FE + ro);
    const double2* pt = reinterpret_cast<const double2*>(pb.FEt + ro);
    const double2* pk = reinterpret_cast<const double2*>(pb.FK + ro);
    const double mud = pb.mu[d];

    const double* qc[NC];
    const double* fc[NC];
    double th[NC][P];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cc = min(c0 + c, ch.n_chains - 1);
        qc[c] = ch.vec + vec_off(pb, cc, V_Q + ch.plan[cc].cur);
        fc[c] = ch.vec + vec_off(pb, cc, V_F) + (size_t)d * N;
#pragma unroll
        for (int k = 0; k < P; ++k) th[c][k] = FVEC ? 0.0 : ch.par[(size_t)cc * PAR_COUNT + PAR_TH + k];
    }
    double ah[NC], ae[NC], at[NC], ak[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ah[c] = 0.0; ae[c] = 0.0; at[c] = 0.0; ak[c] = 0.0; }

    const int n2 = ld >> 1;
#pragma unroll 2
    for (int jj = lane; jj < n2; jj += 64) {
        const double2 h = ph[jj], e = pe[jj], t = pt[jj], k = pk[jj];
        const int j0 = 2 * jj, j1 = min(2 * jj + 1, N - 1);     // ld is N rounded up to even: the pad column is zero
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (FVEC) {
                const double xca = qc[c][d * N + j0] - mud, xcb = qc[c][d * N + j1] - mud;
                const double fa = fc[c][j0], fb = fc[c][j1];
                ah[c] = fma(h.x, xca, ah[c]); ah[c] = fma(h.y, xcb, ah[c]);
                ae[c] = fma(e.x, xca, ae[c]); ae[c] = fma(e.y, xcb, ae[c]);
                at[c] = fma(t.x, fa, at[c]); at[c] = fma(t.y, fb, at[c]);
                ak[c] = fma(k.x, fa, ak[c]); ak[c] = fma(k.y, fb, ak[c]);
                continue;
            }
            double xa[D], xb[D];
#pragma unroll
            for (int dd = 0; dd < D; ++dd) { xa[dd] = qc[c][dd * N + j0]; xb[dd] = qc[c][dd * N + j1]; }
            double xda = xa[0], xdb = xb[0];
#pragma unroll
            for (int dd = 1; dd < D; ++dd) if (d == dd) { xda = xa[dd]; xdb = xb[dd]; }
            const double fa = DR::f1(d, xa, th[c]), fb = DR::f1(d, xb, th[c]);
            const double xca = xda - mud, xcb = xdb - mud;
            ah[c] = fma(h.x, xca, ah[c]); ah[c] = fma(h.y, xcb, ah[c]);
            ae[c] = fma(e.x, xca, ae[c]); ae[c] = fma(e.y, xcb, ae[c]);
            at[c] = fma(t.x, fa, at[c]); at[c] = fma(t.y, fb, at[c]);
            ak[c] = fma(k.x, fa, ak[c]); ak[c] = fma(k.y, fb, ak[c]);
        }
    }
    leap_epilogue<NC, DRIFT>(pb, ch, c0, ri, d, lane, ah, ae, at, ak, res, redk);
}

// ---- banded operators: rows hold columns [i - bf, i + bf] ---------------------------------------------
template <int NC, int DRIFT, bool FVEC>
__global__ __launch_bounds__(64 * DriftT<DRIFT>::D * LEAP_RI) void k_leap_band(DevProblem pb, DevChains ch) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P;
    if (ch.gctl->all_done) return;
    if (blockIdx.x == ch.n_wg) { leap_service<NC>(ch, blockIdx.y * NC); return; }
    __shared__ double res[LEAP_RI * NC * D * 4];
    __shared__ double redk[LEAP_RI * NC * D * PART_K];
    const int N = pb.N, ld = pb.ldf, b = pb.bandf, W = 2 * b + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ri = wave / D, d = wave - ri * D;
    const int i = blockIdx.x * LEAP_RI + ri, ii = min(i, N - 1);
    const int c0 = blockIdx.y * NC;
    const size_t ro = ((size_t)d * N + ii) * ld;
    const double *ph = pb.FH + ro, *pe = pb.FE + ro, *pt = pb.FEt + ro, *pk = pb.FK + ro;
    const double mud = pb.mu[d];
    const double* qc[NC];
    const double* fc[NC];
    double th[NC][P];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cc = min(c0 + c, ch.n_chains - 1);
        qc[c] = ch.vec + vec_off(pb, cc, V_Q + ch.plan[cc].cur);
        fc[c] = ch.vec + vec_off(pb, cc, V_F) + (size_t)d * N;
#pragma unroll
        for (int k = 0; k < P; ++k) th[c][k] = FVEC ? 0.0 : ch.par[(size_t)cc * PAR_COUNT + PAR_TH + k];
    }
    double ah[NC], ae[NC], at[NC], ak[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { ah[c] = 0.0; ae[c] = 0.0; at[c] = 0.0; ak[c] = 0.0; }
#pragma unroll 2
    for (int kk = lane; kk < W; kk += 64) {
        const int j = ii - b + kk;
        const bool in = (j >= 0) && (j < N);
        const int jc = min(max(j, 0), N - 1);
        const double h = in ? ph[kk] : 0.0, e = in ? pe[kk] : 0.0, t = in ? pt[kk] : 0.0, k = in ? pk[kk] : 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (FVEC) {
                const double xc = qc[c][d * N + jc] - mud, fa = fc[c][jc];
                ah[c] = fma(h, xc, ah[c]); ae[c] = fma(e, xc, ae[c]);
                at[c] = fma(t, fa, at[c]); ak[c] = fma(k, fa, ak[c]);
                continue;
            }
            double xa[D];
#pragma unroll
            for (int dd = 0; dd < D; ++dd) xa[dd] = qc[c][dd * N + jc];
            double xd = xa[0];
#pragma unroll
            for (int dd = 1; dd < D; ++dd) if (d == dd) xd = xa[dd];
            const double fa = DR::f1(d, xa, th[c]);
            const double xc = xd - mud;
            ah[c] = fma(h, xc, ah[c]); ae[c] = fma(e, xc, ae[c]);
            at[c] = fma(t, fa, at[c]); ak[c] = fma(k, fa, ak[c]);
        }
    }
    leap_epilogue<NC, DRIFT>(pb, ch, c0, ri, d, lane, ah, ae, at, ak, res, redk);
}

// f(X, theta) of every active chain's evaluated state -> V_F (only launched when n_chains > 1)
template <int DRIFT>
__global__ __launch_bounds__(256) void k_drift(DevProblem pb, DevChains ch) {
    using DR = DriftT<DRIFT>;
    if (ch.gctl->all_done) return;
    const int c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const LeafPlan lp = ch.plan[c];
    if (!lp.active || i >= pb.N) return;
    const double* q = ch.vec + vec_off(pb, c, V_Q + lp.cur);
    double* F = ch.vec + vec_off(pb, c, V_F);
    double x[DR::D], th[DR::P], f[DR::D];
#pragma unroll
    for (int d = 0; d < DR::D; ++d) x[d] = q[d * pb.N + i];
#pragma unroll
    for (int k = 0; k < DR::P; ++k) th[k] = ch.par[(size_t)c * PAR_COUNT + PAR_TH + k];
    DR::f(x, th, f);
#pragma unroll
    for (int d = 0; d < DR::D; ++d) F[d * pb.N + i] = f[d];
}

// validation / bootstrap plan: evaluate buffer 0, no leapfrog
__global__ void k_plan_eval(DevChains ch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ch.n_chains) return;
    LeafPlan p{};
    p.active = 1;
    ch.plan[c] = p;
}

template <int DRIFT>
__global__ __launch_bounds__(MAGI_TAIL_THREADS) void k_leap_finalize(DevProblem pb, DevChains ch, double* out) {
    __shared__ double sh[25 * 16];
    __shared__ double shs[16];
    const int c = blockIdx.x;
    double* vb = ch.vec + vec_off(pb, c, 0);
    const LeafPlan lp = ch.plan[c];
    const ReduceOut ro = leap_reduce<DRIFT>(pb, ch, c, vb, ch.par + (size_t)c * PAR_COUNT, lp, sh, shs);
    if (threadIdx.x == 0 && out) {
        out[c * 8 + 0] = ro.L;
        out[c * 8 + 1] = ro.t12;
        out[c * 8 + 2] = 0.0;
        out[c * 8 + 3] = ro.t3;
        out[c * 8 + 4] = ro.t4;
    }
}

template <int NC, int DRIFT>
int launch_leap_nd(magi_handle* h, int n_chains, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const dim3 grid((pb.N + LEAP_RI - 1) / LEAP_RI + 1, (n_chains + NC - 1) / NC);      // + the service block
    const dim3 block(64 * DriftT<DRIFT>::D * LEAP_RI);
    constexpr bool FVEC = NC > 1;
    if (FVEC) hipLaunchKernelGGL((k_drift<DRIFT>), dim3((pb.N + 255) / 256, n_chains), dim3(256), 0, s, pb, h->ch);
    if (pb.bandf < 0) hipLaunchKernelGGL((k_leap_dense<NC, DRIFT, FVEC>), grid, block, 0, s, pb, h->ch);
    else hipLaunchKernelGGL((k_leap_band<NC, DRIFT, FVEC>), grid, block, 0, s, pb, h->ch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("leap launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

template <int NC>
int launch_leap_nc(magi_handle* h, int n_chains, hipStream_t s) {
    switch (h->pb.drift) {
    case MAGI_DRIFT_SEIR3: return launch_leap_nd<NC, MAGI_DRIFT_SEIR3>(h, n_chains, s);
    case MAGI_DRIFT_SEIR4: return launch_leap_nd<NC, MAGI_DRIFT_SEIR4>(h, n_chains, s);
    default: return launch_leap_nd<NC, MAGI_DRIFT_SIRW>(h, n_chains, s);
    }
}

}  // namespace

int magi_leap_wgs(const DevProblem& pb) { return (pb.N + LEAP_RI - 1) / LEAP_RI; }

int magi_launch_leap(magi_handle* h, int n_chains, hipStream_t s) {
    if (n_chains >= 6) return launch_leap_nc<8>(h, n_chains, s);
    if (n_chains >= 3) return launch_leap_nc<4>(h, n_chains, s);
    if (n_chains == 2) return launch_leap_nc<2>(h, n_chains, s);
    return launch_leap_nc<1>(h, n_chains, s);
}

int magi_launch_plan_eval(magi_handle* h, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(k_plan_eval, dim3((n_chains + 63) / 64), dim3(64), 0, s, h->ch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("plan launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_leap_finalize(magi_handle* h, int n_chains, double* d_out, hipStream_t s) {
    const dim3 g(n_chains), b(MAGI_TAIL_THREADS);
    switch (h->pb.drift) {
    case MAGI_DRIFT_SEIR3: hipLaunchKernelGGL(k_leap_finalize<MAGI_DRIFT_SEIR3>, g, b, 0, s, h->pb, h->ch, d_out); break;
    case MAGI_DRIFT_SEIR4: hipLaunchKernelGGL(k_leap_finalize<MAGI_DRIFT_SEIR4>, g, b, 0, s, h->pb, h->ch, d_out); break;
    default: hipLaunchKernelGGL(k_leap_finalize<MAGI_DRIFT_SIRW>, g, b, 0, s, h->pb, h->ch, d_out); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("leap_finalize launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}
