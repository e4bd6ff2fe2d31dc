// Log-posterior value + gradient (reference: unnormalized_log_prob, magi_v2.py:308-348, and the
// TF autodiff of it that TFP's leapfrog triggers, magi_v2.py:360-364) as three dependent
// mat-vec phases plus one reduce.  HBM/Infinity-Cache bound: each phase streams its matrices once
// with 16-B coalesced loads, one 64-lane wave per matrix row, the source vector staged in LDS and
// shared by the rows / matrices / chains of the workgroup, row sums by wave butterflies.
//
//   phase 1:  cx = Csym xc ,  mx = M xc       -> V_CX,  V_R = f(X,theta) - mx   (:332, :335-336)
//   phase 2:  kr = Ksym r                     -> V_KR                           (:337)
//   phase 3:  mtg = Mt (2 kr)                 -> V_G[X part] (d/dX of the whole log posterior)
//   reduce :  t1..t4, L, d/dsigma_pre, d/dtheta_pre                              (:318-323, :340-348)
//
// Csym = (C^-1 + C^-T)/2 and Ksym likewise: x^T A x = x^T Asym x and grad = 2 Asym x, so one pass
// over each matrix yields both the quadratic form and its gradient for arbitrary (non-symmetric,
// as the reference's pinv outputs are) matrices.  Mt = m^T is stored separately so the transposed
// product also reads rows.
#include "magi_internal.h"

namespace {

template <int PHASE>
__device__ inline double load_src(const DevProblem& pb, const double* vb, int d, int j) {
    if (PHASE == 1) return vb[(size_t)V_Q * pb.dimp + d * pb.N + j] - pb.mu[d];
    if (PHASE == 2) return vb[(size_t)V_R * pb.dimp + d * pb.N + j];
    return 2.0 * vb[(size_t)V_KR * pb.dimp + d * pb.N + j];
}

// Row epilogue shared by the dense and banded kernels: v0 / v1 are the finished row sums.
template <int PHASE>
__device__ inline void row_epilogue(const DevProblem& pb, double* vb, const double* par, int d, int row, double v0, double v1) {
    const int N = pb.N, dimp = pb.dimp;
    const size_t e = (size_t)d * N + row;
    if (PHASE == 2) {
        vb[(size_t)V_KR * dimp + e] = v0;
        return;
    }
    const double* q = vb + (size_t)V_Q * dimp;
    double xg[MAGI_MAX_D], th[MAGI_MAX_P];
#pragma unroll
    for (int dd = 0; dd < MAGI_MAX_D; ++dd) xg[dd] = (dd < pb.D) ? q[dd * N + row] : 0.0;
#pragma unroll
    for (int p = 0; p < MAGI_MAX_P; ++p) th[p] = (p < pb.P) ? par[PAR_TH + p] : 0.0;
    if (PHASE == 1) {
        vb[(size_t)V_CX * dimp + e] = v0;
        vb[(size_t)V_R * dimp + e] = drift_f(pb.drift, d, xg, th) - v1;
    } else {
        double g2[MAGI_MAX_D];
#pragma unroll
        for (int dd = 0; dd < MAGI_MAX_D; ++dd)
            g2[dd] = (dd < pb.D) ? 2.0 * vb[(size_t)V_KR * dimp + dd * N + row] : 0.0;
        const double jt = drift_jt_g(pb.drift, d, xg, th, g2);
        const double d12 = 2.0 * vb[(size_t)V_CX * dimp + e] - v0 + jt;
        const double y = pb.yobs[e];
        double d4 = 0.0;
        if (!isnan(y)) {
            d4 = 2.0 * (q[e] - y) / par[PAR_SIG2 + d];
        }
        vb[(size_t)V_G * dimp + e] = -0.5 * (pb.beta_inv * d12 + d4);
    }
}

// ---- dense: grid (ceil(N / (4R)), D, ceil(n_chains / NC)), 256 threads, LDS NC*TJ doubles -------
template <int PHASE, int NC, int R>
__global__ __launch_bounds__(256) void k_matvec_dense(DevProblem pb, DevChains ch, int TJ) {
    if (ch.gctl->all_done) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int N = pb.N, ld = pb.ld, d = blockIdx.y;
    const int c0 = blockIdx.z * NC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    const double* A0 = (PHASE == 1) ? pb.Csym : (PHASE == 2 ? pb.Ksym : pb.Mt);
    const double* A1 = pb.M;

    double acc0[R][NC], acc1[R][NC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc0[r][c] = 0.0; acc1[r][c] = 0.0; }

    const double2* a0p[R];
    const double2* a1p[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int rr = min(row0 + r, N - 1);
        a0p[r] = reinterpret_cast<const double2*>(A0 + ((size_t)d * N + rr) * ld);
        a1p[r] = reinterpret_cast<const double2*>(A1 + ((size_t)d * N + rr) * ld);
    }

    for (int j0 = 0; j0 < ld; j0 += TJ) {
        const int tjl = min(TJ, ld - j0);
        for (int idx = threadIdx.x; idx < NC * tjl; idx += 256) {
            const int c = idx / tjl, j = idx - c * tjl;
            const int cc = c0 + c;
            double v = 0.0;
            if (cc < ch.n_chains && j0 + j < N)
                v = load_src<PHASE>(pb, ch.vec + vec_off(pb, cc, 0), d, j0 + j);
            lds[c * TJ + j] = v;
        }
        __syncthreads();
        const int n2 = tjl >> 1, jb = j0 >> 1;
#pragma unroll 4
        for (int jj = lane; jj < n2; jj += 64) {
            double2 a0[R], a1[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a0[r] = a0p[r][jb + jj];
                if (PHASE == 1) a1[r] = a1p[r][jb + jj];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double2 xv = reinterpret_cast<const double2*>(lds + c * TJ)[jj];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc0[r][c] = fma(a0[r].x, xv.x, acc0[r][c]);
                    acc0[r][c] = fma(a0[r].y, xv.y, acc0[r][c]);
                    if (PHASE == 1) {
                        acc1[r][c] = fma(a1[r].x, xv.x, acc1[r][c]);
                        acc1[r][c] = fma(a1[r].y, xv.y, acc1[r][c]);
                    }
                }
            }
        }
        __syncthreads();
    }

    double v0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double s0 = wave_sum(acc0[r][c]);
            const double s1 = (PHASE == 1) ? wave_sum(acc1[r][c]) : 0.0;
            if (lane == r * NC + c) { v0 = s0; v1 = s1; }
        }
    if (lane < R * NC) {
        const int r = lane / NC, c = lane - r * NC;
        const int row = row0 + r, cc = c0 + c;
        if (row < N && cc < ch.n_chains)
            row_epilogue<PHASE>(pb, ch.vec + vec_off(pb, cc, 0), ch.par + (size_t)cc * PAR_COUNT, d, row, v0, v1);
    }
}

// ---- banded: rows hold columns [i-b, i+b] (ld >= 2b+1); grid (ceil(N/8), D, ceil(n_chains/NC)) ----
template <int PHASE, int NC>
__global__ __launch_bounds__(256) void k_matvec_band(DevProblem pb, DevChains ch) {
    if (ch.gctl->all_done) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int R = 2, RB = 8;
    const int N = pb.N, ld = pb.ld, b = pb.band, W = 2 * b + 1, d = blockIdx.y;
    const int c0 = blockIdx.z * NC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x * RB;
    const int WL = RB + 2 * b;               // staged window: columns [rb-b, rb+RB+b)
    const double* A0 = (PHASE == 1) ? pb.Csym : (PHASE == 2 ? pb.Ksym : pb.Mt);
    const double* A1 = pb.M;

    for (int idx = threadIdx.x; idx < NC * WL; idx += 256) {
        const int c = idx / WL, w = idx - c * WL;
        const int j = rb - b + w, cc = c0 + c;
        double v = 0.0;
        if (cc < ch.n_chains && j >= 0 && j < N) v = load_src<PHASE>(pb, ch.vec + vec_off(pb, cc, 0), d, j);
        lds[c * WL + w] = v;
    }
    __syncthreads();

    double v0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = rb + wave * R + r;
        const int rr = min(row, N - 1);
        const double* a0 = A0 + ((size_t)d * N + rr) * ld;
        const double* a1 = A1 + ((size_t)d * N + rr) * ld;
        double acc0[NC], acc1[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc0[c] = 0.0; acc1[c] = 0.0; }
        const int xoff = rr - rb;            // window index of column rr-b
#pragma unroll 4
        for (int k = lane; k < W; k += 64) {
            const double e0 = a0[k];
            const double e1 = (PHASE == 1) ? a1[k] : 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double xv = lds[c * WL + xoff + k];
                acc0[c] = fma(e0, xv, acc0[c]);
                if (PHASE == 1) acc1[c] = fma(e1, xv, acc1[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double s0 = wave_sum(acc0[c]);
            const double s1 = (PHASE == 1) ? wave_sum(acc1[c]) : 0.0;
            if (lane == r * NC + c) { v0 = s0; v1 = s1; }
        }
    }
    if (lane < R * NC) {
        const int r = lane / NC, c = lane - r * NC;
        const int row = rb + wave * R + r, cc = c0 + c;
        if (row < N && cc < ch.n_chains)
            row_epilogue<PHASE>(pb, ch.vec + vec_off(pb, cc, 0), ch.par + (size_t)cc * PAR_COUNT, d, row, v0, v1);
    }
}

// ---- reduce: one 1024-thread workgroup per chain; out[chain][8] = L, t1, t2, t3, t4 ------------
__global__ __launch_bounds__(MAGI_TAIL_THREADS) void k_finalize(DevProblem pb, DevChains ch, double* out) {
    __shared__ double sh[(3 + MAGI_MAX_D + MAGI_MAX_P) * 16];
    __shared__ double shs[8];
    const int c = blockIdx.x;
    double* vb = ch.vec + vec_off(pb, c, 0);
    FinalizeOut fo = finalize_gradient(pb, vb, ch.par + (size_t)c * PAR_COUNT, sh, shs);
    if (threadIdx.x == 0 && out) {
        out[c * 8 + 0] = fo.L;
        out[c * 8 + 1] = fo.t1;
        out[c * 8 + 2] = fo.t2;
        out[c * 8 + 3] = fo.t3;
        out[c * 8 + 4] = fo.t4;
    }
}

// transformed parameters of the states in V_Q (API path; the sampler's tail does this itself)
__global__ void k_prepare(DevProblem pb, DevChains ch) {
    const int c = blockIdx.x, j = threadIdx.x;
    if (j < pb.D + pb.P)
        compute_par_entry(pb, j, ch.vec[vec_off(pb, c, V_Q) + pb.ND + j], ch.par + (size_t)c * PAR_COUNT);
    // operand-order mirror of the positions (matrix-core streaming kernel; both slot parities)
    const double* q = ch.vec + vec_off(pb, c, V_Q);
    for (int e = j; e < pb.ND; e += blockDim.x) {
        const int d = e / pb.N, i = e - d * pb.N;
        ch.xop[xop_off(pb, ch.n_chains, 0, c, d, i)] = q[e];
        ch.xop[xop_off(pb, ch.n_chains, 1, c, d, i)] = q[e];
    }
}

template <int PHASE, int NC>
int launch_phase_nc(magi_handle* h, int n_chains, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const int groups = (n_chains + NC - 1) / NC;
    if (pb.band < 0) {
        constexpr int R = 2;
        int TJ = pb.ld;
        const int cap = (64 * 1024) / (8 * NC);
        if (TJ > cap) TJ = cap & ~127;
        dim3 grid((pb.N + 4 * R - 1) / (4 * R), pb.D, groups);
        const size_t lds = (size_t)NC * TJ * sizeof(double);
        hipLaunchKernelGGL((k_matvec_dense<PHASE, NC, R>), grid, dim3(256), lds, s, pb, h->ch, TJ);
    } else {
        dim3 grid((pb.N + 7) / 8, pb.D, groups);
        const size_t lds = (size_t)NC * (8 + 2 * pb.band) * sizeof(double);
        hipLaunchKernelGGL((k_matvec_band<PHASE, NC>), grid, dim3(256), lds, s, pb, h->ch);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("phase launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

template <int PHASE>
int launch_phase(magi_handle* h, int n_chains, hipStream_t s) {
    if (n_chains >= 8) return launch_phase_nc<PHASE, 8>(h, n_chains, s);
    if (n_chains >= 3) return launch_phase_nc<PHASE, 4>(h, n_chains, s);
    if (n_chains == 2) return launch_phase_nc<PHASE, 2>(h, n_chains, s);
    return launch_phase_nc<PHASE, 1>(h, n_chains, s);
}

}  // namespace

int magi_launch_phase(magi_handle* h, int phase, int n_chains, hipStream_t s) {
    if (phase == 1) return launch_phase<1>(h, n_chains, s);
    if (phase == 2) return launch_phase<2>(h, n_chains, s);
    return launch_phase<3>(h, n_chains, s);
}

int magi_launch_gradient(magi_handle* h, int n_chains, hipStream_t s) {
    int rc;
    if ((rc = launch_phase<1>(h, n_chains, s))) return rc;
    if ((rc = launch_phase<2>(h, n_chains, s))) return rc;
    return launch_phase<3>(h, n_chains, s);
}

int magi_launch_prepare(magi_handle* h, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(k_prepare, dim3(n_chains), dim3(64), 0, s, h->pb, h->ch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("prepare launch: ") + hipGetErrorString(e));
    return magi_launch_mirror(h, n_chains, s);          // (separable drifts: the streaming kernel's operand mirror)
}

int magi_launch_finalize(magi_handle* h, int n_chains, double* d_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(n_chains), dim3(MAGI_TAIL_THREADS), 0, s, h->pb, h->ch, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("finalize launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}
