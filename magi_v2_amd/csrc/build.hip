// Kernel-matrix construction on the GPU (reference: MAGI_v2._build_matrices magi_v2.py:774-823,
// the pinv sites :126,128 and the band approximation :271-274).
//
//   K1  k_matern          Kappa, p_Kappa, Kappa_pp on grid x grid via a device modified Bessel
//                         function K_nu for real order (Temme series u <= 2, Steed CF2 u > 2);
//                         scipy/AMOS `kvp` is what the reference calls (magi_v2.py:787).
//   K2  potrf (blocked, right-looking): diagonal 128x128 block factorised + inverted in LDS by one
//                         workgroup, panel solve and trailing SYRK update on v_mfma_f64_16x16x4_f64
//   K3  products           m = p_Kappa Kappa^-1,  K = Kappa_pp - p_Kappa Kappa^-1 Kappa_p   (MFMA GEMM)
//   K4  potri              A^-1 = T^T T with T = L^-1 from a blocked triangular inverse
//
// The reference forms pinv(Kappa) and pinv(K) by SVD; at the conditioning of these problems
// (cond(Kappa) ~ 1e6..1e10, full rank under both libraries' cut-offs: SURVEY section 7) pinv is
// the inverse, and it is computed here through Cholesky.  The Matern derivatives use the
// algebraically simplified forms
//     kappa       = A u^nu K_nu(u)
//     d/ds        = -A c sgn(s-t) u^nu K_{nu-1}(u)
//     d2/dsdt     = A c^2 u^{nu-1} (K_{nu-1}(u) - u K_{nu-2}(u)),   u = c|s-t|, c = sqrt(2nu)/phi2,
// which agree with the reference's kvp-based expressions (:790-815) to their own rounding error.
#include <algorithm>
#include <cstdlib>

#include "magi_internal.h"

namespace {

// =============================================================================================
// K1: modified Bessel function of the second kind, real order
// =============================================================================================
struct BesselConsts {
    double mu;                 // fractional order in [-0.5, 0.5]
    int n;                     // nu = mu + n
    double gam1, gam2, gampl, gammi;   // Temme's gamma-function combinations for mu
};

// e^{x} K_mu(x) and e^{x} K_{mu+1}(x) (exponentially SCALED so that large x does not underflow
// before it is combined with u^nu)
__device__ inline void bessel_k_scaled(double x, const BesselConsts& bc, double& k0, double& k1) {
    const double mu = bc.mu, mu2 = mu * mu;
    if (x <= 2.0) {
        const double b = 0.5 * x;
        double d = -log(b);
        double e = mu * d;
        const double fact2 = (fabs(e) < 1e-16) ? 1.0 : sinh(e) / e;
        const double pimu = 3.141592653589793 * mu;
        const double fact = (fabs(pimu) < 1e-16) ? 1.0 : pimu / sin(pimu);
        double ff = fact * (bc.gam1 * cosh(e) + bc.gam2 * fact2 * d);
        double sum = ff;
        e = exp(e);
        double p = 0.5 * e / bc.gampl;
        double q = 0.5 / (e * bc.gammi);
        double c = 1.0;
        d = b * b;
        double sum1 = p;
        for (int i = 1; i < 500; ++i) {
            ff = (i * ff + p + q) / (i * (double)i - mu2);
            c *= d / i;
            p /= (i - mu);
            q /= (i + mu);
            const double del = c * ff;
            sum += del;
            sum1 += c * (p - i * ff);
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        const double ex = exp(x);
        k0 = sum * ex;
        k1 = sum1 * (2.0 / x) * ex;
    } else {
        double b = 2.0 * (1.0 + x);
        double d = 1.0 / b;
        double h = d, delh = d;
        double q1 = 0.0, q2 = 1.0;
        const double a1 = 0.25 - mu2;
        double q = a1, c = a1, a = -a1;
        double s = 1.0 + q * delh;
        for (int i = 2; i < 10000; ++i) {
            a -= 2 * (i - 1);
            c = -a * c / i;
            const double qnew = (q1 - b * q2) / a;
            q1 = q2;
            q2 = qnew;
            q += c * qnew;
            b += 2.0;
            d = 1.0 / (b + a * d);
            delh = (b * d - 1.0) * delh;
            h += delh;
            const double dels = q * delh;
            s += dels;
            if (fabs(dels / s) < 1e-17) break;
        }
        h = a1 * h;
        k0 = sqrt(3.141592653589793 / (2.0 * x)) / s;
        k1 = k0 * (mu + x + 0.5 - h) / x;
    }
}

struct MaternArgs {
    const double* I;
    double *Kappa, *pKappa, *Kappapp;
    int N;
    double phi1, nu, c;        // c = sqrt(2 nu) / phi2
    double logA;               // log(phi1 2^{1-nu} / Gamma(nu))
    double diag_pp;            // nu phi1 / (phi2^2 (nu - 1))
    BesselConsts bc;
    int rows;                  // grid rows per workgroup (64, fewer on small grids: an entry costs microseconds of dependent
                               // fp64 arithmetic, so a small matrix wants one entry per thread rather than 16)
    const double* dyn;         // non-null: phi1, c, logA, diag_pp come from device memory (FitDyn below) -- the hyper-parameter
                               // fit replays one captured graph per Adam step, so its launch arguments cannot change
    long bsK, bs_dyn;          // grid.z > 1 (the fit's batched step: all components in one launch): element strides of the outputs / dyn blocks
};

// 64 x 64 tile per 256-thread workgroup; the two 64-entry slices of the time grid staged in LDS
// layout of the per-component device block of the hyper-parameter fit (doubles)
enum FitDyn { FD_RAW = 0, FD_M = 3, FD_V = 6, FD_PV = 9 /* phi1, phi2, sigma^2 */, FD_C = 12, FD_LOGA, FD_DIAGPP, FD_SHIFT, FD_COUNT };

__global__ __launch_bounds__(256) void k_matern(MaternArgs a) {
    __shared__ double ts[64], tt[64];
    if (gridDim.z > 1) {
        const long z = blockIdx.z;
        a.Kappa += z * a.bsK; a.pKappa += z * a.bsK; a.Kappapp += z * a.bsK;
        if (a.dyn) a.dyn += z * a.bs_dyn;
    }
    if (a.dyn) { a.phi1 = a.dyn[FD_PV]; a.c = a.dyn[FD_C]; a.logA = a.dyn[FD_LOGA]; a.diag_pp = a.dyn[FD_DIAGPP]; }
    const int i0 = blockIdx.y * a.rows, j0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int i = i0 + threadIdx.x;
        ts[threadIdx.x] = (i < a.N) ? a.I[i] : 0.0;
    } else if (threadIdx.x < 128) {
        const int j = j0 + threadIdx.x - 64;
        tt[threadIdx.x - 64] = (j < a.N) ? a.I[j] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < a.rows; r += 4) {
        const int i = i0 + r, j = j0 + tx;
        if (i >= a.N || j >= a.N) continue;
        double kap, pk, kpp;
        if (i == j) {
            kap = a.phi1; pk = 0.0; kpp = a.diag_pp;                  // magi_v2.py:795, 802, 815
        } else {
            const double dt = ts[r] - tt[tx];
            const double u = a.c * fabs(dt);
            double k0, k1;                       // scaled K_mu, K_{mu+1}
            bessel_k_scaled(u, a.bc, k0, k1);
            // recur to K_{nu-2}, K_{nu-1}, K_nu  (nu = mu + n, n >= 1)
            double km2, km1, kn;
            if (a.bc.n == 1) {
                km1 = k0; kn = k1;
                km2 = k1 - 2.0 * a.bc.mu / u * k0;                    // K_{mu-1} = K_{mu+1} - (2 mu/u) K_mu
            } else {
                km2 = k0; km1 = k1;
                double ord = a.bc.mu + 1.0;
                kn = km2 + 2.0 * ord / u * km1;
                for (int t = 2; t < a.bc.n; ++t) {
                    km2 = km1; km1 = kn; ord += 1.0;
                    kn = km2 + 2.0 * ord / u * km1;
                }
            }
            const double lu = log(u);
            const double f = exp(a.logA + a.nu * lu - u);              // A u^nu e^-u
            kap = f * kn;
            pk = -(dt > 0.0 ? 1.0 : -1.0) * a.c * f * km1;
            kpp = a.c * a.c * (f / u) * (km1 - u * km2);
        }
        const size_t o = (size_t)i * a.N + j;
        a.Kappa[o] = kap;
        a.pKappa[o] = pk;
        a.Kappapp[o] = kpp;
    }
}

// =============================================================================================
// fp64 MFMA GEMM:  C(m,n) = alpha * sum_k A(m,k) B(n,k) + beta * C(m,n)
//   A(m,k) = A[m*sAm + k*sAk],  B(n,k) = B[n*sBn + k*sBk]  (arbitrary strides -> NN / NT / TN)
//   128 x 128 tile per 256-thread workgroup, 4 waves x (64 x 64) = 4x4 v_mfma_f64_16x16x4_f64
//   blocks each, K step 16, operands staged in LDS as [k][m] with an odd pitch (129 doubles).
// =============================================================================================
using d4 = __attribute__((ext_vector_type(4))) double;

struct GemmArgs {
    const double* A; long sAm, sAk;
    const double* B; long sBn, sBk;
    double* C; long ldc;
    int M, N, K;
    double alpha, beta;
    int lower_only;    // 1: skip tiles strictly above the block diagonal (m0 + 127 < n0)
    int kmode;         // 0: k in [0, K) ; 1: k >= min-aligned max(m0, n0) (A, B "lower" in (k, m)) ;
                       // 2: k < m0 + 128 (A lower-triangular in (m, k)) ; 3: k >= n0 (B(n,k) zero for k < n) ;
                       // 4: k < n0 + 128 (B(n,k) zero for k > n)
    long batchA, batchB, batchC;   // element strides between grid.z batches
    double* C2; long ldc2, batchC2t;   // optional second output, TRANSPOSED: C2[n * ldc2 + m] = alpha * sum (the Cholesky panels are kept in
                                   // both orientations: the rank-k updates then read both operands along their unit-stride dimension)
    int remap;                     // set by launch_gemm: XCD-aware super-block tile order (large tile grids)
    int remap_min;                 // > 0: this launch's own threshold (super-blocks) for that order instead of the option's
    int zinner;                    // > 0: grid.z = zinner x outer; batch* step the inner index, batch*2 the outer one
    long batchA2, batchB2, batchC2;
};

constexpr int GT = 128, GK = 16, GP = 129;   // odd pitch: the k-fast staging stores of a 16-lane group hit 16 distinct bank pairs

// 128 x 16 operand tile = 2048 doubles, 8 per thread, walking the unit-stride dimension fastest.  A thread's 8 elements are
// p + i * di with p per thread (loop-invariant: row/column of the thread) and di uniform, so a load costs an SGPR offset and a
// predicate -- written as base[gm * sm + gk * sk] per element it cost three quarter-rate 64-bit multiplies, ~1300 VALU cycles
// per K step next to 4096 MFMA cycles.
struct TileWalk {
    const double* p;     // element (m_t, k_t) of the tile at k0 = 0
    long di;             // element stride between a thread's consecutive elements
    int m_t, k_t, dm, dk, lds0, dlds;
};

__device__ __forceinline__ TileWalk gemm_tile_walk(const double* base, long sm, long sk, int m0) {
    TileWalk w;
    const bool kfast = (sk == 1);
    const int t = threadIdx.x;
    w.m_t = kfast ? (t >> 4) : (t & 127);
    w.k_t = kfast ? (t & 15) : (t >> 7);
    w.dm = kfast ? 16 : 0;
    w.dk = kfast ? 0 : 2;
    w.di = kfast ? 16 * sm : 2 * sk;
    w.p = base + (long)(m0 + w.m_t) * sm + (long)w.k_t * sk;
    w.lds0 = w.k_t * GP + w.m_t;
    w.dlds = kfast ? 16 : 2 * GP;
    return w;
}

// INTERIOR: the whole 128 x 16 tile lies inside the operand (uniform per workgroup and K step): no predicates, no branches
template <bool INTERIOR>
__device__ __forceinline__ void gemm_load_tile(const TileWalk& w, long sk, int m0, int k0, int Mlim, int Klim, double (&reg)[8]) {
    const double* q = w.p + (long)k0 * sk;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (INTERIOR) {
            reg[i] = q[i * w.di];
        } else {
            const bool ok = (m0 + w.m_t + i * w.dm < Mlim) && (k0 + w.k_t + i * w.dk < Klim);
            reg[i] = ok ? q[i * w.di] : 0.0;
        }
    }
}

__device__ __forceinline__ void gemm_store_tile(double* lds, const TileWalk& w, const double (&reg)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) lds[w.lds0 + i * w.dlds] = reg[i];
}

template <bool INTERIOR>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& g, const double* A, const double* B, int m0, int n0, int kbeg, int kend,
                                              double* As, double* Bs, d4 (&acc)[4][4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 15, lk = lane >> 4;
    double ra[8], rb[8];
    const TileWalk wa = gemm_tile_walk(A, g.sAm, g.sAk, m0), wb = gemm_tile_walk(B, g.sBn, g.sBk, n0);
    if (kbeg < kend) {
        gemm_load_tile<INTERIOR>(wa, g.sAk, m0, kbeg, g.M, kend, ra);
        gemm_load_tile<INTERIOR>(wb, g.sBk, n0, kbeg, g.N, kend, rb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        __syncthreads();
        gemm_store_tile(As, wa, ra);
        gemm_store_tile(Bs, wb, rb);
        __syncthreads();
        if (k0 + GK < kend) {
            gemm_load_tile<INTERIOR>(wa, g.sAk, m0, k0 + GK, g.M, kend, ra);
            gemm_load_tile<INTERIOR>(wb, g.sBk, n0, k0 + GK, g.N, kend, rb);
        }
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[(kk * 4 + lk) * GP + wm + i * 16 + lr];
                b[i] = Bs[(kk * 4 + lk) * GP + wn + i * 16 + lr];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
}

// C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.  FULL: the 128 x 128 tile lies inside C, no
// predicates -- with a branch per element the compiler waits for ALL outstanding memory operations (vmcnt counts stores too) in
// front of every store.  Read-modify-write (beta != 0) loads the 16 values of a row block before its 16 stores: element by
// element the load -> store order must be kept, 64 dependent round trips per thread, several times the MFMA time of a K = 128
// update.
template <bool FULL>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, double* C, double* C2, int m0, int n0, const d4 (&acc)[4][4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = lane & 15, lk = lane >> 4;
    double* c0 = C + (long)(m0 + wm + lk) * g.ldc + n0 + wn + lr;          // element (i, j, r) at c0 + (16 i + 4 r) ldc + 16 j
    const int mrel = m0 + wm + lk, nrel = n0 + wn + lr;
    if (g.beta == 0.0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (FULL || (mrel + 16 * i + 4 * r < g.M && nrel + 16 * j < g.N)) c0[(long)(16 * i + 4 * r) * g.ldc + 16 * j] = g.alpha * acc[i][j][r];
        if (C2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (FULL || (mrel + 16 * i + 4 * r < g.M && nrel + 16 * j < g.N))
                            C2[(long)(nrel + 16 * j) * g.ldc2 + mrel + 16 * i + 4 * r] = g.alpha * acc[i][j][r];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double c[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    c[j][r] = (FULL || (mrel + 16 * i + 4 * r < g.M && nrel + 16 * j < g.N)) ? c0[(long)(16 * i + 4 * r) * g.ldc + 16 * j] : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (FULL || (mrel + 16 * i + 4 * r < g.M && nrel + 16 * j < g.N))
                        c0[(long)(16 * i + 4 * r) * g.ldc + 16 * j] = g.alpha * acc[i][j][r] + g.beta * c[j][r];
        }
    }
}

#ifndef MAGI_GEMM_OCC
#define MAGI_GEMM_OCC 2      // 250 VGPRs, two workgroups per CU: one stages while the other issues MFMAs (1 -> 0.53, 2 -> 0.76 of the fp64 MFMA peak at N = 8192)
#endif
// CLS only names the instantiation (BuildClass below): the per-class rows of a rocprofv3 kernel trace of the matrix build
template <int CLS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MAGI_GEMM_OCC, MAGI_GEMM_OCC))) void k_gemm_f64(GemmArgs g) {
    __shared__ double As[GK * GP], Bs[GK * GP];
    // Tile of this workgroup.  The hardware deals workgroups to the 8 XCDs round-robin on the linear id, and an XCD holds 64 of them
    // (32 CUs x 2): XCD x takes the 8 x 8 SUPER-BLOCKS u = x, x + 8, ... of the tile grid, one at a time -- its 64 resident tiles
    // then share 8 row panels of A and 8 column panels of B through its own L2 in step, whatever the shape of the active region
    // (lower-only / triangular k ranges otherwise scatter the co-resident tiles and lose that reuse: 0.55 instead of 0.9 of the
    // MFMA peak).  Lower-only launches enumerate the lower super-blocks only.
    const int tY = (g.M + GT - 1) / GT, tX = (g.N + GT - 1) / GT, sY = (tY + 7) >> 3, sX = (tX + 7) >> 3;
    int by, bx;
    if (!g.remap) {           // small tile grids (fewer than three super-blocks per XCD): plain row-major tiles, long k ranges first
        by = (int)blockIdx.x / tX; bx = (int)blockIdx.x - by * tX;
        if (g.kmode == 2) by = tY - 1 - by;
        if (g.kmode == 4) bx = tX - 1 - bx;
    } else {
    const int i = (int)blockIdx.x, j = i >> 3;
    // (rotated by the round AND by the batch index: an XCD must not keep ONE super-column -- k ranges that depend on the column would all be
    //  long on one XCD.  The round alone is not enough: a 32 x 32-tile launch has 16 super-blocks, two rounds, so an XCD sees two of the four
    //  super-columns, and with the same two for every component of the batch XCDs 0 and 4 carried 1.5 x the mean of trtri's first product --
    //  it took as long as over the full K range, 0.46 of the MFMA peak against 0.85 for its twin (profiles/r04_trtri_by_level.txt))
    const int u = (j >> 6) * 8 + (((i & 7) + (j >> 6) + 3 * (int)blockIdx.z) & 7), w = j & 63;
    int sy, sx;
    if (g.lower_only) {
        sy = (int)((sqrtf(8.0f * (float)u + 1.0f) - 1.0f) * 0.5f);
        while (sy * (sy + 1) / 2 > u) --sy;
        while ((sy + 1) * (sy + 2) / 2 <= u) ++sy;
        sx = u - sy * (sy + 1) / 2;
        if (sy >= sY) return;
    } else {
        if (u >= sY * sX) return;
        sy = u / sX; sx = u - sy * sX;
    }
    // (triangular k ranges grow with m0 (kmode 2) or n0 (kmode 4): the long tiles first, the short ones fill the tail)
    if (g.kmode == 2) sy = sY - 1 - sy;
    if (g.kmode == 4) sx = sX - 1 - sx;
    by = 8 * sy + (w >> 3); bx = 8 * sx + (w & 7);
    if (by >= tY || bx >= tX) return;
    }
    const int m0 = by * GT, n0 = bx * GT;
    if (g.lower_only && m0 + GT - 1 < n0) return;
    const int zi = g.zinner > 0 ? (int)blockIdx.z % g.zinner : (int)blockIdx.z, zo = g.zinner > 0 ? (int)blockIdx.z / g.zinner : 0;
    const double* A = g.A + (long)zi * g.batchA + (long)zo * g.batchA2;
    const double* B = g.B + (long)zi * g.batchB + (long)zo * g.batchB2;
    double* C = g.C + (long)zi * g.batchC + (long)zo * g.batchC2;
    int kbeg = 0, kend = g.K;
    if (g.kmode == 1) kbeg = (max(m0, n0) / GK) * GK;
    else if (g.kmode == 2) kend = min(g.K, m0 + GT);
    else if (g.kmode == 3) kbeg = (n0 / GK) * GK;
    else if (g.kmode == 4) kend = min(g.K, n0 + GT);
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

    // interior workgroups (all 128 rows of both operand tiles exist, whole K steps) run the loop without predicates
    const bool interior = (m0 + GT <= g.M) && (n0 + GT <= g.N) && ((kend - kbeg) % GK == 0);
    if (interior) gemm_mainloop<true>(g, A, B, m0, n0, kbeg, kend, As, Bs, acc);
    else gemm_mainloop<false>(g, A, B, m0, n0, kbeg, kend, As, Bs, acc);
    double* C2 = g.C2 ? g.C2 + (long)zi * g.batchC2t : nullptr;
    if ((m0 + GT <= g.M) && (n0 + GT <= g.N)) gemm_epilogue<true>(g, C, C2, m0, n0, acc);
    else gemm_epilogue<false>(g, C, C2, m0, n0, acc);
}

// =============================================================================================
// diagonal block: Cholesky of an n x n (n <= 128) SPD block + explicit inverse of its factor, one workgroup.
//
// A column-by-column factorisation in LDS is a chain of 128 x 3 barriers (measured 0.48 ms per block, a quarter of the
// N = 8192 build).  Here the block is processed in four 32-wide sub-blocks:
//   1. ONE wave factorises the 32 x 32 diagonal sub-block with a row per lane in registers -- pivots and multipliers
//      travel by v_readlane, no barrier, no LDS -- and inverts the factor the same way (a column per lane);
//   2. the rows below are multiplied by that inverse (one thread per row, operands broadcast from LDS);
//   3. the trailing sub-matrix takes its rank-32 update in 2 x 2 register tiles;
// then the off-diagonal blocks of the inverse follow from 32 x 32 x 32 products (block forward substitution).
// LDS: one 128 x 129 array holds L in its lower triangle and the strict part of Linv, transposed, in its upper
// triangle; the diagonal of Linv and one 32 x 33 temporary sit behind it.  The block is padded to 128 with the
// identity so no loop carries bounds.
// =============================================================================================
constexpr int DG_LD = 129;
constexpr int DG_LDS_DOUBLES = 128 * DG_LD + 128 + 3 * 32 * 33 + 2;      // 158 480 bytes of the CU's 160 KB

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// C (32 x 32, thread tile 2 x 2) = sum over k of  a(i, k) * b(k, j)  with accessors (no bounds: the caller passes K)
// (K is a multiple of 32; unrolled by 8 so that eight iterations' LDS reads are in flight -- one at a time the loop is a chain of
//  LDS latencies, 250 cycles per k)
template <class FA, class FB>
__device__ __forceinline__ void dg_tile_gemm(int K, FA a, FB b, double (&c)[2][2], bool accumulate = false) {
    const int ti = (threadIdx.x >> 4) * 2, tj = (threadIdx.x & 15) * 2;
    if (!accumulate) c[0][0] = c[0][1] = c[1][0] = c[1][1] = 0.0;
#pragma unroll 8
    for (int k = 0; k < K; ++k) {
        const double a0 = a(ti, k), a1 = a(ti + 1, k), b0 = b(k, tj), b1 = b(k, tj + 1);
        c[0][0] = fma(a0, b0, c[0][0]); c[0][1] = fma(a0, b1, c[0][1]);
        c[1][0] = fma(a1, b0, c[1][0]); c[1][1] = fma(a1, b1, c[1][1]);
    }
}

// One wave: acc (16 x 16, v_mfma_f64_16x16x4_f64 layout: lane holds rows (lane >> 4) + 4 r of column lane & 15) += sum over k < K of
// a(m, k) * b(k, n) with the operands read from LDS through accessors (m, n = lane & 15; k = 4 step + (lane >> 4)).  K multiple of 4.
template <class FA, class FB>
__device__ __forceinline__ void dg_mfma(int K, FA a, FB b, d4& acc) {
    const int lane = threadIdx.x & 63, li = lane & 15, lj = lane >> 4;
    // (one k step at a time the loop is a chain of LDS latencies -- operand reads, then the dependent MFMA: ~200 cycles per step, 1.6 k per
    //  rank-32 tile; with the operands of 8 / 4 steps read ahead the MFMAs issue back to back.  Same accumulation order.)
    int k = 0;
    for (; k + 32 <= K; k += 32) {
        double av[8], bv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { av[s] = a(li, k + 4 * s + lj); bv[s] = b(k + 4 * s + lj, li); }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
    }
    for (; k + 16 <= K; k += 16) {
        double av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { av[s] = a(li, k + 4 * s + lj); bv[s] = b(k + 4 * s + lj, li); }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
    }
    for (; k < K; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(li, k + lj), b(k + lj, li), acc, 0, 0, 0);
}

// One wave: Cholesky factor of the 16 x 16 block of S at (b0, b0) and the inverse of that factor, in registers: lane r (& 15) holds
// row r, pivots and multipliers travel by v_readlane (no barrier, no LDS in the loop).  L goes back to the lower triangle of the
// block, X = L^-1 (lower) to the dense array X (pitch 33) at (x0, x0).  Returns the first non-positive pivot's index or -1 (uniform).
__device__ __forceinline__ int dg_factor16(double* S, int b0, double* X, int x0) {
    const int lane = threadIdx.x & 63, r = lane & 15;
    double d[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = S[(b0 + r) * DG_LD + b0 + c];
#pragma unroll
    for (int c = 0; c < 16; ++c) if (c > r) d[c] = 0.0;              // (the upper part holds other data)
    int fail = -1;
    double myrd = 0.0;                                                   // lane k keeps 1 / L_kk
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double piv = readlane_f64(d[k], k);
        if (!(piv > 0.0) && fail < 0) fail = b0 + k;                     // uniform; also catches NaN
        // 1/sqrt(piv): v_rsq_f64 (~23 bits) + two Newton steps, then sqrt = piv * y with one correction (the library sqrt followed by a
        // division is ~70 dependent fp64 instructions per column)
        double y = __builtin_amdgcn_rsq(piv);
        const double hp = 0.5 * piv;
        y = fma(y, fma(-hp * y, y, 0.5), y);
        y = fma(y, fma(-hp * y, y, 0.5), y);
        double dk = piv * y;
        dk = fma(fma(-dk, dk, piv), 0.5 * y, dk);
        if (r == k) myrd = y;
        d[k] = (r == k) ? dk : d[k] * y;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) {
            const double ljk = readlane_f64(d[k], j);
            d[j] = fma(-d[k], ljk, d[j]);          // (rows r < j compute garbage in the upper triangle, which nothing reads: no predicate)
        }
    }
    double x[16];                                                        // lane j holds column j of the inverse
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double sacc = 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) sacc = fma(readlane_f64(d[k], i), x[k], sacc);
        x[i] = (i < r) ? 0.0 : ((i == r) ? 1.0 : -sacc) * readlane_f64(myrd, i);
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c <= r) S[(b0 + r) * DG_LD + b0 + c] = d[c];
            X[(x0 + c) * 33 + x0 + r] = x[c];                            // X[c][r] (zero above the diagonal)
        }
    }
    return fail;
}

#ifdef MAGI_DIAG_STAMPS      // dev: s_memtime at the phase boundaries of k_diag_chol_inv, printed by thread 0
#define DG_STAMP(i) do { if (threadIdx.x == 0) dg_st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define DG_STAMP_PRINT() do { if (threadIdx.x == 0 && block_row0 == 0) { printf("diag stamps (cycles from start):"); for (int q = 1; q < 16; ++q) printf(" %d:%lld", q, (long long)(dg_st[q] - dg_st[0])); printf("\n"); } } while (0)
#else
#define DG_STAMP(i) do {} while (0)
#define DG_STAMP_PRINT() do {} while (0)
#endif
// grid.x = components of a batch: matrix, inverse block and status word of component z at z * bsA / bsL / bsS
//
// Schedule (round 4).  The register factorisation of a 32 x 32 diagonal sub-block (step 1) is a dependent chain on ONE wave -- 25 k of
// the kernel's 183 k cycles per sub-block, 56 % of the kernel -- and everything else used to wait for it at a barrier.  Now wave 0 is the
// FACTOR wave and waves 1..3 are WORKERS that do, beside step 1 of sub-block kb, whatever no longer depends on it:
//     wave 0 :  [update of the next diagonal sub-block (3 tiles of step 3)]  step 1 (kb)
//     workers:  the other tiles of step 3 (kb - 1)  |  the off-diagonal blocks of the inverse's block row kb - 1  |  stores: block column
//               kb - 1 of L, finished block rows of the inverse  (iteration 0: the load of everything but the first sub-block)
// with ONE barrier behind them, then step 2 (rows below the sub-block, all waves) and its barrier.  A job of the inverse is a 32 x 16 half
// of a block on one wave, its intermediate product kept in the accumulators (the C layout of v_mfma_f64_16x16x4_f64 -- lane holds rows
// (lane >> 4) + 4 r of column lane & 15 -- is the B-operand layout of the following product), so workers never synchronise with each other.
// Every element takes the same operations in the same order as before: the outputs are bit-identical to the round-3 kernel's.
__global__ __launch_bounds__(256) void k_diag_chol_inv(double* A, long lda, int n, double* Linv /* [128][128] */, int* status, int block_row0,
                                                       long bsA, long bsL, int bsS) {
    A += (long)blockIdx.x * bsA; Linv += (long)blockIdx.x * bsL; status += (long)blockIdx.x * bsS;
#ifdef MAGI_DIAG_STAMPS
    unsigned long long dg_st[16] = {0};
#endif
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* S = lds;                               // [128][DG_LD]
    double* dinv = lds + 128 * DG_LD;              // diagonal of the inverse
    double* T = dinv + 128;                        // [32][33]
    double* T1 = T + 32 * 33;                      // dense inverse of the diagonal sub-block in flight (step 1 -> step 2)
    double* T2 = T1 + 32 * 33;
    int& bad = *reinterpret_cast<int*>(T2 + 32 * 33);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    if (tid == 0) bad = -1;
    DG_STAMP(0);
    const int nsb = (n + 31) >> 5, nl = 32 * nsb;          // live 32-wide sub-blocks (the identity padding behind them factors to itself)

    // ---- pieces --------------------------------------------------------------------------------------------------------------------
    // elements e = i * 128 + j of the block, e in {t, t + nt, ...}, rows [r0, r1): global -> S (lower triangle; identity padding; zeros above),
    // 16 loads in flight per thread (`S[..] = cond ? A[..] : pad` in a plain loop is a load -> LDS store chain, one round trip per element)
    // (lw = log2 of the region's width: 5 or 7)
    auto load_rows = [&](int r0, int r1, int lw, int t, int nt) {
        const int cnt = (r1 - r0) << lw, wm = (1 << lw) - 1;
        for (int e0 = t; e0 < cnt; e0 += nt * 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = e0 + nt * u, i = r0 + (e >> lw), j = e & wm;
                v[u] = (e < cnt && j <= i && i < n) ? A[(long)i * lda + j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = e0 + nt * u, i = r0 + (e >> lw), j = e & wm;
                if (e < cnt) S[i * DG_LD + j] = (i < n && j < n) ? v[u] : ((i == j) ? 1.0 : 0.0);
            }
        }
    };
    // step 1 of sub-block kb (one wave): two 16 x 16 register factorisations (dg_factor16) with the rank-16 update and the off-diagonal block of
    // the inverse between them on the matrix cores (wave-private: LDS operations of a wave are ordered).  T1 <- dense inv(L11) (32 x 32).
    auto step1 = [&](int c0) {
        const int c1 = c0 + 16;
        int fail = dg_factor16(S, c0, T1, 0);
        {   // L21 = A21 X11a^T ; A22 -= L21 L21^T
            d4 o = d4{0.0, 0.0, 0.0, 0.0};
            dg_mfma(16, [&](int m, int k) { return S[(c1 + m) * DG_LD + c0 + k]; }, [&](int k, int nn) { return T1[nn * 33 + k]; }, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(c1 + lj + 4 * r) * DG_LD + c0 + li] = o[r];
            d4 u = d4{0.0, 0.0, 0.0, 0.0};
            dg_mfma(16, [&](int m, int k) { return S[(c1 + m) * DG_LD + c0 + k]; }, [&](int k, int nn) { return S[(c1 + nn) * DG_LD + c0 + k]; }, u);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(c1 + lj + 4 * r) * DG_LD + c1 + li] -= u[r];
        }
        const int fail2 = dg_factor16(S, c1, T1, 16);
        if (fail < 0) fail = fail2;
        if (fail >= 0 && lane == 0) bad = fail;
        {   // X21 = -X22 (L21 X11a)
            d4 tm = d4{0.0, 0.0, 0.0, 0.0};
            dg_mfma(16, [&](int m, int k) { return S[(c1 + m) * DG_LD + c0 + k]; }, [&](int k, int nn) { return T1[k * 33 + nn]; }, tm);
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(lj + 4 * r) * 33 + li] = tm[r];
            d4 o = d4{0.0, 0.0, 0.0, 0.0};
            dg_mfma(16, [&](int m, int k) { return T1[(16 + m) * 33 + 16 + k]; }, [&](int k, int nn) { return T[k * 33 + nn]; }, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) { T1[(16 + lj + 4 * r) * 33 + li] = -o[r]; T1[(lj + 4 * r) * 33 + 16 + li] = 0.0; }
        }
    };
    // all threads, behind the barrier that ends step 1: T1 -> the form the rest of the kernel reads (diagonal -> dinv, strict lower part of the
    // inverse -> the upper triangle of S, transposed).  (On the factor wave alone: 16 dependent LDS round trips on the critical path.)
    auto publish = [&](int c0) {
        for (int e = tid; e < 32 * 32; e += 256) {
            const int c = e >> 5, k = e & 31;
            const double v = T1[c * 33 + k];
            if (k == c) dinv[c0 + c] = v;
            else if (k < c) S[(c0 + k) * DG_LD + c0 + c] = v;
        }
    };
    // one 16 x 16 tile (I, J) of step 3 after sub-block cp: A22 -= L21 L21^T, rank 32 (a diagonal tile also writes its upper half: those
    // entries belong to later diagonal sub-blocks' upper triangles, which step 1 masks when it reads and overwrites with the inverse)
    auto trail_tile = [&](int cp, int I, int J) {
        const int m0 = cp + 32, i0 = m0 + 16 * I, j0 = m0 + 16 * J;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        dg_mfma(32, [&](int m, int k) { return S[(i0 + m) * DG_LD + cp + k]; }, [&](int k, int nn) { return S[(j0 + nn) * DG_LD + cp + k]; }, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(i0 + lj + 4 * r) * DG_LD + j0 + li] -= acc[r];
    };
    // the inverse's diagonal blocks live as a transposed triangle + a separate diagonal
    auto Xf = [&](int i, int j) -> double { return i == j ? dinv[i] : (i > j ? S[j * DG_LD + i] : 0.0); };
    // one wave: columns [16 h, 16 h + 16) of X_{bi,bj} = -X_ii * sum_{k = bj}^{bi - 1} L_ik X_kj   (block rows in order: X_kj, k < bi, are complete)
    auto xjob = [&](int bi, int bj, int h) {
        const int tj = 16 * h, ri = 32 * bi, rj = 32 * bj;
        d4 w[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ti = 16 * q;
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
            dg_mfma(32, [&](int m, int k) { return S[(ri + ti + m) * DG_LD + rj + k]; }, [&](int k, int nn) { return Xf(rj + k, rj + tj + nn); }, acc);
            dg_mfma((bi - bj - 1) * 32, [&](int m, int k) { return S[(ri + ti + m) * DG_LD + rj + 32 + k]; },
                    [&](int k, int nn) { return S[(rj + tj + nn) * DG_LD + rj + 32 + k]; }, acc);
            w[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ti = 16 * q;
            d4 out = d4{0.0, 0.0, 0.0, 0.0};
            double xa[8];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) xa[s8] = Xf(ri + ti + li, ri + 4 * s8 + lj);
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8)        // B operand of step k = 4 s8: W[k + lj][li] = the accumulator entry that holds row lj + 4 (s8 & 3) of half s8 >> 2
                out = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s8], w[s8 >> 2][s8 & 3], out, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(rj + tj + li) * DG_LD + ri + ti + lj + 4 * r] = -out[r];      // X_ij[ti + row][tj + col], stored transposed
        }
    };
    // block column cb of L (and the zeros above the diagonal) -> global; block row rb of the dense inverse -> global
    auto store_L_cols = [&](int cb, int t, int nt) {
#pragma unroll 8
        for (int e = t; e < 128 * 32; e += nt) {
            const int i = e >> 5, j = 32 * cb + (e & 31);
            const double v = S[i * DG_LD + j];
            if (i < n && j < n) A[(long)i * lda + j] = (j <= i) ? v : 0.0;
        }
    };
    auto store_Linv_rows = [&](int rb, int t, int nt) {
#pragma unroll 8
        for (int e = t; e < 32 * 128; e += nt) {
            const int i = 32 * rb + (e >> 7), j = e & 127;
            const double v = S[min(j, i) * DG_LD + max(j, i)], dv = dinv[i];
            Linv[i * 128 + j] = (i < n && j <= i) ? (i == j ? dv : v) : 0.0;
        }
    };

    // ---- the schedule ---------------------------------------------------------------------------------------------------------------
    const int wt = tid - 64;                        // worker thread index (192 of them)
    for (int kb = 0; kb < nsb; ++kb) {
        const int c0 = 32 * kb;
        if (wave == 0) {
            if (kb == 0) load_rows(0, 32, 5, lane, 64);                            // the first sub-block (the rest arrives beside its factorisation)
            else { trail_tile(c0 - 32, 0, 0); trail_tile(c0 - 32, 1, 0); trail_tile(c0 - 32, 1, 1); }
            step1(c0);
        } else {
            if (kb == 0) {
                for (int e = wt; e < 32 * 96; e += 192) S[(e / 96) * DG_LD + 32 + e % 96] = 0.0;      // rows of the first sub-block right of it: above the diagonal
                load_rows(32, 128, 7, wt, 192);
            } else {
                const int mt = (nl - c0) >> 4;                                       // 16-tiles per side of the trailing matrix behind sub-block kb - 1
                for (int t = 3 + (wave - 1); t < mt * (mt + 1) / 2; t += 3) {        // t = I (I + 1) / 2 + J; tiles 0, 1, 2 (the next sub-block) are wave 0's
                    int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                    while (I * (I + 1) / 2 > t) --I;
                    while ((I + 1) * (I + 2) / 2 <= t) ++I;
                    trail_tile(c0 - 32, I, t - I * (I + 1) / 2);
                }
                // block row kb - 1 of the inverse (its diagonal block was published before the last barrier): 2 (kb - 1) jobs
                for (int job = wave - 1; job < 2 * (kb - 1); job += 3) xjob(kb - 1, job >> 1, job & 1);
                store_L_cols(kb - 1, wt, 192);
                if (kb == 1) store_Linv_rows(0, wt, 192);                            // (block row 0 is its diagonal block)
                else if (kb >= 3) store_Linv_rows(kb - 2, wt, 192);                  // (block row kb - 2 was completed beside step 1 of kb - 1)
            }
        }
        __syncthreads();
        DG_STAMP(2 + 3 * kb);
        if (bad >= 0) break;
        publish(c0);
        if (kb == nsb - 1) break;
        // ---- 2. rows below: L21 = A21 * inv(L11)^T on the matrix cores: 16-row tiles over the waves, both 16-column halves per tile ----
        for (int rt = wave; 16 * rt < nl - c0 - 32; rt += 4) {
            const int i0 = c0 + 32 + 16 * rt;
            d4 o0 = d4{0.0, 0.0, 0.0, 0.0}, o1 = d4{0.0, 0.0, 0.0, 0.0};
            dg_mfma(32, [&](int m, int k) { return S[(i0 + m) * DG_LD + c0 + k]; }, [&](int k, int nn) { return T1[nn * 33 + k]; }, o0);
            dg_mfma(32, [&](int m, int k) { return S[(i0 + m) * DG_LD + c0 + k]; }, [&](int k, int nn) { return T1[(16 + nn) * 33 + k]; }, o1);
            // (the tile's rows are read by this wave only, and all its reads precede these writes: the MFMA results depend on them)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                S[(i0 + lj + 4 * r) * DG_LD + c0 + li] = o0[r];
                S[(i0 + lj + 4 * r) * DG_LD + c0 + 16 + li] = o1[r];
            }
        }
        __syncthreads();
        DG_STAMP(3 + 3 * kb);
    }
    if (bad >= 0) {
        if (tid == 0) atomicCAS(status, -1, block_row0 + bad);
        return;
    }
    // ---- tail: the last block row of the inverse (2 (nsb - 1) jobs over the four waves), then what is not stored yet ----
    __syncthreads();                                    // (the last sub-block's inverse is published)
    // (jobs in the order of their length -- block column 0 takes nsb - 1 products, the last one takes one: the second round goes to the waves
    //  in reverse, so that the wave with the longest first job gets no second one)
    for (int round = 0, base = 0; base < 2 * (nsb - 1); ++round, base += 4) {
        const int job = base + ((round & 1) ? 3 - wave : wave);
        if (job < 2 * (nsb - 1)) xjob(nsb - 1, job >> 1, job & 1);
    }
    __syncthreads();
    DG_STAMP(14);
    store_L_cols(nsb - 1, tid, 256);
    if (nsb == 1) store_Linv_rows(0, tid, 256);
    if (nsb >= 3) store_Linv_rows(nsb - 2, tid, 256);
    if (nsb >= 2) store_Linv_rows(nsb - 1, tid, 256);
    for (int rb = nsb; rb < 4; ++rb) store_Linv_rows(rb, tid, 256);           // rows >= n of the dense inverse are zeros
    DG_STAMP(15);
    DG_STAMP_PRINT();
}

// helpers --------------------------------------------------------------------------------------
__global__ void k_mirror_lower(double* A, int N, long bsA) {           // A[j][i] = A[i][j] for j > i (tile transpose); grid.z = batch
    __shared__ double t[32][33];
    A += (long)blockIdx.z * bsA;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int i = bi * 32 + threadIdx.y, j = bj * 32 + threadIdx.x;
    t[threadIdx.y][threadIdx.x] = (i < N && j < N) ? A[(size_t)i * N + j] : 0.0;
    __syncthreads();
    const int oi = bj * 32 + threadIdx.y, oj = bi * 32 + threadIdx.x;   // transposed tile position
    if (oi < N && oj < N && oj > oi) A[(size_t)oi * N + oj] = t[threadIdx.x][threadIdx.y];
}

// the stored inverses of the diagonal blocks -> the diagonal of the triangular inverse; grid (blocks, batch)
__global__ __launch_bounds__(256) void k_place_dinv(double* A, int N, const double* dinv, long bsA, long bsL) {
    const int jb = blockIdx.x, j0 = jb * 128, n = min(128, N - j0);
    double* dst = A + (long)blockIdx.y * bsA + (long)j0 * N + j0;
    const double* src = dinv + (long)blockIdx.y * bsL + (long)jb * 128 * 128;
    for (int e = threadIdx.x; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        if (i < n && j < n) dst[(long)i * N + j] = src[e];
    }
}

__global__ void k_symmetrize(double* A, int N) {             // A = (A + A^T)/2, written to the lower part
    __shared__ double t[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int ui = bj * 32 + threadIdx.y, uj = bi * 32 + threadIdx.x;   // upper-tile element
    t[threadIdx.y][threadIdx.x] = (ui < N && uj < N) ? A[(size_t)ui * N + uj] : 0.0;
    __syncthreads();
    const int i = bi * 32 + threadIdx.y, j = bj * 32 + threadIdx.x;
    if (i < N && j < N && j <= i) A[(size_t)i * N + j] = 0.5 * (A[(size_t)i * N + j] + t[threadIdx.x][threadIdx.y]);
}


// =============================================================================================
// GP hyper-parameter fit (reference: MAGI_v2._fit_kernel_hparams, magi_v2.py:538-691)
// device pieces: S = Kappa + (sigma^2 + jitter) I, log det from the Cholesky factor, alpha = S^-1 r,
// and the trace terms of d loglik / d(phi1, phi2, sigma^2) = 1/2 tr((alpha alpha^T - S^-1) dS/d.)
// =============================================================================================
__global__ void k_fit_shift(const double* __restrict__ Kap, double* __restrict__ S, int N, double shift, const double* __restrict__ dyn) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)N * N) return;
    Kap += (size_t)blockIdx.y * N * N; S += (size_t)blockIdx.y * N * N;          // (grid.y = component of a batched step)
    if (dyn) dyn += (size_t)blockIdx.y * FD_COUNT;
    if (dyn) shift = dyn[FD_SHIFT];
    const int i = (int)(e / N), j = (int)(e - (size_t)i * N);
    S[e] = Kap[e] + (i == j ? shift : 0.0);
}

// out[0] = sum_i log L_ii (one block)
__global__ __launch_bounds__(256) void k_fit_logdiag(const double* __restrict__ L, int N, double* out) {
    __shared__ double sh[2 * 16];
    L += (size_t)blockIdx.x * N * N; out += (size_t)blockIdx.x * 8;            // (grid.x = component; out blocks are 8 doubles apart)
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < N; i += 256) v[0] += log(L[(size_t)i * N + i]);
    block_sum<1>(v, sh);
    if (threadIdx.x == 0) out[0] = v[0];
}

// alpha = Sinv r ; one wave per row
__global__ __launch_bounds__(256) void k_fit_gemv(const double* __restrict__ A, const double* __restrict__ r, double* __restrict__ y, int N) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    A += (size_t)blockIdx.y * N * N; r += (size_t)blockIdx.y * N; y += (size_t)blockIdx.y * N;
    double s = 0.0;
    for (int j = lane; j < N; j += 64) s = fma(A[(size_t)row * N + j], r[j], s);
    s = wave_sum(s);
    if (lane == 0) y[row] = s;
}

// per-block partials of: [0] sum W_ij Kap_ij  [1] sum W_ij (-pK_ij (t_i - t_j))  [2] tr Sinv  [3] r.alpha  [4] alpha.alpha
// with W = alpha alpha^T - Sinv; grid (ceil(N/64), N/ROWS) ; each block handles 4 rows x 64-wide column strips looped
__global__ __launch_bounds__(256) void k_fit_terms(const double* __restrict__ Kap, const double* __restrict__ pK, const double* __restrict__ Sinv,
                                                   const double* __restrict__ alpha, const double* __restrict__ r, const double* __restrict__ t,
                                                   int N, double* __restrict__ part /* [gridDim.x][5] */) {
    __shared__ double sh[6 * 16];
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    {   // grid.y = component (the time grid t is shared)
        const size_t zn = (size_t)blockIdx.y * N, znn = zn * N;
        Kap += znn; pK += znn; Sinv += znn; alpha += zn; r += zn; part += (size_t)blockIdx.y * gridDim.x * 5;
    }
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (row < N) {
        const double ai = alpha[row], ti = t[row];
        for (int j = lane; j < N; j += 64) {
            const size_t e = (size_t)row * N + j;
            const double w = ai * alpha[j] - Sinv[e];
            v[0] = fma(w, Kap[e], v[0]);
            v[1] = fma(w, -pK[e] * (ti - t[j]), v[1]);
            if (j == row) v[2] += Sinv[e];
        }
        if (lane == 0) { v[3] = r[row] * ai; v[4] = ai * ai; }
    }
    block_sum<5>(v, sh);
    if (threadIdx.x < 5) part[(size_t)blockIdx.x * 5 + threadIdx.x] = v[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_fit_final(const double* __restrict__ part, int nblk, double* out /* [5] */) {
    __shared__ double sh[6 * 16];
    part += (size_t)blockIdx.x * nblk * 5; out += (size_t)blockIdx.x * 8;      // (grid.x = component)
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += part[(size_t)b * 5 + k];
    block_sum<5>(v, sh);
    if (threadIdx.x < 5) out[threadIdx.x] = v[threadIdx.x];
}

// The scalar tail of one Adam step of one component, on the device so that a fit never waits for the host:
// log likelihood and gradient from the reduced sums, TruncatedNormal priors, softplus chain rule, Adam update (tf_keras
// defaults), next hyper-parameters and the Matern constants derived from them.  (reference: magi_v2.py:611-691)
struct FitStepArgs {
    double* dyn;               // [FD_COUNT]
    int* ist;                  // [0] step t (1-based)  [1] failed pivot index (-1: none, sticky)  [2] step of the failure
    const double* out;         // [6] sums of k_fit_final + log-diagonal sum
    const int* status;         // pivot word of this step's Cholesky
    double* trace;             // [iters]: D (loglik + log prior) of the parameters the step started from
    int N, D;
    double nu, lgam_nu, mu_phi2, sd_phi2, sig_loc, lr, jitter;
    // batched step (grid.x = components; blocks of dyn / ist / out / status / trace are FD_COUNT / 4 / 8 / 2 / `iters` apart):
    int batched, iters;
    double mu_phi2_z[8], sd_phi2_z[8], sig_loc_z[8];
};

__device__ inline void fit_derive(double* dyn, double nu, double lgam_nu, double jitter) {
    const double p1 = dyn[FD_PV], p2 = dyn[FD_PV + 1], s2 = dyn[FD_PV + 2];
    dyn[FD_C] = sqrt(2.0 * nu) / p2;
    dyn[FD_LOGA] = log(p1) + (1.0 - nu) * log(2.0) - lgam_nu;
    dyn[FD_DIAGPP] = nu * p1 / ((p2 * p2) * (nu - 1.0));
    dyn[FD_SHIFT] = s2 + jitter;
}

__global__ void k_fit_step(FitStepArgs a) {
    if (threadIdx.x != 0) return;
    if (a.batched) {
        const int z = blockIdx.x;
        a.dyn += (size_t)z * FD_COUNT; a.ist += 4 * z; a.out += 8 * z; a.status += 2 * z; a.trace += (size_t)z * a.iters;
        a.mu_phi2 = a.mu_phi2_z[z]; a.sd_phi2 = a.sd_phi2_z[z]; a.sig_loc = a.sig_loc_z[z];
    } else if (blockIdx.x != 0) return;
    double* dyn = a.dyn;
    const int t = a.ist[0];
    if (t == 0) {                                      // first launch: hyper-parameters from the raw variables
        for (int k = 0; k < 3; ++k) dyn[FD_PV + k] = log1p(exp(dyn[FD_RAW + k]));
        fit_derive(dyn, a.nu, a.lgam_nu, a.jitter);
        a.ist[0] = 1;
        return;
    }
    if (a.ist[1] >= 0) return;
    if (a.status[0] >= 0) { a.ist[1] = a.status[0]; a.ist[2] = t; return; }
    const double D = (double)a.D, sD = sqrt(D);
    const double p1 = dyn[FD_PV], p2 = dyn[FD_PV + 1], s2 = dyn[FD_PV + 2];
    const double* o = a.out;
    const double logdet = 2.0 * o[5];
    const double ll = -0.5 * o[3] - 0.5 * logdet - 0.5 * a.N * log(2.0 * 3.141592653589793);
    const double g3[3] = {0.5 * o[0] / p1, 0.5 * o[1] / p2, 0.5 * (o[4] - o[2])};
    const double sc[3] = {1000.0 * sD, a.sd_phi2 * sD, 1000.0 * sD};
    const double z[3] = {(p1 - 1e-4) / sc[0], (p2 - a.mu_phi2) / sc[1], (s2 - a.sig_loc) / sc[2]};
    const double lp = -0.5 * (z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    a.trace[t - 1] = D * (ll + lp);
    const double b1 = 0.9, b2 = 0.999, eps = 1e-7;
    const double al = a.lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t));
    for (int k = 0; k < 3; ++k) {
        const double raw = dyn[FD_RAW + k];
        const double g = -D * (g3[k] - z[k] / sc[k]) * (1.0 / (1.0 + exp(-raw)));
        const double m = b1 * dyn[FD_M + k] + (1.0 - b1) * g;
        const double v = b2 * dyn[FD_V + k] + (1.0 - b2) * g * g;
        const double nraw = raw - al * m / (sqrt(v) + eps);
        dyn[FD_M + k] = m; dyn[FD_V + k] = v; dyn[FD_RAW + k] = nraw;
        dyn[FD_PV + k] = log1p(exp(nraw));
    }
    fit_derive(dyn, a.nu, a.lgam_nu, a.jitter);
    a.ist[0] = t + 1;
}

struct Linalg {
    magi_handle* h;
    hipStream_t s;
    int N;
    int batch = 1;              // components factorised together: every launch carries them on a grid axis
    long bsA = 0;               // element stride between the components' matrices (potrf / trtri / lauum operate in place on A + z bsA)
    double* dinv = nullptr;     // [batch][nb][128*128]
    double* panel = nullptr;    // [batch][panel_elems]
    long bs_dinv = 0, bs_panel = 0;
    int* status = nullptr;      // [batch][2]
    bool pooled = false;        // dinv / panel belong to the handle's work space
    bool lookahead = false;     // potrf may fork the rank-k updates to the handle's CU-masked stream (dense build only; never inside a capture)
};

// optional per-class timing of the build (MAGI_BUILD_PROFILE=1): HIP events around every launch, so the
// build is serialised and slower -- diagnostics only
enum BuildClass { BC_MATERN = 0, BC_DIAG, BC_PANEL, BC_TRAIL, BC_TRTRI, BC_LAUUM, BC_PROD, BC_FUSED, BC_COUNT };
struct BuildProfile { double flops[BC_COUNT]; double ms[BC_COUNT]; long calls[BC_COUNT]; bool on; hipEvent_t e0, e1; };
BuildProfile g_prof{};

void prof_begin(hipStream_t s) { if (g_prof.on) (void)hipEventRecord(g_prof.e0, s); }
void prof_end(hipStream_t s, int cls, double flops) {
    if (!g_prof.on) return;
    (void)hipEventRecord(g_prof.e1, s);
    (void)hipEventSynchronize(g_prof.e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_prof.e0, g_prof.e1);
    g_prof.ms[cls] += ms; g_prof.flops[cls] += flops; g_prof.calls[cls] += 1;
}

// executed flops of a launch: tiles actually computed x 2 * 128 * 128 * (k range)
double gemm_flops(const GemmArgs& g, int batch) {
    double f = 0.0;
    for (int m0 = 0; m0 < g.M; m0 += GT)
        for (int n0 = 0; n0 < g.N; n0 += GT) {
            if (g.lower_only && m0 + GT - 1 < n0) continue;
            int kb = 0, ke = g.K;
            if (g.kmode == 1) kb = (std::max(m0, n0) / GK) * GK;
            else if (g.kmode == 2) ke = std::min(g.K, m0 + GT);
            else if (g.kmode == 3) kb = (n0 / GK) * GK;
            else if (g.kmode == 4) ke = std::min(g.K, n0 + GT);
            f += 2.0 * std::min(GT, g.M - m0) * std::min(GT, g.N - n0) * std::max(0, ke - kb);
        }
    return f * batch;
}

int launch_gemm(magi_handle* h, hipStream_t s, const GemmArgs& g_in, int batch = 1, int cls = BC_PROD) {
    GemmArgs g = g_in;
    if (g.M <= 0 || g.N <= 0 || batch <= 0) return MAGI_OK;
    const int tY = (g.M + GT - 1) / GT, tX = (g.N + GT - 1) / GT, sY = (tY + 7) / 8, sX = (tX + 7) / 8;
    const int nsuper = g.lower_only ? sY * (sY + 1) / 2 : sY * sX;           // (lower-only: square tile grids)
    // (an option of the handle: tests/test_fullsize_gpu.py forces the super-block order on small tile grids with gemm_remap_min = 1)
    // (a launch's own threshold holds while the option is at its default; an explicit option value -- the tests force the order with 1 -- rules them all)
    const int remap_min = (g.remap_min > 0 && h->opt.gemm_remap_min == MAGI_GEMM_REMAP_MIN_DEFAULT) ? g.remap_min : h->opt.gemm_remap_min;
    g.remap = (nsuper >= remap_min && (!g.lower_only || tY == tX)) ? 1 : 0;
    dim3 grid(g.remap ? ((nsuper + 7) / 8) * 8 * 64 : tY * tX, 1, batch);
    prof_begin(s);
    switch (cls) {
    case BC_PANEL: hipLaunchKernelGGL(k_gemm_f64<BC_PANEL>, grid, dim3(256), 0, s, g); break;
    case BC_TRAIL: hipLaunchKernelGGL(k_gemm_f64<BC_TRAIL>, grid, dim3(256), 0, s, g); break;
    case BC_TRTRI: hipLaunchKernelGGL(k_gemm_f64<BC_TRTRI>, grid, dim3(256), 0, s, g); break;
    case BC_LAUUM: hipLaunchKernelGGL(k_gemm_f64<BC_LAUUM>, grid, dim3(256), 0, s, g); break;
    case BC_FUSED: hipLaunchKernelGGL(k_gemm_f64<BC_FUSED>, grid, dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL(k_gemm_f64<BC_PROD>, grid, dim3(256), 0, s, g); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("gemm launch: ") + hipGetErrorString(e));
    if (g_prof.on) prof_end(s, cls, gemm_flops(g, batch));
    return MAGI_OK;
}

// In-place lower Cholesky of the la.batch matrices A + z la.bsA (N x N row-major, ld = N), all of them in every launch.
// Right-looking over block columns of 128 * opt.potrf_panels (384 = three 128-wide panels by default), left-looking inside a block
// column: the 128 x 128 diagonal blocks are factorised AND inverted by k_diag_chol_inv (one workgroup per component), a panel is
// solved as a GEMM with that inverse, panel c of a block column first takes the rank-128c update of the panels before it (a tall
// 128-wide GEMM), and the trailing matrix takes all the block column's panels in ONE rank-384 SYRK update on the matrix cores --
// three times the flops per byte of the trailing matrix's read-modify-write of a rank-128 sweep (four and more panels: fewer
// bytes still, but a tile of the update then lives longer than the chain's launches it shares the CUs with under look-ahead).
// The inverses of the diagonal blocks stay in la.dinv for the triangular inverse that follows.
// The handle's second stream for the look-ahead of potrf: its hardware queue may use every CU except the first one of each XCD
// (CU-mask bit b names CU b / 8 of XCD b % 8 on this part: tools/micro/cumask.hip), plus the two fork / join events.  nullptr when
// such a stream cannot be had (the factorisation then runs on one stream, as for small grids).
hipStream_t magi_trail_stream(magi_handle* h) {
    if (h->stream_trail) return h->stream_trail;
    if (h->trail_unavailable) return nullptr;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || ncu < 64 || ncu % 8) { h->trail_unavailable = true; return nullptr; }
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int c = 8; c < ncu; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
    hipStream_t st = nullptr, sc = nullptr;
    int pr_lo = 0, pr_hi = 0;
    bool ok = hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi) == hipSuccess && hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) == hipSuccess &&
              hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, pr_hi) == hipSuccess;
    for (int i = 0; i < 3 && ok; ++i) ok = hipEventCreateWithFlags(&h->ev_la[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (st) (void)hipStreamDestroy(st);
        if (sc) (void)hipStreamDestroy(sc);
        for (int i = 0; i < 3; ++i) { if (h->ev_la[i]) (void)hipEventDestroy(h->ev_la[i]); h->ev_la[i] = nullptr; }
        h->trail_unavailable = true;
        return nullptr;
    }
    h->stream_chain = sc;
    h->stream_trail = st;
    return st;
}

int potrf_status(Linalg& la, const char* what, int status_slot = 0) {
    magi_handle* h = la.h;
    std::vector<int> st((size_t)2 * la.batch, -1);
    MAGI_HIP_CHECK(h, hipMemcpyAsync(st.data(), la.status, sizeof(int) * 2 * la.batch, hipMemcpyDeviceToHost, la.s));
    MAGI_HIP_CHECK(h, hipStreamSynchronize(la.s));
    for (int z = 0; z < la.batch; ++z)
        if (st[(size_t)2 * z + status_slot] >= 0)
            return magi_fail(h, MAGI_E_NOTSPD, std::string("Cholesky of ") + what + ": non-positive pivot at index " + std::to_string(st[(size_t)2 * z + status_slot]));
    return MAGI_OK;
}

// defer_status: only enqueue (no host synchronisation); the caller asks potrf_status later
// status_slot: which of the two status words per component records a failed pivot (a deferred caller may have two
// factorisations in flight on the stream)
int potrf(Linalg& la, double* A, const char* what, bool defer_status = false, int status_slot = 0) {
    magi_handle* h = la.h;
    const int N = la.N, NB = 128;
    const size_t lds = (size_t)DG_LDS_DOUBLES * sizeof(double);
    hipStream_t sc = la.s;                      // the stream of the block columns' chains (look-ahead: the handle's high-priority stream)
    for (int z = 0; z < la.batch; ++z)          // status words of this slot <- -1
        MAGI_HIP_CHECK(h, hipMemsetAsync(la.status + 2 * z + status_slot, 0xFF, sizeof(int), la.s));
    auto diag = [&](int j0, int n) {
        prof_begin(sc);
        hipLaunchKernelGGL(k_diag_chol_inv, dim3(la.batch), dim3(256), lds, sc, A + (size_t)j0 * N + j0, (long)N, n,
                           la.dinv + (size_t)(j0 / NB) * 128 * 128, la.status + status_slot, j0, la.bsA, la.bs_dinv, 2);
        prof_end(sc, BC_DIAG, (double)n * n * n * la.batch);        // n^3/3 factor + 2 n^3/3 inverse
    };
    auto panel = [&](int jblk, int j0, int n, int row0) -> int {     // rows >= row0 of block column j0 <- . Linv_jj^T
        (void)jblk;
        GemmArgs g{};
        double* P = A + (size_t)row0 * N + j0;
        g.A = P; g.sAm = N; g.sAk = 1;
        g.B = la.dinv + (size_t)(j0 / NB) * 128 * 128; g.sBn = 128; g.sBk = 1;
        g.C = P; g.ldc = N; g.M = N - row0; g.N = n; g.K = n; g.alpha = 1.0; g.beta = 0.0;
        g.batchA = la.bsA; g.batchB = la.bs_dinv; g.batchC = la.bsA;
        g.remap_min = 24;           // (thin launches: the plain order is the better one until the tile grid is large, profiles/r04_gemm_remap_min_sweep.txt)
        return launch_gemm(h, sc, g, la.batch, BC_PANEL);
    };
    // (a transposed copy of the panels, so that this update reads both operands along their unit-stride dimension, was measured:
    //  no gain -- 34.6 against 34.1 ms at N = 8192 -- the rank-k updates are bound by the tall thin launches inside a block column)
    auto syrk = [&](hipStream_t st, int row0, int ncols, int k0, int K) -> int {     // A[row0.., row0 .. row0 + ncols) -= A[row0.., k0 .. k0+K) A[row0 .. row0+ncols, k0 .. k0+K)^T (lower tiles)
        GemmArgs t{};
        double* P = A + (size_t)row0 * N + k0;
        t.A = P; t.sAm = N; t.sAk = 1;
        t.B = P; t.sBn = N; t.sBk = 1;
        t.C = A + (size_t)row0 * N + row0; t.ldc = N; t.M = N - row0; t.N = ncols; t.K = K; t.alpha = -1.0; t.beta = 1.0;
        t.lower_only = ncols > GT ? 1 : 0;          // (a single tile column has no tile above the diagonal)
        t.batchA = la.bsA; t.batchB = la.bsA; t.batchC = la.bsA;
        t.remap_min = 24;
        return launch_gemm(h, st, t, la.batch, BC_TRAIL);
    };
    // outer block column = NPAN panels of 128 (left-looking inside it: panel c first takes the rank-128c update of the panels
    // before it), then the rank-(128 NPAN) update of the trailing matrix.
    //
    // Look-ahead (grids of opt.potrf_lookahead_min points and more, dense build only): the update is split by columns into U1 = the
    // NEXT block column (all the chain of that column needs) and U2 = the rest, and U2 goes to a second stream while this stream
    // continues with U1 and the next column's chain of tall updates / diagonal blocks / panels:
    //     this stream :  chain(J) | wait U2(J-1) | U1(J) | chain(J+1) | wait U2(J) | U1(J+1) ...
    //     second one  :            ... U2(J-1)   | wait chain(J) | U2(J) ...
    // A plain second stream gains nothing (measured in round 3: 347 against 345 ms): the update's workgroups hold every CU's
    // registers and LDS, the diagonal kernel needs 158 KB of ONE CU's LDS and does not start before the update has drained.  The
    // second stream is therefore created with a CU mask that leaves out one CU of every XCD (magi_trail_stream;
    // tools/micro/cumask.hip: a 158 KB workgroup then starts within 10-20 us beside a kernel that fills the masked queue, against
    // "when that kernel ends" without the mask): the update loses 8 of 256 CUs, the diagonal kernel always finds a home.
    const int NPAN = std::max(1, std::min(h->opt.potrf_panels, 16));
    const int W = NPAN * NB;
    hipStream_t s2 = nullptr;
    if (la.lookahead && !g_prof.on && h->opt.potrf_lookahead_min > 0 && N >= h->opt.potrf_lookahead_min && N > 2 * W) s2 = magi_trail_stream(h);
    if (s2) {           // fork: the chains continue on the high-priority stream (their short launches then win the CUs an update's workgroups free)
        MAGI_HIP_CHECK(h, hipEventRecord(h->ev_la[2], la.s));
        MAGI_HIP_CHECK(h, hipStreamWaitEvent(h->stream_chain, h->ev_la[2], 0));
        MAGI_HIP_CHECK(h, hipStreamWaitEvent(s2, h->ev_la[2], 0));
        sc = h->stream_chain;
    }
    bool u2_pending = false;
    int rc = MAGI_OK;
    for (int j0 = 0; j0 < N && rc == MAGI_OK; j0 += W) {
        for (int c = 0; c < NPAN && rc == MAGI_OK; ++c) {
            const int jc = j0 + c * NB;
            if (jc >= N) break;
            const int nc = std::min(NB, N - jc);
            if (c > 0 && (rc = syrk(sc, jc, nc, j0, c * NB))) break;
            diag(jc, nc);
            if (jc + NB < N) rc = panel(j0, jc, nc, jc + NB);
        }
        const int jn = j0 + W;
        if (rc || jn >= N) break;
        if (!s2) { rc = syrk(sc, jn, N - jn, j0, W); continue; }
        const int w1 = std::min(W, N - jn);
        MAGI_HIP_CHECK(h, hipEventRecord(h->ev_la[0], sc));                                     // chain(J) done
        if (u2_pending) MAGI_HIP_CHECK(h, hipStreamWaitEvent(sc, h->ev_la[1], 0));             // U1(J) touches what U2(J-1) updated
        u2_pending = false;
        if (jn + w1 < N) {
            MAGI_HIP_CHECK(h, hipStreamWaitEvent(s2, h->ev_la[0], 0));
            if ((rc = syrk(s2, jn + w1, N - jn - w1, j0, W))) break;
            MAGI_HIP_CHECK(h, hipEventRecord(h->ev_la[1], s2));
            u2_pending = true;
        }
        rc = syrk(sc, jn, w1, j0, W);
    }
    if (s2) {           // join (also on an error path: nothing is left on the side streams unobserved)
        if (u2_pending) MAGI_HIP_CHECK(h, hipStreamWaitEvent(sc, h->ev_la[1], 0));
        MAGI_HIP_CHECK(h, hipEventRecord(h->ev_la[2], sc));
        MAGI_HIP_CHECK(h, hipStreamWaitEvent(la.s, h->ev_la[2], 0));
    }
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("potrf launch: ") + hipGetErrorString(e));
    return defer_status ? MAGI_OK : potrf_status(la, what, status_slot);
}

// In-place T = L^-1 (lower) from the factors left by potrf, bottom-up over block sizes s = 128, 256, ...:
//   [T11 0; T21 T22] with T21 = -T22 (L21 T11).  Every level is two batched MFMA GEMMs over all block pairs of all
//   components, so the large levels expose (s/128)^2 x pairs tiles instead of the N/128 of a column sweep
//   (measured at N=8192: 429 ms -> see profiles/).  The diagonal 128-blocks come from k_diag_chol_inv.
int trtri(Linalg& la, double* A) {
    magi_handle* h = la.h;
    const int N = la.N, NB = 128;
    const int nb = (N + NB - 1) / NB;
    hipLaunchKernelGGL(k_place_dinv, dim3(nb, la.batch), dim3(256), 0, la.s, A, N, la.dinv, la.bsA, la.bs_dinv);
    for (long s = NB; s < N; s *= 2) {
        const int nfull = (int)(N / (2 * s));
        const long rem = N - (long)nfull * 2 * s;            // rows left after the full pairs
        for (int pass = 0; pass < 2; ++pass) {
            int pairs; long b0; int M2;
            if (pass == 0) { pairs = nfull; b0 = 0; M2 = (int)s; }
            else { pairs = (rem > s) ? 1 : 0; b0 = (long)nfull * 2 * s; M2 = (int)(rem - s); }
            if (pairs <= 0) continue;
            const long pstride = 2 * s * ((long)N + 1);
            double* base = A + b0 * ((long)N + 1);
            GemmArgs g{};   // tmp <- L21 T11                 (T11 lower: B(n,k) = T11[k][n] = 0 for k < n)
            g.A = base + s * N; g.sAm = N; g.sAk = 1;
            g.B = base; g.sBn = 1; g.sBk = N;
            g.C = la.panel; g.ldc = s; g.M = M2; g.N = (int)s; g.K = (int)s; g.alpha = 1.0; g.beta = 0.0; g.kmode = 3;
            g.zinner = pairs;
            // (triangular k ranges: in plain tile order the tiles of a row start their K walks at different places and each streams its own stretch
            //  of the shared operand panel -- this product ran at 0.46 of the MFMA peak at every level, as long as over the FULL K range, beside
            //  0.85 for its twin; in super-block order, which the top level of a large grid now gets (gemm_remap_min 24 -> 10), 0.72 for the class:
            //  profiles/r04_trtri_by_level.txt, r04_gemm_remap_min_sweep.txt)
            g.batchA = pstride; g.batchB = pstride; g.batchC = s * s;
            g.batchA2 = la.bsA; g.batchB2 = la.bsA; g.batchC2 = la.bs_panel;
            int rc = launch_gemm(h, la.s, g, pairs * la.batch, BC_TRTRI);
            if (rc) return rc;
            GemmArgs t{};   // T21 <- -T22 tmp                 (T22 lower: k < m0 + 128)
            t.A = base + s * N + s; t.sAm = N; t.sAk = 1;
            t.B = la.panel; t.sBn = 1; t.sBk = s;
            t.C = base + s * N; t.ldc = N; t.M = M2; t.N = (int)s; t.K = M2; t.alpha = -1.0; t.beta = 0.0; t.kmode = 2;
            t.zinner = pairs;
            t.batchA = pstride; t.batchB = s * s; t.batchC = pstride;
            t.batchA2 = la.bsA; t.batchB2 = la.bs_panel; t.batchC2 = la.bsA;
            if ((rc = launch_gemm(h, la.s, t, pairs * la.batch, BC_TRTRI))) return rc;
        }
    }
    return MAGI_OK;
}

// out + z bs_out = T^T T (full symmetric), T lower in A + z la.bsA
int lauum_tt(Linalg& la, const double* T, double* out, long bs_out) {
    GemmArgs g{};
    const int N = la.N;
    g.A = T; g.sAm = 1; g.sAk = N;      // A(m,k) = T[k][m]
    g.B = T; g.sBn = 1; g.sBk = N;      // B(n,k) = T[k][n]
    g.C = out; g.ldc = N; g.M = N; g.N = N; g.K = N; g.alpha = 1.0; g.beta = 0.0;
    g.lower_only = 1; g.kmode = 1;
    g.batchA = la.bsA; g.batchB = la.bsA; g.batchC = bs_out;
    int rc = launch_gemm(la.h, la.s, g, la.batch, BC_LAUUM);
    if (rc) return rc;
    dim3 grid((N + 31) / 32, (N + 31) / 32, la.batch);
    hipLaunchKernelGGL(k_mirror_lower, grid, dim3(32, 32), 0, la.s, out, N, bs_out);
    return MAGI_OK;
}

// pooled: dinv / panel come from the handle's grow-only work space (the matrix build) instead of fresh allocations
int linalg_init(Linalg& la, magi_handle* h, int N, int batch = 1, long bsA = 0, bool pooled = false) {
    la.h = h; la.s = h->stream; la.N = N; la.batch = batch; la.bsA = bsA; la.pooled = pooled;
    const int nb = (N + 127) / 128;
    la.bs_dinv = (long)nb * 128 * 128;
    la.bs_panel = (long)std::max((size_t)N * 128, (size_t)N * N / 2 + 128 * 128);
    if (pooled) {
        la.dinv = magi_workspace(h, magi_handle::WS_DINV, (size_t)la.bs_dinv * batch);
        la.panel = magi_workspace(h, magi_handle::WS_PANEL, (size_t)la.bs_panel * batch);
        if (!la.dinv || !la.panel) return MAGI_E_HIP;
    } else {
        MAGI_HIP_CHECK(h, hipMalloc(&la.dinv, (size_t)la.bs_dinv * batch * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&la.panel, (size_t)la.bs_panel * batch * sizeof(double)));
    }
    MAGI_HIP_CHECK(h, hipMalloc(&la.status, (size_t)2 * batch * sizeof(int)));
    MAGI_HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_diag_chol_inv), hipFuncAttributeMaxDynamicSharedMemorySize, DG_LDS_DOUBLES * (int)sizeof(double)));
    return MAGI_OK;
}

void linalg_free(Linalg& la) {
    if (!la.pooled) {
        if (la.dinv) (void)hipFree(la.dinv);
        if (la.panel) (void)hipFree(la.panel);
    }
    if (la.status) (void)hipFree(la.status);
    la.dinv = la.panel = nullptr; la.status = nullptr;
}

// Temme's gamma combinations for |mu| <= 1/2, evaluated in 80-bit long double on the host
BesselConsts bessel_consts(double nu) {
    BesselConsts bc{};
    int n = (int)std::floor(nu + 0.5);
    double mu = nu - n;
    if (n < 1) { n = 1; mu = nu - 1.0; }
    bc.n = n; bc.mu = mu;
    const long double m = (long double)mu;
    const long double gp = 1.0L / tgammal(1.0L + m), gm = 1.0L / tgammal(1.0L - m);
    bc.gampl = (double)gp;
    bc.gammi = (double)gm;
    bc.gam2 = (double)(0.5L * (gm + gp));
    if (fabsl(m) < 1e-6L) {
        // (1/G(1-m) - 1/G(1+m)) / (2m) -> -gamma_E + O(m^2)
        bc.gam1 = (double)(-0.57721566490153286060651209L + 0.0L);
    } else {
        bc.gam1 = (double)((gm - gp) / (2.0L * m));
    }
    return bc;
}

int launch_matern(magi_handle* h, const double* dI, int N, double phi1, double phi2, double nu, double* dK, double* dP, double* dPP, const double* dyn = nullptr,
                  int nz = 1) {
    MaternArgs a{};
    a.dyn = dyn;
    a.bsK = (long)N * N; a.bs_dyn = FD_COUNT;      // (used when nz > 1: the batched fit step, hyper-parameters from the components' dyn blocks)
    a.I = dI; a.Kappa = dK; a.pKappa = dP; a.Kappapp = dPP; a.N = N;
    a.phi1 = phi1; a.nu = nu; a.c = std::sqrt(2.0 * nu) / phi2;
    a.logA = std::log(phi1) + (1.0 - nu) * std::log(2.0) - std::lgamma(nu);
    a.diag_pp = nu * phi1 / ((phi2 * phi2) * (nu - 1.0));
    a.bc = bessel_consts(nu);
    a.rows = 64;
    while (a.rows > 4 && (long)((N + 63) / 64) * ((N + a.rows - 1) / a.rows) < 1024) a.rows /= 2;
    dim3 grid((N + 63) / 64, (N + a.rows - 1) / a.rows, nz);
    prof_begin(h->stream);
    hipLaunchKernelGGL(k_matern, grid, dim3(256), 0, h->stream, a);
    prof_end(h->stream, BC_MATERN, 0.0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("matern launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

struct DevBuf {
    double* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n * sizeof(double)); }
};


// One evaluation of the GP marginal log likelihood of component data x (mean mu) and its gradient with
// respect to (phi1, phi2, sigma^2):  S = phi1 R(phi2) + (sigma^2 + jitter) I
//   ll = -1/2 r^T S^-1 r - 1/2 log|S| - N/2 log 2pi ,  d ll/d. = 1/2 tr((a a^T - S^-1) dS/d.) , a = S^-1 r
struct FitWork {          // one component: its own work space and stream, so the D independent fits overlap on the GPU
    DevBuf I, r, Kap, pK, Kpp, S, Sinv, alpha, part, out;
    Linalg la{};
    double* host = nullptr;          // pinned: 6 outputs
    int* host_status = nullptr;      // pinned
    hipStream_t stream = nullptr;
    int N = 0, nblk = 0;
    DevBuf dyn, trace;               // device-resident Adam loop: FitDyn block, per-step objective
    int* ist = nullptr;              // device: step counter, sticky failure
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

// enqueue one evaluation of component `w` on ITS stream (w.la.s); nothing here waits for the device
// dyn != nullptr: hyper-parameters come from the component's device block and nothing is read back (graph capture)
int fit_issue(magi_handle* h, FitWork& w, double phi1, double phi2, double sig2, double nu, double jitter, const double* dyn = nullptr) {
    const int N = w.N;
    hipStream_t keep = h->stream;
    h->stream = w.la.s;                       // the helpers below launch on the handle's stream
    int rc = launch_matern(h, w.I.p, N, phi1, phi2, nu, w.Kap.p, w.pK.p, w.Kpp.p, dyn);
    const size_t nn = (size_t)N * N;
    if (rc == MAGI_OK) {
        hipLaunchKernelGGL(k_fit_shift, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, h->stream, w.Kap.p, w.S.p, N, sig2 + jitter, dyn);
        rc = potrf(w.la, w.S.p, "GP marginal covariance", true);
    }
    if (rc == MAGI_OK) {
        hipLaunchKernelGGL(k_fit_logdiag, dim3(1), dim3(256), 0, h->stream, w.S.p, N, w.out.p + 5);
        rc = trtri(w.la, w.S.p);
    }
    if (rc == MAGI_OK) rc = lauum_tt(w.la, w.S.p, w.Sinv.p, 0);
    if (rc == MAGI_OK) {
        hipLaunchKernelGGL(k_fit_gemv, dim3((N + 3) / 4), dim3(256), 0, h->stream, w.Sinv.p, w.r.p, w.alpha.p, N);
        hipLaunchKernelGGL(k_fit_terms, dim3(w.nblk), dim3(256), 0, h->stream, w.Kap.p, w.pK.p, w.Sinv.p, w.alpha.p, w.r.p, w.I.p, N, w.part.p);
        hipLaunchKernelGGL(k_fit_final, dim3(1), dim3(256), 0, h->stream, w.part.p, w.nblk, w.out.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, std::string("fit launch: ") + hipGetErrorString(e));
    }
    if (rc == MAGI_OK && !dyn) {
        hipError_t e = hipMemcpyAsync(w.host, w.out.p, 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(w.host_status, w.la.status, sizeof(int), hipMemcpyDeviceToHost, h->stream);
        if (e != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, std::string("fit readback: ") + hipGetErrorString(e));
    }
    h->stream = keep;
    return rc;
}

int fit_finish(magi_handle* h, FitWork& w, double phi1, double phi2, double* ll, double* g3) {
    MAGI_HIP_CHECK(h, hipStreamSynchronize(w.la.s));
    if (*w.host_status >= 0)
        return magi_fail(h, MAGI_E_NOTSPD, "Cholesky of GP marginal covariance: non-positive pivot at index " + std::to_string(*w.host_status));
    const double* o = w.host;
    const double logdet = 2.0 * o[5];
    *ll = -0.5 * o[3] - 0.5 * logdet - 0.5 * w.N * std::log(2.0 * 3.141592653589793);
    g3[0] = 0.5 * o[0] / phi1;
    g3[1] = 0.5 * o[1] / phi2;
    g3[2] = 0.5 * (o[4] - o[2]);
    return MAGI_OK;
}

}  // namespace

int magi_fused_operators(magi_handle* h, int N, int D, double* dCs_inout_H, const double* dM, const double* dKs, double* dE) {
    const long nn = (long)N * N;
    GemmArgs g{};     // E = Ks M
    g.A = dKs; g.sAm = N; g.sAk = 1;
    g.B = dM; g.sBn = 1; g.sBk = N;
    g.C = dE; g.ldc = N; g.M = N; g.N = N; g.K = N; g.alpha = 1.0; g.beta = 0.0;
    g.batchA = nn; g.batchB = nn; g.batchC = nn;
    int rc = launch_gemm(h, h->stream, g, D, BC_FUSED);
    if (rc) return rc;
    GemmArgs t{};     // H = M^T E + Cs   (block rows at or below the block diagonal)
    t.A = dM; t.sAm = 1; t.sAk = N;
    t.B = dE; t.sBn = 1; t.sBk = N;
    t.C = dCs_inout_H; t.ldc = N; t.M = N; t.N = N; t.K = N; t.alpha = 1.0; t.beta = 1.0;
    t.batchA = nn; t.batchB = nn; t.batchC = nn;
    t.lower_only = 1;     // H is symmetric and the tile packer reads its lower block triangle only (pack.hip)
    return launch_gemm(h, h->stream, t, D, BC_FUSED);
}

// Adam on the softplus-reparameterised (phi1, phi2, sigma^2) of every component, objective
// D * sum_d [ GP marginal_d + TruncatedNormal priors_d ]  (the [D, D] broadcast of the reference's
// log_prob, magi_v2.py:604-608, 649-665, summed by tape.gradient), tf_keras Adam(lr) defaults.
int magi_fit_hparams_device(magi_handle* h, const double* I, int N, int D, const double* X /* [N][D] */, const double* mu,
                            const double* mu_phi2, const double* sd_phi2, const double* sig_loc, double nu, int iters, double lr,
                            double jitter, double* phi1, double* phi2, double* sig2, double* loss_trace) {
    std::vector<FitWork> ws(D);
    const size_t nn = (size_t)N * N;
    int rc = MAGI_OK;
    auto cleanup = [&]() {
        for (auto& w : ws) {
            linalg_free(w.la);
            if (w.host) (void)hipHostFree(w.host);
            if (w.host_status) (void)hipHostFree(w.host_status);
            if (w.exec) (void)hipGraphExecDestroy(w.exec);
            if (w.graph) (void)hipGraphDestroy(w.graph);
            if (w.ist) (void)hipFree(w.ist);
            if (w.stream) (void)hipStreamDestroy(w.stream);
        }
    };
    const bool batched_fit = !h->opt.fit_host_loop && !h->opt.fit_per_component && D <= 8;
    for (int d = 0; d < D && rc == MAGI_OK && !batched_fit; ++d) {
        FitWork& w = ws[d];
        w.N = N;
        w.nblk = (N + 3) / 4;
        hipError_t e = hipSuccess;
        auto chk = [&](hipError_t x) { if (e == hipSuccess) e = x; };
        chk(w.I.alloc(N)); chk(w.r.alloc(N)); chk(w.Kap.alloc(nn)); chk(w.pK.alloc(nn)); chk(w.Kpp.alloc(nn)); chk(w.S.alloc(nn));
        chk(w.Sinv.alloc(nn)); chk(w.alpha.alloc(N)); chk(w.part.alloc((size_t)w.nblk * 5)); chk(w.out.alloc(8));
        chk(hipHostMalloc(reinterpret_cast<void**>(&w.host), 8 * sizeof(double)));
        chk(hipHostMalloc(reinterpret_cast<void**>(&w.host_status), sizeof(int)));
        chk(hipStreamCreate(&w.stream));
        if (e == hipSuccess) chk(hipMemcpy(w.I.p, I, sizeof(double) * N, hipMemcpyHostToDevice));
        std::vector<double> r((size_t)N);
        for (int i = 0; i < N; ++i) r[i] = X[(size_t)i * D + d] - mu[d];
        if (e == hipSuccess) chk(hipMemcpy(w.r.p, r.data(), sizeof(double) * N, hipMemcpyHostToDevice));
        if (e != hipSuccess) { rc = magi_fail(h, MAGI_E_HIP, std::string("fit setup: ") + hipGetErrorString(e)); break; }
        rc = linalg_init(w.la, h, N);
        w.la.s = w.stream;
    }
    if (rc) { cleanup(); return rc; }

    auto softplus = [](double x) { return std::log1p(std::exp(x)); };
    auto softplus_inv = [](double y) { return std::log(std::expm1(y)); };
    auto sigmoid = [](double x) { return 1.0 / (1.0 + std::exp(-x)); };
    // raw variables, order [phi1(D), phi2(D), sig2(D)] ; Adam state
    std::vector<double> raw(3 * D), m(3 * D, 0.0), v(3 * D, 0.0), grad(3 * D);
    for (int d = 0; d < D; ++d) { raw[d] = softplus_inv(phi1[d]); raw[D + d] = softplus_inv(phi2[d]); raw[2 * D + d] = softplus_inv(sig2[d]); }
    if (batched_fit) {
        // Round 3 -- ONE captured graph per Adam step for ALL components: the components go through the same launches, so every kernel of
        // the step carries them on a grid axis (Matern blocks, shift, the batched Cholesky / inverse of the matrix build, the reductions,
        // the scalar tail), their hyper-parameters in per-component FitDyn blocks FD_COUNT apart.  (Round 1 / 2: one graph per component
        // on four streams that share two hardware queues -- MAGI_FIT_PER_COMPONENT=1 keeps that path for comparison.)
        g_prof.on = false;
        ws.clear();                                                   // (no per-component work spaces on this path)
        DevBuf bI, br, bKap, bpK, bKpp, bS, bSinv, balpha, bpart, bout, bdyn, btrace;
        int* ist = nullptr;
        hipStream_t st = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        Linalg la{};
        const int nblk = (N + 3) / 4;
        hipError_t e = hipSuccess;
        auto chk = [&](hipError_t x) { if (e == hipSuccess) e = x; };
        chk(bI.alloc(N)); chk(br.alloc((size_t)D * N)); chk(bKap.alloc(nn * D)); chk(bpK.alloc(nn * D)); chk(bKpp.alloc(nn * D)); chk(bS.alloc(nn * D));
        chk(bSinv.alloc(nn * D)); chk(balpha.alloc((size_t)D * N)); chk(bpart.alloc((size_t)D * nblk * 5)); chk(bout.alloc((size_t)D * 8));
        chk(bdyn.alloc((size_t)D * FD_COUNT)); chk(btrace.alloc((size_t)D * std::max(iters, 1)));
        chk(hipMalloc(reinterpret_cast<void**>(&ist), (size_t)D * 4 * sizeof(int)));
        chk(hipStreamCreate(&st));
        std::vector<double> rr((size_t)D * N), init((size_t)D * FD_COUNT, 0.0);
        std::vector<int> ist0((size_t)D * 4, 0);
        for (int d = 0; d < D; ++d) {
            for (int i = 0; i < N; ++i) rr[(size_t)d * N + i] = X[(size_t)i * D + d] - mu[d];
            for (int k = 0; k < 3; ++k) init[(size_t)d * FD_COUNT + FD_RAW + k] = raw[(size_t)k * D + d];
            ist0[(size_t)d * 4 + 1] = -1;
        }
        if (e == hipSuccess) chk(hipMemcpy(bI.p, I, sizeof(double) * N, hipMemcpyHostToDevice));
        if (e == hipSuccess) chk(hipMemcpy(br.p, rr.data(), sizeof(double) * rr.size(), hipMemcpyHostToDevice));
        if (e == hipSuccess) chk(hipMemcpy(bdyn.p, init.data(), sizeof(double) * init.size(), hipMemcpyHostToDevice));
        if (e == hipSuccess) chk(hipMemcpy(ist, ist0.data(), sizeof(int) * ist0.size(), hipMemcpyHostToDevice));
        if (e == hipSuccess) chk(hipMemset(btrace.p, 0, (size_t)D * std::max(iters, 1) * sizeof(double)));
        if (e != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, std::string("fit setup: ") + hipGetErrorString(e));
        if (rc == MAGI_OK) { rc = linalg_init(la, h, N, D, (long)nn); la.s = st; }
        FitStepArgs sa{};
        if (rc == MAGI_OK) {
            sa.dyn = bdyn.p; sa.ist = ist; sa.out = bout.p; sa.status = la.status; sa.trace = btrace.p; sa.N = N; sa.D = D;
            sa.nu = nu; sa.lgam_nu = std::lgamma(nu); sa.lr = lr; sa.jitter = jitter; sa.batched = 1; sa.iters = std::max(iters, 1);
            for (int d = 0; d < D; ++d) { sa.mu_phi2_z[d] = mu_phi2[d]; sa.sd_phi2_z[d] = sd_phi2[d]; sa.sig_loc_z[d] = sig_loc[d]; }
            hipLaunchKernelGGL(k_fit_step, dim3(D), dim3(64), 0, st, sa);             // t = 0: derive the first hyper-parameters
            hipStream_t keep = h->stream;
            h->stream = st;                                                           // (the helpers below launch on the handle's stream)
            chk(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            if (e == hipSuccess) {
                rc = launch_matern(h, bI.p, N, 1.0, 1.0, nu, bKap.p, bpK.p, bKpp.p, bdyn.p, D);
                if (rc == MAGI_OK) {
                    hipLaunchKernelGGL(k_fit_shift, dim3((unsigned)((nn + 255) / 256), D), dim3(256), 0, st, bKap.p, bS.p, N, 1.0, bdyn.p);
                    rc = potrf(la, bS.p, "GP marginal covariance", true);
                }
                if (rc == MAGI_OK) {
                    hipLaunchKernelGGL(k_fit_logdiag, dim3(D), dim3(256), 0, st, bS.p, N, bout.p + 5);
                    rc = trtri(la, bS.p);
                }
                if (rc == MAGI_OK) rc = lauum_tt(la, bS.p, bSinv.p, (long)nn);
                if (rc == MAGI_OK) {
                    hipLaunchKernelGGL(k_fit_gemv, dim3((N + 3) / 4, D), dim3(256), 0, st, bSinv.p, br.p, balpha.p, N);
                    hipLaunchKernelGGL(k_fit_terms, dim3(nblk, D), dim3(256), 0, st, bKap.p, bpK.p, bSinv.p, balpha.p, br.p, bI.p, N, bpart.p);
                    hipLaunchKernelGGL(k_fit_final, dim3(D), dim3(256), 0, st, bpart.p, nblk, bout.p);
                    hipLaunchKernelGGL(k_fit_step, dim3(D), dim3(64), 0, st, sa);
                }
                chk(hipStreamEndCapture(st, &graph));
            }
            h->stream = keep;
            if (e == hipSuccess && rc == MAGI_OK) chk(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            if (e != hipSuccess && rc == MAGI_OK) rc = magi_fail(h, MAGI_E_HIP, std::string("fit graph: ") + hipGetErrorString(e));
        }
        for (int t = 1; t <= iters && rc == MAGI_OK; ++t)
            if (hipGraphLaunch(exec, st) != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, "fit graph launch");
        std::vector<double> tr((size_t)std::max(iters, 1) * D, 0.0), fin((size_t)D * FD_COUNT, 0.0);
        std::vector<int> istv((size_t)D * 4, 0);
        if (st) (void)hipStreamSynchronize(st);
        if (rc == MAGI_OK) {
            hipError_t e2 = hipMemcpy(fin.data(), bdyn.p, sizeof(double) * fin.size(), hipMemcpyDeviceToHost);
            if (e2 == hipSuccess) e2 = hipMemcpy(istv.data(), ist, sizeof(int) * istv.size(), hipMemcpyDeviceToHost);
            if (e2 == hipSuccess && iters > 0) e2 = hipMemcpy(tr.data(), btrace.p, (size_t)iters * D * sizeof(double), hipMemcpyDeviceToHost);
            if (e2 != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, std::string("fit: ") + hipGetErrorString(e2));
        }
        for (int d = 0; d < D && rc == MAGI_OK; ++d) {
            if (istv[(size_t)d * 4 + 1] >= 0)
                rc = magi_fail(h, MAGI_E_NOTSPD, "Cholesky of GP marginal covariance (component " + std::to_string(d) + ", Adam step " + std::to_string(istv[(size_t)d * 4 + 2]) +
                               "): non-positive pivot at index " + std::to_string(istv[(size_t)d * 4 + 1]));
            for (int k = 0; k < 3; ++k) raw[(size_t)k * D + d] = fin[(size_t)d * FD_COUNT + FD_RAW + k];
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        linalg_free(la);
        if (ist) (void)hipFree(ist);
        if (st) (void)hipStreamDestroy(st);
        if (rc) return rc;
        if (loss_trace)
            for (int t = 0; t < iters; ++t) {
                double loss = 0.0;
                for (int d = 0; d < D; ++d) loss -= tr[(size_t)d * iters + t];
                loss_trace[t] = loss;
            }
        for (int d = 0; d < D; ++d) { phi1[d] = softplus(raw[d]); phi2[d] = softplus(raw[D + d]); sig2[d] = softplus(raw[2 * D + d]); }
        return MAGI_OK;
    }
    if (!h->opt.fit_host_loop) {
        // Device-resident loop: one Adam step of one component = one captured graph (Matern blocks -> Cholesky -> inverse ->
        // reductions -> k_fit_step) whose inputs live in the component's FitDyn block; the host replays it `iters` times on
        // the component's stream and synchronises once at the end.
        g_prof.on = false;                                            // (its events would synchronise inside the capture)
        const double lgam = std::lgamma(nu);
        for (int d = 0; d < D && rc == MAGI_OK; ++d) {
            FitWork& w = ws[d];
            hipError_t e = hipSuccess;
            auto chk = [&](hipError_t x) { if (e == hipSuccess) e = x; };
            chk(w.dyn.alloc(FD_COUNT)); chk(w.trace.alloc((size_t)std::max(iters, 1)));
            chk(hipMalloc(reinterpret_cast<void**>(&w.ist), 4 * sizeof(int)));
            double init[FD_COUNT] = {0.0};
            for (int k = 0; k < 3; ++k) init[FD_RAW + k] = raw[(size_t)k * D + d];
            const int ist0[4] = {0, -1, 0, 0};
            if (e == hipSuccess) chk(hipMemcpy(w.dyn.p, init, sizeof(init), hipMemcpyHostToDevice));
            if (e == hipSuccess) chk(hipMemcpy(w.ist, ist0, sizeof(ist0), hipMemcpyHostToDevice));
            if (e == hipSuccess) chk(hipMemset(w.trace.p, 0, (size_t)std::max(iters, 1) * sizeof(double)));
            if (e != hipSuccess) { rc = magi_fail(h, MAGI_E_HIP, std::string("fit setup: ") + hipGetErrorString(e)); break; }
            FitStepArgs sa{};
            sa.dyn = w.dyn.p; sa.ist = w.ist; sa.out = w.out.p; sa.status = w.la.status; sa.trace = w.trace.p; sa.N = N; sa.D = D;
            sa.nu = nu; sa.lgam_nu = lgam; sa.mu_phi2 = mu_phi2[d]; sa.sd_phi2 = sd_phi2[d]; sa.sig_loc = sig_loc[d]; sa.lr = lr; sa.jitter = jitter;
            hipLaunchKernelGGL(k_fit_step, dim3(1), dim3(64), 0, w.stream, sa);          // t = 0: derive the first hyper-parameters
            chk(hipStreamBeginCapture(w.stream, hipStreamCaptureModeThreadLocal));
            if (e == hipSuccess) {
                rc = fit_issue(h, w, 1.0, 1.0, 1.0, nu, jitter, w.dyn.p);
                hipLaunchKernelGGL(k_fit_step, dim3(1), dim3(64), 0, w.stream, sa);
                chk(hipStreamEndCapture(w.stream, &w.graph));
            }
            if (e == hipSuccess && rc == MAGI_OK) chk(hipGraphInstantiate(&w.exec, w.graph, nullptr, nullptr, 0));
            if (e != hipSuccess && rc == MAGI_OK) rc = magi_fail(h, MAGI_E_HIP, std::string("fit graph: ") + hipGetErrorString(e));
        }
        for (int t = 1; t <= iters && rc == MAGI_OK; ++t)
            for (int d = 0; d < D && rc == MAGI_OK; ++d)
                if (hipGraphLaunch(ws[d].exec, ws[d].stream) != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, "fit graph launch");
        std::vector<double> tr((size_t)std::max(iters, 1) * D, 0.0);
        for (int d = 0; d < D && rc == MAGI_OK; ++d) {
            FitWork& w = ws[d];
            hipError_t e = hipStreamSynchronize(w.stream);
            double fin[FD_COUNT]; int ist[4] = {0, -1, 0, 0};
            if (e == hipSuccess) e = hipMemcpy(fin, w.dyn.p, sizeof(fin), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(ist, w.ist, sizeof(ist), hipMemcpyDeviceToHost);
            if (e == hipSuccess && iters > 0) e = hipMemcpy(tr.data() + (size_t)d * iters, w.trace.p, (size_t)iters * sizeof(double), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = magi_fail(h, MAGI_E_HIP, std::string("fit: ") + hipGetErrorString(e)); break; }
            if (ist[1] >= 0) {
                rc = magi_fail(h, MAGI_E_NOTSPD, "Cholesky of GP marginal covariance (component " + std::to_string(d) + ", Adam step " + std::to_string(ist[2]) +
                               "): non-positive pivot at index " + std::to_string(ist[1]));
                break;
            }
            for (int k = 0; k < 3; ++k) raw[(size_t)k * D + d] = fin[FD_RAW + k];
        }
        for (auto& w : ws) (void)hipStreamSynchronize(w.stream);
        cleanup();
        if (rc) return rc;
        if (loss_trace)
            for (int t = 0; t < iters; ++t) {
                double loss = 0.0;
                for (int d = 0; d < D; ++d) loss -= tr[(size_t)d * iters + t];
                loss_trace[t] = loss;
            }
        for (int d = 0; d < D; ++d) { phi1[d] = softplus(raw[d]); phi2[d] = softplus(raw[D + d]); sig2[d] = softplus(raw[2 * D + d]); }
        return MAGI_OK;
    }
    // host loop (MAGI_FIT_HOST_LOOP=1): the same arithmetic with the scalar tail and Adam on the host, one synchronisation per step
    const double sD = std::sqrt((double)D);
    const double b1 = 0.9, b2 = 0.999, eps = 1e-7;
    std::vector<double> pv(3 * D);
    for (int t = 1; t <= iters && rc == MAGI_OK; ++t) {
        double loss = 0.0;
        for (int d = 0; d < D && rc == MAGI_OK; ++d) {             // enqueue all components, then collect
            pv[d] = softplus(raw[d]); pv[D + d] = softplus(raw[D + d]); pv[2 * D + d] = softplus(raw[2 * D + d]);
            rc = fit_issue(h, ws[d], pv[d], pv[D + d], pv[2 * D + d], nu, jitter);
        }
        for (int d = 0; d < D && rc == MAGI_OK; ++d) {
            const double p1 = pv[d], p2 = pv[D + d], s2 = pv[2 * D + d];
            double ll, g3[3];
            rc = fit_finish(h, ws[d], p1, p2, &ll, g3);
            if (rc) break;
            // TruncatedNormal(loc, scale, low = 1e-6) priors: only the quadratic depends on the value (magi_v2.py:611-627)
            const double sc1 = 1000.0 * sD, sc2 = sd_phi2[d] * sD, sc3 = 1000.0 * sD;
            const double z1 = (p1 - 1e-4) / sc1, z2 = (p2 - mu_phi2[d]) / sc2, z3 = (s2 - sig_loc[d]) / sc3;
            const double lp = -0.5 * (z1 * z1 + z2 * z2 + z3 * z3);
            loss -= D * (ll + lp);
            grad[d] = -D * (g3[0] - z1 / sc1) * sigmoid(raw[d]);
            grad[D + d] = -D * (g3[1] - z2 / sc2) * sigmoid(raw[D + d]);
            grad[2 * D + d] = -D * (g3[2] - z3 / sc3) * sigmoid(raw[2 * D + d]);
        }
        if (rc) break;
        if (loss_trace) loss_trace[t - 1] = loss;
        const double a = lr * std::sqrt(1.0 - std::pow(b2, t)) / (1.0 - std::pow(b1, t));
        for (int k = 0; k < 3 * D; ++k) {
            m[k] = b1 * m[k] + (1.0 - b1) * grad[k];
            v[k] = b2 * v[k] + (1.0 - b2) * grad[k] * grad[k];
            raw[k] -= a * m[k] / (std::sqrt(v[k]) + eps);
        }
    }
    for (auto& w : ws) (void)hipStreamSynchronize(w.stream);
    cleanup();
    if (rc) return rc;
    for (int d = 0; d < D; ++d) { phi1[d] = softplus(raw[d]); phi2[d] = softplus(raw[D + d]); sig2[d] = softplus(raw[2 * D + d]); }
    return MAGI_OK;
}

int magi_matern_blocks_device(magi_handle* h, const double* I, int N, double phi1, double phi2, double nu, double* Kappa,
                              double* p_Kappa, double* Kappa_pp) {
    if (!(phi1 > 0.0) || !(phi2 > 0.0) || !(nu > 1.0)) return magi_fail(h, MAGI_E_BADARG, "need phi1, phi2 > 0 and nu > 1");
    DevBuf dI, dK, dP, dPP;
    const size_t nn = (size_t)N * N;
    MAGI_HIP_CHECK(h, dI.alloc(N));
    MAGI_HIP_CHECK(h, dK.alloc(nn));
    MAGI_HIP_CHECK(h, dP.alloc(nn));
    MAGI_HIP_CHECK(h, dPP.alloc(nn));
    MAGI_HIP_CHECK(h, hipMemcpy(dI.p, I, sizeof(double) * N, hipMemcpyHostToDevice));
    int rc = launch_matern(h, dI.p, N, phi1, phi2, nu, dK.p, dP.p, dPP.p);
    if (rc) return rc;
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (Kappa) MAGI_HIP_CHECK(h, hipMemcpy(Kappa, dK.p, nn * sizeof(double), hipMemcpyDeviceToHost));
    if (p_Kappa) MAGI_HIP_CHECK(h, hipMemcpy(p_Kappa, dP.p, nn * sizeof(double), hipMemcpyDeviceToHost));
    if (Kappa_pp) MAGI_HIP_CHECK(h, hipMemcpy(Kappa_pp, dPP.p, nn * sizeof(double), hipMemcpyDeviceToHost));
    return MAGI_OK;
}

int magi_build_profile_get(const magi_handle* h, double* flops, double* ms, long* calls) {
    for (int i = 0; i < BC_COUNT; ++i) { flops[i] = g_prof.flops[i]; ms[i] = g_prof.ms[i]; calls[i] = g_prof.calls[i]; }
    // one more row, filled by EVERY dense build of this handle: its two factorisations as a whole (N^3 / 3 flops per matrix)
    flops[BC_COUNT] = h ? h->potrf_wall_flops : 0.0; ms[BC_COUNT] = h ? h->potrf_wall_ms : 0.0; calls[BC_COUNT] = h && h->potrf_wall_ms > 0.0 ? 2 : 0;
    return BC_COUNT + 1;
}

int magi_ensure_dense(magi_handle* h, int N, int D) {
    if (h->dDense[0] && h->dense_N == N && h->dense_D == D) return MAGI_OK;
    const size_t nn = (size_t)N * N;
    for (int k = 0; k < 3; ++k) { if (h->dDense[k]) (void)hipFree(h->dDense[k]); h->dDense[k] = nullptr; }
    h->dense_N = h->dense_D = 0;
    for (int k = 0; k < 3; ++k) {
        hipError_t e = hipMalloc(&h->dDense[k], nn * D * sizeof(double));
        if (e == hipSuccess) e = hipMemset(h->dDense[k], 0, nn * D * sizeof(double));         // components never built read as zero matrices
        if (e != hipSuccess) {       // all three or none: a partial set must not pass the callers' `dDense[0] != null` guards
            (void)hipGetLastError();
            for (int j = 0; j < 3; ++j) { if (h->dDense[j]) (void)hipFree(h->dDense[j]); h->dDense[j] = nullptr; }
            return magi_fail(h, MAGI_E_HIP, std::string("dense stacks (3 x ") + std::to_string(nn * D * sizeof(double) >> 20) + " MiB): " + hipGetErrorString(e));
        }
    }
    h->dense_N = N; h->dense_D = D;
    return MAGI_OK;
}

// Eqn. 6 matrices of the components sel[0 .. n_sel) (phi1 / phi2 indexed like sel) into the handle's dense stacks
int magi_build_dense_device(magi_handle* h, const double* I, int N, int D, int n_sel, const int* sel, const double* phi1, const double* phi2,
                            double nu) {
    g_prof.on = h->opt.build_profile != 0;
    if (g_prof.on) {
        if (!g_prof.e0) { (void)hipEventCreate(&g_prof.e0); (void)hipEventCreate(&g_prof.e1); }
        for (int i = 0; i < BC_COUNT; ++i) { g_prof.flops[i] = 0.0; g_prof.ms[i] = 0.0; g_prof.calls[i] = 0; }
    }
    for (int d = 0; d < n_sel; ++d) {
        if (!(phi1[d] > 0.0) || !(phi2[d] > 0.0)) return magi_fail(h, MAGI_E_BADARG, "phi1 and phi2 must be positive");
        if (sel[d] < 0 || sel[d] >= D) return magi_fail(h, MAGI_E_BADARG, "component index out of range");
    }
    int rc0 = magi_ensure_dense(h, N, D);
    if (rc0) return rc0;
    const size_t nn = (size_t)N * N;
    // The D components are independent and go through the SAME sequence of launches, so they are batched: every kernel of the
    // chain carries all components of a group on a grid axis (one stream, D times fewer launches, D times the tiles per launch:
    // the one-workgroup diagonal-block kernels of the components run side by side and the thin panels fill the GPU together).
    // B < D components per group when memory is short (or MAGI_BUILD_SERIAL is set): the groups follow each other.
    DevBuf dI;
    MAGI_HIP_CHECK(h, dI.alloc(N));
    MAGI_HIP_CHECK(h, hipMemcpy(dI.p, I, sizeof(double) * N, hipMemcpyHostToDevice));
    // the batched launches address the components of a group at a constant stride: a group = a run of consecutive indices
    int B = n_sel;
    {
        size_t free_b = 0, total_b = 0;
        const size_t per = (3 * nn + std::max((size_t)N * 128, nn / 2 + 128 * 128) + (size_t)((N + 127) / 128) * 128 * 128) * sizeof(double);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            B = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_sel, (free_b / 10 * 9) / per));
        if (h->opt.build_serial) B = 1;
    }
    struct { double* p; } Kap{magi_workspace(h, magi_handle::WS_KAP, nn * B)}, P{magi_workspace(h, magi_handle::WS_P, nn * B)},
        PP{magi_workspace(h, magi_handle::WS_PP, nn * B)};
    if (!Kap.p || !P.p || !PP.p) return MAGI_E_HIP;
    Linalg la{};
    int rc = linalg_init(la, h, N, B, (long)nn, true);
    la.lookahead = true;
    for (int i = 0; i < 4 && rc == MAGI_OK; ++i)
        if (!h->ev_pw[i] && hipEventCreate(&h->ev_pw[i]) != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, "build: event");
    h->potrf_wall_ms = 0.0; h->potrf_wall_flops = 0.0;
    std::vector<int> status((size_t)2 * n_sel, -1);
    for (int d0 = 0; d0 < n_sel && rc == MAGI_OK;) {
        int nb = 1;
        while (nb < B && d0 + nb < n_sel && sel[d0 + nb] == sel[d0 + nb - 1] + 1) ++nb;
        la.batch = nb;
        for (int z = 0; z < nb && rc == MAGI_OK; ++z)
            rc = launch_matern(h, dI.p, N, phi1[d0 + z], phi2[d0 + z], nu, Kap.p + nn * z, P.p + nn * z, PP.p + nn * z);
        double* Cd = h->dDense[0] + nn * sel[d0];
        double* Md = h->dDense[1] + nn * sel[d0];
        double* Kd = h->dDense[2] + nn * sel[d0];
        // Kappa = L L^T, T = L^-1, C^-1 = Kappa^-1 = T^T T   (Kappa is consumed)            magi_v2.py:818, 126
        if (!rc) (void)hipEventRecord(h->ev_pw[0], la.s);
        if (!rc) rc = potrf(la, Kap.p, "Kappa", true, 0);
        if (!rc) (void)hipEventRecord(h->ev_pw[1], la.s);
        if (!rc) rc = trtri(la, Kap.p);
        if (!rc) rc = lauum_tt(la, Kap.p, Cd, (long)nn);
        // Wt = (p_Kappa T^T)^T = T p_Kappa^T = -T p_Kappa   (p_Kappa is exactly antisymmetric: the sign of s - t is its only odd
        // factor, magi_v2.py:798-802).  T lower: k < m0 + 128.  Kept in the K^-1 output buffer until that is formed.  Formed as the
        // TRANSPOSE of W so that the two products below read both operands along their unit-stride dimension (A m-fast,
        // B n-fast): the k-fast x k-fast forms of W T and W W^T ran at 0.55 of the MFMA peak.
        if (!rc) {
            GemmArgs g{};
            g.A = Kap.p; g.sAm = N; g.sAk = 1;
            g.B = P.p; g.sBn = 1; g.sBk = N;
            g.C = Kd; g.ldc = N; g.M = N; g.N = N; g.K = N; g.alpha = -1.0; g.beta = 0.0; g.kmode = 2;
            g.batchA = (long)nn; g.batchB = (long)nn; g.batchC = (long)nn;
            rc = launch_gemm(h, la.s, g, nb);
        }
        // m = p_Kappa Kappa^-1 = W T = Wt^T T  (B(n,k) = T[k][n] = 0 for k < n)              magi_v2.py:819
        if (!rc) {
            GemmArgs g{};
            g.A = Kd; g.sAm = 1; g.sAk = N;
            g.B = Kap.p; g.sBn = 1; g.sBk = N;
            g.C = Md; g.ldc = N; g.M = N; g.N = N; g.K = N; g.alpha = 1.0; g.beta = 0.0; g.kmode = 3;
            g.batchA = (long)nn; g.batchB = (long)nn; g.batchC = (long)nn;
            rc = launch_gemm(h, la.s, g, nb);
        }
        // K = Kappa_pp - p_Kappa Kappa^-1 Kappa_p = Kappa_pp - W W^T = Kappa_pp - Wt^T Wt  (Kappa_p = -p_Kappa = p_Kappa^T,
        // magi_v2.py:805, 820): a SYRK on the lower tiles -- exactly symmetric by construction; its Cholesky reads the lower
        // triangle only
        if (!rc) {
            GemmArgs g{};
            g.A = Kd; g.sAm = 1; g.sAk = N;
            g.B = Kd; g.sBn = 1; g.sBk = N;
            g.C = PP.p; g.ldc = N; g.M = N; g.N = N; g.K = N; g.alpha = -1.0; g.beta = 1.0; g.lower_only = 1;
            g.batchA = (long)nn; g.batchB = (long)nn; g.batchC = (long)nn;
            rc = launch_gemm(h, la.s, g, nb);
        }
        // K^-1                                                                                magi_v2.py:128
        if (!rc) (void)hipEventRecord(h->ev_pw[2], la.s);
        if (!rc) rc = potrf(la, PP.p, "K_d", true, 1);
        if (!rc) (void)hipEventRecord(h->ev_pw[3], la.s);
        if (!rc) rc = trtri(la, PP.p);
        if (!rc) rc = lauum_tt(la, PP.p, Kd, (long)nn);
        if (!rc && hipMemcpyAsync(&status[(size_t)2 * d0], la.status, (size_t)2 * nb * sizeof(int), hipMemcpyDeviceToHost, la.s) != hipSuccess)
            rc = magi_fail(h, MAGI_E_HIP, "build: status readback");
        if (!rc && hipStreamSynchronize(la.s) != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, "build: synchronize");
        if (!rc) {      // whole factorisations on the device clock (with the per-launch profile on they are serialised, like everything else)
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, h->ev_pw[0], h->ev_pw[1]) == hipSuccess && hipEventElapsedTime(&b, h->ev_pw[2], h->ev_pw[3]) == hipSuccess) {
                h->potrf_wall_ms += (double)a + (double)b;
                h->potrf_wall_flops += 2.0 * nb * ((double)N * N * N / 3.0);
            }
        }
        d0 += nb;
    }
    linalg_free(la);
    if (rc) return rc;
    for (int d = 0; d < n_sel; ++d)
        for (int k = 0; k < 2; ++k)
            if (status[(size_t)2 * d + k] >= 0)
                return magi_fail(h, MAGI_E_NOTSPD, std::string("Cholesky of ") + (k ? "K_d" : "Kappa") + " (component " + std::to_string(sel[d]) +
                                 "): non-positive pivot at index " + std::to_string(status[(size_t)2 * d + k]));
    return MAGI_OK;
}

int magi_build_matrices_device(magi_handle* h, const double* I, int N, int D, const double* phi1, const double* phi2,
                               double nu, int bandsize, double* C_inv, double* m, double* K_inv) {
    std::vector<int> sel(D);
    for (int d = 0; d < D; ++d) sel[d] = d;
    int rc = magi_build_dense_device(h, I, N, D, D, sel.data(), phi1, phi2, nu);
    if (rc) return rc;
    const size_t nn = (size_t)N * N;
    if (C_inv) MAGI_HIP_CHECK(h, hipMemcpy(C_inv, h->dDense[0], nn * D * sizeof(double), hipMemcpyDeviceToHost));
    if (m) MAGI_HIP_CHECK(h, hipMemcpy(m, h->dDense[1], nn * D * sizeof(double), hipMemcpyDeviceToHost));
    if (K_inv) MAGI_HIP_CHECK(h, hipMemcpy(K_inv, h->dDense[2], nn * D * sizeof(double), hipMemcpyDeviceToHost));
    return magi_pack_matrices(h, N, D, bandsize, h->dDense[0], h->dDense[1], h->dDense[2]);
}
