// Internal declarations shared by the HIP translation units of libmagi_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/magi_hip.h"

#ifndef MAGI_MAX_D
#define MAGI_MAX_D 4        // compiled-in drifts have D <= 4, P <= 5; a library built for a traced drift with 5..8 components sets 8
#endif
#ifndef MAGI_MAX_P
#define MAGI_MAX_P 6        // a library built for a traced drift with 7 or 8 parameters sets 8
#endif
#define MAGI_MAX_DEPTH 12   // checkpoint slots for the iterative NUTS U-turn checks
#define MAGI_TAIL_THREADS 256
#define MAGI_WAVE 64

// a[d] for a runtime d without a runtime-indexed register array (which would live in scratch)
#if MAGI_MAX_D == 4
#define MAGI_SEL_D(a, d) (((d) == 0) ? (a)[0] : ((d) == 1) ? (a)[1] : ((d) == 2) ? (a)[2] : (a)[3])
#else
template <int N>
__host__ __device__ __forceinline__ double sel_n(const double (&a)[N], int d) {
    double v = a[N - 1];
#pragma unroll
    for (int k = N - 2; k >= 0; --k) v = (d == k) ? a[k] : v;
    return v;
}
#define MAGI_SEL_D(a, d) sel_n(a, d)
#endif

// ------------------------------------------------------------------------------------------
// Device-visible problem description (passed by value to kernels)
// ------------------------------------------------------------------------------------------
struct DevProblem {
    int N, D, P;
    int ld;        // row pitch (doubles) of the four matrices; even
    int band;      // -1 dense; else half-width b, rows hold columns [i-b, i+b]
    int drift;
    int ND;        // N*D
    int dim;       // N*D + D + P
    int dimp;      // dim rounded up to a multiple of 8
    double beta_inv;                       // 1.0 / beta              (magi_v2.py:348)
    double mu[MAGI_MAX_D];                 // magi_v2.py:114
    double N_ds[MAGI_MAX_D];               // magi_v2.py:53
    double LB[MAGI_MAX_D];                 // magi_v2.py:300
    const double* Csym;   // [D][N][ld]  (C^-1 + C^-T)/2
    const double* M;      // [D][N][ld]  m
    const double* Mt;     // [D][N][ld]  m^T
    const double* Ksym;   // [D][N][ld]  (K^-1 + K^-T)/2
    const double* yobs;   // [D][N], NaN = not observed        (magi_v2.py:96-100)
    // single-phase ("fused") operators of the sampler (see leap.hip / pack.hip):
    //   FH = Csym + m^T Ksym m (symmetric),  FE = Ksym m,  FK = Ksym (symmetric)
    // stored as packed TB x TB blocks, one per task: the lower block triangle of FH and FK (a block
    // serves A x and A^T x in one pass), every block of FE (serves FE xc and FE^T f) -- 2 N^2 D values
    // instead of the 4 N^2 D of four row-major stacks.
    const double* tiles;   // [n_tasks][TB][TB]
    const int* tasks;      // [n_tasks][4] = {d, kind, bi, bj}
    int n_tasks;
    const int* stasks;     // [n_stasks][8] = {d, kind, bi, bj, tile, tile1, kind1, 0}: the tasks of k_stream_sep -- the tasks above with the two
                           // diagonal blocks FH_bb, FK_bb of a component (half the work each) paired into one (tile1 >= 0), so all are equal
    int n_stasks;
    int nb;        // blocks per side, ceil(N / TB)
    int Np;        // nb * TB
    int wb;        // block half-band: blocks with |bi - bj| <= wb exist (nb when dense)
    int bandf;     // -1 dense; else half-width 3b of the fused operators (products of band-b matrices)
};

// Per-chain vector slots (each dimp doubles) -------------------------------------------------
enum VecSlot {
    V_Q = 0,     // position buffer 0: X comp-major [D][N], sigma_pre[D], theta_pre[P]
    V_Q1,        // position buffer 1 (the sampler ping-pongs: plan.cur selects the one being evaluated)
    V_P,         // half-step momentum buffer 0
    V_P1,        // half-step momentum buffer 1
    V_PLEAF,     // momentum at the leaf (after the full step)
    V_G,         // UNtempered gradient of L at the evaluated position
    V_CX,        // Csym * xc           [D][N]
    V_R,         // f - m xc            [D][N]
    V_KR,        // Ksym * r            [D][N]   (fused path: Ksym * f)
    V_ETF,       // (validation kernels only)
    V_F,         // (unused)
    V_PL, V_QL, V_GL,       // left end of the trajectory
    V_PR, V_QR, V_GR,       // right end
    V_CANDQ, V_CANDG,       // trajectory-level proposal
    V_SUBQ, V_SUBG,         // subtree-level proposal
    V_RHO, V_RHOSUB,        // momentum sums (generalised U-turn)
    V_CKP0,                 // MAGI_MAX_DEPTH checkpoint momenta
    V_CKRHO0 = V_CKP0 + MAGI_MAX_DEPTH,
    V_COUNT = V_CKRHO0 + MAGI_MAX_DEPTH
};

enum ChainPhase { PH_INIT = 0, PH_LEAF = 1, PH_IDLE = 2,
                  PH_MERGED = 3,    // a subtree's end is being merged by the point phase (boundary op): the decisions of the next slot read its U-turn sums
                  PH_DRAWN = 4 };   // a transition's momentum is being drawn by the point phase: the decisions of the next slot read p.p

// Boundary work of a chain at a subtree / transition end: state-sized, element by element -- done by the POINT kernel of the slot (all its
// workgroups; leap_point.h: boundary_block) on the decisions' order (LeafPlan::vop), not by the decisions' single workgroup.
enum BoundaryOp {
    VOP_ENDS = 1,        // the finished subtree's last leaf becomes the trajectory's end `vdir`; rho += rho_sub; U-turn sums rho.p_other, rho.p_end
    VOP_CAND = 2,        // trajectory proposal <- subtree proposal (or the last leaf itself: VOP_TAKE_LEAF)
    VOP_TAKE_LEAF = 4,
    VOP_OUT = 8,         // sample row `vout` <- trajectory proposal
    VOP_DRAW = 16,       // transition `step_k` starts: momentum ~ N(0, I) at both ends, ends <- proposal, sum p.p
    VOP_DOUBLE = 32      // a doubling starts from end `ndir`: first half / full step into buffer `cur` (+ operand mirrors, parameter block)
};

// Per-chain scalar state of the device-resident sampler ---------------------------------------
struct ChainCtl {
    int phase;
    int k;            // index of the transition in progress (0-based, burn-in included)
    int depth;        // doublings finished in this transition
    int it;           // leaf index inside the current subtree
    int nsteps;       // leaves in the current subtree (1 << depth)
    int dir;          // +1 forward / -1 backward
    int leaf_ctr;     // leaves so far in this transition (RNG counter)
    int cont;         // subtree still growing
    int nd;           // no divergence so far (subtree running value)
    int not_div;      // no divergence, committed at subtree ends
    int is_accepted;
    int lf_count;     // leapfrogs committed in this transition
    int sub_lf;
    int da_step;
    int done_epoch;   // last run epoch in which this chain reported itself idle
    int cur;          // position buffer holding the state being evaluated
    long long chain_id;
    long long total_leapfrogs;
    double eps;         // step size of this transition (> 0)
    double beta_k;      // temperature of this transition          (magi_v2.py:855)
    double beta_cache;  // temperature the cached target/grad of the proposal were computed at
    double init_energy;
    double e_sum_sub, e_sum;
    double cand_L, cand_energy, cand_weight, cand_bfac;
    double sub_L, sub_energy, sub_weight;
    double L_cur;
    double LL, LR, bfacL, bfacR;
    double bfac_cur;
    double da_step_size, da_error_sum, da_log_avg, da_log_shrink;
};

struct SamplerCfgDev {
    int total, burnin, n_adapt, max_depth, mode, hmc_L, anneal, stale;
    double step_size, target_accept, max_energy_diff, min_temp;
    unsigned long long seed;
};

struct GlobalCtl {
    int done_chains;   // chains idle in the current run epoch
    int n_chains;
    int all_done;      // every chain idle: the mat-vec kernels and the tail return immediately
    int stop_k;        // chains pause once they have finished this many transitions
    int epoch;         // incremented by the host at every magi_sampler_run
    int slots;         // leapfrog slots issued in this run before every chain was idle (counted by chain 0's decision workgroup)
    int pad[2];
};

// What the streaming kernel's epilogue must do for a chain in the next leapfrog slot; written by the
// tail (or the host for API calls), read by every workgroup of k_leap_*.
struct LeafPlan {
    int active;        // 0: leave the chain's state alone (idle chain)
    int skip;          // 1: the state in buffer `cur` was set up by the decisions while this slot's stream was already
                       //    running on something else: the point phase does nothing, the NEXT stream evaluates `cur` as is
    int leaf;          // 1: complete the leapfrog (momentum, sums, checkpoint, speculative next state)
    int cur;           // position / half-momentum buffer being evaluated (0/1)
    int even;          // write the U-turn checkpoint of this leaf
    int ck_slot;       // checkpoint slot to write (popcount of the leaf index)
    int nchk;          // U-turn checks folded into the epilogue (0..4)
    int chk_slot[4];   // their checkpoint slots
    unsigned leaf_ctr; // Philox counters of the two uniforms this leaf may need (STREAM_LEAF / STREAM_MERGE)
    unsigned depth;
    unsigned step_k;
    unsigned chain_id;
    double hs;         // 0.5 * eps * beta_k (signed)
    double eps;        // signed step
    unsigned long long seed;
    // boundary op (vop != 0: `leaf` is 0; with VOP_DOUBLE active = skip = 1 and `cur` is the buffer the NEXT stream evaluates, else active = 0)
    int vop;           // BoundaryOp bits
    int vleaf;         // position buffer of the subtree's last leaf (VOP_ENDS, VOP_TAKE_LEAF)
    int vdir;          // +1 / -1: the end the finished subtree grew (VOP_ENDS)
    int ndir;          // +1 / -1: the end the next doubling starts from (VOP_DOUBLE)
    long long vout;    // row of ch.samples (VOP_OUT)
    int sub_copy;      // leaf plans: the PREVIOUS leaf (position buffer cur ^ 1, gradient V_G as it stands) was accepted as the subtree's proposal:
                       // the point phase copies it to V_SUBQ / V_SUBG before it overwrites both (the D + P parameter entries: the decisions)
};

constexpr int MAGI_TB = 128;  // block edge of the packed single-phase operators
enum TileKind { TK_FH = 0, TK_FK = 1, TK_FE = 2 };
enum TileVec { TV_HX = 0, TV_EX = 1, TV_ETF = 2, TV_KF = 3 };

constexpr int PART_K = (1 + MAGI_MAX_D + MAGI_MAX_P + 1 + 8 + 3) / 4 * 4 < 24 ? 24 : (1 + MAGI_MAX_D + MAGI_MAX_P + 1 + 8 + 3) / 4 * 4;   // per workgroup and chain: t12, ss[D], tp[P], pp, 4 x (dA, dB)

struct DevChains {
    double* vec;          // [n_chains][V_COUNT][dimp]
    ChainCtl* ctl;        // [n_chains]
    LeafPlan* plan;       // [2][n_chains] ring by slot parity: the point phase of slot s executes plan[s & 1]
    double* part;         // [n_chains][PART_K][n_wg] partial sums of the point kernel
    int n_wg;             // workgroups along the grid axis of k_point
    double* tpart;        // [n_chains][4 (hx, ex, etf, kf)][D][nb][Np] block partials of the streaming kernel
    double* xop;          // [2 slot parities][ceil(n_chains / 16)][D][Np][16 chains]: the positions the stream of slot s evaluates, in the operand
                          // order of the matrix-core streaming kernel (a block's slice of 16 chains is contiguous), at xop[s & 1]: written by whoever
                          // sets that state up during slot s - 1 (the point phase's speculative next leaf, the decisions' new subtree start,
                          // k_prepare) -- the stream needs no plan to find it
    double* vop;          // separable drifts: operand mirror of the streaming kernel (SepLayout), written like xop; tpart then holds product slots
    double* par;          // [n_chains][PAR_COUNT] transformed parameters of the state in V_Q
    GlobalCtl* gctl;
    int n_chains;
    int mc;               // the matrix-core streaming kernel k_stream_mc serves this batch (three or more chains of a NON-separable drift): keep xop up to date
    int sep;              // the separable streaming kernel k_stream_sep serves this batch (separable drift and three or more chains, or
                          // MAGI_STREAM_FAMILY=mc): keep vop up to date, tpart holds product slots (SepLayout)
    // outputs
    double* samples;      // [n_chains][num_results][dimp]
    double* d_step_size;  // diag arrays [n_chains][total]
    double* d_lar;
    double* d_target;
    double* d_energy;
    double* d_beta;
    int* d_leapfrogs;
    int* d_depth;
    int* d_flags;         // bit0 has_divergence, bit1 reach_max_depth, bit2 is_accepted
};

// element (slot parity b, chain, component d, grid index i) of the operand-order mirror.  A grid point's chains are contiguous:
// 16 per point (one 128-B line), or 8 (64 B) when there are at most 8 chains -- every stream workgroup reads its operand slices of
// all chains from here, and with half-empty lines that traffic (L2 hits, in one burst at the kernel's start) is twice what it must be.
__host__ __device__ inline int xop_width(int n_chains) { return n_chains <= 8 ? 8 : 16; }
__host__ __device__ inline size_t xop_off(const DevProblem& pb, int n_chains, int b, int chain, int d, int i) {
    const size_t groups = (size_t)((n_chains + 15) >> 4);
    return ((((size_t)b * groups + (size_t)(chain >> 4)) * pb.D + d) * (size_t)pb.Np + i) * xop_width(n_chains) + (chain & 15);
}

__host__ __device__ inline size_t vec_off(const DevProblem& pb, int chain, int slot) {
    return ((size_t)chain * V_COUNT + (size_t)slot) * (size_t)pb.dimp;
}

// ------------------------------------------------------------------------------------------
// Kernel arguments are ~0.5 KB of structs passed by value; the compiler fetches their cache lines lazily, next to the
// first use, and every first touch of a line is a scalar-cache miss on the critical path of a 5 us kernel (k_point had
// four such waits in a row before its first vector load).  Touching every line up front turns them into one.
// ------------------------------------------------------------------------------------------
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const unsigned __attribute__((address_space(4))) * ka_ptr;
    ka_ptr ka = (ka_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned t = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) t |= ka[o / 4];
    asm volatile("" ::"s"(t));
#endif
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), identical to oracle/magi_oracle.py::philox4x32
// ------------------------------------------------------------------------------------------
struct Philox4 { unsigned int x, y, z, w; };

__host__ __device__ inline Philox4 philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2,
                                                 unsigned int c3, unsigned long long key) {
    unsigned int k0 = (unsigned int)(key & 0xFFFFFFFFull), k1 = (unsigned int)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = 0xD2511F53ull * (unsigned long long)c0;
        unsigned long long p1 = 0xCD9E8D57ull * (unsigned long long)c2;
        unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
        unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
        unsigned int n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__host__ __device__ inline double u01_53(unsigned int hi, unsigned int lo) {
    unsigned long long x = (((unsigned long long)hi << 32) | (unsigned long long)lo) >> 11;
    return ((double)x + 0.5) * 1.1102230246251565e-16;   // 2^-53
}

enum { STREAM_MOMENTUM = 0, STREAM_DIRECTION = 1, STREAM_LEAF = 2, STREAM_MERGE = 3, STREAM_HMC = 4 };

__device__ inline double rng_uniform(unsigned int index, unsigned int step, unsigned int chain,
                                     unsigned int stream, unsigned long long key) {
    Philox4 r = philox4x32_10(index, step, chain, stream, key);
    return u01_53(r.x, r.y);
}

// ------------------------------------------------------------------------------------------
// scalar helpers.  fp64 transcendentals are ~100-300-instruction sequences; the sampler's tail is
// a run-once-per-launch kernel whose speed is set by instruction fetch, so each function exists
// ONCE per translation unit (noinline) instead of once per call site.
// ------------------------------------------------------------------------------------------
static __device__ __noinline__ double m_log(double x) { return log(x); }
static __device__ __noinline__ double m_exp(double x) { return exp(x); }
static __device__ __noinline__ double m_log1p(double x) { return log1p(x); }

__device__ inline double softplus_ref(double x) { return m_log(1.0 + m_exp(x)); }   // magi_v2.py:318
__device__ inline double sigmoid(double x) { return 1.0 / (1.0 + m_exp(-x)); }

__device__ inline double logaddexp(double a, double b) {
    if (a == -INFINITY && b == -INFINITY) return -INFINITY;
    double t = a - b;
    if (t > 0) return a + m_log1p(m_exp(-t));
    if (t <= 0) return b + m_log1p(m_exp(t));
    return t;   // NaN
}

__device__ inline double temperature(int step, double min_temp) {   // magi_v2.py:833-835
    return fmax(1.0 / m_log((double)step + 2.0), min_temp);
}

__device__ inline double rng_normal_elem(unsigned int e, unsigned int step, unsigned int chain,
                                         unsigned long long key) {
    Philox4 r = philox4x32_10(e >> 1, step, chain, STREAM_MOMENTUM, key);
    double u1 = u01_53(r.x, r.y), u2 = u01_53(r.z, r.w);
    double rad = sqrt(-2.0 * m_log(u1));
    double ang = 2.0 * 3.141592653589793 * u2;
    return (e & 1) ? rad * sin(ang) : rad * cos(ang);
}

// both normals of Philox pair j (elements 2 j and 2 j + 1): the same values as rng_normal_elem, with the Philox block, the
// logarithm and the square root evaluated once per pair instead of once per element
__device__ inline void rng_normal_pair(unsigned int j, unsigned int step, unsigned int chain, unsigned long long key, double& z0, double& z1) {
    Philox4 r = philox4x32_10(j, step, chain, STREAM_MOMENTUM, key);
    double u1 = u01_53(r.x, r.y), u2 = u01_53(r.z, r.w);
    double rad = sqrt(-2.0 * m_log(u1));
    double ang = 2.0 * 3.141592653589793 * u2;
    z0 = rad * cos(ang);
    z1 = rad * sin(ang);
}

// ------------------------------------------------------------------------------------------
// Transformed parameters of a state (magi_v2.py:318-323), computed ONCE per state by whoever
// writes V_Q (fp64 exp/log are ~100-instruction sequences: keeping them out of the per-row
// epilogues and out of the 1024-thread reduce is worth tens of microseconds per gradient).
// ------------------------------------------------------------------------------------------
enum ParOff { PAR_TH = 0, PAR_SGT = 8, PAR_LJT = 16, PAR_SIG2 = 24, PAR_SGS = 24 + MAGI_MAX_D, PAR_LJS = 24 + 2 * MAGI_MAX_D,
              PAR_LOG2PIS = 24 + 3 * MAGI_MAX_D,
              PAR_ULEAF = 56, PAR_UMERGE = 57 /* (unused) */,
              PAR_COUNT = 64 };

// entry j of the parameter block: j < D -> sigma_pre[j], else theta_pre[j - D]
// lb_tab: optional LDS copy of pb.LB (a per-lane select on the kernel-argument array compiles into a scratch table = a global
// round trip; the sampler's decision workgroup stages N_ds and LB in LDS once)
__device__ inline void compute_par_entry(const DevProblem& pb, int j, double pre, double* par, bool with_log2pis = true, const double* lb_tab = nullptr) {
    const double e = m_exp(pre);
    const double sp = m_log(1.0 + e);          // magi_v2.py:318-319
    const double sg = e / (1.0 + e);           // d softplus / d pre (= 1/(1+exp(-pre)) up to rounding)
    if (j < pb.D) {
        const double lb = lb_tab ? lb_tab[j] : MAGI_SEL_D(pb.LB, j);
        const double s2 = sp + lb;
        par[PAR_SIG2 + j] = s2;
        par[PAR_SGS + j] = sg;
        par[PAR_LJS + j] = pre - sp;
        if (with_log2pis) par[PAR_LOG2PIS + j] = m_log(2.0 * 3.141592653589793 * s2);      // (leap_reduce evaluates it at the consumer)
    } else {
        const int p = j - pb.D;
        par[PAR_TH + p] = sp;
        par[PAR_SGT + p] = sg;
        par[PAR_LJT + p] = pre - sp;
    }
}

// ------------------------------------------------------------------------------------------
// Drifts.  x[] = state at one grid point, th[] = softplus'ed parameters.
// A library built with -DMAGI_USER_DRIFT_HEADER="<file>" carries ONE drift traced from the caller's f_vec
// (magi_v2_amd/drift.py, jit.py) and instantiates the sampler's kernels for it alone.
// ------------------------------------------------------------------------------------------
template <int DRIFT> struct DriftT;
#ifdef MAGI_USER_DRIFT_HEADER
#include MAGI_USER_DRIFT_HEADER
#define MAGI_DRIFT_DISPATCH(drift, CALL) do { CALL(MAGI_DRIFT_USER); } while (0)
#else
#define MAGI_DRIFT_DISPATCH(drift, CALL)                         \
    do {                                                         \
        switch (drift) {                                         \
        case MAGI_DRIFT_SEIR3: CALL(MAGI_DRIFT_SEIR3); break;    \
        case MAGI_DRIFT_SEIR4: CALL(MAGI_DRIFT_SEIR4); break;    \
        default: CALL(MAGI_DRIFT_SIRW); break;                   \
        }                                                        \
    } while (0)
#endif
__device__ __forceinline__ double drift_f(int drift, int d, const double* x, const double* th) {
    switch (drift) {
#ifdef MAGI_USER_DRIFT_HEADER
    case MAGI_DRIFT_USER: return user_drift_f(d, x, th);
#endif
    case MAGI_DRIFT_SEIR3: {   // vignette.ipynb cell 3
        double E = x[0], I = x[1], R = x[2], S = 1.0 - ((E + I) + R);
        if (d == 0) return (th[0] * S * I) - (th[2] * E);
        if (d == 1) return (th[2] * E) - (th[1] * I);
        return th[1] * I;
    }
    case MAGI_DRIFT_SEIR4: {
        double S = x[0], E = x[1], I = x[2];
        if (d == 0) return -th[0] * S * I;
        if (d == 1) return th[0] * S * I - th[2] * E;
        if (d == 2) return th[2] * E - th[1] * I;
        return th[1] * I;
    }
    default: {                 // MAGI_DRIFT_SIRW, test_magi_script.py:19-45
        double S = x[0], I = x[1], R = x[2], W = x[3];
        if (d == 0) return -th[0] * S * I + th[4] * W;
        if (d == 1) return th[0] * S * I - th[1] * I;
        if (d == 2) return th[1] * I - th[2] * R + th[3] * I * W;
        return th[2] * R - th[3] * I * W - th[4] * W;
    }
    }
}

// sum_d' g[d'] * d f_d' / d x_d   (column d of the Jacobian contracted with g)
__device__ __forceinline__ double drift_jt_g(int drift, int d, const double* x, const double* th, const double* g) {
    switch (drift) {
#ifdef MAGI_USER_DRIFT_HEADER
    case MAGI_DRIFT_USER: {
        double c[MAGI_MAX_D];
#pragma unroll
        for (int k = 0; k < MAGI_MAX_D; ++k) c[k] = 0.0;
        user_drift_jt(x, th, g, c, nullptr);
        return MAGI_SEL_D(c, d);
    }
#endif
    case MAGI_DRIFT_SEIR3: {
        double E = x[0], I = x[1], R = x[2], S = 1.0 - ((E + I) + R);
        double b = th[0], gm = th[1], s = th[2];
        if (d == 0) return g[0] * (-b * I - s) + g[1] * s;
        if (d == 1) return g[0] * (b * S - b * I) - g[1] * gm + g[2] * gm;
        return -g[0] * b * I;
    }
    case MAGI_DRIFT_SEIR4: {
        double S = x[0], I = x[2];
        double b = th[0], gm = th[1], s = th[2];
        if (d == 0) return (g[1] - g[0]) * b * I;
        if (d == 1) return (g[2] - g[1]) * s;
        if (d == 2) return (g[1] - g[0]) * b * S + (g[3] - g[2]) * gm;
        return 0.0;
    }
    default: {
        double S = x[0], I = x[1], W = x[3];
        double be = th[0], ph = th[1], xi = th[2], ch = th[3], ka = th[4];
        if (d == 0) return (g[1] - g[0]) * be * I;
        if (d == 1) return -g[0] * be * S + g[1] * (be * S - ph) + g[2] * (ph + ch * W) - g[3] * ch * W;
        if (d == 2) return (g[3] - g[2]) * xi;
        return g[0] * ka + g[2] * ch * I + g[3] * (-ch * I - ka);
    }
    }
}

// out[p] += sum_d g[d] * d f_d / d theta_p
__device__ __forceinline__ void drift_tt_g_acc(int drift, const double (&x)[MAGI_MAX_D], const double (&th)[MAGI_MAX_P],
                                               const double (&g)[MAGI_MAX_D], double (&out)[MAGI_MAX_P]) {
    // every case fills the same five scalars so the accumulation below keeps static indices
    double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0, o4 = 0.0;
#ifdef MAGI_USER_DRIFT_HEADER
    if (drift == MAGI_DRIFT_USER) {
        double t[MAGI_MAX_P];
#pragma unroll
        for (int k = 0; k < MAGI_MAX_P; ++k) t[k] = 0.0;
        user_drift_jt(x, th, g, nullptr, t);
#pragma unroll
        for (int k = 0; k < MAGI_MAX_P; ++k) out[k] += t[k];
        return;
    }
#endif
    switch (drift) {
    case MAGI_DRIFT_SEIR3: {
        const double E = x[0], I = x[1], R = x[2], S = 1.0 - ((E + I) + R);
        o0 = g[0] * S * I;
        o1 = (g[2] - g[1]) * I;
        o2 = (g[1] - g[0]) * E;
        break;
    }
    case MAGI_DRIFT_SEIR4: {
        const double S = x[0], E = x[1], I = x[2];
        o0 = (g[1] - g[0]) * S * I;
        o1 = (g[3] - g[2]) * I;
        o2 = (g[2] - g[1]) * E;
        break;
    }
    default: {
        const double S = x[0], I = x[1], R = x[2], W = x[3];
        o0 = (g[1] - g[0]) * S * I;
        o1 = (g[2] - g[1]) * I;
        o2 = (g[3] - g[2]) * R;
        o3 = (g[2] - g[3]) * I * W;
        o4 = (g[0] - g[3]) * W;
        break;
    }
    }
    out[0] += o0; out[1] += o1; out[2] += o2; out[3] += o3; out[4] += o4;
}

// ------------------------------------------------------------------------------------------
// Compile-time drifts for the sampler's kernels (D, P known -> no guards, no switch, small code)
// ------------------------------------------------------------------------------------------
// Separable form (SEP): f_d(x, theta) = sum_{k < nbasis(d)} coef_{d,k}(theta) phi_{d,k}(x).  The sampler's streaming kernel applies the
// operators to the theta-FREE basis vectors phi_{d,k}(x) (written by whoever sets a state up) and the point phase combines the products
// with coef(theta): the stream then needs no parameter that hangs on a global sum of the slot before (leap.hip, k_stream_sep).
// basis / coefs fill [D][NBMAX] with zeros in unused entries.
template <> struct DriftT<MAGI_DRIFT_SEIR3> {
    static constexpr int D = 3, P = 3;
    static constexpr bool SEP = true;
    static constexpr int NBMAX = 2;
    __host__ __device__ static constexpr int nbasis(int d) { return d == 2 ? 1 : 2; }
    static __device__ __forceinline__ void basis(const double (&x)[3], double (&ph)[3][2]) {
        const double E = x[0], I = x[1], R = x[2], S = 1.0 - ((E + I) + R);
        ph[0][0] = S * I; ph[0][1] = E;
        ph[1][0] = E; ph[1][1] = I;
        ph[2][0] = I; ph[2][1] = 0.0;
    }
    static __device__ __forceinline__ void coefs(const double (&th)[3], double (&c)[3][2]) {
        c[0][0] = th[0]; c[0][1] = -th[2];
        c[1][0] = th[2]; c[1][1] = -th[1];
        c[2][0] = th[1]; c[2][1] = 0.0;
    }
    static __device__ __forceinline__ void f(const double (&x)[3], const double (&th)[3], double (&o)[3]) {
        const double E = x[0], I = x[1], R = x[2], S = 1.0 - ((E + I) + R);
        o[0] = (th[0] * S * I) - (th[2] * E);
        o[1] = (th[2] * E) - (th[1] * I);
        o[2] = th[1] * I;
    }
    static __device__ __forceinline__ double f1(int d, const double (&x)[3], const double (&th)[3]) {
        const double E = x[0], I = x[1], R = x[2], S = 1.0 - ((E + I) + R);
        if (d == 0) return (th[0] * S * I) - (th[2] * E);
        if (d == 1) return (th[2] * E) - (th[1] * I);
        return th[1] * I;
    }
    // c[d] = sum_d' g[d'] df_d'/dx_d ; t[p] += sum_d g[d] df_d/dtheta_p
    static __device__ __forceinline__ void jt(const double (&x)[3], const double (&th)[3], const double (&g)[3], double (&c)[3], double (&t)[3]) {
        const double E = x[0], I = x[1], R = x[2], S = 1.0 - ((E + I) + R);
        const double b = th[0], gm = th[1], s = th[2];
        c[0] = g[0] * (-b * I - s) + g[1] * s;
        c[1] = g[0] * (b * S - b * I) - g[1] * gm + g[2] * gm;
        c[2] = -g[0] * b * I;
        t[0] += g[0] * S * I;
        t[1] += (g[2] - g[1]) * I;
        t[2] += (g[1] - g[0]) * E;
    }
};

template <> struct DriftT<MAGI_DRIFT_SEIR4> {
    static constexpr int D = 4, P = 3;
    static constexpr bool SEP = true;
    static constexpr int NBMAX = 2;
    __host__ __device__ static constexpr int nbasis(int d) { return (d == 0 || d == 3) ? 1 : 2; }
    static __device__ __forceinline__ void basis(const double (&x)[4], double (&ph)[4][2]) {
        const double S = x[0], E = x[1], I = x[2];
        ph[0][0] = S * I; ph[0][1] = 0.0;
        ph[1][0] = S * I; ph[1][1] = E;
        ph[2][0] = E; ph[2][1] = I;
        ph[3][0] = I; ph[3][1] = 0.0;
    }
    static __device__ __forceinline__ void coefs(const double (&th)[3], double (&c)[4][2]) {
        c[0][0] = -th[0]; c[0][1] = 0.0;
        c[1][0] = th[0]; c[1][1] = -th[2];
        c[2][0] = th[2]; c[2][1] = -th[1];
        c[3][0] = th[1]; c[3][1] = 0.0;
    }
    static __device__ __forceinline__ void f(const double (&x)[4], const double (&th)[3], double (&o)[4]) {
        const double S = x[0], E = x[1], I = x[2];
        o[0] = -th[0] * S * I;
        o[1] = th[0] * S * I - th[2] * E;
        o[2] = th[2] * E - th[1] * I;
        o[3] = th[1] * I;
    }
    static __device__ __forceinline__ double f1(int d, const double (&x)[4], const double (&th)[3]) {
        const double S = x[0], E = x[1], I = x[2];
        if (d == 0) return -th[0] * S * I;
        if (d == 1) return th[0] * S * I - th[2] * E;
        if (d == 2) return th[2] * E - th[1] * I;
        return th[1] * I;
    }
    static __device__ __forceinline__ void jt(const double (&x)[4], const double (&th)[3], const double (&g)[4], double (&c)[4], double (&t)[3]) {
        const double S = x[0], E = x[1], I = x[2];
        const double b = th[0], gm = th[1], s = th[2];
        c[0] = (g[1] - g[0]) * b * I;
        c[1] = (g[2] - g[1]) * s;
        c[2] = (g[1] - g[0]) * b * S + (g[3] - g[2]) * gm;
        c[3] = 0.0;
        t[0] += (g[1] - g[0]) * S * I;
        t[1] += (g[3] - g[2]) * I;
        t[2] += (g[2] - g[1]) * E;
    }
};

template <> struct DriftT<MAGI_DRIFT_SIRW> {
    static constexpr int D = 4, P = 5;
    static constexpr bool SEP = true;
    static constexpr int NBMAX = 3;
    __host__ __device__ static constexpr int nbasis(int d) { return d < 2 ? 2 : 3; }
    static __device__ __forceinline__ void basis(const double (&x)[4], double (&ph)[4][3]) {
        const double S = x[0], I = x[1], R = x[2], W = x[3];
        ph[0][0] = S * I; ph[0][1] = W; ph[0][2] = 0.0;
        ph[1][0] = S * I; ph[1][1] = I; ph[1][2] = 0.0;
        ph[2][0] = I; ph[2][1] = R; ph[2][2] = I * W;
        ph[3][0] = R; ph[3][1] = I * W; ph[3][2] = W;
    }
    static __device__ __forceinline__ void coefs(const double (&th)[5], double (&c)[4][3]) {
        c[0][0] = -th[0]; c[0][1] = th[4]; c[0][2] = 0.0;
        c[1][0] = th[0]; c[1][1] = -th[1]; c[1][2] = 0.0;
        c[2][0] = th[1]; c[2][1] = -th[2]; c[2][2] = th[3];
        c[3][0] = th[2]; c[3][1] = -th[3]; c[3][2] = -th[4];
    }
    static __device__ __forceinline__ void f(const double (&x)[4], const double (&th)[5], double (&o)[4]) {
        const double S = x[0], I = x[1], R = x[2], W = x[3];
        o[0] = -th[0] * S * I + th[4] * W;
        o[1] = th[0] * S * I - th[1] * I;
        o[2] = th[1] * I - th[2] * R + th[3] * I * W;
        o[3] = th[2] * R - th[3] * I * W - th[4] * W;
    }
    static __device__ __forceinline__ double f1(int d, const double (&x)[4], const double (&th)[5]) {
        const double S = x[0], I = x[1], R = x[2], W = x[3];
        if (d == 0) return -th[0] * S * I + th[4] * W;
        if (d == 1) return th[0] * S * I - th[1] * I;
        if (d == 2) return th[1] * I - th[2] * R + th[3] * I * W;
        return th[2] * R - th[3] * I * W - th[4] * W;
    }
    static __device__ __forceinline__ void jt(const double (&x)[4], const double (&th)[5], const double (&g)[4], double (&c)[4], double (&t)[5]) {
        const double S = x[0], I = x[1], R = x[2], W = x[3];
        const double be = th[0], ph = th[1], xi = th[2], ch = th[3], ka = th[4];
        c[0] = (g[1] - g[0]) * be * I;
        c[1] = -g[0] * be * S + g[1] * (be * S - ph) + g[2] * (ph + ch * W) - g[3] * ch * W;
        c[2] = (g[3] - g[2]) * xi;
        c[3] = g[0] * ka + g[2] * ch * I + g[3] * (-ch * I - ka);
        t[0] += (g[1] - g[0]) * S * I;
        t[1] += (g[2] - g[1]) * I;
        t[2] += (g[3] - g[2]) * R;
        t[3] += (g[2] - g[3]) * I * W;
        t[4] += (g[0] - g[3]) * W;
    }
};

// ------------------------------------------------------------------------------------------
// Storage of the separable streaming path (k_stream_sep, leap.hip).
//   product slots of component d (tpart[chain][slot][other block][Np]; unused k stay zero):
//       d PS + 0: hx = FH xc     + 1: ex = FE xc     + 2 + k: etf_k = FE^T phi_{d,k}     + 2 + NBMAX + k: kf_k = FK phi_{d,k}
//   operand mirror vop[slot parity][chain group][d][plane][Np / 2][16][2]: plane 0 = xc_d of the group's chains (column = chain & 15),
//       plane 1 + z = basis group z: 16 matrix-core columns = (k, chain); with at most 8 chains (CW = 8) two basis functions share a
//       plane (k = 2 z + (col >> 3), chain = col & 7), else one per plane (k = z, chain = col).  Entries never written stay zero.
//       Inside a plane the grid points come in PAIRS: element (point i, column c) sits at vop_elem(i, c) = ((i >> 1) 16 + c) 2 + (i & 1), so
//       that the two points a lane of the streaming kernel needs for one matrix-core operand register pair are ONE 16-byte load (round 4:
//       the prologue of k_stream_sep is bound by the number of vector-memory instructions its CU has to issue, not by their latency).
// ------------------------------------------------------------------------------------------
template <int DRIFT> struct SepLayout {
    using DR = DriftT<DRIFT>;
    static constexpr int D = DR::D, NBMAX = DR::NBMAX;
    static constexpr int PS = 2 + 2 * NBMAX, PS_TOTAL = D * PS;
    __host__ __device__ static constexpr int slot_hx(int d) { return d * PS; }
    __host__ __device__ static constexpr int slot_ex(int d) { return d * PS + 1; }
    __host__ __device__ static constexpr int slot_etf(int d, int k) { return d * PS + 2 + k; }
    __host__ __device__ static constexpr int slot_kf(int d, int k) { return d * PS + 2 + NBMAX + k; }
    __host__ __device__ static constexpr bool slot_used(int slot) {
        const int d = slot / PS, r = slot - d * PS;
        return r < 2 || ((r - 2) % NBMAX) < DR::nbasis(d);
    }
    __host__ __device__ static constexpr int gz(int cw) { return (NBMAX * cw + 15) / 16; }          // basis planes per component
    __host__ __device__ static constexpr int planes(int cw) { return 1 + gz(cw); }
    __host__ __device__ static constexpr int plane_of(int cw, int k) { return 1 + (cw == 8 ? (k >> 1) : k); }
    __host__ __device__ static constexpr int col_of(int cw, int k, int chain_local) { return cw == 8 ? ((k & 1) * 8 + chain_local) : chain_local; }
};
// start of the (even) point i's pair inside plane `plane` (i even: block starts are multiples of 128)
__host__ __device__ inline size_t vop_off(int D, int planes, int Np, int groups, int b, int group, int d, int plane, int i) {
    return (((((size_t)b * groups + group) * D + d) * planes + plane) * (size_t)Np + i) * 16;
}
// element (point i, column c) relative to a plane's start
__host__ __device__ inline size_t vop_elem(int i, int c) { return ((size_t)(i >> 1) * 16 + (size_t)c) * 2 + (size_t)(i & 1); }

// ------------------------------------------------------------------------------------------
// reductions: 64-lane butterfly, then a fixed-order sum over the block's waves (deterministic)
// ------------------------------------------------------------------------------------------
// All cross-lane traffic stays on the VALU (DPP row operations + gfx950's v_permlane{16,32}_swap);
// the ds_bpermute path that __shfl_xor lowers to costs an LDS round trip per 32-bit half per step.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// butterfly inside each row of 16 lanes: every lane ends with its row's sum
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
    v = row16_sum(v);
    {   // rows 0<->1, 2<->3
        unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {   // lower 32 <-> upper 32
        unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v;
}

// Block-wide sums of K values; every thread returns with all K totals.  sh: (K + 1) * 16 doubles.
// Stage 1: per-wave butterflies, one partial per wave.  Stage 2: wave 0 adds the <= 16 partials of
// each value with a row butterfly and publishes K totals (fixed order -> deterministic).
template <int K>
__device__ inline void block_sum(double (&v)[K], double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) sh[k * 16 + wave] = s;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double x = (lane < nw) ? sh[k * 16 + (lane & 15)] : 0.0;
            if (lane >= 16) x = 0.0;
            x = row16_sum(x);
            if (lane == 0) sh[K * 16 + k] = x;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = sh[K * 16 + k];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Reduce step of one gradient evaluation (all threads of the block must call it).
// Reads V_Q, V_CX, V_R, V_KR of the chain, writes the sigma_pre / theta_pre parts of V_G and
// returns the UNtempered log posterior L and t1..t4 of magi_v2.py:332-345.
// ------------------------------------------------------------------------------------------
struct FinalizeOut { double L, t1, t2, t3, t4; };

template <int NSEL>
__device__ __forceinline__ double select_lane(const double (&a)[NSEL], int base, int n, int j) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < NSEL; ++k)
        if (k >= base && k < base + n && j == k - base) v = a[k];
    return v;
}

// sh: [(2 + MAX_D + MAX_P) * 16] doubles, shs: [8] doubles of LDS scratch
__device__ inline FinalizeOut finalize_gradient(const DevProblem& pb, double* vb, const double* par, double* sh, double* shs) {
    const int N = pb.N, D = pb.D, P = pb.P, ND = pb.ND, dimp = pb.dimp;
    const double* q = vb + (size_t)V_Q * dimp;
    const double* Cx = vb + (size_t)V_CX * dimp;
    const double* r = vb + (size_t)V_R * dimp;
    const double* Kr = vb + (size_t)V_KR * dimp;
    double* g = vb + (size_t)V_G * dimp;

    double th[MAGI_MAX_P];
#pragma unroll
    for (int p = 0; p < MAGI_MAX_P; ++p) th[p] = (p < P) ? par[PAR_TH + p] : 0.0;

    constexpr int K = 2 + MAGI_MAX_D + MAGI_MAX_P;
    double t1s = 0.0, t2s = 0.0, ss[MAGI_MAX_D], tp[MAGI_MAX_P];
#pragma unroll
    for (int d = 0; d < MAGI_MAX_D; ++d) ss[d] = 0.0;
#pragma unroll
    for (int p = 0; p < MAGI_MAX_P; ++p) tp[p] = 0.0;

    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        double x[MAGI_MAX_D], g2[MAGI_MAX_D];
#pragma unroll
        for (int d = 0; d < MAGI_MAX_D; ++d) {
            if (d < D) {
                const int idx = d * N + i;
                const double xv = q[idx];
                x[d] = xv;
                t1s = fma(xv - pb.mu[d], Cx[idx], t1s);
                const double kr = Kr[idx];
                t2s = fma(r[idx], kr, t2s);
                g2[d] = 2.0 * kr;
                const double y = pb.yobs[idx];
                if (!isnan(y)) { const double df = xv - y; ss[d] = fma(df, df, ss[d]); }
            } else { x[d] = 0.0; g2[d] = 0.0; }
        }
        drift_tt_g_acc(pb.drift, x, th, g2, tp);
    }
    double red[K];
    red[0] = t1s;
    red[1] = t2s;
#pragma unroll
    for (int d = 0; d < MAGI_MAX_D; ++d) red[2 + d] = ss[d];
#pragma unroll
    for (int p = 0; p < MAGI_MAX_P; ++p) red[2 + MAGI_MAX_D + p] = tp[p];
    block_sum<K>(red, sh);

    // scalar part: one wave, lane j <-> parameter j (no transcendental left: see compute_par_entry)
    if (threadIdx.x < 64) {
        const int j = threadIdx.x;
        double t3 = 0.0, t4 = 0.0, lj = 0.0;
        if (j < D) {
            const double s2 = par[PAR_SIG2 + j], sg = par[PAR_SGS + j];
            const double ssd = select_lane<K>(red, 2, MAGI_MAX_D, j);
            const double nds = MAGI_SEL_D(pb.N_ds, j);
            t3 = nds * par[PAR_LOG2PIS + j];
            t4 = ssd * (1.0 / s2);
            lj = par[PAR_LJS + j];
            const double dsig = nds / s2 - ssd / (s2 * s2);
            g[ND + j] = -0.5 * dsig * sg + (1.0 - sg);
        } else if (j < D + P) {
            const int p = j - D;
            const double sg = par[PAR_SGT + p];
            const double tpp = select_lane<K>(red, 2 + MAGI_MAX_D, MAGI_MAX_P, p);
            lj = par[PAR_LJT + p];
            g[ND + D + p] = -0.5 * pb.beta_inv * tpp * sg + (1.0 - sg);
        }
        t3 = wave_sum(t3);
        t4 = wave_sum(t4);
        lj = wave_sum(lj);
        if (j == 0) {
            shs[0] = -0.5 * ((pb.beta_inv * (red[0] + red[1])) + (t3 + t4)) + lj;
            shs[1] = t3;
            shs[2] = t4;
        }
    }
    __syncthreads();   // V_G and shs complete for the whole block
    FinalizeOut o;
    o.t1 = red[0];
    o.t2 = red[1];
    o.L = shs[0];
    o.t3 = shs[1];
    o.t4 = shs[2];
    return o;
}

// ------------------------------------------------------------------------------------------
// Host-side handle
// ------------------------------------------------------------------------------------------
// Tuning / test switches of a handle.  The environment variables of the same names (MAGI_STREAM_FAMILY, ...) are read ONCE, when the
// handle is created; afterwards only magi_set_option changes them (no getenv on any compute path: a concurrent setenv cannot race a
// launch, and a stray variable cannot change what a running job computes between two calls).
#define MAGI_GEMM_REMAP_MIN_DEFAULT 10
struct MagiOptions {
    int stream_family = 0;              // 0 auto (leap.hip: magi_stream_family_mc), 1 "mc": every batch on the matrix-core kernel, 2 "valu"
    int family_chains = 0;              // > 0: "auto" decides as if the batch had this many chains (the largest per-GPU share of a sharded job:
                                        // every rank then runs the same kernel family, whatever its own share -- shard.family_chains_for)
    int sep_pair_min = 256;             // pack.hip: pair the diagonal blocks FH_bb + FK_bb when there are more tasks than this
    int fused_parity = 0;               // magi_logpost_grad_fused evaluates as an even (0) / odd (1) leapfrog slot
    int gemm_remap_min = MAGI_GEMM_REMAP_MIN_DEFAULT;      // build.hip: super-block tile order from this many super-blocks per launch (24 until round 4;
                                        // while it is at its default, potrf's thin launches keep 24: GemmArgs::remap_min)
    int potrf_panels = 3;               // build.hip: 128-wide panels per block column of the Cholesky factorisation (3: best of 2..8 at N = 1024..8192, profiles/r04_potrf_lookahead_ab.txt)
    int potrf_lookahead_min = 4096;     // build.hip: grids from this size on factorise with look-ahead (0: never), see potrf
    long long slot_budget_graphs = 0;   // TEST HOOK: cap on the graph launches of one magi_sampler_run (0 = the computed bound)
    int no_graph = 0;                   // launch the leapfrog slots directly (debugging, long rocprofv3 kernel traces)
    int fit_host_loop = 0, fit_per_component = 0;      // build.hip: A/B paths of the hyper-parameter fit
    int build_profile = 0, build_serial = 0;           // build.hip: per-class device times (serialises), one component per group
};
void magi_options_from_env(MagiOptions& o);               // capi.hip

struct magi_handle {
    int device = 0;
    MagiOptions opt;
    hipStream_t stream = nullptr;
    std::string err;

    // matrices
    bool have_matrices = false;
    bool have_problem = false;
    DevProblem pb{};
    double *dCsym = nullptr, *dM = nullptr, *dMt = nullptr, *dKsym = nullptr, *dYobs = nullptr;
    size_t mat_elems = 0;
    // dense C^-1, m, K^-1 ([D][N][N], unmasked) as built or uploaded: authoritative until the caller uploads others; packed
    // (band mask, symmetrised / transposed copies, single-phase operator blocks) by magi_pack_matrices
    double* dDense[3] = {nullptr, nullptr, nullptr};
    int dense_N = 0, dense_D = 0;
    double* dTiles = nullptr;
    int* dTasks = nullptr;
    size_t tiles_cap = 0, tasks_cap = 0;
    // work space of the matrix build and the packing, kept between calls (grow-only): a repeated build at N = 8192 otherwise spends
    // more host time in hipMalloc / hipFree of ~35 GB than the GPU spends on the build
    enum { WS_KAP = 0, WS_P, WS_PP, WS_DINV, WS_PANEL, WS_TCS, WS_TM, WS_TKS, WS_TE, WS_APPLY, WS_TI, WS_COUNT };
    double* ws[WS_COUNT] = {};
    size_t ws_cap[WS_COUNT] = {};

    // chains
    int n_chains = 0;
    int cap_chains = 0;
    DevChains ch{};
    SamplerCfgDev cfg{};
    bool sampler_ready = false;
    bool family_mc = false;          // magi_stream_family_mc(n_chains) at the last magi_ensure_chains
    int num_results = 0;
    // graph of G leapfrog slots
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_slots = 0;
    int graph_chains = 0;
    bool graph_valid = false;
    GlobalCtl* h_gctl = nullptr;     // pinned, 4 snapshots
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    int epoch = 0;
    long long* d_chain_ids = nullptr;
    double* d_fin = nullptr;         // [cap_chains][8] finalize outputs
    size_t samples_cap = 0, diag_cap = 0, vop_elems = 0;
    // magi_sampler_profile: when set, the launchers of k_stream / k_point attach these events to the launch (hipExtLaunchKernel:
    // they take the kernel's own begin / end time stamps, what rocprofv3's kernel trace reports)
    hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;
    long long last_slots = 0, last_graphs = 0;      // of the last magi_sampler_run
    double* apply_pin = nullptr;     // pinned staging of magi_dense_apply (V | Y)
    // look-ahead of the blocked Cholesky (build.hip: potrf): a second stream whose queue may NOT use one CU of every XCD -- the
    // rank-k updates it carries leave those CUs to the diagonal-block kernel of the next block column -- and the fork / join events
    hipStream_t stream_trail = nullptr, stream_chain = nullptr;      // (the chain of a block column runs on a stream of the highest priority)
    hipEvent_t ev_la[3] = {nullptr, nullptr, nullptr};
    bool trail_unavailable = false;  // the masked stream could not be created: factorise without look-ahead
    hipEvent_t ev_pw[4] = {nullptr, nullptr, nullptr, nullptr};      // begin / end of the two factorisations of a dense build
    double potrf_wall_ms = 0.0, potrf_wall_flops = 0.0;              // of the last dense build (whole factorisations, not serialised)
    size_t apply_pin_cap = 0;
};

// work-space slot `k` with room for n doubles (grow-only; freed by magi_destroy); nullptr + error on failure
double* magi_workspace(magi_handle* h, int k, size_t n);

// error helpers -------------------------------------------------------------------------------
extern thread_local std::string g_magi_last_error;
int magi_fail(magi_handle* h, int code, const std::string& msg);

#define MAGI_HIP_CHECK(h, expr)                                                                   \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return magi_fail((h), MAGI_E_HIP,                                                     \
                             std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                                 ":" + std::to_string(__LINE__) + ")");                           \
    } while (0)

// launchers implemented in logpost.hip / sampler.hip ---------------------------------------------
int magi_launch_gradient(magi_handle* h, int n_chains, hipStream_t s);        // phases 1-3
int magi_launch_phase(magi_handle* h, int phase, int n_chains, hipStream_t s);
int magi_launch_finalize(magi_handle* h, int n_chains, double* d_out, hipStream_t s);
int magi_launch_stream(magi_handle* h, int n_chains, int parity, bool with_decisions, hipStream_t s);   // k_stream: block mat-vecs of slot `parity` (+ decisions of the previous slot)
int magi_launch_point(magi_handle* h, int n_chains, int parity, hipStream_t s);                            // k_point: leapfrog epilogue per grid point
int magi_launch_leap_finalize(magi_handle* h, int n_chains, double* d_out, hipStream_t s, int parity = 0);
int magi_leap_wgs(const DevProblem& pb);
bool magi_stream_family_mc(const magi_handle* h, int n_chains);        // leap.hip
int magi_build_profile_get(const magi_handle* h, double* flops, double* ms, long* calls);           // build.hip
int magi_fit_hparams_device(magi_handle* h, const double* I, int N, int D, const double* X, const double* mu, const double* mu_phi2,
                            const double* sd_phi2, const double* sig_loc, double nu, int iters, double lr, double jitter,
                            double* phi1, double* phi2, double* sig2, double* loss_trace);
int magi_launch_read_tiles(magi_handle* h, int rev, hipStream_t s);                     // load-only pass over the packed operator blocks
int magi_launch_plan_eval(magi_handle* h, int n_chains, hipStream_t s, int parity = 0);        // plan: evaluate buffer 0, no leapfrog                                       // workgroups along the grid axis
// build.hip: E = Ks M, H = Cs + M^T E for D dense [N][N] components (H overwrites Cs)
int magi_fused_operators(magi_handle* h, int N, int D, double* dCs_inout_H, const double* dM, const double* dKs, double* dE);
int magi_launch_prepare(magi_handle* h, int n_chains, hipStream_t s);   // fills par from V_Q
int magi_launch_mirror(magi_handle* h, int n_chains, hipStream_t s);    // leap.hip: separable drifts: operand mirror (both slot parities) from V_Q
bool magi_drift_separable(int drift);                                   // leap.hip
size_t magi_sep_tpart_elems(const DevProblem& pb, int n_chains);        // leap.hip: doubles in tpart / vop for the separable path
size_t magi_sep_vop_elems(const DevProblem& pb, int n_chains);
void magi_sep_traffic(const DevProblem& pb, int n_chains, double* stores, double* operands, double* point_reads, double* mirror_writes);   // leap.hip
int magi_launch_init_chains(magi_handle* h, const long long* d_chain_ids, hipStream_t s);
int magi_ensure_chains(magi_handle* h, int n_chains);
// build.hip: dense device matrices -> packed device storage (sym / transpose / band)
int magi_pack_matrices(magi_handle* h, int N, int D, int bandsize, const double* dC_inv, const double* dM,
                       const double* dK_inv);
int magi_build_matrices_device(magi_handle* h, const double* I, int N, int D, const double* phi1,
                               const double* phi2, double nu, int bandsize, double* C_inv, double* m,
                               double* K_inv);
// build.hip: the listed components of the resident dense stacks (allocated / zeroed when N, D change); no packing
int magi_build_dense_device(magi_handle* h, const double* I, int N, int D, int n_sel, const int* sel, const double* phi1,
                            const double* phi2, double nu);
int magi_ensure_dense(magi_handle* h, int N, int D);
// pack.hip: Y[d] = A_d V[d] or A_d^T V[d] for the resident dense stack `which` (0 C^-1, 1 m, 2 K^-1); V, Y device [D][N][nv]
int magi_dense_apply_device(magi_handle* h, int which, int trans, int nv, const double* dV, double* dY);
// thetainit.hip: Adam x iters on the theta objective of magi_v2.py:148-158 with the resident UNbanded stacks, one captured graph per step
int magi_theta_init_device(magi_handle* h, int drift, int P, const double* Xhat, const double* mu, int iters, double lr, double* theta, double* loss_trace);
int magi_matern_blocks_device(magi_handle* h, const double* I, int N, double phi1, double phi2, double nu,
                              double* Kappa, double* p_Kappa, double* Kappa_pp);
