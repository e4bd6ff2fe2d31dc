// Reduce half of a leapfrog slot, shared by the sampler's decisions (decide.h) and by the validation
// entry point magi_logpost_grad_fused (k_leap_finalize).
//
// k_stream + k_point (leap.hip, leap_point.h) have already, on all CUs: applied the four single-phase operators,
// assembled dL/dX, completed the momentum step of the X entries, extended the subtree momentum sum, written
// the U-turn checkpoint, accumulated the checkpointed U-turn dot products and written the speculative next
// position -- and left PART_K partial sums per point workgroup.  What remains is O(#WG): add the partials,
// finish the D + P parameter entries (whose gradients are global sums), and hand the totals to the caller's
// decision logic.
//
// PART layout: [0] t12  [1 .. MAX_D] ss_d  [PK_TP ..] tp_p  [PK_PP] p.p  [PK_DOT + 2k, + 2k + 1] U-turn dots of check k
#pragma once
#include "magi_internal.h"

#ifdef MAGI_TAIL_STAMPS
// timing stamps (dev builds): staged in LDS, copied to par[40..55] only when a HOT leaf finishes, so a run that
// ends on a slow path still reports the last hot leaf
static __shared__ double g_stamps[16];
#define MAGI_STAMP(par, i) do { if (threadIdx.x == 0) g_stamps[(i)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#define MAGI_STAMP_FLUSH(par) do { if (threadIdx.x == 0) for (int _i = 0; _i < 11; ++_i) (par)[40 + _i] = g_stamps[_i]; } while (0)   /* 11, 12: written by the stream workgroups */
#else
#define MAGI_STAMP_FLUSH(par) do { } while (0)
#define MAGI_STAMP(par, i) do { } while (0)
#endif

constexpr int PK_T12 = 0, PK_SS = 1, PK_TP = 1 + MAGI_MAX_D, PK_PP = PK_TP + MAGI_MAX_P, PK_DOT = PK_PP + 1;
static_assert(PK_DOT + 8 <= PART_K, "PART layout");

// ---- pieces shared bit for bit by the decisions (below) and by k_stream, which derives the next theta itself ----
// sum of one PART row over the point workgroups: whole wave, lane = workgroup, fixed order
__device__ __forceinline__ double part_row_sum(const double* part, int nwg, int row, int lane) {
    double v = 0.0;
    for (int w = lane; w < nwg; w += 64) v += part[(size_t)row * nwg + w];
    return wave_sum(v);
}
// P rows at once, all loads in flight together (the stream's derivation of theta')
template <int P>
__device__ __forceinline__ void part_rows_sum(const double* part, int nwg, int row0, int lane, double (&out)[P]) {
#pragma unroll
    for (int k = 0; k < P; ++k) out[k] = (lane < nwg) ? part[(size_t)(row0 + k) * nwg + lane] : 0.0;
    for (int w0 = 64; w0 < nwg; w0 += 64) {
#pragma unroll
        for (int k = 0; k < P; ++k) if (w0 + lane < nwg) out[k] += part[(size_t)(row0 + k) * nwg + w0 + lane];
    }
#pragma unroll
    for (int k = 0; k < P; ++k) out[k] = wave_sum(out[k]);
}

// d L / d theta_pre_p from the global sum tp_p (magi_v2.py:318-323, 335-337 chained through softplus)
__device__ __forceinline__ double theta_entry_grad(double beta_inv, double tpp, double sg) { return -0.5 * beta_inv * tpp * sg + (1.0 - sg); }
// position of a parameter entry after completing this leaf's momentum step and taking the next half step
__device__ __forceinline__ double next_entry_pre(double ph, double q, double hs, double eps, double gj) {
    const double pn = ph + hs * gj;
    return q + eps * (pn + hs * gj);
}

struct ReduceOut {
    double L, t12, t3, t4, pp;
    double dA[4], dB[4];
};

// The workgroup partials of one chain, loaded with NO dependence on the plan (so the decisions can issue them together
// with their control state): wave w owns values k = w, w + 4, ...; lane = workgroup.  4 waves per block.
constexpr int RED_WAVES = 4;
template <int DRIFT> struct RedLayout {
    static constexpr int D = DriftT<DRIFT>::D, P = DriftT<DRIFT>::P;
    static constexpr int K0 = 2 + D + P;            // t12, ss, tp, pp  (red[]: [0] t12, [1..D] ss, [1+D..] tp, [1+D+P] pp, [K0..] dots)
    static constexpr int PER_WAVE = (K0 + 8 + RED_WAVES - 1) / RED_WAVES;
    static __device__ __forceinline__ int row(int k) {
        return (k == 0) ? PK_T12 : (k <= D) ? PK_SS + (k - 1) : (k <= D + P) ? PK_TP + (k - 1 - D) : (k == 1 + D + P) ? PK_PP : PK_DOT + (k - K0);
    }
};

template <int DRIFT>
__device__ __forceinline__ void leap_reduce_issue(const DevChains& ch, int chain, double (&v)[RedLayout<DRIFT>::PER_WAVE]) {
    using RL = RedLayout<DRIFT>;
    const double* part = ch.part + (size_t)chain * PART_K * ch.n_wg;      // [PART_K][n_wg]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwg = ch.n_wg;
    // every load of the first pass is issued before anything is added: a loop "a += part[...]" per value would make each
    // value its own round trip (measured: 5 x 1.1 us next to the saturating stream)
#pragma unroll
    for (int i = 0; i < RL::PER_WAVE; ++i) {
        const int k = wave + RED_WAVES * i;
        v[i] = (k < RL::K0 + 8 && lane < nwg) ? part[(size_t)RL::row(k) * nwg + lane] : 0.0;
    }
    for (int w0 = 64; w0 < nwg; w0 += 64) {                                // (N > 1024; same order of additions as part_row_sum)
#pragma unroll
        for (int i = 0; i < RL::PER_WAVE; ++i) {
            const int k = wave + RED_WAVES * i;
            if (k < RL::K0 + 8 && w0 + lane < nwg) v[i] += part[(size_t)RL::row(k) * nwg + w0 + lane];
        }
    }
}

// The parameter entries (index >= N D) of every vector the reduce may need, whatever the plan says: both momentum /
// position buffers, the subtree momentum sum and all checkpoints -> LDS [OPS_COUNT][OPS_W].
constexpr int OPS_P = 0, OPS_Q = 2, OPS_RHO = 4, OPS_CKP = 5, OPS_CKR = OPS_CKP + MAGI_MAX_DEPTH, OPS_COUNT = OPS_CKR + MAGI_MAX_DEPTH;
constexpr int OPS_W = MAGI_MAX_D + MAGI_MAX_P + 2;      // entries per vector
template <int PER>
__device__ __forceinline__ void reduce_prefetch_ops_load(const DevProblem& pb, const double* vb, int n_entries, double (&tmp)[PER]) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = threadIdx.x + u * MAGI_TAIL_THREADS;
        const int slot = idx / OPS_W, j = idx - slot * OPS_W;
        tmp[u] = 0.0;
        if (idx < OPS_COUNT * OPS_W && j < n_entries) {
            const int vs = slot < OPS_Q ? V_P + slot : slot < OPS_RHO ? V_Q + (slot - OPS_Q) : slot == OPS_RHO ? V_RHOSUB
                         : slot < OPS_CKR ? V_CKP0 + (slot - OPS_CKP) : V_CKRHO0 + (slot - OPS_CKR);
            tmp[u] = vb[(size_t)vs * pb.dimp + pb.ND + j];
        }
    }
}
template <int PER>
__device__ __forceinline__ void reduce_prefetch_ops_store(const double (&tmp)[PER], double* ops) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = threadIdx.x + u * MAGI_TAIL_THREADS;
        if (idx < OPS_COUNT * OPS_W) ops[idx] = tmp[u];
    }
}

// sh: (21 + 4) * 16 doubles (scratch + a 64-double parking block), shs: >= 16 doubles.  `lp` is the plan the point phase just
// executed; `pre` the values of leap_reduce_issue; `par_r` the state's parameter block (may be an LDS copy), `par` the global one.
template <int DRIFT>
__device__ __forceinline__ ReduceOut leap_reduce(const DevProblem& pb, const DevChains& ch, int chain, double* vb, double* par, const double* par_r,
                                                 const LeafPlan& lp, const double (&pre)[RedLayout<DRIFT>::PER_WAVE], double* sh, double* shs,
                                                 const double* ops = nullptr /* LDS copy made by reduce_prefetch_ops, or null */,
                                                 const double* cst = nullptr /* LDS: N_ds[MAX_D], LB[MAX_D], or null */) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P;
    const int ND = pb.ND, dimp = pb.dimp;

    // parameter entries (lanes j < D + P of wave 0): operands fetched early, used after the reduce
    const int j = threadIdx.x;
    const bool plane = j < D + P;
    const bool leaf = lp.leaf != 0;
    double* g = vb + (size_t)V_G * dimp;
    double* q = vb + (size_t)(V_Q + lp.cur) * dimp;
    double* qn = vb + (size_t)(V_Q + (lp.cur ^ 1)) * dimp;
    double* ph = vb + (size_t)(V_P + lp.cur) * dimp;
    double* phn = vb + (size_t)(V_P + (lp.cur ^ 1)) * dimp;
    double* pleaf = vb + (size_t)V_PLEAF * dimp;
    double* rho = vb + (size_t)V_RHOSUB * dimp;
    double* ckp = vb + (size_t)V_CKP0 * dimp;
    double* ckr = vb + (size_t)V_CKRHO0 * dimp;
    double pj = 0.0, rj = 0.0, qj = 0.0, cpj[4] = {0.0, 0.0, 0.0, 0.0}, crj[4] = {0.0, 0.0, 0.0, 0.0};
    if (plane && leaf) {
        if (ops) {
            pj = ops[(OPS_P + lp.cur) * OPS_W + j];
            qj = ops[(OPS_Q + lp.cur) * OPS_W + j];
            rj = ops[OPS_RHO * OPS_W + j];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < lp.nchk) { cpj[k] = ops[(OPS_CKP + lp.chk_slot[k]) * OPS_W + j]; crj[k] = ops[(OPS_CKR + lp.chk_slot[k]) * OPS_W + j]; }
        } else {
            pj = ph[ND + j];
            rj = rho[ND + j];
            qj = q[ND + j];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < lp.nchk) { cpj[k] = ckp[(size_t)lp.chk_slot[k] * dimp + ND + j]; crj[k] = ckr[(size_t)lp.chk_slot[k] * dimp + ND + j]; }
        }
    }

    // ---- add the workgroup partials (fixed order: lane = workgroup, butterfly) ------------------------------------------
    constexpr int K0 = 2 + D + P;
    double red[K0 + 8];
    const bool dots = leaf && lp.nchk > 0;
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < RedLayout<DRIFT>::PER_WAVE; ++i) {
            const int k = wave + RED_WAVES * i;
            const double v = wave_sum(pre[i]);
            if (lane == 0 && k < K0 + 8) sh[k] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K0 + 8; ++k) red[k] = (k < K0 || dots) ? sh[k] : 0.0;
        __syncthreads();                     // sh is reused below (parking block) and by the caller
    }
    MAGI_STAMP(par, 3);

    // ---- parameter entries: gradient, momentum step, checkpoint, U-turn terms, speculative next state ----
    if (threadIdx.x < 64) {
        double t4 = 0.0, lj = 0.0, gj = 0.0;
        if (j < D) {
            const double sg = par_r[PAR_SGS + j], sj = par_r[PAR_SIG2 + j];
            const double ssd = select_lane<K0 + 8>(red, 1, D, j);
            const double nds = cst ? cst[j] : MAGI_SEL_D(pb.N_ds, j);
            t4 = ssd * (1.0 / sj);
            lj = par_r[PAR_LJS + j];
            gj = -0.5 * (nds / sj - ssd / (sj * sj)) * sg + (1.0 - sg);
        } else if (plane) {
            const double sg = par_r[PAR_SGT + (j - D)];
            const double tpp = select_lane<K0 + 8>(red, 1 + D, P, j - D);
            lj = par_r[PAR_LJT + (j - D)];
            gj = theta_entry_grad(pb.beta_inv, tpp, sg);
        }
        double ppj = 0.0, a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
        if (plane) {
            g[ND + j] = gj;
            if (leaf) {
                const double pn = pj + lp.hs * gj;
                const double rs = rj + pn;
                pleaf[ND + j] = pn;
                rho[ND + j] = rs;
                ppj = pn * pn;
                if (lp.even) { ckp[(size_t)lp.ck_slot * dimp + ND + j] = pn; ckr[(size_t)lp.ck_slot * dimp + ND + j] = rs; }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < lp.nchk) { const double df = rs - crj[k]; a[k] = df * cpj[k]; b[k] = df * pn; }
                const double pnext = pn + lp.hs * gj;          // speculative next half step
                phn[ND + j] = pnext;
                qn[ND + j] = qj + lp.eps * pnext;
            }
        }
        t4 = row16_sum(t4); lj = row16_sum(lj); ppj = row16_sum(ppj);
        if (dots) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = row16_sum(a[k]); b[k] = row16_sum(b[k]); }
        }
        if (j == 0) {
            shs[0] = lj;
            shs[2] = t4;
            shs[3] = red[1 + D + P] + ppj;
#pragma unroll
            for (int k = 0; k < 4; ++k) { shs[4 + 2 * k] = red[K0 + 2 * k] + a[k]; shs[5 + 2 * k] = red[K0 + 2 * k + 1] + b[k]; }
        }
    } else if (threadIdx.x < 128) {
        // wave 1, concurrently: the parameter entries of the speculative next state (pre-transform).  It re-derives the entry's gradient
        // from the same totals, so it does not wait for wave 0.  Their transform (compute_par_entry: an exp -> log chain per entry) is
        // the CALLER's, next to its own transcendental work (decide.h) -- here it made every barrier of the reduce wait for this wave.
        const int jj = threadIdx.x - 64;
        if (leaf && jj < D + P) {
            double gj;
            if (jj < D) {
                const double sg = par_r[PAR_SGS + jj], sj = par_r[PAR_SIG2 + jj];
                const double ssd = select_lane<K0 + 8>(red, 1, D, jj);
                const double nds = cst ? cst[jj] : MAGI_SEL_D(pb.N_ds, jj);
                gj = -0.5 * (nds / sj - ssd / (sj * sj)) * sg + (1.0 - sg);
            } else {
                const double sg = par_r[PAR_SGT + (jj - D)];
                const double tpp = select_lane<K0 + 8>(red, 1 + D, P, jj - D);
                gj = theta_entry_grad(pb.beta_inv, tpp, sg);
            }
            const double phv = ops ? ops[(OPS_P + lp.cur) * OPS_W + jj] : ph[ND + jj], qv = ops ? ops[(OPS_Q + lp.cur) * OPS_W + jj] : q[ND + jj];
            sh[21 * 16 + jj] = next_entry_pre(phv, qv, lp.hs, lp.eps, gj);      // parking block of 64 doubles behind block_sum's scratch: read by leap_next_par
        }
    } else if (threadIdx.x < 192) {
        // wave 2, concurrently: t3 = sum_d N_d log(2 pi sigma_d^2) of the evaluated state (one m_log; combined after the barrier)
        const int jd = threadIdx.x - 128;
        double t3 = 0.0;
        if (jd < D) {
            const double nds = cst ? cst[jd] : MAGI_SEL_D(pb.N_ds, jd);
            t3 = nds * m_log(2.0 * 3.141592653589793 * par_r[PAR_SIG2 + jd]);
        }
        t3 = row16_sum(t3);
        if (jd == 0) shs[1] = t3;
    }
    __syncthreads();
    ReduceOut o;
    o.t12 = red[0]; o.t3 = shs[1]; o.t4 = shs[2]; o.pp = shs[3];
    o.L = -0.5 * ((pb.beta_inv * red[0]) + (o.t3 + o.t4)) + shs[0];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.dA[k] = shs[4 + 2 * k]; o.dB[k] = shs[5 + 2 * k]; }
    __syncthreads();   // shs may be reused by the caller
    return o;
}

// Transformed parameters of the speculative next state (read by the next slot's point phase if the subtree continues; rewritten by a
// boundary op otherwise): threads 64 .. 64 + D + P - 1, from the entries leap_reduce parked in sh.  Straight to the global block --
// nothing in the kernel it rides in reads `par` of a chain whose plan is a leaf (the stream derives theta' itself).
template <int DRIFT>
__device__ __forceinline__ void leap_next_par(const DevProblem& pb, double* par, const double* sh, const double* lb_tab) {
    const int jj = (int)threadIdx.x - 64;
    if (jj >= 0 && jj < DriftT<DRIFT>::D + DriftT<DRIFT>::P) compute_par_entry(pb, jj, sh[21 * 16 + jj], par, false, lb_tab);
}
