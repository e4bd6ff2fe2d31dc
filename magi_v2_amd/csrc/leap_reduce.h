// Reduce half of a leapfrog slot, shared by the sampler's tail (k_tail) and by the validation
// entry point magi_logpost_grad_fused (k_leap_finalize).
//
// k_leap_* (leap.hip) has already, on all CUs: applied the four single-phase operators, assembled
// dL/dX, completed the momentum step of the X entries, extended the subtree momentum sum, written
// the U-turn checkpoint, accumulated the checkpointed U-turn dot products and written the
// speculative next position -- and left PART_K partial sums per workgroup.  What remains is O(#WG):
// add the partials, finish the D + P parameter entries (whose gradients are global sums), and hand
// the totals to the caller's decision logic.
//
// PART layout: [0] t12  [1..4] ss_d  [5..10] tp_p  [11] p.p  [12+2k, 13+2k] U-turn dots of check k
#pragma once
#include "magi_internal.h"

#ifdef MAGI_TAIL_STAMPS
// timing stamps (dev builds): staged in LDS, copied to par[40..55] only when a HOT leaf finishes, so a run that
// ends on a slow path still reports the last hot leaf
static __shared__ double g_stamps[16];
#define MAGI_STAMP(par, i) do { if (threadIdx.x == 0) g_stamps[(i)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#define MAGI_STAMP_FLUSH(par) do { if (threadIdx.x == 0) for (int _i = 0; _i < 16; ++_i) (par)[40 + _i] = g_stamps[_i]; } while (0)
#else
#define MAGI_STAMP_FLUSH(par) do { } while (0)
#define MAGI_STAMP(par, i) do { } while (0)
#endif

constexpr int PK_T12 = 0, PK_SS = 1, PK_TP = 5, PK_PP = 11, PK_DOT = 12;

struct ReduceOut {
    double L, t12, t3, t4, pp;
    double dA[4], dB[4];
};

// sh: (21 + 4) * 16 doubles (block_sum scratch + a 64-double parking block), shs: >= 16 doubles.  `lp` is the plan the streaming kernel just executed.
template <int DRIFT>
__device__ __forceinline__ ReduceOut leap_reduce(const DevProblem& pb, const DevChains& ch, int chain, double* vb, double* par,
                                                 const LeafPlan& lp, double* sh, double* shs) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P;
    const int ND = pb.ND, dimp = pb.dimp;
    const double* part = ch.part + (size_t)chain * PART_K * ch.n_wg;      // [PART_K][n_wg]
    const int nwg = ch.n_wg;

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
        pj = ph[ND + j];
        rj = rho[ND + j];
        qj = q[ND + j];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < lp.nchk) { cpj[k] = ckp[(size_t)lp.chk_slot[k] * dimp + ND + j]; crj[k] = ckr[(size_t)lp.chk_slot[k] * dimp + ND + j]; }
    }

    // ---- add the workgroup partials: wave w owns values k = w, w + nw, ...; lane = workgroup (fixed order) ---------
    constexpr int K0 = 2 + D + P;            // t12, ss, tp, pp  (layout of red[]: [0] t12, [1..D] ss, [1+D..] tp, [1+D+P] pp, [K0..] dots)
    double red[K0 + 8];
    const bool dots = leaf && lp.nchk > 0;
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
        const int nval = dots ? K0 + 8 : K0;
        for (int k = wave; k < nval; k += nw) {
            // red index -> PART row
            const int row = (k == 0) ? PK_T12 : (k <= D) ? PK_SS + (k - 1) : (k <= D + P) ? PK_TP + (k - 1 - D) : (k == 1 + D + P) ? PK_PP : PK_DOT + (k - K0);
            double v = 0.0;
            for (int w = lane; w < nwg; w += 64) v += ld_agent(&part[(size_t)row * nwg + w]);
            v = wave_sum(v);
            if (lane == 0) sh[k] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K0 + 8; ++k) red[k] = (k < K0 || dots) ? sh[k] : 0.0;
        __syncthreads();                     // sh is reused below (parking block) and by the caller
    }
    MAGI_STAMP(par, 3);

    // ---- parameter entries: gradient, momentum step, checkpoint, U-turn terms, speculative next state ----
    if (threadIdx.x < 64) {
        double t3 = 0.0, t4 = 0.0, lj = 0.0, gj = 0.0;
        if (j < D) {
            const double sg = par[PAR_SGS + j], sj = par[PAR_SIG2 + j];
            const double ssd = select_lane<K0 + 8>(red, 1, D, j);
            const double nds = (j == 0) ? pb.N_ds[0] : (j == 1) ? pb.N_ds[1] : (j == 2) ? pb.N_ds[2] : pb.N_ds[3];
            t3 = nds * par[PAR_LOG2PIS + j];
            t4 = ssd * (1.0 / sj);
            lj = par[PAR_LJS + j];
            gj = -0.5 * (nds / sj - ssd / (sj * sj)) * sg + (1.0 - sg);
        } else if (plane) {
            const double sg = par[PAR_SGT + (j - D)];
            const double tpp = select_lane<K0 + 8>(red, 1 + D, P, j - D);
            lj = par[PAR_LJT + (j - D)];
            gj = -0.5 * pb.beta_inv * tpp * sg + (1.0 - sg);
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
        t3 = row16_sum(t3); t4 = row16_sum(t4); lj = row16_sum(lj); ppj = row16_sum(ppj);
        if (dots) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] = row16_sum(a[k]); b[k] = row16_sum(b[k]); }
        }
        if (j == 0) {
            shs[0] = -0.5 * ((pb.beta_inv * red[0]) + (t3 + t4)) + lj;
            shs[1] = t3;
            shs[2] = t4;
            shs[3] = red[1 + D + P] + ppj;
#pragma unroll
            for (int k = 0; k < 4; ++k) { shs[4 + 2 * k] = red[K0 + 2 * k] + a[k]; shs[5 + 2 * k] = red[K0 + 2 * k + 1] + b[k]; }
        }
    } else if (threadIdx.x < 128 && leaf) {
        // wave 1, concurrently: transformed parameters of the speculative next state (used by the next
        // k_leap_* if the subtree continues; overwritten by the tail's slow paths otherwise).  It re-derives
        // the entry's gradient from the same totals, so it does not wait for wave 0.
        const int jj = threadIdx.x - 64;
        if (jj < D + P) {
            double gj;
            if (jj < D) {
                const double sg = par[PAR_SGS + jj], sj = par[PAR_SIG2 + jj];
                const double ssd = select_lane<K0 + 8>(red, 1, D, jj);
                const double nds = (jj == 0) ? pb.N_ds[0] : (jj == 1) ? pb.N_ds[1] : (jj == 2) ? pb.N_ds[2] : pb.N_ds[3];
                gj = -0.5 * (nds / sj - ssd / (sj * sj)) * sg + (1.0 - sg);
            } else {
                const double sg = par[PAR_SGT + (jj - D)];
                const double tpp = select_lane<K0 + 8>(red, 1 + D, P, jj - D);
                gj = -0.5 * pb.beta_inv * tpp * sg + (1.0 - sg);
            }
            const double pn = ph[ND + jj] + lp.hs * gj;
            const double qnx = q[ND + jj] + lp.eps * (pn + lp.hs * gj);
            // the old entries are still needed by wave 0 -> park the new ones, publish after the barrier
            compute_par_entry(pb, jj, qnx, sh + 21 * 16);      // parking block of 64 doubles behind block_sum's scratch
        }
    }
    __syncthreads();
    if (leaf && threadIdx.x >= 64 && threadIdx.x < 64 + PAR_ULEAF) {
        // publish par' (entries below PAR_ULEAF; the rest of the block is not parameter data)
        const int k = threadIdx.x - 64;
        const bool used = (k < PAR_TH + P) || (k >= PAR_SGT && k < PAR_SGT + P) || (k >= PAR_LJT && k < PAR_LJT + P) ||
                          (k >= PAR_SIG2 && k < PAR_SIG2 + D) || (k >= PAR_SGS && k < PAR_SGS + D) ||
                          (k >= PAR_LJS && k < PAR_LJS + D) || (k >= PAR_LOG2PIS && k < PAR_LOG2PIS + D);
        if (used) par[k] = sh[21 * 16 + k];
    }
    ReduceOut o;
    o.L = shs[0]; o.t12 = red[0]; o.t3 = shs[1]; o.t4 = shs[2]; o.pp = shs[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.dA[k] = shs[4 + 2 * k]; o.dB[k] = shs[5 + 2 * k]; }
    __syncthreads();   // shs may be reused by the caller
    return o;
}
