// Point half of a leapfrog slot: everything elementwise, PT_POINTS grid points per workgroup.
//
// k_stream (leap.hip) has left, per chain, the block partials of the four single-phase products
//     hx = FH xc,  ex = FE xc,  etf = FE^T f,  kf = FK f        in tpart[chain][vec][d][slot][i].
// point_block adds them in fixed slot order and ONE lane per (grid point, component) finishes the point:
//     Ksym r = kf - ex,   dL/dx_d = -1/2 (beta^-1 (2 hx - 2 etf + J^T 2 Ksym r)_d + dt4/dx_d)
//     p_leaf = p_half + hs g,  rho_sub += p_leaf,  checkpoint,  U-turn partial dots,
//     speculative next leaf:  p_half' = p_leaf + hs g,  x' = x + eps p_half'   (other buffer)
// and leaves PART_K partial sums per workgroup for the reduce (leap_reduce.h).
// (reference arithmetic: magi_v2.py:308-348 and the leapfrog of TFP's NoUTurnSampler)
#pragma once
#include "magi_internal.h"
#include "leap_reduce.h"

constexpr int PT_DSLOT = MAGI_MAX_D;            // component lanes per grid point (4, or 8 in a library built for 5..8 components)
constexpr int PT_POINTS = 64 / PT_DSLOT;       // grid points per workgroup
constexpr int PT_THREADS = 256;                // = PT_POINTS x PT_DSLOT components x 4 products

template <int DRIFT>
struct GridPoint {     // component d of grid index i of one chain (one lane)
    using DR = DriftT<DRIFT>;
    static constexpr int D = DR::D, P = DR::P;
    struct Ops { double y, phe, rhoe, cpk[4], crk[4], x[D], th[P], sig2, qprev, gold; };

    // the lane's own operands: independent of the products, so their latency overlaps the partial sums
    static __device__ __forceinline__ Ops load(const DevProblem& pb, const DevChains& ch, const LeafPlan& lp, int cc, int i, int d) {
        Ops o;
        const int N = pb.N, dimp = pb.dimp;
        const double* vb = ch.vec + vec_off(pb, cc, 0);
        const double* par = ch.par + (size_t)cc * PAR_COUNT;
        const double* q = vb + (size_t)(V_Q + lp.cur) * dimp;
        const int e = d * N + i;
        o.y = pb.yobs[e];
        o.phe = 0.0; o.rhoe = 0.0; o.qprev = 0.0; o.gold = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o.cpk[k] = 0.0; o.crk[k] = 0.0; }
        if (lp.leaf) {
            if (lp.sub_copy) {       // the previous leaf becomes the subtree's proposal: its position and gradient, before this phase overwrites them
                o.qprev = (vb + (size_t)(V_Q + (lp.cur ^ 1)) * dimp)[e];
                o.gold = (vb + (size_t)V_G * dimp)[e];
            }
            o.phe = (vb + (size_t)(V_P + lp.cur) * dimp)[e];
            o.rhoe = (vb + (size_t)V_RHOSUB * dimp)[e];
            const double* ckp = vb + (size_t)V_CKP0 * dimp;
            const double* ckr = vb + (size_t)V_CKRHO0 * dimp;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < lp.nchk) { o.cpk[k] = ckp[(size_t)lp.chk_slot[k] * dimp + e]; o.crk[k] = ckr[(size_t)lp.chk_slot[k] * dimp + e]; }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) o.th[k] = par[PAR_TH + k];
#pragma unroll
        for (int dd = 0; dd < D; ++dd) o.x[dd] = q[dd * N + i];
        o.sig2 = par[PAR_SIG2 + d];
        return o;
    }

    // `mud` = mu[d], read by the caller from an LDS copy: selecting pb.mu[d] with a per-lane d makes the compiler spill the
    // kernel-argument array to scratch and index it there -- a global-memory round trip in the middle of the finishing lane
    static __device__ __forceinline__ void finish(const DevProblem& pb, const DevChains& ch, const LeafPlan& lp, int cc, int i, int d, const Ops& o,
                                                  const double* res /* [D][4] */, double mud, double* pk /* [PART_K] */, int xop_buf) {
#pragma unroll
        for (int k = 0; k < PART_K; ++k) pk[k] = 0.0;
        const int N = pb.N, dimp = pb.dimp;
        double* vb = ch.vec + vec_off(pb, cc, 0);
        const int e = d * N + i;
        double f[D], g2[D], jt[D], tp[P];
#pragma unroll
        for (int k = 0; k < P; ++k) tp[k] = 0.0;
        DR::f(o.x, o.th, f);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) g2[dd] = 2.0 * (res[dd * 4 + TV_KF] - res[dd * 4 + TV_EX]);
        DR::jt(o.x, o.th, g2, jt, tp);
        // select this lane's component
        double xd = o.x[0], fd = f[0], jtd = jt[0];
#pragma unroll
        for (int dd = 1; dd < D; ++dd) if (d == dd) { xd = o.x[dd]; fd = f[dd]; jtd = jt[dd]; }
        const double hx = res[d * 4 + TV_HX], ex = res[d * 4 + TV_EX], etf = res[d * 4 + TV_ETF], kf = res[d * 4 + TV_KF];
        pk[PK_T12] = (xd - mud) * hx + fd * (kf - 2.0 * ex);
        if (d == 0) {
#pragma unroll
            for (int k = 0; k < P; ++k) pk[PK_TP + k] = tp[k];
        }
        double d4 = 0.0;
        if (!isnan(o.y)) {
            const double df = xd - o.y;
            pk[PK_SS + d] = df * df;
            d4 = 2.0 * df / o.sig2;
        }
        const double gx = -0.5 * (pb.beta_inv * (2.0 * hx - 2.0 * etf + jtd) + d4);
        *(vb + (size_t)V_G * dimp + e) = gx;
        if (lp.leaf) {
            double* rho = vb + (size_t)V_RHOSUB * dimp;
            double* ckp = vb + (size_t)V_CKP0 * dimp;
            double* ckr = vb + (size_t)V_CKRHO0 * dimp;
            const double pn = o.phe + lp.hs * gx;
            *(vb + (size_t)V_PLEAF * dimp + e) = pn;
            const double rs = o.rhoe + pn;
            *(rho + e) = rs;
            pk[PK_PP] = pn * pn;
            if (lp.sub_copy) { *(vb + (size_t)V_SUBQ * dimp + e) = o.qprev; *(vb + (size_t)V_SUBG * dimp + e) = o.gold; }
            if (lp.even) { *(ckp + (size_t)lp.ck_slot * dimp + e) = pn; *(ckr + (size_t)lp.ck_slot * dimp + e) = rs; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < lp.nchk) { const double df = rs - o.crk[k]; pk[PK_DOT + 2 * k] = df * o.cpk[k]; pk[PK_DOT + 2 * k + 1] = df * pn; }
            const double pnext = pn + lp.hs * gx;
            *(vb + (size_t)(V_P + (lp.cur ^ 1)) * dimp + e) = pnext;
            const double xn = xd + lp.eps * pnext;
            *(vb + (size_t)(V_Q + (lp.cur ^ 1)) * dimp + e) = xn;
            if (ch.mc) ch.xop[xop_off(pb, ch.n_chains, xop_buf, cc, d, i)] = xn;      // (three or more chains: the matrix-core stream of the NEXT slot reads this mirror)
        }
    }
};

// All PT_THREADS threads of the block must call it.  res: PT_POINTS*PT_DSLOT*4 doubles, redk: 64*PART_K doubles of LDS.
// Writes part[cc][k][blk].  `lp`: the chain's plan for this slot (active, not skip).
#ifdef MAGI_PT_STAMPS      // dev: 100 MHz stamps of workgroup MAGI_PT_STAMPS of k_point, kept in scalar registers, written at the end to par[40 ..] of chain 0
#define PT_STAMP(i) do { pt_st[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PT_STAMP(i) do { } while (0)
#endif
template <int DRIFT>
__device__ __forceinline__ void point_block(const DevProblem& pb, const DevChains& ch, const LeafPlan& lp, int cc, int blk, double* res, double* redk, double* s_mu /* MAGI_MAX_D */,
                                            int xop_buf /* slot parity ^ 1 */) {
    using GP = GridPoint<DRIFT>;
    constexpr int D = GP::D, TB = MAGI_TB;
    const unsigned t = threadIdx.x;
#ifdef MAGI_PT_STAMPS
    unsigned long long pt_st[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    PT_STAMP(0);
    // finishing lanes: t < 64 = (component, point)
    const int fpt = t & (PT_POINTS - 1), fd = (t / PT_POINTS) & (PT_DSLOT - 1);
    const int fi = blk * PT_POINTS + fpt;
    const bool fvalid = (t < 64) && (fd < D) && (fi < pb.N);
    typename GP::Ops ops;
    if (fvalid) ops = GP::load(pb, ch, lp, cc, fi, fd);
    // product lanes: t = (component, product, point)
    {
        const int pt = t & (PT_POINTS - 1), v = (t / PT_POINTS) & 3, d = t / (PT_POINTS * 4);
        const int i = blk * PT_POINTS + pt;
        double sum = 0.0;
        if (d < D && i < pb.N) {
            const int b = i / TB, s0 = max(0, b - pb.wb), s1 = min(pb.nb - 1, b + pb.wb);
            const double* src = ch.tpart + (size_t)cc * 4 * D * pb.nb * pb.Np + ((size_t)(v * D + d) * pb.nb) * pb.Np + i;
            int sl = s0;
            for (; sl + 7 <= s1; sl += 8) {
                double u[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) u[k] = src[(size_t)(sl + k) * pb.Np];
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += u[k];
            }
            if (sl <= s1) {        // ragged tail (1..7 slots): one batch from clamped addresses -- a `sum += src[..]` loop waits per slot
                double u[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) u[k] = src[(size_t)min(sl + k, s1) * pb.Np];
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += (sl + k <= s1) ? u[k] : 0.0;
            }
        }
        res[(pt * PT_DSLOT + d) * 4 + v] = sum;
    }
    PT_STAMP(1);
    if (t == PT_THREADS - 1) {
#pragma unroll
        for (int k = 0; k < MAGI_MAX_D; ++k) s_mu[k] = pb.mu[k];       // (static indices: one thread copies the kernel-argument array)
    }
    __syncthreads();
    PT_STAMP(2);
    if (t < 64) {
        double* pk = redk + (size_t)t * PART_K;
        if (fvalid) {
            GP::finish(pb, ch, lp, cc, fi, fd, ops, res + fpt * PT_DSLOT * 4, s_mu[fd], pk, xop_buf);
        } else {
#pragma unroll
            for (int k = 0; k < PART_K; ++k) pk[k] = 0.0;
        }
    }
    PT_STAMP(3);
    __syncthreads();
    PT_STAMP(4);
    // PART_K sums over the 64 finishing lanes: four threads per sum, 16 lanes each in order, then (s0 + s1) + (s2 + s3) by two quad
    // exchanges (one thread per sum, 64 dependent additions: 0.64 us of the kernel's 2.5 us behind the plan)
    if (t < 4 * PART_K) {
        const int k = t >> 2, q = t & 3;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += redk[(16 * q + u) * PART_K + k];
        s += dpp_f64<0xB1>(s);    // quad_perm [1,0,3,2]
        s += dpp_f64<0x4E>(s);    // quad_perm [2,3,0,1]
        if (q == 0) *(&ch.part[((size_t)cc * PART_K + k) * ch.n_wg + blk]) = s;
    }
    PT_STAMP(5);
#ifdef MAGI_PT_STAMPS
    if (blk == (MAGI_PT_STAMPS) && cc == 0 && t == 0)
        for (int _i = 0; _i < 8; ++_i) reinterpret_cast<unsigned long long*>(ch.par + 40)[_i] = pt_st[_i];
#endif
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Point phase of the SEPARABLE streaming path (k_stream_sep): the stream has left, per chain, the block partials of
//     hx = FH xc,  ex = FE xc,  etf_k = FE^T phi_{d,k},  kf_k = FK phi_{d,k}        in tpart[chain][slot][other block][i]  (SepLayout)
// -- products of theta-free vectors.  This phase adds them, combines etf = sum_k coef_k(theta) etf_k, kf = sum_k coef_k(theta) kf_k with
// the parameters of the evaluated state (par, complete before this kernel starts), finishes the point exactly as point_block does, and
// writes the operand mirror of the speculative next state: xc' = x' - mu and phi_{d,k}(x') for the stream of the NEXT slot.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int DRIFT>
__device__ __forceinline__ void point_block_sep(const DevProblem& pb, const DevChains& ch, const LeafPlan& lp, int cc, int blk,
                                                double* res /* PT_POINTS * PS_TOTAL */, double* redk /* 64 * PART_K */, double* s_mu /* MAGI_MAX_D */,
                                                double* s_x /* PT_POINTS * PT_DSLOT */, int mirror_buf /* slot parity ^ 1 */) {
    using GP = GridPoint<DRIFT>;
    using DR = DriftT<DRIFT>;
    using SL = SepLayout<DRIFT>;
    constexpr int D = GP::D, P = GP::P, TB = MAGI_TB, PST = SL::PS_TOTAL, NBM = DR::NBMAX;
    const unsigned t = threadIdx.x;
    // finishing lanes: t < 64 = (component, point)
    const int fpt = t & (PT_POINTS - 1), fd = (t / PT_POINTS) & (PT_DSLOT - 1);
    const int fi = blk * PT_POINTS + fpt;
    const bool fvalid = (t < 64) && (fd < D) && (fi < pb.N);
    typename GP::Ops ops;
    if (fvalid) ops = GP::load(pb, ch, lp, cc, fi, fd);
    // product lanes: item = (slot, point); every used slot's nb block partials in flight together
    {
        const double* base = ch.tpart + (size_t)cc * PST * pb.nb * pb.Np;
#pragma unroll
        for (int it0 = 0; it0 < PT_POINTS * PST; it0 += PT_THREADS) {
            const int item = it0 + (int)t;
            const int pt = item & (PT_POINTS - 1), slot = item / PT_POINTS;
            const int i = blk * PT_POINTS + pt;
            if (slot < PST) {
                double sum = 0.0;
                if (SL::slot_used(slot) && i < pb.N) {
                    const int b = i / TB, s0 = max(0, b - pb.wb), s1 = min(pb.nb - 1, b + pb.wb);
                    const double* src = base + ((size_t)slot * pb.nb) * pb.Np + i;
                    int sl = s0;
                    for (; sl + 7 <= s1; sl += 8) {
                        double u[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) u[k] = src[(size_t)(sl + k) * pb.Np];
#pragma unroll
                        for (int k = 0; k < 8; ++k) sum += u[k];
                    }
                    if (sl <= s1) {
                        double u[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) u[k] = src[(size_t)min(sl + k, s1) * pb.Np];
#pragma unroll
                        for (int k = 0; k < 8; ++k) sum += (sl + k <= s1) ? u[k] : 0.0;
                    }
                }
                res[pt * PST + slot] = sum;
            }
        }
    }
    if (t == PT_THREADS - 1) {
#pragma unroll
        for (int k = 0; k < MAGI_MAX_D; ++k) s_mu[k] = pb.mu[k];
    }
    __syncthreads();
    if (t < 64) {
        double* pk = redk + (size_t)t * PART_K;
#pragma unroll
        for (int k = 0; k < PART_K; ++k) pk[k] = 0.0;
        double xn = 0.0;
        if (fvalid) {
            const double* r = res + fpt * PST;
            const int N = pb.N, dimp = pb.dimp, d = fd, i = fi;
            double* vb = ch.vec + vec_off(pb, cc, 0);
            const int e = d * N + i;
            const double mud = s_mu[fd];
            double cf[D][NBM];
            DR::coefs(ops.th, cf);
            double f[D], g2[D], jt[D], tp[P];
#pragma unroll
            for (int k = 0; k < P; ++k) tp[k] = 0.0;
            DR::f(ops.x, ops.th, f);
            double hx = 0.0, ex = 0.0, etf = 0.0, kf = 0.0;
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                double kfd = 0.0, etd = 0.0;
#pragma unroll
                for (int k = 0; k < NBM; ++k) {
                    if (k < DR::nbasis(dd)) {
                        kfd = fma(cf[dd][k], r[SL::slot_kf(dd, k)], kfd);
                        etd = fma(cf[dd][k], r[SL::slot_etf(dd, k)], etd);
                    }
                }
                const double exd = r[SL::slot_ex(dd)];
                g2[dd] = 2.0 * (kfd - exd);
                if (d == dd) { hx = r[SL::slot_hx(dd)]; ex = exd; etf = etd; kf = kfd; }
            }
            DR::jt(ops.x, ops.th, g2, jt, tp);
            double xd = ops.x[0], fdv = f[0], jtd = jt[0];
#pragma unroll
            for (int dd = 1; dd < D; ++dd) if (d == dd) { xd = ops.x[dd]; fdv = f[dd]; jtd = jt[dd]; }
            pk[PK_T12] = (xd - mud) * hx + fdv * (kf - 2.0 * ex);
            if (d == 0) {
#pragma unroll
                for (int k = 0; k < P; ++k) pk[PK_TP + k] = tp[k];
            }
            double d4 = 0.0;
            if (!isnan(ops.y)) {
                const double df = xd - ops.y;
                pk[PK_SS + d] = df * df;
                d4 = 2.0 * df / ops.sig2;
            }
            const double gx = -0.5 * (pb.beta_inv * (2.0 * hx - 2.0 * etf + jtd) + d4);
            *(vb + (size_t)V_G * dimp + e) = gx;
            if (lp.leaf) {
                double* rho = vb + (size_t)V_RHOSUB * dimp;
                double* ckp = vb + (size_t)V_CKP0 * dimp;
                double* ckr = vb + (size_t)V_CKRHO0 * dimp;
                const double pn = ops.phe + lp.hs * gx;
                *(vb + (size_t)V_PLEAF * dimp + e) = pn;
                const double rs = ops.rhoe + pn;
                *(rho + e) = rs;
                pk[PK_PP] = pn * pn;
                if (lp.sub_copy) { *(vb + (size_t)V_SUBQ * dimp + e) = ops.qprev; *(vb + (size_t)V_SUBG * dimp + e) = ops.gold; }
                if (lp.even) { *(ckp + (size_t)lp.ck_slot * dimp + e) = pn; *(ckr + (size_t)lp.ck_slot * dimp + e) = rs; }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < lp.nchk) { const double df = rs - ops.crk[k]; pk[PK_DOT + 2 * k] = df * ops.cpk[k]; pk[PK_DOT + 2 * k + 1] = df * pn; }
                const double pnext = pn + lp.hs * gx;
                *(vb + (size_t)(V_P + (lp.cur ^ 1)) * dimp + e) = pnext;
                xn = xd + lp.eps * pnext;
                *(vb + (size_t)(V_Q + (lp.cur ^ 1)) * dimp + e) = xn;
            }
        }
        // operand mirror of the speculative next state: the component lanes of a point exchange x' through LDS (one wave: its LDS
        // operations execute in order, no workgroup barrier), then every lane writes xc' and the basis values of ITS component
        if (lp.leaf) {
            s_x[fpt * PT_DSLOT + fd] = xn;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (fvalid) {
                double xq[D], ph[D][NBM];
#pragma unroll
                for (int dd = 0; dd < D; ++dd) xq[dd] = s_x[fpt * PT_DSLOT + dd];
                DR::basis(xq, ph);
                const int cw = xop_width(ch.n_chains), groups = (ch.n_chains + 15) >> 4, cl = cc & 15;
                const int planes = 1 + (NBM * cw + 15) / 16;
                double* m0 = ch.vop + vop_off(D, planes, pb.Np, groups, mirror_buf, cc >> 4, fd, 0, 0);      // plane 0 of this component
                m0[vop_elem(fi, cl)] = xn - s_mu[fd];
#pragma unroll
                for (int dd = 0; dd < D; ++dd) {
                    if (fd == dd) {
#pragma unroll
                        for (int k = 0; k < NBM; ++k)
                            if (k < DR::nbasis(dd))
                                m0[(size_t)(cw == 8 ? 1 + (k >> 1) : 1 + k) * pb.Np * 16 + vop_elem(fi, cw == 8 ? (k & 1) * 8 + cl : cl)] = ph[dd][k];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (t < 4 * PART_K) {
        const int k = t >> 2, q = t & 3;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += redk[(16 * q + u) * PART_K + k];
        s += dpp_f64<0xB1>(s);
        s += dpp_f64<0x4E>(s);
        if (q == 0) *(&ch.part[((size_t)cc * PART_K + k) * ch.n_wg + blk]) = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Boundary op of a chain (LeafPlan::vop, BoundaryOp): the state-sized work at a subtree's or a transition's end -- the finished subtree's
// last leaf becomes an end of the trajectory, rho += rho_sub and the trajectory-level U-turn sums; the proposal copy; the sample row; the
// momentum draw of the next transition; the first half / full step of the next doubling with its operand mirror and parameter block.
// All of it is element by element over the state vector (TFP's NoUTurnSampler: the tree merge of its loop_tree_doubling, and
// _start_trajectory_batched / the leapfrog's first step; wiring magi_v2.py:357-396), so it runs here on every workgroup of the point
// kernel -- one lane per entry, every load of a lane in ONE round trip -- instead of on the decisions' single workgroup, where the same
// passes were loops of four dependent round trips next to the saturating stream (25-100 us, the slot waiting for them).  The decisions
// (decide.h) keep the scalars; the sums this op leaves (rows PK_DOT, PK_DOT + 1: U-turn; PK_PP: p.p) reach them with the next slot.
// Workgroup 0 also takes the D + P parameter entries (a second pass of its first lanes).
// ------------------------------------------------------------------------------------------------------------------------------------
template <int DRIFT>
__device__ __forceinline__ void boundary_block(const DevProblem& pb, const DevChains& ch, const LeafPlan& lp, int cc, int blk,
                                               double* redk /* 64 * PART_K */, double* s_mu /* MAGI_MAX_D */, double* s_x /* PT_POINTS * PT_DSLOT */,
                                               int mirror_buf /* slot parity ^ 1 */) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P;
    const unsigned t = threadIdx.x;
    const int N = pb.N, ND = pb.ND;
    const size_t dimp = pb.dimp;
    const int vop = lp.vop;
    const bool ends = (vop & VOP_ENDS) != 0, cand = (vop & VOP_CAND) != 0, takel = (vop & VOP_TAKE_LEAF) != 0, outp = (vop & VOP_OUT) != 0;
    const bool draw = (vop & VOP_DRAW) != 0, dbl = (vop & VOP_DOUBLE) != 0;
    double* vb = ch.vec + vec_off(pb, cc, 0);
    if (t == PT_THREADS - 1) {
#pragma unroll
        for (int k = 0; k < MAGI_MAX_D; ++k) s_mu[k] = pb.mu[k];
    }
    const int fpt = t & (PT_POINTS - 1), fd = (t / PT_POINTS) & (PT_DSLOT - 1);
    const int fi = blk * PT_POINTS + fpt;
    const bool fvalid = (t < 64) && (fd < D) && (fi < N);
    double xn_grid = 0.0;
    if (t < 64) {
        double* pk = redk + (size_t)t * PART_K;
#pragma unroll
        for (int k = 0; k < PART_K; ++k) pk[k] = 0.0;
        const double* qleaf = vb + (size_t)(V_Q + lp.vleaf) * dimp;
        const double* g = vb + (size_t)V_G * dimp;
        const double* pleaf = vb + (size_t)V_PLEAF * dimp;
        double* pL = vb + (size_t)V_PL * dimp; double* qL = vb + (size_t)V_QL * dimp; double* gL = vb + (size_t)V_GL * dimp;
        double* pR = vb + (size_t)V_PR * dimp; double* qR = vb + (size_t)V_QR * dimp; double* gR = vb + (size_t)V_GR * dimp;
        double* candq = vb + (size_t)V_CANDQ * dimp; double* candg = vb + (size_t)V_CANDG * dimp;
        const double* subq = vb + (size_t)V_SUBQ * dimp; const double* subg = vb + (size_t)V_SUBG * dimp;
        double* rho = vb + (size_t)V_RHO * dimp; double* rhosub = vb + (size_t)V_RHOSUB * dimp;
        double* pE = lp.vdir > 0 ? pR : pL; double* qE = lp.vdir > 0 ? qR : qL; double* gE = lp.vdir > 0 ? gR : gL;
        const double* pO = lp.vdir > 0 ? pL : pR;                          // the other end
        const double* pS = lp.ndir > 0 ? pR : pL; const double* qS = lp.ndir > 0 ? qR : qL; const double* gS = lp.ndir > 0 ? gR : gL;
        const bool from_new_end = ends && lp.ndir == lp.vdir;              // the doubling starts from the end this op writes: its values are in registers
        const bool start_loads = dbl && !draw && !from_new_end;
        const int npass = (blk == 0) ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            const bool grid = pass == 0;
            const bool valid = grid ? fvalid : ((int)t < D + P);
            const int e = grid ? fd * N + fi : ND + (int)t;
            if (valid) {
                // ---- every load of this entry, one round trip ----
                double ql = 0.0, gl = 0.0, pn = 0.0, rho_v = 0.0, rs_v = 0.0, po = 0.0, sq = 0.0, sg = 0.0, cq = 0.0, cg = 0.0, p0 = 0.0, q0 = 0.0, g0 = 0.0;
                if (ends || (cand && takel)) { ql = qleaf[e]; gl = g[e]; }
                if (ends) { pn = pleaf[e]; rho_v = rho[e]; rs_v = rhosub[e]; po = pO[e]; }
                if (cand && !takel) { sq = subq[e]; sg = subg[e]; }
                if ((outp || draw) && !cand) { cq = candq[e]; cg = candg[e]; }
                if (start_loads) { p0 = pS[e]; q0 = qS[e]; g0 = gS[e]; }
                // ---- merge ----
                if (ends) {
                    pE[e] = pn; qE[e] = ql; gE[e] = gl;
                    const double rr = rho_v + rs_v;
                    rho[e] = rr;
                    pk[PK_DOT] += rr * po;
                    pk[PK_DOT + 1] += rr * pn;
                    if (from_new_end) { p0 = pn; q0 = ql; g0 = gl; }
                }
                if (cand) {
                    cq = takel ? ql : sq; cg = takel ? gl : sg;
                    candq[e] = cq; candg[e] = cg;
                }
                if (outp) ch.samples[(size_t)lp.vout * dimp + e] = cq;
                // ---- next transition: momentum, both ends = the proposal ----
                if (draw) {
                    const double z = rng_normal_elem((unsigned)e, lp.step_k, lp.chain_id, lp.seed);
                    pL[e] = z; pR[e] = z; rho[e] = z;
                    qL[e] = cq; qR[e] = cq; gL[e] = cg; gR[e] = cg;
                    pk[PK_PP] += z * z;
                    p0 = z; q0 = cq; g0 = cg;
                }
                // ---- next doubling: first half / full step from the selected end (identity mass: p_half = p + eps/2 grad, q' = q + eps p_half) ----
                if (dbl) {
                    const double ph = p0 + lp.hs * g0;
                    *(vb + (size_t)(V_P + lp.cur) * dimp + e) = ph;
                    const double qn = q0 + lp.eps * ph;
                    *(vb + (size_t)(V_Q + lp.cur) * dimp + e) = qn;
                    rhosub[e] = 0.0;
                    if (grid) {
                        xn_grid = qn;
                        if (ch.mc) ch.xop[xop_off(pb, ch.n_chains, mirror_buf, cc, fd, fi)] = qn;      // (read by the next slot's matrix-core stream)
                    } else {
                        compute_par_entry(pb, (int)t, qn, ch.par + (size_t)cc * PAR_COUNT, true, nullptr);
                    }
                }
            }
        }
        if constexpr (DR::SEP) {
            // operand mirror of the new state for the next slot's separable stream: as in point_block_sep (one wave, LDS exchange of a point's components)
            if (ch.sep && dbl) {
                constexpr int NBM = DR::NBMAX;
                s_x[fpt * PT_DSLOT + fd] = xn_grid;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (fvalid) {
                    double xq[D], ph[D][NBM];
#pragma unroll
                    for (int dd = 0; dd < D; ++dd) xq[dd] = s_x[fpt * PT_DSLOT + dd];
                    DR::basis(xq, ph);
                    const int cw = xop_width(ch.n_chains), groups = (ch.n_chains + 15) >> 4, cl = cc & 15;
                    const int planes = 1 + (NBM * cw + 15) / 16;
                    double* m0 = ch.vop + vop_off(D, planes, pb.Np, groups, mirror_buf, cc >> 4, fd, 0, 0);
                    m0[vop_elem(fi, cl)] = xn_grid - s_mu[fd];
#pragma unroll
                    for (int dd = 0; dd < D; ++dd) {
                        if (fd == dd) {
#pragma unroll
                            for (int k = 0; k < NBM; ++k)
                                if (k < DR::nbasis(dd))
                                    m0[(size_t)(cw == 8 ? 1 + (k >> 1) : 1 + k) * pb.Np * 16 + vop_elem(fi, cw == 8 ? (k & 1) * 8 + cl : cl)] = ph[dd][k];
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (t < 4 * PART_K) {
        const int k = t >> 2, q = t & 3;
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += redk[(16 * q + u) * PART_K + k];
        s += dpp_f64<0xB1>(s);
        s += dpp_f64<0x4E>(s);
        if (q == 0) *(&ch.part[((size_t)cc * PART_K + k) * ch.n_wg + blk]) = s;
    }
}
