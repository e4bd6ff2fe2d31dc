// k_stream: the sampler's streaming kernel -- the four single-phase mat-vecs of a gradient
//     hx = FH xc,  ex = FE xc,  etf = FE^T f,  kf = FK f        (xc = X - mu, f = drift(X, theta))
// over the packed 128 x 128 operator blocks (pack.hip): the lower block triangle of the symmetric FH and FK
// and every block of FE, each block serving a row-type and a column-type product in one pass.  A leapfrog slot
// is the kernel pair [k_stream, k_point]; the decisions of slot s - 1 (decide.h) ride in k_stream(s) as one extra
// workgroup per chain, off the critical path, and the stream itself assumes "same subtree, next leaf".
// (reference arithmetic: magi_v2.py:308-348)
#include "magi_internal.h"
#include "leap_reduce.h"
#include "leap_point.h"
#include "decide.h"

namespace {


// Transposed butterfly: v[0..8) per lane -> every lane returns the 64-lane sum of v[lane >> 3].
// Halving steps hand half of the values to the partner (v_permlane32/16_swap move both halves in one
// instruction pair), so 8 row sums cost 7 exchanges + 3 plain steps instead of 8 x 6.
__device__ __forceinline__ double swap_add32(double a, double b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double tsum8(const double (&v)[8], int lane) {
    // lanes 32..63 keep rows 4..7, lanes 0..31 rows 0..3
    const double s0 = swap_add32(v[0], v[4]), s1 = swap_add32(v[1], v[5]), s2 = swap_add32(v[2], v[6]), s3 = swap_add32(v[3], v[7]);
    // odd rows of 16 lanes keep the upper two of those
    const double u0 = swap_add16(s0, s2), u1 = swap_add16(s1, s3);
    // lanes with bit 3 set keep u1 (partner: row_mirror, which flips bit 3)
    const bool hi8 = (lane & 8) != 0;
    const double keep = hi8 ? u1 : u0, send = hi8 ? u0 : u1;
    double w = keep + dpp_f64<0x140>(send);
    w += dpp_f64<0x141>(w);   // row_half_mirror (stays inside the 8-lane group)
    w += dpp_f64<0x4E>(w);
    w += dpp_f64<0xB1>(w);
    return w;
}

#ifndef MAGI_ST_WAVES
#define MAGI_ST_WAVES 4
#endif
constexpr int ST_WAVES = MAGI_ST_WAVES;        // waves per block task
constexpr int ST_RW = MAGI_TB / ST_WAVES;      // rows of the block per wave

// ---- streaming kernel: one TB x TB block of FH / FK / FE per workgroup ----------------------------------
// grid (n_tasks + 1, ceil(n_chains / NC)), block 64 * ST_WAVES.  Wave w streams rows [w*RW, (w+1)*RW) of the block with
// 16-B coalesced loads (lane = two columns) and forms, for up to NC chains sharing the bytes,
//   row-type products   (A v_col)[r]   -> transposed butterflies, one partial per block row
//   column-type products (A^T v_row)[c] -> per-lane accumulators, combined over the waves in LDS
// where v is xc = X_d - mu_d or f_d = drift_d(X, theta) as the operator requires (evaluated on the fly from the
// state vector).  The partials go to tpart[chain][vec][d][other block][i]; k_point adds them in fixed order.
template <int NC, int DRIFT>
__global__ __launch_bounds__(64 * ST_WAVES) __attribute__((amdgpu_waves_per_eu(NC <= 2 ? 3 : 2)))
void k_stream(DevProblem pb, DevChains ch, SamplerCfgDev cfg, int parity) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P, TB = MAGI_TB;
    // (the "all chains idle" flag is fetched here but tested after the other first loads are on the wire: an early return on it
    //  would put one more dependent round trip in front of every workgroup of every slot)
    const int all_done = ch.gctl->all_done;
    kernarg_prefetch<sizeof(DevProblem) + sizeof(DevChains) + sizeof(SamplerCfgDev) + sizeof(int)>();
    const int c0 = blockIdx.y * NC;
    __shared__ double vcol[NC][TB], vrow[NC][TB], rowout[NC][TB], colacc[ST_WAVES][NC][TB];
    __shared__ double th_s[NC][MAGI_MAX_P];
    const int n_dec = (int)gridDim.x - pb.n_tasks;        // decision workgroups come FIRST in dispatch order: their one round of
    if ((int)blockIdx.x < n_dec) {                         // loads is then on the wire before the stream saturates the memory system
        // ---- the decisions of the previous slot, one workgroup per chain, next to this slot's stream (decide.h) ----
        __shared__ double dsh[25 * 16], dshs[24];
        __shared__ ChainCtl s_ctl;
        __shared__ int s_g[2];
        __shared__ double s_par[PAR_COUNT];
        __shared__ double s_ops[OPS_COUNT * OPS_W];
        __shared__ double s_cst[2 * MAGI_MAX_D];
        const int chain = c0 + (int)blockIdx.x;
        if (chain < ch.n_chains) decide_block<DRIFT>(pb, ch, cfg, chain, parity, all_done, dsh, dshs, &s_ctl, s_g, s_par, s_ops, s_cst);
        return;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tix = (int)blockIdx.x - n_dec;
    // (the descriptor table is read-only for the lifetime of the matrices: fetched through the constant address space it is a
    //  scalar load on its own counter, so waiting for it does not wait for the tile loads issued below and vice versa)
    typedef const int __attribute__((address_space(4))) * const_int_ptr;
    const_int_ptr tk = (const_int_ptr)(unsigned long long)(pb.tasks + 4 * (size_t)tix);
    int4 task;
    task.x = tk[0]; task.y = tk[1]; task.z = tk[2]; task.w = tk[3];
    // the wave's rows in chunks of 8, two chunks in flight (a0 / a1): 16 KB per wave on the wire while one chunk is in the ALUs.
    // tile loads need nothing but the block index: on the wire before the plan-dependent loads (except in the waves that derive theta')
    const double2* A = reinterpret_cast<const double2*>(pb.tiles + (size_t)tix * TB * TB + (size_t)((threadIdx.x >> 6) * ST_RW) * TB) + (threadIdx.x & 63);
    constexpr int NCK = ST_RW / 8;
    double2 a0[8], a1[8];
    if ((threadIdx.x >> 6) >= NC) {
#pragma unroll
        for (int r = 0; r < 8; ++r) a0[r] = A[(size_t)r * (TB / 2)];
#pragma unroll
        for (int r = 0; r < 8; ++r) a1[r] = A[(size_t)(8 + r) * (TB / 2)];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int d = task.x, kind = task.y, bi = task.z, bj = task.w;
    const int N = pb.N;

    // What to evaluate for each chain, from the plan the point phase executed LAST (the decisions that complete it run
    // concurrently and may not be read): a leaf -> assume the subtree continues: the speculative state in the other
    // buffer, with theta' derived here exactly as the decisions derive it; a skip-type plan -> the buffer as is.
    const bool isrow = t >= TB;
    const int loc = (isrow ? t - TB : t) & (TB - 1);
    const int gi = (isrow ? bi : bj) * TB + loc;
    const bool wantf = isrow ? (kind != TK_FH) : (kind == TK_FK);
    const double mud = MAGI_SEL_D(pb.mu, d);
    double xin[NC][D];
    bool act[NC];
    // (small loads first, the tile stream behind them: their wait then does not cover the row loads)
    // (blocks of FH multiply xc on both sides: they need no theta and do not wait for it -- their traffic fills the window
    //  in which the other workgroups derive theta')
    if (wave < NC && kind != TK_FH) {
        const int c = wave, cc = min(c0 + c, ch.n_chains - 1);
        const LeafPlan* lp = ch.plan + (size_t)(parity ^ 1) * ch.n_chains + cc;
        const bool derive = lp->active && !lp->skip && lp->leaf;
        double thp = 0.0;
        if (derive) {
            const double* vb = ch.vec + vec_off(pb, cc, 0);
            const double* part = ch.part + (size_t)cc * PART_K * ch.n_wg;
            // (every load of the derivation is issued before the first wait)
            const int e = pb.ND + D + min(lane, P - 1);
            const double qv = (vb + (size_t)(V_Q + lp->cur) * pb.dimp)[e], pv = (vb + (size_t)(V_P + lp->cur) * pb.dimp)[e];
            double rows[P];
            part_rows_sum<P>(part, ch.n_wg, PK_TP, lane, rows);
            double tpp = 0.0;
#pragma unroll
            for (int k = 0; k < P; ++k) if (lane == k) tpp = rows[k];
            if (lane < P) {
                const double ex = m_exp(qv);
                const double sg = ex / (1.0 + ex);                       // == par[PAR_SGT] of that state (compute_par_entry)
                const double qnx = next_entry_pre(pv, qv, lp->hs, lp->eps, theta_entry_grad(pb.beta_inv, tpp, sg));
                thp = m_log(1.0 + m_exp(qnx));                           // == par'[PAR_TH] (compute_par_entry)
            }
        } else if (lane < P) {
            thp = ch.par[(size_t)cc * PAR_COUNT + PAR_TH + lane];
        }
        if (lane < P) th_s[c][lane] = thp;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cc = min(c0 + c, ch.n_chains - 1);
        const LeafPlan* lp = ch.plan + (size_t)(parity ^ 1) * ch.n_chains + cc;
        act[c] = (c0 + c < ch.n_chains) && lp->active != 0;
        const int buf = (lp->skip || !lp->leaf) ? lp->cur : (lp->cur ^ 1);
        const double* q = ch.vec + vec_off(pb, cc, V_Q + buf);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) xin[c][dd] = q[dd * N + min(gi, N - 1)];
    }

    if (all_done) return;
#ifdef MAGI_TAIL_STAMPS
    if (tix == 0 && threadIdx.x == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(ch.par + (size_t)c0 * PAR_COUNT + 40 + 11);
        st[1] = __builtin_amdgcn_s_memrealtime();      // [12] first stream workgroup's start
        st[0] = 0ull;                                    // [11] latest stream workgroup end (atomicMax below)
    }
#endif
    if (wave < NC) {                 // (the waves that derived theta' issue their first chunks now)
#pragma unroll
        for (int r = 0; r < 8; ++r) a0[r] = A[(size_t)r * (TB / 2)];
#pragma unroll
        for (int r = 0; r < 8; ++r) a1[r] = A[(size_t)(8 + r) * (TB / 2)];
    }
    __builtin_amdgcn_sched_barrier(0);
    double thv[NC][P];
    if (kind != TK_FH) {
        __syncthreads();             // th_s
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int k = 0; k < P; ++k) thv[c][k] = th_s[c][k];
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int k = 0; k < P; ++k) thv[c][k] = 0.0;
    }

#pragma unroll
    for (int c = 0; c < NC; ++c) {
        double xd = xin[c][0];
#pragma unroll
        for (int dd = 1; dd < D; ++dd) if (d == dd) xd = xin[c][dd];
        double val = wantf ? DR::f1(d, xin[c], thv[c]) : xd - mud;
        if (gi >= N) val = 0.0;
        if (t < 2 * TB) (isrow ? vrow : vcol)[c][loc] = val;
    }
    __syncthreads();

    double2 vc[NC], cacc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        vc[c] = *reinterpret_cast<const double2*>(&vcol[c][2 * lane]);
        cacc[c].x = 0.0; cacc[c].y = 0.0;
    }
#pragma unroll
    for (int ck = 0; ck < NCK; ++ck) {
        double2 (&a)[8] = (ck & 1) ? a1 : a0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double p[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double2 ar = a[r];
                p[r] = fma(ar.y, vc[c].y, ar.x * vc[c].x);
                const double xr = vrow[c][wave * ST_RW + ck * 8 + r];
                cacc[c].x = fma(ar.x, xr, cacc[c].x);
                cacc[c].y = fma(ar.y, xr, cacc[c].y);
            }
            const double s = tsum8(p, lane);
            // all 8 lanes of a group hold the same bits (commutative butterflies): an unconditional store keeps the loop
            // free of branches (with them LLVM sinks the column accumulators behind the loop and the tile stays live)
            rowout[c][wave * ST_RW + ck * 8 + (lane >> 3)] = s;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ck + 2 < NCK) {
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = A[(size_t)((ck + 2) * 8 + r) * (TB / 2)];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) *reinterpret_cast<double2*>(&colacc[wave][c][2 * lane]) = cacc[c];
    __syncthreads();

    // partials: threads [0, TB) the row-type output (block row bi, slot bj), threads [TB, 2 TB) the
    // column-type output (block row bj, slot bi; the diagonal blocks of FH / FK are complete by rows)
    const int rvec = kind == TK_FH ? TV_HX : kind == TK_FK ? TV_KF : TV_EX;
    const int cvec = kind == TK_FH ? TV_HX : kind == TK_FK ? TV_KF : TV_ETF;
    const bool colout = (kind == TK_FE) || (bi != bj);
    const size_t cstride = (size_t)4 * D * pb.nb * pb.Np;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (!act[c] || t >= 2 * TB) continue;
        double* tp = ch.tpart + (size_t)(c0 + c) * cstride;
        if (!isrow) {
            tp[((size_t)(rvec * D + d) * pb.nb + bj) * pb.Np + bi * TB + loc] = rowout[c][loc];
        } else if (colout) {
            double sum = colacc[0][c][loc];
#pragma unroll
            for (int w = 1; w < ST_WAVES; ++w) sum += colacc[w][c][loc];
            tp[((size_t)(cvec * D + d) * pb.nb + bi) * pb.Np + bj * TB + loc] = sum;
        }
    }
#ifdef MAGI_TAIL_STAMPS
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned long long*>(ch.par + (size_t)c0 * PAR_COUNT + 40 + 11), (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// ---- point kernel: the elementwise half of slot `parity` (leap_point.h), N / 16 workgroups per chain -------------------------
template <int DRIFT>
__global__ __launch_bounds__(PT_THREADS) void k_point(DevProblem pb, DevChains ch, int parity) {
    __shared__ double res[PT_POINTS * PT_DSLOT * 4];
    __shared__ double redk[64 * PART_K];
    __shared__ double s_mu[MAGI_MAX_D];
    // (flag and plan are fetched together and combined arithmetically: `a || b` would fetch b only after a has arrived --
    //  one more dependent round trip at the head of a 5 us kernel)
    const int all_done = ch.gctl->all_done;
    const LeafPlan lp = ch.plan[(size_t)parity * ch.n_chains + blockIdx.y];
    kernarg_prefetch<sizeof(DevProblem) + sizeof(DevChains) + sizeof(int)>();
    const int gate = all_done | (lp.active ^ 1) | lp.skip;
    if (gate != 0) return;
    point_block<DRIFT>(pb, ch, lp, blockIdx.y, blockIdx.x, res, redk, s_mu);
}

// load-only twin of k_stream's tile stream (bench.py's ceiling leg): every workgroup reads its 128 KB block with the same
// 16-B-per-lane pattern and does nothing else -- what the memory system delivers for this layout and working set
__global__ __launch_bounds__(256) void k_read_tiles(const double2* __restrict__ p, double* out) {
    const double2* q = p + (size_t)blockIdx.x * (MAGI_TB * MAGI_TB / 2) + threadIdx.x;
    double a = 0.0;
#pragma unroll
    for (int r = 0; r < MAGI_TB * MAGI_TB / 2 / 256; ++r) { const double2 v = q[r * 256]; a += v.x + v.y; }
    if (a == 12345.678) out[0] = a;
}

// validation plan (magi_logpost_grad_fused / timing): slot 0 evaluates buffer 0 as is, no leapfrog
__global__ void k_plan_eval(DevChains ch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ch.n_chains) return;
    LeafPlan prev{};
    prev.active = 1; prev.skip = 1;           // what k_stream(parity 0) reads: "evaluate buffer 0 as is"
    ch.plan[(size_t)ch.n_chains + c] = prev;
    LeafPlan p{};
    p.active = 1;                             // what k_point(parity 0) executes: gradient only
    ch.plan[c] = p;
}

template <int DRIFT>
__global__ __launch_bounds__(MAGI_TAIL_THREADS) void k_leap_finalize(DevProblem pb, DevChains ch, double* out) {
    __shared__ double sh[25 * 16];
    __shared__ double shs[16];
    const int c = blockIdx.x;
    double* vb = ch.vec + vec_off(pb, c, 0);
    const LeafPlan lp = ch.plan[c];
    double pre[RedLayout<DRIFT>::PER_WAVE];
    leap_reduce_issue<DRIFT>(ch, c, pre);
    double* par = ch.par + (size_t)c * PAR_COUNT;
    const ReduceOut ro = leap_reduce<DRIFT>(pb, ch, c, vb, par, par, lp, pre, sh, shs);
    if (threadIdx.x == 0 && out) {
        out[c * 8 + 0] = ro.L;
        out[c * 8 + 1] = ro.t12;
        out[c * 8 + 2] = 0.0;
        out[c * 8 + 3] = ro.t3;
        out[c * 8 + 4] = ro.t4;
    }
}

template <int NC, int DRIFT>
int launch_stream_nd(magi_handle* h, int n_chains, int parity, bool with_decisions, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const dim3 grid(pb.n_tasks + (with_decisions ? NC : 0), (n_chains + NC - 1) / NC);      // + one decision workgroup per chain
    hipLaunchKernelGGL((k_stream<NC, DRIFT>), grid, dim3(64 * ST_WAVES), 0, s, pb, h->ch, h->cfg, parity);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("stream launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

template <int NC>
int launch_stream_nc(magi_handle* h, int n_chains, int parity, bool with_decisions, hipStream_t s) {
#define MAGI_CALL(DR) return launch_stream_nd<NC, DR>(h, n_chains, parity, with_decisions, s)
    MAGI_DRIFT_DISPATCH(h->pb.drift, MAGI_CALL);
#undef MAGI_CALL
    return MAGI_OK;
}

}  // namespace

int magi_leap_wgs(const DevProblem& pb) { return (pb.N + PT_POINTS - 1) / PT_POINTS; }

int magi_launch_stream(magi_handle* h, int n_chains, int parity, bool with_decisions, hipStream_t s) {
    if (n_chains >= 3) return launch_stream_nc<4>(h, n_chains, parity, with_decisions, s);
    if (n_chains == 2) return launch_stream_nc<2>(h, n_chains, parity, with_decisions, s);
    return launch_stream_nc<1>(h, n_chains, parity, with_decisions, s);
}

int magi_launch_point(magi_handle* h, int n_chains, int parity, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const dim3 g(magi_leap_wgs(pb), n_chains), b(PT_THREADS);
#define MAGI_CALL(DR) hipLaunchKernelGGL(k_point<DR>, g, b, 0, s, pb, h->ch, parity)
    MAGI_DRIFT_DISPATCH(pb.drift, MAGI_CALL);
#undef MAGI_CALL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("point launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_read_tiles(magi_handle* h, hipStream_t s) {
    hipLaunchKernelGGL(k_read_tiles, dim3(h->pb.n_tasks), dim3(256), 0, s, reinterpret_cast<const double2*>(h->pb.tiles), h->d_fin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("read_tiles launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_plan_eval(magi_handle* h, int n_chains, hipStream_t s) {
    hipLaunchKernelGGL(k_plan_eval, dim3((n_chains + 63) / 64), dim3(64), 0, s, h->ch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("plan launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_leap_finalize(magi_handle* h, int n_chains, double* d_out, hipStream_t s) {
    const dim3 g(n_chains), b(MAGI_TAIL_THREADS);
#define MAGI_CALL(DR) hipLaunchKernelGGL(k_leap_finalize<DR>, g, b, 0, s, h->pb, h->ch, d_out)
    MAGI_DRIFT_DISPATCH(h->pb.drift, MAGI_CALL);
#undef MAGI_CALL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("leap_finalize launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}
