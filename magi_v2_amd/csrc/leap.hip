// k_stream: the sampler's streaming kernel -- the four single-phase mat-vecs of a gradient
//     hx = FH xc,  ex = FE xc,  etf = FE^T f,  kf = FK f        (xc = X - mu, f = drift(X, theta))
// over the packed 128 x 128 operator blocks (pack.hip): the lower block triangle of the symmetric FH and FK
// and every block of FE, each block serving a row-type and a column-type product in one pass.  A leapfrog slot
// is the kernel pair [k_stream, k_point]; the decisions of slot s - 1 (decide.h) ride in k_stream(s) as one extra
// workgroup per chain, off the critical path, and the stream itself assumes "same subtree, next leaf".
// (reference arithmetic: magi_v2.py:308-348)
#include <cstdlib>
#include <utility>

#include "magi_internal.h"
#include "leap_reduce.h"
#include "leap_point.h"
#include "decide.h"

namespace {

// dev (-DMAGI_WG_TRACE): begin / end (100 MHz real-time counter), hardware placement and a tag of EVERY workgroup of the last launch of
// the streaming kernel [0] and of k_point [1] (tools/exp_wg_trace.py: where a launch's time goes across the grid)
#ifdef MAGI_WG_TRACE
__device__ unsigned long long g_wg_trace[2][4096][4];
struct WgTrace {
    int kern, wg;
    __device__ __forceinline__ WgTrace(int kern_, int tag) : kern(kern_) {
        wg = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
        if (threadIdx.x == 0 && wg < 4096) {
            g_wg_trace[kern][wg][0] = __builtin_amdgcn_s_memrealtime();
            g_wg_trace[kern][wg][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
            g_wg_trace[kern][wg][3] = (unsigned long long)(long long)tag;
        }
    }
    __device__ __forceinline__ ~WgTrace() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the workgroup's stores have retired)
        if ((threadIdx.x & 63) == 0 && wg < 4096) atomicMax(&g_wg_trace[kern][wg][1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
};
#define WG_TRACE(kern, tag) WgTrace wg_trace_((kern), (tag))
#else
#define WG_TRACE(kern, tag) do { } while (0)
#endif

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
#ifdef MAGI_ST_STAMPS      // dev: 100 MHz time stamps of ONE stream workgroup (task MAGI_ST_STAMPS, wave MAGI_ST_STAMP_WAVE) into par[40 ..] of chain 0
#ifndef MAGI_ST_STAMP_WAVE //      (tools/exp_st_stamps.py); kept in scalar registers and written once at the end -- a store per stamp spills the kernel (12 -> 27 us)
#define MAGI_ST_STAMP_WAVE 1
#endif
#define ST_STAMP_DECL unsigned long long st_stamps[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}
#define ST_STAMP(i) do { st_stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ST_STAMP_FLUSH() do { if ((int)blockIdx.x - n_dec == (MAGI_ST_STAMPS) && blockIdx.y == 0 && threadIdx.x == 64 * (MAGI_ST_STAMP_WAVE)) \
    for (int _i = 0; _i < 8; ++_i) reinterpret_cast<unsigned long long*>(ch.par + 40)[_i] = st_stamps[_i]; } while (0)
#else
#define ST_STAMP_DECL do { } while (0)
#define ST_STAMP(i) do { } while (0)
#define ST_STAMP_FLUSH() do { } while (0)
#endif
template <int NC, int DRIFT>
__global__ __launch_bounds__(64 * ST_WAVES) __attribute__((amdgpu_waves_per_eu(3)))
void k_stream(DevProblem pb, DevChains ch, SamplerCfgDev cfg, int parity) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P, TB = MAGI_TB;
    WG_TRACE(0, 0);
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
        __shared__ double s_cst[3 * MAGI_MAX_D];
        const int chain = c0 + (int)blockIdx.x;
        if (chain < ch.n_chains) decide_block<DRIFT>(pb, ch, cfg, chain, parity, all_done, dsh, dshs, &s_ctl, s_g, s_par, s_ops, s_cst);
        return;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tix = (int)blockIdx.x - n_dec;
    ST_STAMP_DECL;
    ST_STAMP(0);
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
    // Odd slots walk the wave's row chunks backwards: what a slot read LAST is what the next one reads FIRST, so the tail of the
    // block stream is still in the XCD's L2 (4 MB against 9 MB of blocks per XCD; tools/micro/readshape.hip: 5-15 % on the
    // load-only twin).  Chunk ck of the walk is physical chunk pc(ck); results are summed per PHYSICAL chunk so that they do
    // not depend on the direction.  (Only the one- and two-chain instantiations: four chains have no registers for it.)
    static_assert(NC <= 2, "three or more chains per pass run k_stream_mc");
    constexpr bool ALT = true;
    const int pc0 = (ALT && (parity & 1)) ? NCK - 1 : 0, pcs = (ALT && (parity & 1)) ? -1 : 1;
    double2 a0[8], a1[8];
    if ((threadIdx.x >> 6) >= NC) {
#pragma unroll
        for (int r = 0; r < 8; ++r) a0[r] = A[(size_t)(pc0 * 8 + r) * (TB / 2)];
#pragma unroll
        for (int r = 0; r < 8; ++r) a1[r] = A[(size_t)((pc0 + pcs) * 8 + r) * (TB / 2)];
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
        ST_STAMP(1);
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
        for (int r = 0; r < 8; ++r) a0[r] = A[(size_t)(pc0 * 8 + r) * (TB / 2)];
#pragma unroll
        for (int r = 0; r < 8; ++r) a1[r] = A[(size_t)((pc0 + pcs) * 8 + r) * (TB / 2)];
    }
    __builtin_amdgcn_sched_barrier(0);
    double thv[NC][P];
    ST_STAMP(2);
    if (kind != TK_FH) {
        __syncthreads();             // th_s
        ST_STAMP(3);
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
    ST_STAMP(4);

    constexpr int NACC = ALT ? NCK : 1;            // column accumulators per physical chunk (direction-independent sums)
    double2 vc[NC], cacc[NC][NACC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        vc[c] = *reinterpret_cast<const double2*>(&vcol[c][2 * lane]);
#pragma unroll
        for (int k = 0; k < NACC; ++k) { cacc[c][k].x = 0.0; cacc[c][k].y = 0.0; }
    }
#pragma unroll
    for (int ck = 0; ck < NCK; ++ck) {
        double2 (&a)[8] = (ck & 1) ? a1 : a0;
        const int row0 = wave * ST_RW + (pc0 + pcs * ck) * 8;          // first row of this chunk inside the block
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double p[8];
            double2& acc = cacc[c][ALT ? ck : 0];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const double2 ar = a[r];
                p[r] = fma(ar.y, vc[c].y, ar.x * vc[c].x);
                const double xr = vrow[c][row0 + r];
                acc.x = fma(ar.x, xr, acc.x);
                acc.y = fma(ar.y, xr, acc.y);
            }
            const double s = tsum8(p, lane);
            // all 8 lanes of a group hold the same bits (commutative butterflies): an unconditional store keeps the loop
            // free of branches (with them LLVM sinks the column accumulators behind the loop and the tile stays live)
            rowout[c][row0 + (lane >> 3)] = s;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ck + 2 < NCK) {
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = A[(size_t)((pc0 + pcs * (ck + 2)) * 8 + r) * (TB / 2)];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        double2 tot = cacc[c][0];
        if (ALT) {
            // physical chunk order 0, 1, .., NCK - 1 whatever the walk was: walk index of physical chunk k is k (even slots) or NCK - 1 - k
            const bool rev = (parity & 1) != 0;
            tot = rev ? cacc[c][NCK - 1] : cacc[c][0];
#pragma unroll
            for (int k = 1; k < NACC; ++k) {
                const double2 nx = rev ? cacc[c][NCK - 1 - k] : cacc[c][k];
                tot.x += nx.x; tot.y += nx.y;
            }
        }
        *reinterpret_cast<double2*>(&colacc[wave][c][2 * lane]) = tot;
    }
    ST_STAMP(5);
    __syncthreads();
    ST_STAMP(6);

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
    ST_STAMP(7);
    ST_STAMP_FLUSH();
#ifdef MAGI_TAIL_STAMPS
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned long long*>(ch.par + (size_t)c0 * PAR_COUNT + 40 + 11), (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// ---- multi-chain streaming kernel on the matrix cores -----------------------------------------------------------------------
// For 3 .. 16 chains per pass the chains are the 16-wide dimension of v_mfma_f64_16x16x4_f64: a 16 x 16 piece S of the operator
// block gives   row-type  (S v)[r, chain]   = sum_c S[r][c] V[c][chain]   and   column-type (S^T w)[c, chain] = sum_r S[r][c] W[r][chain]
// as 4 + 4 MFMAs whose accumulation IS the reduction -- no per-chain butterflies, and every chain shares every matrix byte.
// The two products contract over different tile indices, so the tile is needed in two lane layouts (an MFMA operand has its
// contraction index on lane >> 4 and its free index on lane & 15):
//   * the global load is shaped for the column-type product -- one instruction = 4 rows x 32 columns, lane = (row j = lane >> 4,
//     column pair i = lane & 15): four fully used 256-B segments -- and feeds those MFMAs straight from the registers;
//   * the same registers are written to a wave-private 16 x 32 LDS patch (272-B rows) and read back transposed (lane = (row i,
//     column pair j)) for the row-type MFMAs; wave-private, so no barrier: LDS operations of one wave execute in order.
// Wave w owns rows [32 w, 32 w + 32) of the block = 2 chunks of 16 rows x 4 column groups of 32: eight steps, a ring of
// MC_RING steps' loads in flight.
// The operand vectors xc = X - mu and f = drift(X, theta') are formed per workgroup from ch.xop, the mirror of the positions a
// slot evaluates in the order [slot parity][d][grid index][8 or 16 chains] (written by the point phase and the decisions of the
// slot before): a block's slice of all chains is a contiguous, fully coalesced read found without a plan look-up -- gathering 16
// chains x D components from the per-chain state vectors instead moved as many L2 bytes as the operator blocks themselves.
// theta' is derived per workgroup as in k_stream.
constexpr int MC = 16;                                   // chain columns per pass
constexpr int MC_PITCH = 34;                             // doubles per staged row: 32 columns + 2 (272 B: conflict-free transposed reads)
constexpr int MC_SM_V = 0, MC_SM_ST = MC_SM_V + MAGI_TB * MC, MC_SM_CS = MC_SM_ST + 4 * 16 * MC_PITCH, MC_SM_TH = MC_SM_CS + MC * MAGI_TB,
              MC_SM_DOUBLES = MC_SM_TH + MC * 8;
using mc_d4 = __attribute__((ext_vector_type(4))) double;
#ifndef MAGI_MC_RING
#define MAGI_MC_RING 3
#endif
constexpr int MC_RING = MAGI_MC_RING;                    // steps of tile loads in flight per wave + 1 (ring of register buffers; 3: 22.7 us, 4: 23.2, 5: 23.6 at 8 chains)

#ifdef MAGI_MC_STAMPS      // dev: 100 MHz time stamps of ONE stream workgroup (task MAGI_MC_STAMPS) into par[40 ..] of chain 0 (tools/exp_mc_stamps.py)
#define MC_STAMP(i) do { if ((int)blockIdx.x - n_dec == (MAGI_MC_STAMPS) && blockIdx.y == 0 && threadIdx.x == 0) \
    reinterpret_cast<unsigned long long*>(ch.par + 40)[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MC_STAMP(i) do { } while (0)
#endif
template <int DRIFT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_stream_mc(DevProblem pb, DevChains ch, SamplerCfgDev cfg, int parity) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P, TB = MAGI_TB;
    static_assert(P <= 8, "theta slots");
    WG_TRACE(0, 0);
    const int all_done = ch.gctl->all_done;
    kernarg_prefetch<sizeof(DevProblem) + sizeof(DevChains) + sizeof(SamplerCfgDev) + sizeof(int)>();
    const int c0 = blockIdx.y * MC;
    // one LDS block: the stream workgroups carve their operand image and staging patches out of it, the decision workgroups their scratch
    __shared__ __attribute__((aligned(16))) double smem[MC_SM_DOUBLES];
    const int n_dec = (int)gridDim.x - pb.n_tasks;
    if ((int)blockIdx.x < n_dec) {
        double* dsh = smem;                                   // 25 * 16
        double* dshs = dsh + 25 * 16;                         // 24
        double* s_par = dshs + 24;                            // PAR_COUNT
        double* s_ops = s_par + PAR_COUNT;                    // OPS_COUNT * OPS_W
        double* s_cst = s_ops + OPS_COUNT * OPS_W;            // 3 * MAGI_MAX_D
        int* s_g = reinterpret_cast<int*>(s_cst + 3 * MAGI_MAX_D);
        ChainCtl* s_ctl = reinterpret_cast<ChainCtl*>(s_cst + 3 * MAGI_MAX_D + 2);
        static_assert(25 * 16 + 24 + PAR_COUNT + OPS_COUNT * OPS_W + 3 * MAGI_MAX_D + 2 + (sizeof(ChainCtl) + 7) / 8 <= MC_SM_DOUBLES, "decision scratch");
        const int chain = c0 + (int)blockIdx.x;
        if (chain < ch.n_chains) decide_block<DRIFT>(pb, ch, cfg, chain, parity, all_done, dsh, dshs, s_ctl, s_g, s_par, s_ops, s_cst);
        return;
    }
    const double2* vimg = reinterpret_cast<const double2*>(smem + MC_SM_V);    // [column pair][chain] = (v[2 cp], v[2 cp + 1]) of the block's column slice
    double* colsum = smem + MC_SM_CS;                                            // [chain][column]: running column-type sums
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lj = lane >> 4;
    double* stage = smem + MC_SM_ST + wave * 16 * MC_PITCH;
    const int tix = (int)blockIdx.x - n_dec;
    MC_STAMP(0);
    typedef const int __attribute__((address_space(4))) * const_int_ptr;
    const_int_ptr tk = (const_int_ptr)(unsigned long long)(pb.tasks + 4 * (size_t)tix);
    const int d = tk[0], kind = tk[1], bi = tk[2], bj = tk[3];
    const int nch = ch.n_chains;

    // Tile stream.  Wave w owns rows [32 w, 32 w + 32); its eight steps are ordered by column group: phase p = s >> 1 works on
    // group g = (w + p) & 3, chunk s & 1.  In every phase the four waves work on four DIFFERENT column groups, so the running
    // column-type sums of a group live in LDS and are owned by one wave at a time (fixed hand-over order: deterministic), and a
    // wave carries the accumulators of ONE group instead of four -- the registers go to the load ring instead.
    // lane = (row 4 q + lj, column pair li) of a 16 x 32 piece.
    const double2* A = reinterpret_cast<const double2*>(pb.tiles + (size_t)tix * TB * TB) + (size_t)(32 * wave + lj) * (TB / 2) + li;
    auto ld = [&](int s, int q) { return A[(size_t)(16 * (s & 1) + 4 * q) * (TB / 2) + 16 * ((wave + (s >> 1)) & 3)]; };
    double2 tl[MC_RING][4];
    const bool coltype = (kind == TK_FE) || (bi != bj);      // (diagonal blocks of FH / FK are stored full: complete by rows)
    const bool colf = kind == TK_FK, rowf = kind != TK_FH;   // which operand slices are drift values (else xc = x - mu)
    MC_STAMP(13);

    // Order of the prologue's loads.  Every workgroup of the grid starts at about the same time, and once the tile loads are out
    // (tens of MB across the device) anything issued behind them -- by anyone -- queues for microseconds; a wave's loads also
    // complete in order.  Measured with the tile ring issued first: theta' 4.7 us, operands 5.4 us, first MFMA 10 us after the
    // workgroup's start, half of its life.  So everything the operands depend on is issued FIRST, in one round with no dependent
    // address: the inputs of theta' (the parameter entries of BOTH position buffers: which one counts is in the plan, fetched in
    // the same round), the plans' `active` bits, and the raw positions of both operand slices from the operand-order mirror --
    // which is indexed by SLOT PARITY (whoever sets up the state a slot evaluates writes xop[that slot & 1]), not by the
    // ping-pong buffer the plan names.  The tile ring goes out behind them.
    // ---- (1) inputs of theta' (as k_stream: derived from the point phase's partial sums with the decisions' own functions when
    //      the last plan was a leaf, else the state's parameter block).  Wave w: chains 4 w .. 4 w + 3, one per 16-lane row of
    //      the wave for the transcendental part.  Blocks of FH multiply xc on both sides and need none. ----
    double* th_s = smem + MC_SM_TH;                                   // [chain][8]
    // (The prologue is instruction-bound, not latency-bound: 1 400 instructions per wave in its first version -- 64-bit address
    //  arithmetic per load, an exec-mask region per element -- took 3.5 us to ISSUE on a CU that runs twelve such waves.  Hence:
    //  wave-uniform bases + 32-bit byte offsets, the uniform cases hoisted out of the loops, the group's chains spread over the
    //  four waves for theta'.)
    const int ngrp = min(nch - c0, MC), kpw = (ngrp + 3) >> 2;       // chains of this group; chains per wave for theta'
    auto ldb = [](const double* sbase, unsigned boff) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(sbase) + boff); };
    double rows[4][P];
    int lcur = 0; double lhs = 0.0, leps = 0.0, qv0 = 0.0, qv1 = 0.0, pv0 = 0.0, pv1 = 0.0, parv = 0.0; bool lder = false;
    const int cg = wave * kpw + lj;                                    // this lane's own (chain cg of the group, parameter li)
    const bool mine = lj < kpw && cg < ngrp && li < P;
    if (kind != TK_FH) {
        const unsigned lo8 = (unsigned)min(lane, ch.n_wg - 1) * 8u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < kpw) {
                const double* part = ch.part + ((size_t)min(c0 + wave * kpw + k, nch - 1) * PART_K + PK_TP) * ch.n_wg;      // (wave-uniform)
#pragma unroll
                for (int r = 0; r < P; ++r) { const double v = ldb(part + (size_t)r * ch.n_wg, lo8); rows[k][r] = (lane < ch.n_wg) ? v : 0.0; }
                for (int w0 = 64; w0 < ch.n_wg; w0 += 64) {
#pragma unroll
                    for (int r = 0; r < P; ++r) if (w0 + lane < ch.n_wg) rows[k][r] += part[(size_t)r * ch.n_wg + w0 + lane];
                }
            }
        }
        if (mine) {
            const int cc = c0 + cg;
            const LeafPlan* lp = ch.plan + (size_t)(parity ^ 1) * nch + cc;
            lcur = lp->cur;
            lhs = lp->hs; leps = lp->eps;
            lder = lp->active && !lp->skip && lp->leaf;
            const double* vb = ch.vec + vec_off(pb, cc, 0) + pb.ND + D + li;
            qv0 = vb[(size_t)V_Q * pb.dimp]; qv1 = vb[(size_t)V_Q1 * pb.dimp];
            pv0 = vb[(size_t)V_P * pb.dimp]; pv1 = vb[(size_t)V_P1 * pb.dimp];
            parv = ch.par[(size_t)cc * PAR_COUNT + PAR_TH + li];
        }
    }
    MC_STAMP(14);
    // which of the 16 chain columns take part in this slot (bit c): every lane fetches its own chain's flag now, not at the stores
    const bool valid = li < ngrp;
    const unsigned long long actb = __ballot(valid && ch.plan[(size_t)(parity ^ 1) * nch + min(c0 + li, nch - 1)].active != 0);
    // ---- (2) raw positions: column slice of block column bj (thread = (chain li, 4 of the 64 column pairs)), this wave's rows
    //      of block row bi (column-type product only).  xop has Np rows per component: no clamp at the ragged end (those entries
    //      are selected away below). ----
    const int cw = xop_width(nch);
    const double* xg = ch.xop + xop_off(pb, nch, parity, c0, 0, 0);                               // the group's mirror (wave-uniform)
    const unsigned vcol = ((unsigned)(bj * TB + 2 * (t >> 4)) * (unsigned)cw + (unsigned)li) * 8u;   // + (32 k + hh) cw 8
    const unsigned vrow = ((unsigned)(bi * TB + 32 * wave + lj) * (unsigned)cw + (unsigned)li) * 8u; // + (16 cidx + 4 q) cw 8
    const unsigned cw8 = (unsigned)cw * 8u;
    double xcol[4][2][D], xrow[2][4][D];
    if (valid) {
        if (colf) {
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const double* xd = xg + (size_t)dd * pb.Np * cw;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) xcol[k][hh][dd] = ldb(xd, vcol + (unsigned)(32 * k + hh) * cw8);
            }
        } else {
            const double* xd = xg + (size_t)d * pb.Np * cw;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) xcol[k][hh][0] = ldb(xd, vcol + (unsigned)(32 * k + hh) * cw8);
        }
        if (coltype) {
            if (rowf) {
#pragma unroll
                for (int dd = 0; dd < D; ++dd) {
                    const double* xd = xg + (size_t)dd * pb.Np * cw;
#pragma unroll
                    for (int cidx = 0; cidx < 2; ++cidx)
#pragma unroll
                        for (int q = 0; q < 4; ++q) xrow[cidx][q][dd] = ldb(xd, vrow + (unsigned)(16 * cidx + 4 * q) * cw8);
                }
            } else {
                const double* xd = xg + (size_t)d * pb.Np * cw;
#pragma unroll
                for (int cidx = 0; cidx < 2; ++cidx)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xrow[cidx][q][0] = ldb(xd, vrow + (unsigned)(16 * cidx + 4 * q) * cw8);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    MC_STAMP(10);
    // ---- (3) theta' ----
    if (kind != TK_FH) {
        double tpp = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < kpw) {
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const double sres = wave_sum(rows[k][r]);
                    if (lj == k && li == r) tpp = sres;
                }
            }
        }
        MC_STAMP(11);
        double thp = parv;
        if (lder && mine) {
            const double qv = lcur ? qv1 : qv0, pv = lcur ? pv1 : pv0;
            const double ex = m_exp(qv);
            const double sg = ex / (1.0 + ex);                       // == par[PAR_SGT] of that state (compute_par_entry)
            const double qnx = next_entry_pre(pv, qv, lhs, leps, theta_entry_grad(pb.beta_inv, tpp, sg));
            thp = m_log(1.0 + m_exp(qnx));                           // == par'[PAR_TH] (compute_par_entry)
        }
        if (mine) th_s[cg * 8 + li] = thp;
    }
    if (all_done) return;
    MC_STAMP(1);
    __syncthreads();                 // th_s
    MC_STAMP(2);

    // ---- (4) column slice -> LDS, [column pair][chain]; row slice W[32 wave + 16 cidx + 4 q + lj][chain li] -> registers ----
    const double mud = MAGI_SEL_D(pb.mu, d);
    double thv[P];
#pragma unroll
    for (int k = 0; k < P; ++k) thv[k] = (kind != TK_FH) ? th_s[li * 8 + k] : 0.0;
    {
        double2 vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i0 = bj * TB + 2 * ((t >> 4) + 16 * k);
            double v0, v1;
            if (colf) { v0 = DR::f1(d, xcol[k][0], thv); v1 = DR::f1(d, xcol[k][1], thv); }
            else { v0 = xcol[k][0][0] - mud; v1 = xcol[k][1][0] - mud; }
            vv[k].x = (valid && i0 < pb.N) ? v0 : 0.0;
            vv[k].y = (valid && i0 + 1 < pb.N) ? v1 : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<double2*>(smem + MC_SM_V)[((t >> 4) + 16 * k) * MC + li] = vv[k];
    }
    // ---- the tile ring: behind every operand load (the raw column slice has left its registers to it; one or two steps of it
    //      in front of the operand loads were measured: 23.2 / 26.2 us against 22.7 -- they delay every workgroup's operands) ----
    __builtin_amdgcn_sched_barrier(0);
    MC_STAMP(12);
#pragma unroll
    for (int s = 0; s < MC_RING - 1; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) tl[s][q] = ld(s, q);
    __builtin_amdgcn_sched_barrier(0);
    double wf[2][4];
#pragma unroll
    for (int cidx = 0; cidx < 2; ++cidx)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = bi * TB + 32 * wave + 16 * cidx + 4 * q + lj;
            double w0 = 0.0;
            if (valid && coltype) w0 = rowf ? DR::f1(d, xrow[cidx][q], thv) : xrow[cidx][q][0] - mud;
            wf[cidx][q] = (i < pb.N) ? w0 : 0.0;
        }
    MC_STAMP(3);
    __syncthreads();                 // operand image complete
    MC_STAMP(4);

    mc_d4 accc[2], accr[2];
    accr[0] = mc_d4{0.0, 0.0, 0.0, 0.0}; accr[1] = mc_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int cidx = s & 1, g = (wave + (s >> 1)) & 3;
        if (s + MC_RING - 1 < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tl[(s + MC_RING - 1) % MC_RING][q] = ld(s + MC_RING - 1, q);
        }
        double2 (&tt)[4] = tl[s % MC_RING];
        if (cidx == 0) {
            // take over the column sums of group g (register r of lane (li, lj) = chain lj + 4 r, columns 32 g + 2 li + {0, 1})
            if (s == 0 || !coltype) { accc[0] = mc_d4{0.0, 0.0, 0.0, 0.0}; accc[1] = mc_d4{0.0, 0.0, 0.0, 0.0}; }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 u = *reinterpret_cast<const double2*>(&colsum[(lj + 4 * r) * TB + 32 * g + 2 * li]);
                    accc[0][r] = u.x; accc[1][r] = u.y;
                }
            }
        }
        // column-type: A = W^T fragment (m = chain, k = row), B = tile (k = row, n = column)
        if (coltype) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                accc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[cidx][q], tt[q].x, accc[0], 0, 0, 0);
                accc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[cidx][q], tt[q].y, accc[1], 0, 0, 0);
            }
        }
        // transpose through the wave's LDS patch
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(&stage[(4 * q + lj) * MC_PITCH + 2 * li]) = tt[q];
        // row-type: A = V^T fragment (m = chain, k = column), B = tile (k = column, n = row)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 b = *reinterpret_cast<const double2*>(&stage[li * MC_PITCH + 2 * (4 * q + lj)]);
            const double2 v = vimg[(16 * g + 4 * q + lj) * MC + li];
            accr[cidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, b.x, accr[cidx], 0, 0, 0);
            accr[cidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, b.y, accr[cidx], 0, 0, 0);
        }
        if (cidx == 1 && coltype) {
            // hand the column sums of group g on: the next phase's owner of g is another wave
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<double2*>(&colsum[(lj + 4 * r) * TB + 32 * g + 2 * li]) = double2{accc[0][r], accc[1][r]};
            __syncthreads();
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the next steps' fragment reads out of this one: they would not fit the register file)
        if (s & 1) MC_STAMP(5 + (s >> 1));
    }

    // ---- partials.  Accumulator layout: register r of lane (li, lj) = chain lj + 4 r, free index li. ----
    const int rvec = kind == TK_FH ? TV_HX : kind == TK_FK ? TV_KF : TV_EX;
    const int cvec = kind == TK_FH ? TV_HX : kind == TK_FK ? TV_KF : TV_ETF;
    const size_t cstride = (size_t)4 * D * pb.nb * pb.Np;
    // row-type: block row bi, slot bj; rows 32 wave + 16 cidx + li
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = lj + 4 * r;
        const bool act = ((actb >> c) & 1ull) != 0;
#pragma unroll
        for (int cidx = 0; cidx < 2; ++cidx)
            if (act)
                ch.tpart[(size_t)(c0 + c) * cstride + ((size_t)(rvec * D + d) * pb.nb + bj) * pb.Np + bi * TB + 32 * wave + 16 * cidx + li] = accr[cidx][r];
    }
    if (!coltype) { MC_STAMP(9); return; }
    // column-type: complete in LDS after the last phase's barrier; thread = (chain t >> 4, eight columns)
    {
        const int c = t >> 4, col = 8 * (t & 15);
        const bool act = ((actb >> c) & 1ull) != 0;
        if (act) {
            double2* dst = reinterpret_cast<double2*>(&ch.tpart[(size_t)(c0 + c) * cstride + ((size_t)(cvec * D + d) * pb.nb + bi) * pb.Np + bj * TB + col]);
            const double2* src = reinterpret_cast<const double2*>(&colsum[c * TB + col]);
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k] = src[k];
        }
    }
    MC_STAMP(9);
}

// ---- streaming kernel for SEPARABLE drifts: no parameter, no plan, no drift evaluation in front of the tile stream ------------------
// f_d(x, theta) = sum_k coef_{d,k}(theta) phi_{d,k}(x)  (DriftT<>::SEP): the operators are applied to the theta-FREE vectors xc_d and
// phi_{d,k}(x) of the state a slot evaluates, which whoever set that state up has already written -- in the kernel's own operand order --
// to the mirror ch.vop[slot parity] (SepLayout, magi_internal.h): the point phase's speculative next leaf, the decisions' subtree start,
// k_mirror.  The point phase of THIS slot combines the products with coef(theta) (leap_point.h, point_block_sep): theta of the evaluated
// state is complete by then (the decisions riding in this kernel finish it), so nothing in front of the tile stream hangs on the global
// sums of the slot before.  Prologue = ONE round of loads (operand slices, the chains' active bits), no transcendental.
// The 16 matrix-core columns of a pass are (basis function k, chain): up to 8 chains x 2 basis functions (CW = 8) or 16 chains x 1
// (CW = 16); further basis planes are served by further workgroups on grid.z (they re-read the tile: rare shapes only).
// Work per workgroup is EQUAL: a workgroup's time is its stream of tile bytes with the MFMAs hidden under it (30-31 ns per
// v_mfma_f64_16x16x4_f64 and SIMD, tools/micro/mfma_rate.hip: 7.7 us of matrix pipe at two workgroups per CU), a diagonal block of FH / FK
// has only the row-type product (half the MFMAs), and 544 blocks put a third workgroup on 32 of the 256 CUs -- so the task table
// (pack.hip: stasks) pairs the two diagonal blocks FH_bb, FK_bb of a component into ONE task of 16 row-type steps: dense N = 1024,
// D = 4: 480 + 32 = 512 tasks of 128 MFMAs per wave, two per CU.
// Tile stream, lane layouts, LDS transposition and the rotation of the column sums through LDS are those of k_stream_mc above.
// compile-time loop (the tile ring is a register array indexed by the step: the steps MUST be distinct straight-line code; a
// `#pragma unroll` over a body of this size is only a request, and a loop the optimiser keeps puts the ring in scratch)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifndef MAGI_SEP_RING
#define MAGI_SEP_RING 2      // steps of tile loads per wave in the ring.  8 chains, standalone: 2: 18.7-19.0 us; 3: 24.6 and 4: 30.7 as the code stands --
                             // at 168 registers the deeper rings SPILL inside the streaming body; 3 without spills (the pair's second operand slice
                             // loaded at step 5 instead of the prologue): 19.8, with the operand loads in front of the ring 20.3.  More bytes in
                             // flight per wave do not help this memory system (k_stream: 12.0 / 12.1 / 12.7 / 14.6 us for 1 / 2 / 3 / 4 row chunks).
#endif
constexpr int SEP_RING = MAGI_SEP_RING;
#ifndef MAGI_SEP_OCC
#define MAGI_SEP_OCC 3
#endif
template <int DRIFT, int CW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MAGI_SEP_OCC, MAGI_SEP_OCC)))
void k_stream_sep(DevProblem pb, DevChains ch, SamplerCfgDev cfg, int parity) {
    using DR = DriftT<DRIFT>;
    using SL = SepLayout<DRIFT>;
    constexpr int D = DR::D, TB = MAGI_TB, PLANES = SL::planes(CW), PST = SL::PS_TOTAL;
#ifdef MAGI_SEP_STAMPS      // dev: 100 MHz time stamps of ONE task's workgroup, kept in scalar registers, written at the end to par[40 ..] of chain 0 (tools/exp_sep_stamps.py)
    unsigned long long sst[16] = {0ull};
#define SEP_STAMP(i) do { sst[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SEP_STAMP(i) do { } while (0)
#endif
    SEP_STAMP(0);
    WG_TRACE(0, 0);
    const int all_done = ch.gctl->all_done;
    kernarg_prefetch<sizeof(DevProblem) + sizeof(DevChains) + sizeof(SamplerCfgDev) + sizeof(int)>();
    const int c0 = blockIdx.y * MC, z = blockIdx.z;
    __shared__ __attribute__((aligned(16))) double smem[MC_SM_DOUBLES];
    const int n_dec = (int)gridDim.x - pb.n_stasks;
    if ((int)blockIdx.x < n_dec) {
        if (z != 0) return;
        const int chain = c0 + (int)blockIdx.x;
        // (Measured as a function of its own -- not inlined, the kernel arguments re-read from the kernarg segment: the streaming body
        //  then allocates for itself, 0 spills -- but the decisions then run 40 us: 500 scratch accesses, each a memory round trip next
        //  to the saturating stream, and the slot waits for them: 45 us against 27.  Inlined, their spills stay off their hot path.)
        double* dsh = smem;                                   // 25 * 16
        double* dshs = dsh + 25 * 16;                         // 24
        double* s_par = dshs + 24;                            // PAR_COUNT
        double* s_ops = s_par + PAR_COUNT;                    // OPS_COUNT * OPS_W
        double* s_cst = s_ops + OPS_COUNT * OPS_W;            // 3 * MAGI_MAX_D
        int* s_g = reinterpret_cast<int*>(s_cst + 3 * MAGI_MAX_D);
        ChainCtl* s_ctl = reinterpret_cast<ChainCtl*>(s_cst + 3 * MAGI_MAX_D + 2);
        if (chain < ch.n_chains) decide_block<DRIFT>(pb, ch, cfg, chain, parity, all_done, dsh, dshs, s_ctl, s_g, s_par, s_ops, s_cst);
        return;
    }
    double* colsum = smem + MC_SM_CS;                                            // [matrix-core column][block column]: running column-type sums
    const double2* vimg = reinterpret_cast<const double2*>(smem + MC_SM_V);    // [column pair][matrix-core column]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lj = lane >> 4;
    double* stage = smem + MC_SM_ST + wave * 16 * MC_PITCH;
    const int tix = (int)blockIdx.x - n_dec;
#ifdef MAGI_SEP_STAGGER      // dev A/B: half of the workgroups enter MAGI_SEP_STAGGER x 64 cycles late (mode 0: odd tasks, 1: the second half of the grid)
#ifndef MAGI_SEP_STAGGER_MODE
#define MAGI_SEP_STAGGER_MODE 0
#endif
    if (MAGI_SEP_STAGGER_MODE == 0 ? (tix & 1) != 0 : tix >= pb.n_stasks / 2) {
        for (int k_ = 0; k_ < (MAGI_SEP_STAGGER); k_ += 100) __builtin_amdgcn_s_sleep(100);
    }
#endif
    typedef const int __attribute__((address_space(4))) * const_int_ptr;
    const_int_ptr tk = (const_int_ptr)(unsigned long long)(pb.stasks + 8 * (size_t)tix);
    const int d = tk[0], bi = tk[2], bj = tk[3];
    int kind = tk[1], tile0 = tk[4], tile1 = tk[5], kind1 = tk[6];
    SEP_STAMP(1);              // task descriptor known
    const int nch = ch.n_chains, ngrp = min(nch - c0, MC);
    // which products this (task, basis plane) has: the row-type product multiplies xc (FH, FE: plane 0 only) or the basis plane z (FK);
    // the column-type product xc (FH) or the basis plane z (FK, FE); diagonal blocks of FH / FK are complete by rows.  A pair task
    // (tile1 >= 0) is two diagonal blocks, FH then FK, of block row bi = bj.
    const int nbd = DR::nbasis(d), gzd = (nbd * CW + 15) >> 4;
    const bool phi_ok = z < gzd;
    if (tile1 >= 0) {
        const bool r0 = (kind == TK_FK) ? phi_ok : (z == 0), r1 = (kind1 == TK_FK) ? phi_ok : (z == 0);
        if (!r0) { tile0 = tile1; kind = kind1; tile1 = -1; }       // (only the second block has work on this plane)
        else if (!r1) tile1 = -1;
    }
    const bool pair = tile1 >= 0;
    const bool rowt = (kind == TK_FK) ? phi_ok : (z == 0);
    const bool colt = ((kind == TK_FE) || (bi != bj)) && ((kind == TK_FH) ? (z == 0) : phi_ok);
    if (!rowt && !colt) return;
    const bool rowphi = kind == TK_FK, colphi = kind != TK_FH;      // operand of the row- / column-type product is a basis plane (else xc)
    const bool rowphi1 = kind1 == TK_FK;

    // ---- tile stream (as k_stream_mc): wave w owns rows [32 w, 32 w + 32) of a block; eight steps per block, phase p = s >> 1 works on
    //      column group (w + p) & 3; a pair task streams its second block behind the first without a gap in the ring ----
    // (explicitly GLOBAL pointers: through the lambdas below the compiler no longer infers the address space, and a flat load counts on
    //  the LDS counter too -- every wait for a tile would also wait for the staging traffic, and the ring would run one step deep)
    typedef double __attribute__((ext_vector_type(2))) d2v;
    typedef const d2v __attribute__((address_space(1))) * gd2_ptr;
    // Rows of a 16-row piece are dealt to the lanes as row = 4 lj + q (load instruction q fetches rows q, 4 + q, 8 + q, 12 + q: still four
    // full 256-byte segments): a lane's four row-operand values of a piece are then FOUR CONSECUTIVE grid points = two 16-byte loads from
    // the pair-interleaved mirror instead of four 8-byte ones (round 3 dealt row = 4 q + lj).
    const gd2_ptr A0 = (gd2_ptr)(unsigned long long)(reinterpret_cast<const double2*>(pb.tiles + (size_t)tile0 * TB * TB) + (size_t)(32 * wave + 4 * lj) * (TB / 2) + li);
    const gd2_ptr A1 = (gd2_ptr)(unsigned long long)(reinterpret_cast<const double2*>(pb.tiles + (size_t)max(tile1, 0) * TB * TB) + (size_t)(32 * wave + 4 * lj) * (TB / 2) + li);
    auto ld = [&](int s, int q) -> double2 {
        const d2v v = ((s >> 3) ? A1 : A0)[(size_t)(16 * (s & 1) + q) * (TB / 2) + 16 * ((wave + ((s & 7) >> 1)) & 3)];
        return double2{v.x, v.y};
    };
    // (The ring goes out FIRST, right behind the task descriptor, then the operand slices.  Device time stamps of a workgroup
    //  (-DMAGI_SEP_STAMPS, tools/exp_sep_stamps.py; 8 chains, kernel 19-20 us): task known 0.5 us after entry, the ~30 vector-memory
    //  instructions of the prologue take 2.4 us to ISSUE on a CU whose other waves are streaming -- which is why a deeper ring makes
    //  the kernel slower (3 steps: 23.7 us, 4: 29.8) --, operands in LDS at 3.4 us, then 1.0-1.8 us per step (4 KB per wave and step:
    //  the CU's 8 waves draw ~26 GB/s, the device 6.7 TB/s: the steps run at the memory system's rate), stores out at 15 us.)
    const int nsteps = pair ? 16 : 8;
    double2 tl[SEP_RING][4];
#pragma unroll
    for (int s = 0; s < SEP_RING - 1; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) tl[s][q] = ld(s, q);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the one round of loads in front of the tile stream: active bits, column slice(s) (block bj), this wave's row slice (block bi) ----
    // (the chains' active flags are LOADED here and balloted at the stores: a ballot in front of the operand loads made every workgroup
    //  wait 1.5 us for the plan's round trip before its first load -- device time stamps, tools/exp_sep_stamps.py)
    //  -- and loaded LATE, two steps before the first store: held from the prologue on, the flag was the one register the streaming
    //  body spilled)
    typedef const int __attribute__((address_space(1))) * gi_ptr;       // (a GLOBAL load: a flat one would also sit on the LDS counter)
    const gi_ptr act_ptr = (gi_ptr)(unsigned long long)&ch.plan[(size_t)(parity ^ 1) * nch + min(c0 + li, nch - 1)].active;
    int act_flag = 0;
    const int groups = (nch + 15) >> 4;
    typedef const double __attribute__((address_space(1))) * gd_ptr;
    typedef const char __attribute__((address_space(1))) * gc_ptr;
    // (16-byte loads from the pair-interleaved mirror, vop_elem: the prologue's ~30 vector-memory instructions per wave took 2.2-2.4 us to
    //  ISSUE -- 240 of them on a CU, one address-processing pipe -- so their NUMBER is what counts: 12-16 now)
    auto ld16 = [](const double* sbase, unsigned boff) -> double2 { const d2v v = *(gd2_ptr)((gc_ptr)(unsigned long long)sbase + boff); return double2{v.x, v.y}; };
    const double* mcol = ch.vop + vop_off(D, PLANES, pb.Np, groups, parity, (int)blockIdx.y, d, rowphi ? 1 + z : 0, bj * TB);     // (wave-uniform)
    const double* mcol1 = ch.vop + vop_off(D, PLANES, pb.Np, groups, parity, (int)blockIdx.y, d, rowphi1 ? 1 + z : 0, bj * TB);
    const double* mrow = ch.vop + vop_off(D, PLANES, pb.Np, groups, parity, (int)blockIdx.y, d, colphi ? 1 + z : 0, bi * TB);
    const bool lcol = rowphi || li < CW, lcol1 = rowphi1 || li < CW, lrow = colphi || li < CW;      // (an xc plane holds CW columns)
    double2 vv[4], vv1[4];
    double wf[2][4];
    {
        // column slice: point pair cp = (t >> 4) + 16 k of the block, column li: one 16-byte element of the mirror
        const unsigned o = ((unsigned)(t >> 4) * 16u + (unsigned)li) * 16u;
        const double2 zero2 = double2{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vv[k] = (rowt && lcol) ? ld16(mcol, o + (unsigned)(16 * k) * 256u) : zero2;
            vv1[k] = (pair && lcol1) ? ld16(mcol1, o + (unsigned)(16 * k) * 256u) : zero2;
        }
        // row slice: rows 32 wave + 16 cidx + 4 lj + q, q = 0 .. 3: point pairs (8 wave + 4 cidx + lj) 2 + {0, 1}
        const unsigned orow = ((unsigned)(16 * wave + 2 * lj) * 16u + (unsigned)li) * 16u;
#pragma unroll
        for (int cidx = 0; cidx < 2; ++cidx) {
            const double2 w01 = (colt && lrow) ? ld16(mrow, orow + (unsigned)(8 * cidx) * 256u) : zero2;
            const double2 w23 = (colt && lrow) ? ld16(mrow, orow + (unsigned)(8 * cidx + 1) * 256u) : zero2;
            wf[cidx][0] = w01.x; wf[cidx][1] = w01.y; wf[cidx][2] = w23.x; wf[cidx][3] = w23.y;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    SEP_STAMP(2);              // operand loads issued
    if (all_done) return;

    SEP_STAMP(3);              // ring issued
    if (rowt) {
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<double2*>(smem + MC_SM_V)[((t >> 4) + 16 * k) * MC + li] = vv[k];
    }
    SEP_STAMP(4);              // operands arrived, image written
    __syncthreads();                 // operand image complete
    SEP_STAMP(5);

    unsigned long long actb = 0ull;          // bit c: chain c of the group takes part in this slot (set before the first store)
    // matrix-core column c of a product -> (basis function k, chain of the group)
    auto col_chain = [&](bool phi, int c, int& k, int& cl) {
        if (phi && CW == 8) { k = 2 * z + (c >> 3); cl = c & 7; }
        else { k = phi ? z : 0; cl = c; }
        return cl < ngrp && (!phi || k < nbd) && (phi || c < CW) && ((actb >> cl) & 1ull) != 0;
    };
    const size_t cstride = (size_t)PST * pb.nb * pb.Np;
    mc_d4 accc[2], accr[2];
    accr[0] = mc_d4{0.0, 0.0, 0.0, 0.0}; accr[1] = mc_d4{0.0, 0.0, 0.0, 0.0};
    accc[0] = mc_d4{0.0, 0.0, 0.0, 0.0}; accc[1] = mc_d4{0.0, 0.0, 0.0, 0.0};
    auto store_rows = [&](int knd, bool phi) {          // block row bi, slot bj; rows 32 wave + 16 cidx + li
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int k, cl;
            const bool ok = col_chain(phi, lj + 4 * r, k, cl);
            const int slot = knd == TK_FH ? SL::slot_hx(d) : knd == TK_FK ? SL::slot_kf(d, k) : SL::slot_ex(d);
#pragma unroll
            for (int cidx = 0; cidx < 2; ++cidx)
                if (ok) ch.tpart[(size_t)(c0 + cl) * cstride + ((size_t)slot * pb.nb + bj) * pb.Np + bi * TB + 32 * wave + 16 * cidx + li] = accr[cidx][r];
        }
    };
    static_for<16>([&](auto S) {
        constexpr int s = decltype(S)::value;
        if (s < nsteps) {
        if (s == 6) act_flag = *act_ptr;
        constexpr int sl = s & 7, cidx = sl & 1;
        const int g = (wave + (sl >> 1)) & 3;
        if (s == 8) {
            // second block of a pair: the first one's rows go out, its operand image is replaced
            actb = __ballot(li < ngrp && act_flag != 0);
            store_rows(kind, rowphi);
            accr[0] = mc_d4{0.0, 0.0, 0.0, 0.0}; accr[1] = mc_d4{0.0, 0.0, 0.0, 0.0};
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) reinterpret_cast<double2*>(smem + MC_SM_V)[((t >> 4) + 16 * k) * MC + li] = vv1[k];
            __syncthreads();
        }
        if (s + SEP_RING - 1 < nsteps) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tl[(s + SEP_RING - 1) % SEP_RING][q] = ld(s + SEP_RING - 1, q);
        }
        double2 (&tt)[4] = tl[s % SEP_RING];
        if (cidx == 0 && sl != 0 && colt) {
            // take over the column sums of group g (register r of lane (li, lj) = column lj + 4 r, block columns 32 g + 2 li + {0, 1})
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 u = *reinterpret_cast<const double2*>(&colsum[(lj + 4 * r) * TB + 32 * g + 2 * li]);
                accc[0][r] = u.x; accc[1][r] = u.y;
            }
        }
        // The transposition for the row-type product goes FIRST: its LDS round trip (write, transposed read, operand read) then runs
        // under the column-type MFMAs, which take the tile piece straight from the registers.
        double2 rb[4], rv[4];
        if (rowt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(&stage[(4 * lj + q) * MC_PITCH + 2 * li]) = tt[q];      // (row of the piece = 4 lj + q)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rb[q] = *reinterpret_cast<const double2*>(&stage[li * MC_PITCH + 2 * (4 * q + lj)]);
                rv[q] = vimg[(16 * g + 4 * q + lj) * MC + li];
            }
        }
#ifdef MAGI_SEP_PIPE
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (colt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                accc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[cidx][q], tt[q].x, accc[0], 0, 0, 0);
                accc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[cidx][q], tt[q].y, accc[1], 0, 0, 0);
            }
        }
        if (rowt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                accr[cidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[q].x, rb[q].x, accr[cidx], 0, 0, 0);
                accr[cidx] = __builtin_amdgcn_mfma_f64_16x16x4f64(rv[q].y, rb[q].y, accr[cidx], 0, 0, 0);
            }
        }
        if (cidx == 1 && colt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<double2*>(&colsum[(lj + 4 * r) * TB + 32 * g + 2 * li]) = double2{accc[0][r], accc[1][r]};
            __syncthreads();
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s < 8) SEP_STAMP(6 + s);      // steps 0 .. 7 done: 6 .. 13
    });

    // ---- partials ----
    actb = __ballot(li < ngrp && act_flag != 0);
    if (rowt) store_rows(pair ? kind1 : kind, pair ? rowphi1 : rowphi);
    if (colt) {          // block row bj, slot bi: complete in LDS after the last phase's barrier; thread = (column t >> 4, eight block columns)
        int k, cl;
        const bool ok = col_chain(colphi, t >> 4, k, cl);
        const int col = 8 * (t & 15);
        const int slot = kind == TK_FH ? SL::slot_hx(d) : kind == TK_FK ? SL::slot_kf(d, k) : SL::slot_etf(d, k);
        if (ok) {
            double2* dst = reinterpret_cast<double2*>(&ch.tpart[(size_t)(c0 + cl) * cstride + ((size_t)slot * pb.nb + bi) * pb.Np + bj * TB + col]);
            const double2* src = reinterpret_cast<const double2*>(&colsum[(t >> 4) * TB + col]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dst[kk] = src[kk];
        }
    }
#ifdef MAGI_SEP_STAMPS
    SEP_STAMP(14);
    __builtin_amdgcn_s_waitcnt(0);
    SEP_STAMP(15);             // stores retired
    if (tix == (MAGI_SEP_STAMPS) && blockIdx.y == 0 && z == 0 && threadIdx.x == 0)
        for (int i_ = 0; i_ < 16; ++i_) reinterpret_cast<unsigned long long*>(ch.par + 40)[i_ < 12 ? i_ : i_] = sst[i_];
#endif
}

// operand mirror (both slot parities) of the states in V_Q: API path and sampler start (the stream then finds its operands as it does
// inside a trajectory)
template <int DRIFT>
__global__ __launch_bounds__(256) void k_mirror(DevProblem pb, DevChains ch) {
    using DR = DriftT<DRIFT>;
    if constexpr (DR::SEP) {
        constexpr int D = DR::D, NBM = DR::NBMAX;
        const int c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
        if (i >= pb.N) return;
        const double* q = ch.vec + vec_off(pb, c, V_Q);
        double x[D], ph[D][NBM];
#pragma unroll
        for (int dd = 0; dd < D; ++dd) x[dd] = q[dd * pb.N + i];
        DR::basis(x, ph);
        const int cw = xop_width(ch.n_chains), groups = (ch.n_chains + 15) >> 4, cl = c & 15;
        const int planes = 1 + (NBM * cw + 15) / 16;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                double* m0 = ch.vop + vop_off(D, planes, pb.Np, groups, b, c >> 4, dd, 0, 0);
                m0[vop_elem(i, cl)] = x[dd] - pb.mu[dd];
#pragma unroll
                for (int k = 0; k < NBM; ++k)
                    if (k < DR::nbasis(dd))
                        m0[(size_t)(cw == 8 ? 1 + (k >> 1) : 1 + k) * pb.Np * 16 + vop_elem(i, cw == 8 ? (k & 1) * 8 + cl : cl)] = ph[dd][k];
            }
    }
}

// ---- point kernel: the elementwise half of slot `parity` (leap_point.h), N / 16 workgroups per chain -------------------------
template <int DRIFT> struct PointRes {
    static constexpr int SEPN = DriftT<DRIFT>::SEP ? PT_POINTS * SepLayout<DRIFT>::PS_TOTAL : 0;
    static constexpr int N = SEPN > PT_POINTS * PT_DSLOT * 4 ? SEPN : PT_POINTS * PT_DSLOT * 4;
};
template <int DRIFT>
__global__ __launch_bounds__(PT_THREADS) void k_point(DevProblem pb, DevChains ch, int parity) {
    __shared__ double res[PointRes<DRIFT>::N];
    __shared__ double redk[64 * PART_K];
    __shared__ double s_mu[MAGI_MAX_D];
    __shared__ double s_x[PT_POINTS * PT_DSLOT];
    WG_TRACE(1, 0);
    // (flag and plan are fetched together and combined arithmetically: `a || b` would fetch b only after a has arrived --
    //  one more dependent round trip at the head of a 5 us kernel)
    const int all_done = ch.gctl->all_done;
    const LeafPlan lp = ch.plan[(size_t)parity * ch.n_chains + blockIdx.y];
    kernarg_prefetch<sizeof(DevProblem) + sizeof(DevChains) + sizeof(int)>();
    if (all_done != 0) return;
    if (lp.vop != 0) { boundary_block<DRIFT>(pb, ch, lp, blockIdx.y, blockIdx.x, redk, s_mu, s_x, parity ^ 1); return; }     // a subtree / transition end (decide.h)
    const int gate = (lp.active ^ 1) | lp.skip;
    if (gate != 0) return;
    if constexpr (DriftT<DRIFT>::SEP) {
        if (ch.sep) { point_block_sep<DRIFT>(pb, ch, lp, blockIdx.y, blockIdx.x, res, redk, s_mu, s_x, parity ^ 1); return; }
    }
    point_block<DRIFT>(pb, ch, lp, blockIdx.y, blockIdx.x, res, redk, s_mu, parity ^ 1);
}

// load-only twin of k_stream's tile stream (bench.py's ceiling leg): every workgroup reads its 128 KB block with the same
// 16-B-per-lane pattern and does nothing else -- what the memory system delivers for this layout and working set
__global__ __launch_bounds__(256) void k_read_tiles(const double2* __restrict__ p, double* out, int rev) {
    const double2* q = p + (size_t)blockIdx.x * (MAGI_TB * MAGI_TB / 2) + threadIdx.x;
    double a = 0.0;
    constexpr int R = MAGI_TB * MAGI_TB / 2 / 256;
#pragma unroll
    for (int r = 0; r < R; ++r) { const double2 v = q[(rev ? R - 1 - r : r) * 256]; a += v.x + v.y; }     // (odd launches walk backwards, as k_stream does)
    if (a == 12345.678) out[0] = a;
}

// validation plan (magi_logpost_grad_fused / timing): slot 0 evaluates buffer 0 as is, no leapfrog
__global__ void k_plan_eval(DevChains ch, int parity) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ch.n_chains) return;
    LeafPlan prev{};
    prev.active = 1; prev.skip = 1;           // what k_stream(parity) reads: "evaluate buffer 0 as is"
    ch.plan[(size_t)(parity ^ 1) * ch.n_chains + c] = prev;
    LeafPlan p{};
    p.active = 1;                             // what k_point(parity) executes: gradient only
    ch.plan[(size_t)parity * ch.n_chains + c] = p;
}

template <int DRIFT>
__global__ __launch_bounds__(MAGI_TAIL_THREADS) void k_leap_finalize(DevProblem pb, DevChains ch, double* out, int parity) {
    __shared__ double sh[25 * 16];
    __shared__ double shs[16];
    const int c = blockIdx.x;
    double* vb = ch.vec + vec_off(pb, c, 0);
    const LeafPlan lp = ch.plan[(size_t)parity * ch.n_chains + c];
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
    {
    if (h->prof_e0) hipExtLaunchKernelGGL((k_stream<NC, DRIFT>), grid, dim3(64 * ST_WAVES), 0, s, h->prof_e0, h->prof_e1, 0, pb, h->ch, h->cfg, parity);
    else hipLaunchKernelGGL((k_stream<NC, DRIFT>), grid, dim3(64 * ST_WAVES), 0, s, pb, h->ch, h->cfg, parity);
    }
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

template <int DRIFT>
int launch_stream_mc(magi_handle* h, int n_chains, int parity, bool with_decisions, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const int groups = (n_chains + MC - 1) / MC;
    if constexpr (DriftT<DRIFT>::SEP) {
        // separable drifts: ONE kernel family for every chain count; grid.z = basis planes
        if (n_chains <= 8) {
            const dim3 grid(pb.n_stasks + (with_decisions ? MC : 0), groups, SepLayout<DRIFT>::gz(8));
            if (h->prof_e0) hipExtLaunchKernelGGL((k_stream_sep<DRIFT, 8>), grid, dim3(256), 0, s, h->prof_e0, h->prof_e1, 0, pb, h->ch, h->cfg, parity);
            else hipLaunchKernelGGL((k_stream_sep<DRIFT, 8>), grid, dim3(256), 0, s, pb, h->ch, h->cfg, parity);
        } else {
            const dim3 grid(pb.n_stasks + (with_decisions ? MC : 0), groups, SepLayout<DRIFT>::gz(16));
            if (h->prof_e0) hipExtLaunchKernelGGL((k_stream_sep<DRIFT, 16>), grid, dim3(256), 0, s, h->prof_e0, h->prof_e1, 0, pb, h->ch, h->cfg, parity);
            else hipLaunchKernelGGL((k_stream_sep<DRIFT, 16>), grid, dim3(256), 0, s, pb, h->ch, h->cfg, parity);
        }
    } else {
    const dim3 grid(pb.n_tasks + (with_decisions ? MC : 0), groups);
    if (h->prof_e0) hipExtLaunchKernelGGL((k_stream_mc<DRIFT>), grid, dim3(256), 0, s, h->prof_e0, h->prof_e1, 0, pb, h->ch, h->cfg, parity);
    else hipLaunchKernelGGL((k_stream_mc<DRIFT>), grid, dim3(256), 0, s, pb, h->ch, h->cfg, parity);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("stream (matrix-core) launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

// Which kernel family streams the operator blocks: one or two chains -> the VALU kernel k_stream<1 | 2>, three or more -> a
// matrix-core kernel (k_stream_sep for separable drifts, k_stream_mc otherwise).  The two sum in different orders, so a chain's rounding depends on the size of the batch it
// runs in (1-2 against >= 3).  MAGI_STREAM_FAMILY=mc routes EVERY batch size through the matrix-core kernel: a chain's samples
// are then bit-identical whatever shares the GPU with it (uneven shards, e.g. 5 chains on 2 GPUs = 3 + 2), at the price of the
// slower kernel for one or two chains.  (Option stream_family of the handle -- the variable is read when the handle is created, magi_set_option
// afterwards -- evaluated at every magi_ensure_chains.)
// Small problems stay on the VALU kernel whatever the batch: with about one workgroup per CU (blocks x chain pairs <= 320) a launch is
// as long as its longest workgroup, and the matrix-core kernels' workgroups are long (prologue + 8 dependent steps + stores).  Slot time,
// VALU against matrix-core kernel: N = 161, b = 80, 8 chains 12.9 / 16.0 us; N = 256, 8 chains 13.0 / 16.0; N = 512, 3 and 4 chains
// (288 workgroups) 14.6 / 16.0 and 14.9 / 16.2; N = 512, 8 chains (576) 20.0 / 17.0; N = 1024, 3 chains 30.7 / 23.5.
// stream_family = valu forces the VALU kernel for every batch (A/B).
bool magi_stream_family_mc(const magi_handle* h, int n_chains) {
    if (h->opt.stream_family == 1) return true;
    if (h->opt.stream_family == 2) return false;
    const int n = h->opt.family_chains > 0 ? std::max(n_chains, h->opt.family_chains) : n_chains;
    return n >= 3 && (long)h->pb.n_tasks * ((n + 1) / 2) > 320;
}

bool magi_drift_separable(int drift) {
#define MAGI_CALL(DR) return DriftT<DR>::SEP
    MAGI_DRIFT_DISPATCH(drift, MAGI_CALL);
#undef MAGI_CALL
    return false;
}

template <int DRIFT> size_t sep_tpart_elems(const DevProblem& pb, int n) {
    if constexpr (DriftT<DRIFT>::SEP) return (size_t)n * SepLayout<DRIFT>::PS_TOTAL * pb.nb * pb.Np;
    else return 0;
}
template <int DRIFT> size_t sep_vop_elems(const DevProblem& pb, int n) {
    if constexpr (DriftT<DRIFT>::SEP) return (size_t)2 * ((n + 15) / 16) * DriftT<DRIFT>::D * SepLayout<DRIFT>::planes(16) * pb.Np * 16;     // (CW = 16 has the most planes)
    else return 0;
}
size_t magi_sep_tpart_elems(const DevProblem& pb, int n_chains) {
#define MAGI_CALL(DR) return sep_tpart_elems<DR>(pb, n_chains)
    MAGI_DRIFT_DISPATCH(pb.drift, MAGI_CALL);
#undef MAGI_CALL
    return 0;
}
size_t magi_sep_vop_elems(const DevProblem& pb, int n_chains) {
#define MAGI_CALL(DR) return sep_vop_elems<DR>(pb, n_chains)
    MAGI_DRIFT_DISPATCH(pb.drift, MAGI_CALL);
#undef MAGI_CALL
    return 0;
}

// Memory-side byte model of one launch of the separable streaming kernel and of its point kernel (magi_gradient_bytes): what the PMC
// passes of profiles/ count at the fabric -- the operator blocks once, the operand planes of one slot parity once per XCD (every XCD's L2
// fetches the 16-column slices its workgroups share), one 128-entry block vector per (task, product, matrix-core column in use) stored,
// and the point kernel's re-read of those product slots.
template <int DRIFT>
void sep_traffic(const DevProblem& pb, int n, double* stores, double* operands, double* point_reads, double* mirror_writes) {
    *stores = *operands = *point_reads = *mirror_writes = 0.0;
    if constexpr (DriftT<DRIFT>::SEP) {
        using DR = DriftT<DRIFT>;
        using SL = SepLayout<DRIFT>;
        double vecs = 0.0;       // block vectors written per chain
        for (int d = 0; d < DR::D; ++d)
            for (int kind = 0; kind < 3; ++kind)
                for (int bi = 0; bi < pb.nb; ++bi)
                    for (int bj = 0; bj < pb.nb; ++bj) {
                        if (kind != TK_FE && bj > bi) continue;
                        if (std::abs(bi - bj) > pb.wb) continue;
                        const int nbd = DR::nbasis(d);
                        vecs += kind == TK_FK ? nbd : 1;                                            // row-type product
                        if (kind == TK_FE || bi != bj) vecs += kind == TK_FH ? 1 : nbd;             // column-type product
                    }
        *stores = vecs * n * MAGI_TB * 8.0;
        const int cw = xop_width(n), groups = (n + 15) >> 4;
        *operands = 8.0 * groups * DR::D * SL::planes(cw) * (double)pb.Np * 16 * 8.0;
        int used = 0, basis = 0;
        for (int sl = 0; sl < SL::PS_TOTAL; ++sl) used += SL::slot_used(sl) ? 1 : 0;
        for (int d = 0; d < DR::D; ++d) basis += 1 + DR::nbasis(d);
        const double nslot = (double)std::min(pb.nb, 2 * pb.wb + 1);
        *point_reads = (double)n * used * nslot * pb.N * 8.0;
        *mirror_writes = (double)n * basis * pb.N * 8.0;
    }
}
void magi_sep_traffic(const DevProblem& pb, int n_chains, double* stores, double* operands, double* point_reads, double* mirror_writes) {
#define MAGI_CALL(DR) return sep_traffic<DR>(pb, n_chains, stores, operands, point_reads, mirror_writes)
    MAGI_DRIFT_DISPATCH(pb.drift, MAGI_CALL);
#undef MAGI_CALL
}

int magi_launch_mirror(magi_handle* h, int n_chains, hipStream_t s) {
    if (!h->ch.sep) return MAGI_OK;
    const dim3 g((h->pb.N + 255) / 256, n_chains), b(256);
#define MAGI_CALL(DR) hipLaunchKernelGGL(k_mirror<DR>, g, b, 0, s, h->pb, h->ch)
    MAGI_DRIFT_DISPATCH(h->pb.drift, MAGI_CALL);
#undef MAGI_CALL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("mirror launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_stream(magi_handle* h, int n_chains, int parity, bool with_decisions, hipStream_t s) {
    if (h->family_mc) {          // (separable drifts: k_stream_sep, else k_stream_mc)
#define MAGI_CALL(DR) return launch_stream_mc<DR>(h, n_chains, parity, with_decisions, s)
        MAGI_DRIFT_DISPATCH(h->pb.drift, MAGI_CALL);
#undef MAGI_CALL
    }
    if (n_chains >= 2) return launch_stream_nc<2>(h, n_chains, parity, with_decisions, s);       // (chain pairs on grid.y)
    return launch_stream_nc<1>(h, n_chains, parity, with_decisions, s);
}

int magi_launch_point(magi_handle* h, int n_chains, int parity, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const dim3 g(magi_leap_wgs(pb), n_chains), b(PT_THREADS);
#define MAGI_CALL(DR) do { if (h->prof_e0) hipExtLaunchKernelGGL(k_point<DR>, g, b, 0, s, h->prof_e0, h->prof_e1, 0, pb, h->ch, parity); \
                           else hipLaunchKernelGGL(k_point<DR>, g, b, 0, s, pb, h->ch, parity); } while (0)
    MAGI_DRIFT_DISPATCH(pb.drift, MAGI_CALL);
#undef MAGI_CALL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("point launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_read_tiles(magi_handle* h, int rev, hipStream_t s) {
    hipLaunchKernelGGL(k_read_tiles, dim3(h->pb.n_tasks), dim3(256), 0, s, reinterpret_cast<const double2*>(h->pb.tiles), h->d_fin, rev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("read_tiles launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_plan_eval(magi_handle* h, int n_chains, hipStream_t s, int parity) {
    hipLaunchKernelGGL(k_plan_eval, dim3((n_chains + 63) / 64), dim3(64), 0, s, h->ch, parity);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("plan launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_leap_finalize(magi_handle* h, int n_chains, double* d_out, hipStream_t s, int parity) {
    const dim3 g(n_chains), b(MAGI_TAIL_THREADS);
#define MAGI_CALL(DR) hipLaunchKernelGGL(k_leap_finalize<DR>, g, b, 0, s, h->pb, h->ch, d_out, parity)
    MAGI_DRIFT_DISPATCH(h->pb.drift, MAGI_CALL);
#undef MAGI_CALL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("leap_finalize launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

#ifdef MAGI_WG_TRACE
// dev: `warm` untimed [stream, point] pairs (validation plans, no decisions: as magi_time_gradient), then ONE traced pair; out = [2][4096][4]
extern "C" int magi_debug_wg_trace(magi_handle* h, int n_chains, int warm, int with_point, unsigned long long* out) {
    if (!h || !out) return MAGI_E_BADARG;
    (void)hipSetDevice(h->device);
    int rc = magi_ensure_chains(h, n_chains);
    if (rc) return rc;
    MAGI_HIP_CHECK(h, hipMemsetAsync(h->ch.gctl, 0, sizeof(GlobalCtl), h->stream));
    if ((rc = magi_launch_prepare(h, n_chains, h->stream))) return rc;
    if ((rc = magi_launch_plan_eval(h, n_chains, h->stream))) return rc;
    h->sampler_ready = false;
    void* sym = nullptr;
    MAGI_HIP_CHECK(h, hipGetSymbolAddress(&sym, HIP_SYMBOL(g_wg_trace)));
    for (int i = 0; i <= warm; ++i) {
        if (i == warm) MAGI_HIP_CHECK(h, hipMemsetAsync(sym, 0, sizeof(unsigned long long) * 2 * 4096 * 4, h->stream));
        if ((rc = magi_launch_stream(h, n_chains, i & 1, false, h->stream))) return rc;
        if (with_point && (rc = magi_launch_point(h, n_chains, 0, h->stream))) return rc;
    }
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));
    MAGI_HIP_CHECK(h, hipMemcpy(out, sym, sizeof(unsigned long long) * 2 * 4096 * 4, hipMemcpyDeviceToHost));
    return MAGI_OK;
}
#endif
