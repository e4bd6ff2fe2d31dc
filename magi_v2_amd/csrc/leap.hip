// k_stream: the sampler's streaming kernel -- the four single-phase mat-vecs of a gradient
//     hx = FH xc,  ex = FE xc,  etf = FE^T f,  kf = FK f        (xc = X - mu, f = drift(X, theta))
// over the packed 128 x 128 operator blocks (pack.hip): the lower block triangle of the symmetric FH and FK
// and every block of FE, each block serving a row-type and a column-type product in one pass.  The point
// phase (leap_point.h) and the decisions (sampler.hip) follow in k_tail.
// (reference arithmetic: magi_v2.py:308-348)
#include "magi_internal.h"
#include "leap_reduce.h"
#include "leap_point.h"

namespace {

// extra block of the streaming kernel: the data-independent uniform draws of the leaf in flight, so the
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

__device__ __forceinline__ double sel4(const double (&a)[MAGI_MAX_D], int d) {
    return d == 0 ? a[0] : d == 1 ? a[1] : d == 2 ? a[2] : a[3];
}

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
__global__ __launch_bounds__(64 * ST_WAVES) __attribute__((amdgpu_waves_per_eu(ST_WAVES == 8 ? (NC <= 2 ? 4 : 2) : (NC == 1 ? 4 : NC == 2 ? 3 : 2)))) void k_stream(DevProblem pb, DevChains ch) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P, TB = MAGI_TB;
    if (ch.gctl->all_done) return;
    const int c0 = blockIdx.y * NC;
    if ((int)blockIdx.x == pb.n_tasks) { leap_service<NC>(ch, c0); return; }
#ifdef MAGI_TAIL_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) ch.par[(size_t)c0 * PAR_COUNT + 40 + 7] = (double)__builtin_amdgcn_s_memrealtime();
    if ((int)blockIdx.x == pb.n_tasks - 1 && threadIdx.x == 0) ch.par[(size_t)c0 * PAR_COUNT + 40 + 10] = (double)__builtin_amdgcn_s_memrealtime();
#endif
    __shared__ double vcol[NC][TB], vrow[NC][TB], rowout[NC][TB], colacc[ST_WAVES][NC][TB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int4 task = reinterpret_cast<const int4*>(pb.tasks)[blockIdx.x];
    const int d = task.x, kind = task.y, bi = task.z, bj = task.w;
    const int N = pb.N;

    // operands of this thread's vector entry (issued before the tile stream so that their wait does not
    // cover the 32 row loads behind them)
    const bool isrow = t >= TB;
    const int loc = (isrow ? t - TB : t) & (TB - 1);
    const int gi = (isrow ? bi : bj) * TB + loc;
    const bool wantf = isrow ? (kind != TK_FH) : (kind == TK_FK);
    const double mud = sel4(pb.mu, d);
    double xin[NC][D], thv[NC][P];
    bool act[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int cc = min(c0 + c, ch.n_chains - 1);
        const LeafPlan* lp = ch.plan + cc;
        act[c] = (c0 + c < ch.n_chains) && lp->active != 0;
        const double* q = ch.vec + vec_off(pb, cc, V_Q + lp->cur);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) xin[c][dd] = q[dd * N + min(gi, N - 1)];
#pragma unroll
        for (int k = 0; k < P; ++k) thv[c][k] = ch.par[(size_t)cc * PAR_COUNT + PAR_TH + k];
    }

    // the wave's rows in chunks of 8, two chunks in flight (a0 / a1): 16 KB per wave on the wire while one chunk is in the ALUs
    const double2* A = reinterpret_cast<const double2*>(pb.tiles + (size_t)blockIdx.x * TB * TB + (size_t)(wave * ST_RW) * TB) + lane;
    constexpr int NCK = ST_RW / 8;
    double2 a0[8], a1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) a0[r] = A[(size_t)r * (TB / 2)];
#pragma unroll
    for (int r = 0; r < 8; ++r) a1[r] = A[(size_t)(8 + r) * (TB / 2)];
    __builtin_amdgcn_sched_barrier(0);

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
}

// ---- point kernel (validation entry; the sampler runs the same point_block inside k_tail) ---------------------------
template <int DRIFT>
__global__ __launch_bounds__(PT_THREADS) void k_point(DevProblem pb, DevChains ch) {
    if (ch.gctl->all_done) return;
    __shared__ double res[PT_POINTS * 4 * 4];
    __shared__ double redk[64 * PART_K];
    if (!ch.plan[blockIdx.y].active) return;
    point_block<DRIFT>(pb, ch, blockIdx.y, blockIdx.x, res, redk);
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
int launch_stream_nd(magi_handle* h, int n_chains, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const dim3 grid(pb.n_tasks + 1, (n_chains + NC - 1) / NC);      // + the service block
    hipLaunchKernelGGL((k_stream<NC, DRIFT>), grid, dim3(64 * ST_WAVES), 0, s, pb, h->ch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("stream launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

template <int NC>
int launch_stream_nc(magi_handle* h, int n_chains, hipStream_t s) {
#define MAGI_CALL(DR) return launch_stream_nd<NC, DR>(h, n_chains, s)
    MAGI_DRIFT_DISPATCH(h->pb.drift, MAGI_CALL);
#undef MAGI_CALL
    return MAGI_OK;
}

}  // namespace

int magi_leap_wgs(const DevProblem& pb) { return (pb.N + PT_POINTS - 1) / PT_POINTS; }

int magi_launch_stream(magi_handle* h, int n_chains, hipStream_t s) {
    if (n_chains >= 3) return launch_stream_nc<4>(h, n_chains, s);
    if (n_chains == 2) return launch_stream_nc<2>(h, n_chains, s);
    return launch_stream_nc<1>(h, n_chains, s);
}

int magi_launch_point(magi_handle* h, int n_chains, hipStream_t s) {
    const DevProblem& pb = h->pb;
    const dim3 g(magi_leap_wgs(pb), n_chains), b(PT_THREADS);
#define MAGI_CALL(DR) hipLaunchKernelGGL(k_point<DR>, g, b, 0, s, pb, h->ch)
    MAGI_DRIFT_DISPATCH(pb.drift, MAGI_CALL);
#undef MAGI_CALL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("point launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_launch_leap(magi_handle* h, int n_chains, hipStream_t s) {
    int rc = magi_launch_stream(h, n_chains, s);
    if (rc == MAGI_OK) rc = magi_launch_point(h, n_chains, s);
    return rc;
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
