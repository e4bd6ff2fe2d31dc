// Packing of dense device matrices C^-1, m, K^-1 ([D][N][N] row-major) into the four streaming
// operands of the log-posterior kernels, with the reference's band approximation applied
// (tf.linalg.band_part(., b, b), magi_v2.py:271-274 / 459-462):
//     Csym = (C^-1 + C^-T)/2,  M = m,  Mt = m^T,  Ksym = (K^-1 + K^-T)/2
// Dense storage:  [D][N][ld], ld = N rounded up to even, pad column = 0 (keeps rows 16-B aligned).
// Banded storage: [D][N][ld], ld = 2b+1 rounded up to even, row i holds columns i-b .. i+b
//                 (zeros where the column falls outside [0, N)).  Used when 2b+1 < N.
#include "magi_internal.h"

#include <algorithm>

namespace {

enum PackMode { PACK_SYM = 0, PACK_COPY = 1, PACK_TRANS = 2 };

template <int MODE>
__global__ void k_pack(const double* __restrict__ src, double* __restrict__ dst, int N, int ld, int band, int banded) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;   // storage column
    const int i = blockIdx.y;
    const int d = blockIdx.z;
    if (k >= ld) return;
    const int j = banded ? (i - band + k) : k;
    double v = 0.0;
    const bool inside = (j >= 0) && (j < N) && (!banded || k < 2 * band + 1);
    if (inside && (band < 0 || abs(i - j) <= band)) {
        const double* A = src + (size_t)d * N * N;
        const double a = A[(size_t)i * N + j];
        if (MODE == PACK_COPY) v = a;
        else if (MODE == PACK_TRANS) v = A[(size_t)j * N + i];
        else v = 0.5 * (a + A[(size_t)j * N + i]);
    }
    dst[((size_t)d * N + i) * ld + k] = v;
}

// The same for DENSE storage (row i holds columns 0 .. ld): 32 x 32 tiles through LDS, so that the transposed operand A[j][i] of
// the symmetrised / transposed copies is read along its rows too (k_pack reads it down a column: 2.3 ms per N = 8192 x 4 stack,
// 0.9 TB/s -- four such passes were 9 ms of the 336 ms build).  grid (ceil(ld / 32), ceil(N / 32), D), block (32, 8).
template <int MODE>
__global__ __launch_bounds__(256) void k_pack_dense(const double* __restrict__ src, double* __restrict__ dst, int N, int ld, int band) {
    __shared__ double t1[32][33], t2[32][33];
    const double* A = src + (size_t)blockIdx.z * N * N;
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32, tx = threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ty = threadIdx.y + 8 * r;
        const int i = i0 + ty, j = j0 + tx;                 // this tile
        if (MODE != PACK_TRANS) t1[ty][tx] = (i < N && j < N) ? A[(size_t)i * N + j] : 0.0;
        const int it = j0 + ty, jt = i0 + tx;               // the partner tile: rows j0.., columns i0..
        if (MODE != PACK_COPY) t2[ty][tx] = (it < N && jt < N) ? A[(size_t)it * N + jt] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ty = threadIdx.y + 8 * r;
        const int i = i0 + ty, j = j0 + tx;
        if (i >= N || j >= ld) continue;
        double v = 0.0;
        if (j < N && (band < 0 || abs(i - j) <= band))
            v = MODE == PACK_COPY ? t1[ty][tx] : MODE == PACK_TRANS ? t2[tx][ty] : 0.5 * (t1[ty][tx] + t2[tx][ty]);
        dst[((size_t)blockIdx.z * N + i) * ld + j] = v;
    }
}

// one TB x TB block of FH / FK / FE per workgroup -> packed tile storage, zero beyond N and beyond the band
__global__ __launch_bounds__(256) void k_pack_tiles(const double* __restrict__ H, const double* __restrict__ K, const double* __restrict__ E,
                                                    const int* __restrict__ tasks, double* __restrict__ tiles, int N, int fb) {
    const int t = blockIdx.x;
    const int d = tasks[4 * t], kind = tasks[4 * t + 1], bi = tasks[4 * t + 2], bj = tasks[4 * t + 3];
    const double* A = (kind == TK_FH ? H : kind == TK_FK ? K : E) + (size_t)d * N * N;
    double* T = tiles + (size_t)t * MAGI_TB * MAGI_TB;
    for (int e = threadIdx.x; e < MAGI_TB * MAGI_TB; e += 256) {
        const int r = e / MAGI_TB, c = e - r * MAGI_TB;
        const int i = bi * MAGI_TB + r, j = bj * MAGI_TB + c;
        double v = 0.0;
        if (i < N && j < N && (fb < 0 || abs(i - j) <= fb)) v = A[(size_t)i * N + j];
        T[e] = v;
    }
}

// Y[d][i][v] = sum_j A_d[i][j] V[d][j][v]  (TRANS: A_d[j][i]), v < NV <= 8.  One wave per output row: the plain product sums
// along the wave (lane = column, butterfly), the transposed one keeps lane = row of the output and walks the rows of A
// (both read A along its unit-stride dimension).  grid (ceil(N / 4), D), 256 threads.
template <int TRANS>
__global__ __launch_bounds__(256) void k_dense_apply(const double* __restrict__ A, const double* __restrict__ V, double* __restrict__ Y, int N, int nv) {
    const int d = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* Ad = A + (size_t)d * N * N;
    const double* Vd = V + (size_t)d * N * nv;
    double* Yd = Y + (size_t)d * N * nv;
    if (!TRANS) {
        const int row = blockIdx.x * 4 + wave;
        if (row >= N) return;
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j = lane; j < N; j += 64) {
            const double a = Ad[(size_t)row * N + j];
#pragma unroll
            for (int v = 0; v < 8; ++v) if (v < nv) acc[v] = fma(a, Vd[(size_t)j * nv + v], acc[v]);
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            if (v >= nv) break;
            const double sres = wave_sum(acc[v]);
            if (lane == 0) Yd[(size_t)row * nv + v] = sres;
        }
    } else {
        // 256 consecutive output rows (= columns of A) per workgroup: thread = column, loop over the rows of A
        const int col = blockIdx.x * 256 + threadIdx.x;
        if (col >= N) return;
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < N; ++i) {
            const double a = Ad[(size_t)i * N + col];
#pragma unroll
            for (int v = 0; v < 8; ++v) if (v < nv) acc[v] = fma(a, Vd[(size_t)i * nv + v], acc[v]);
        }
#pragma unroll
        for (int v = 0; v < 8; ++v) if (v < nv) Yd[(size_t)col * nv + v] = acc[v];
    }
}

}  // namespace

int magi_dense_apply_device(magi_handle* h, int which, int trans, int nv, const double* dV, double* dY) {
    const int N = h->dense_N, D = h->dense_D;
    if (trans) hipLaunchKernelGGL(k_dense_apply<1>, dim3((N + 255) / 256, D), dim3(256), 0, h->stream, h->dDense[which], dV, dY, N, nv);
    else hipLaunchKernelGGL(k_dense_apply<0>, dim3((N + 3) / 4, D), dim3(256), 0, h->stream, h->dDense[which], dV, dY, N, nv);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("dense apply launch: ") + hipGetErrorString(e));
    return MAGI_OK;
}

int magi_pack_matrices(magi_handle* h, int N, int D, int bandsize, const double* dC_inv, const double* dM,
                       const double* dK_inv) {
    DevProblem& pb = h->pb;
    const bool banded = bandsize >= 0 && (2 * bandsize + 1) < N;
    const int W = banded ? 2 * bandsize + 1 : N;
    const int ld = (W + 1) & ~1;
    const size_t elems = (size_t)D * N * ld;
    const bool shape_changed = !h->have_matrices || pb.N != N || pb.D != D;
    if (!h->have_matrices || elems != h->mat_elems) {
        if (h->dCsym) (void)hipFree(h->dCsym);
        if (h->dM) (void)hipFree(h->dM);
        if (h->dMt) (void)hipFree(h->dMt);
        if (h->dKsym) (void)hipFree(h->dKsym);
        h->dCsym = h->dM = h->dMt = h->dKsym = nullptr;
        MAGI_HIP_CHECK(h, hipMalloc(&h->dCsym, elems * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&h->dM, elems * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&h->dMt, elems * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&h->dKsym, elems * sizeof(double)));
        h->mat_elems = elems;
    }
    const int mask = bandsize >= 0 ? bandsize : -1;
    dim3 block(256), grid((ld + 255) / 256, N, D);
    const dim3 dblock(32, 8);
    auto dgrid = [&](int width) { return dim3((width + 31) / 32, (N + 31) / 32, D); };
    if (banded) {
        hipLaunchKernelGGL(k_pack<PACK_SYM>, grid, block, 0, h->stream, dC_inv, h->dCsym, N, ld, mask, 1);
        hipLaunchKernelGGL(k_pack<PACK_COPY>, grid, block, 0, h->stream, dM, h->dM, N, ld, mask, 1);
        hipLaunchKernelGGL(k_pack<PACK_TRANS>, grid, block, 0, h->stream, dM, h->dMt, N, ld, mask, 1);
        hipLaunchKernelGGL(k_pack<PACK_SYM>, grid, block, 0, h->stream, dK_inv, h->dKsym, N, ld, mask, 1);
    } else {
        hipLaunchKernelGGL(k_pack_dense<PACK_SYM>, dgrid(ld), dblock, 0, h->stream, dC_inv, h->dCsym, N, ld, mask);
        hipLaunchKernelGGL(k_pack_dense<PACK_COPY>, dgrid(ld), dblock, 0, h->stream, dM, h->dM, N, ld, mask);
        hipLaunchKernelGGL(k_pack_dense<PACK_TRANS>, dgrid(ld), dblock, 0, h->stream, dM, h->dMt, N, ld, mask);
        hipLaunchKernelGGL(k_pack_dense<PACK_SYM>, dgrid(ld), dblock, 0, h->stream, dK_inv, h->dKsym, N, ld, mask);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("pack launch: ") + hipGetErrorString(e));

    // ---- single-phase operators of the sampler ----------------------------------------------------
    //   t1 + t2 = xc^T FH xc - 2 f^T FE xc + f^T FK f,   FH = Csym + m^T Ksym m,  FE = Ksym m,  FK = Ksym
    // formed from the MASKED matrices, so the reference's band semantics carry over exactly: the
    // products of band-b matrices have band <= 3b, and only blocks that intersect that band are kept.
    {
        const int fb = (bandsize >= 0 && (6 * bandsize + 1) < N) ? 3 * bandsize : -1;
        const int nb = (N + MAGI_TB - 1) / MAGI_TB;
        const int wb = fb < 0 ? nb : std::min(nb, (std::max(fb, 1) - 1) / MAGI_TB + 1);
        std::vector<int> tasks;
        for (int d = 0; d < D; ++d)
            for (int kind = 0; kind < 3; ++kind)
                for (int bi = 0; bi < nb; ++bi)
                    for (int bj = 0; bj < nb; ++bj) {
                        if (kind != TK_FE && bj > bi) continue;            // symmetric: lower block triangle
                        if (std::abs(bi - bj) > wb) continue;
                        tasks.push_back(d); tasks.push_back(kind); tasks.push_back(bi); tasks.push_back(bj);
                    }
        // Launch order = task order, and the order decides which tasks share a CU: the hardware deals the workgroups of a launch that is resident all
        // at once round-robin -- launch indices w, w + C, w + 2 C land on the same CU (C CUs; tools/exp_wg_trace.py) -- so with 1 + 544 workgroups on
        // 256 CUs (N = 1024, dense) the first 33 CUs host THREE of them, stream 384 KB instead of 256 KB and end last (profiles/r04_wg_trace_1chain.txt:
        // 1.0 us after the others).  The blocks of FH are the light tasks (no theta' in front; the diagonal ones a row-type product only: 8.1 against 8.8 us
        // of life): they take the launch positions of those CUs in every round, the rest keeps its order.  Same-box A/Bs (profiles/r04_task_order_ab.txt):
        // diagonal FH blocks last 103.6-104.5 -> 106.5-106.9 samples/s; light tasks on all three positions of the triple CUs 108.9-109.3, slot 17.6 -> 16.8 us
        // (the exact variants of the pattern agree within the noise; this one measured best).  The order has no bearing on the arithmetic: a block's
        // partial sums are addressed by its indices.  Only launches of exactly three rounds are re-ordered: fewer have no third workgroups, more are
        // placed dynamically.
        {
            int ncu = 0;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
            const int T = (int)(tasks.size() / 4) + 1;                       // + the decision workgroup (launch index 0)
            const int rounds = ncu > 0 ? (T + ncu - 1) / ncu : 0, R = ncu > 0 ? T % ncu : 0;
            if (rounds == 3 && R > 1) {
                std::vector<int> heavy, off, diag;
                for (size_t i = 0; i < tasks.size(); i += 4) {
                    std::vector<int>& dst = tasks[i + 1] != TK_FH ? heavy : (tasks[i + 2] == tasks[i + 3] ? diag : off);
                    dst.insert(dst.end(), tasks.begin() + i, tasks.begin() + i + 4);
                }
                std::vector<int> out;
                auto take = [&](std::vector<int>& src, size_t& pos, size_t n) { const size_t m = std::min(n * 4, src.size() - pos); out.insert(out.end(), src.begin() + pos, src.begin() + pos + m); pos += m; };
                size_t ph = 0, po = 0, pd = 0;
                const size_t r = (size_t)R - 1, c = (size_t)ncu;
                take(off, po, r); take(heavy, ph, c - r);                    // round 0: the triple CUs' first workgroups are light
                take(diag, pd, r); take(off, po, r);                         // round 1: their second ones the lightest; (then one more stretch of light ones: measured)
                take(heavy, ph, heavy.size());                               // everything heavy
                take(off, po, off.size()); take(diag, pd, diag.size());      // round 2 = the tail of the launch: light
                tasks.swap(out);
            }
        }
        const int n_tasks = (int)(tasks.size() / 4);
        // tasks of the separable streaming kernel: as above, with FH_bb + FK_bb of a component paired (equal work per workgroup) -- when there
        // are more tasks than CUs.  On a small grid every task has a CU of its own and the kernel lasts as long as its longest workgroup: a
        // pair's 16 dependent steps would be that workgroup (N = 161, b = 80, 8 chains: option sep_pair_min = 0 forces the pairs for A/B).
        const bool pair_diag = n_tasks > h->opt.sep_pair_min;
        std::vector<int> stasks, spairs;
        for (int i = 0; i < n_tasks; ++i) {
            const int d = tasks[4 * i], kind = tasks[4 * i + 1], bi = tasks[4 * i + 2], bj = tasks[4 * i + 3];
            int partner = -1;
            if (pair_diag && kind != TK_FE && bi == bj) {
                if (kind == TK_FK) continue;          // (taken by its FH partner)
                for (int j = 0; j < n_tasks; ++j)
                    if (tasks[4 * j] == d && tasks[4 * j + 1] == TK_FK && tasks[4 * j + 2] == bi && tasks[4 * j + 3] == bi) { partner = j; break; }
            }
            const int e[8] = {d, kind, bi, bj, i, partner, TK_FK, 0};
            // (the pairs stream two blocks: they are dispatched FIRST -- with the diagonal FH blocks at the end of `tasks` they would start last
            //  and end the launch 0.5 us later, measured)
            if (partner >= 0) spairs.insert(spairs.end(), e, e + 8);
            else stasks.insert(stasks.end(), e, e + 8);
        }
        stasks.insert(stasks.begin(), spairs.begin(), spairs.end());
        const int n_stasks = (int)(stasks.size() / 8);
        const size_t n_task_ints = tasks.size();
        tasks.insert(tasks.end(), stasks.begin(), stasks.end());
        const size_t telems = (size_t)n_tasks * MAGI_TB * MAGI_TB;
        if (telems > h->tiles_cap) {
            if (h->dTiles) (void)hipFree(h->dTiles);
            h->dTiles = nullptr; h->tiles_cap = 0;
            MAGI_HIP_CHECK(h, hipMalloc(&h->dTiles, telems * sizeof(double)));
            h->tiles_cap = telems;
        }
        if (tasks.size() > h->tasks_cap) {
            if (h->dTasks) (void)hipFree(h->dTasks);
            h->dTasks = nullptr; h->tasks_cap = 0;
            MAGI_HIP_CHECK(h, hipMalloc(&h->dTasks, tasks.size() * sizeof(int)));
            h->tasks_cap = tasks.size();
        }
        MAGI_HIP_CHECK(h, hipMemcpyAsync(h->dTasks, tasks.data(), tasks.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        const size_t nn = (size_t)D * N * N;
        double *tCs = magi_workspace(h, magi_handle::WS_TCS, nn), *tM = magi_workspace(h, magi_handle::WS_TM, nn),
               *tKs = magi_workspace(h, magi_handle::WS_TKS, nn), *tE = magi_workspace(h, magi_handle::WS_TE, nn);
        if (!tCs || !tM || !tKs || !tE) return MAGI_E_HIP;
        dim3 gd((N + 255) / 256, N, D);
        (void)gd;
        hipLaunchKernelGGL(k_pack_dense<PACK_SYM>, dgrid(N), dblock, 0, h->stream, dC_inv, tCs, N, N, mask);
        hipLaunchKernelGGL(k_pack_dense<PACK_COPY>, dgrid(N), dblock, 0, h->stream, dM, tM, N, N, mask);
        hipLaunchKernelGGL(k_pack_dense<PACK_SYM>, dgrid(N), dblock, 0, h->stream, dK_inv, tKs, N, N, mask);
        int rc = magi_fused_operators(h, N, D, tCs, tM, tKs, tE);
        if (rc == MAGI_OK) {
            hipLaunchKernelGGL(k_pack_tiles, dim3(n_tasks), dim3(256), 0, h->stream, tCs, tKs, tE, h->dTasks, h->dTiles, N, fb);
            e = hipGetLastError();
        }
        hipError_t se = hipStreamSynchronize(h->stream);     // (also keeps `tasks` alive until the copy is done)
        if (rc) return rc;
        if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("fused pack launch: ") + hipGetErrorString(e));
        if (se != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("fused pack: ") + hipGetErrorString(se));
        pb.tiles = h->dTiles;
        pb.tasks = h->dTasks;
        pb.n_tasks = n_tasks;
        pb.stasks = h->dTasks + n_task_ints;
        pb.n_stasks = n_stasks;
        pb.nb = nb;
        pb.Np = nb * MAGI_TB;
        pb.wb = wb;
        pb.bandf = fb;
    }

    pb.N = N;
    pb.D = D;
    pb.ND = N * D;
    pb.ld = ld;
    pb.band = banded ? bandsize : -1;
    pb.Csym = h->dCsym;
    pb.M = h->dM;
    pb.Mt = h->dMt;
    pb.Ksym = h->dKsym;
    h->have_matrices = true;
    if (shape_changed) h->have_problem = false;   // yobs / dim depend on N, D
    // kernel arguments are captured by value in the leapfrog graph
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    h->graph_valid = false;
    h->sampler_ready = false;
    return MAGI_OK;
}
