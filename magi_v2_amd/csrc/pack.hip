// Packing of dense device matrices C^-1, m, K^-1 ([D][N][N] row-major) into the four streaming
// operands of the log-posterior kernels, with the reference's band approximation applied
// (tf.linalg.band_part(., b, b), magi_v2.py:271-274 / 459-462):
//     Csym = (C^-1 + C^-T)/2,  M = m,  Mt = m^T,  Ksym = (K^-1 + K^-T)/2
// Dense storage:  [D][N][ld], ld = N rounded up to even, pad column = 0 (keeps rows 16-B aligned).
// Banded storage: [D][N][ld], ld = 2b+1 rounded up to even, row i holds columns i-b .. i+b
//                 (zeros where the column falls outside [0, N)).  Used when 2b+1 < N.
#include "magi_internal.h"

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

}  // namespace

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
    hipLaunchKernelGGL(k_pack<PACK_SYM>, grid, block, 0, h->stream, dC_inv, h->dCsym, N, ld, mask, banded ? 1 : 0);
    hipLaunchKernelGGL(k_pack<PACK_COPY>, grid, block, 0, h->stream, dM, h->dM, N, ld, mask, banded ? 1 : 0);
    hipLaunchKernelGGL(k_pack<PACK_TRANS>, grid, block, 0, h->stream, dM, h->dMt, N, ld, mask, banded ? 1 : 0);
    hipLaunchKernelGGL(k_pack<PACK_SYM>, grid, block, 0, h->stream, dK_inv, h->dKsym, N, ld, mask, banded ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("pack launch: ") + hipGetErrorString(e));

    // ---- single-phase operators of the sampler ----------------------------------------------------
    //   t1 + t2 = xc^T FH xc - 2 f^T FE xc + f^T FK f,   FH = Csym + m^T Ksym m,  FE = Ksym m
    // formed from the MASKED matrices, so the reference's band semantics carry over exactly: the
    // products of band-b matrices have band <= 3b and are stored without truncation.
    {
        const bool fbanded = bandsize >= 0 && (6 * bandsize + 1) < N;
        const int fW = fbanded ? 6 * bandsize + 1 : N;
        const int ldf = (fW + 1) & ~1;
        const size_t felems = (size_t)D * N * ldf;
        if (felems != h->fused_elems || !h->dFH) {
            if (h->dFH) (void)hipFree(h->dFH);
            if (h->dFE) (void)hipFree(h->dFE);
            if (h->dFEt) (void)hipFree(h->dFEt);
            if (h->dFK) (void)hipFree(h->dFK);
            h->dFH = h->dFE = h->dFEt = h->dFK = nullptr;
            MAGI_HIP_CHECK(h, hipMalloc(&h->dFH, felems * sizeof(double)));
            MAGI_HIP_CHECK(h, hipMalloc(&h->dFE, felems * sizeof(double)));
            MAGI_HIP_CHECK(h, hipMalloc(&h->dFEt, felems * sizeof(double)));
            MAGI_HIP_CHECK(h, hipMalloc(&h->dFK, felems * sizeof(double)));
            h->fused_elems = felems;
        }
        const size_t nn = (size_t)D * N * N;
        double *tCs = nullptr, *tM = nullptr, *tKs = nullptr, *tE = nullptr;
        MAGI_HIP_CHECK(h, hipMalloc(&tCs, nn * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&tM, nn * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&tKs, nn * sizeof(double)));
        MAGI_HIP_CHECK(h, hipMalloc(&tE, nn * sizeof(double)));
        dim3 gd((N + 255) / 256, N, D);
        hipLaunchKernelGGL(k_pack<PACK_SYM>, gd, block, 0, h->stream, dC_inv, tCs, N, N, mask, 0);
        hipLaunchKernelGGL(k_pack<PACK_COPY>, gd, block, 0, h->stream, dM, tM, N, N, mask, 0);
        hipLaunchKernelGGL(k_pack<PACK_SYM>, gd, block, 0, h->stream, dK_inv, tKs, N, N, mask, 0);
        int rc = magi_fused_operators(h, N, D, tCs, tM, tKs, tE);
        if (rc == MAGI_OK) {
            const int fb = fbanded ? 3 * bandsize : -1;
            dim3 gf((ldf + 255) / 256, N, D);
            hipLaunchKernelGGL(k_pack<PACK_COPY>, gf, block, 0, h->stream, tCs, h->dFH, N, ldf, fb, fbanded ? 1 : 0);
            hipLaunchKernelGGL(k_pack<PACK_COPY>, gf, block, 0, h->stream, tE, h->dFE, N, ldf, fb, fbanded ? 1 : 0);
            hipLaunchKernelGGL(k_pack<PACK_TRANS>, gf, block, 0, h->stream, tE, h->dFEt, N, ldf, fb, fbanded ? 1 : 0);
            hipLaunchKernelGGL(k_pack<PACK_COPY>, gf, block, 0, h->stream, tKs, h->dFK, N, ldf, fb, fbanded ? 1 : 0);
            e = hipGetLastError();
        }
        hipError_t se = hipStreamSynchronize(h->stream);
        (void)hipFree(tCs); (void)hipFree(tM); (void)hipFree(tKs); (void)hipFree(tE);
        if (rc) return rc;
        if (e != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("fused pack launch: ") + hipGetErrorString(e));
        if (se != hipSuccess) return magi_fail(h, MAGI_E_HIP, std::string("fused pack: ") + hipGetErrorString(se));
        pb.FH = h->dFH; pb.FE = h->dFE; pb.FEt = h->dFEt; pb.FK = h->dFK;
        pb.ldf = ldf;
        pb.bandf = fbanded ? 3 * bandsize : -1;
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
