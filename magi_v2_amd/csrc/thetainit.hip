// Device-resident theta initialiser (reference magi_v2.py:133-179): Adam(lr = 0.01) x `iters` from theta = 1 on
//     theta_objective(theta) = sum_d toNorm_d^T K_d^-1 toNorm_d,   toNorm = reshape(f(Xhat, theta), [D, N, 1]) - m_d (Xhat - mu)_d
// INCLUDING the reference's tf.reshape of the [N, D] drift values to [D, N, 1] (magi_v2.py:155-156 re-interprets the row-major
// buffer instead of transposing it: entry (d', i') of the reshaped array is element d' N + i' of the flattened [N, D] array).
// The reference runs the 10 000 steps as one XLA loop; round 2 ran the general (not linear-in-theta) branch as a host loop with a
// blocking device mat-vec per step.  Here one Adam step is four launches on the handle's stream -- drift values + theta-Jacobian per
// flat entry, K^-1 r, K^-T r on the resident UNbanded stacks (the initialiser runs before the band approximation, :271-274), one
// reduction + Adam update block -- captured ONCE as a graph and replayed `iters` times; the host waits once, at the end.
// The gradient is the derivative tf.GradientTape forms at :164-166, written out:
//     d/dtheta_p = sum_{d'} reshape(df/dtheta_p)_{d'}^T (K_{d'}^-1 + K_{d'}^-T) toNorm_{d'}
// tf_keras Adam restated from its documented defaults (beta1 .9, beta2 .999, epsilon 1e-7): parity unpinned, as in oracle/.
#include "magi_internal.h"

namespace {

enum { TI_TH = 0, TI_M = 8, TI_V = 16, TI_B1T = 24, TI_B2T = 25, TI_STEP = 26, TI_LOSS = 27, TI_COUNT = 32 };

template <int DRIFT>
__global__ __launch_bounds__(256) void k_ti_eval(int N, const double* __restrict__ Xh /* [N][D] */, const double* __restrict__ st,
                                                 const double* __restrict__ bvec /* [D N] */, double* __restrict__ r, double* __restrict__ Tq /* [P][D N] */) {
    using DR = DriftT<DRIFT>;
    constexpr int D = DR::D, P = DR::P;
    const int n = N * D, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int i = e / D, d = e - i * D;              // flat entry e of the row-major [N][D] drift buffer = (grid point i, component d)
    double x[D], th[P], g[D], c[D], t[P];
#pragma unroll
    for (int k = 0; k < D; ++k) { x[k] = Xh[(size_t)i * D + k]; g[k] = (k == d) ? 1.0 : 0.0; }
#pragma unroll
    for (int p = 0; p < P; ++p) { th[p] = st[TI_TH + p]; t[p] = 0.0; }
    const double f = DR::f1(d, x, th);
    DR::jt(x, th, g, c, t);                          // g = e_d: t[p] = d f_d / d theta_p
    r[e] = f - bvec[e];                              // (bvec is [D][N]: the SAME flat index -- the reshape of :155-156)
#pragma unroll
    for (int p = 0; p < P; ++p) Tq[(size_t)p * n + e] = t[p];
}

// one workgroup: grad_p = sum_e Tq[p][e] (g1 + g2)[e], loss = sum_e r[e] g1[e]; Adam update of theta (threads 0 .. P - 1)
__global__ __launch_bounds__(256) void k_ti_step(int n, int P, const double* __restrict__ Tq, const double* __restrict__ g1, const double* __restrict__ g2,
                                                 const double* __restrict__ r, double* st, double lr, double* loss_trace) {
    __shared__ double sh[(MAGI_MAX_P + 2) * 16];
    double acc[MAGI_MAX_P + 1];
#pragma unroll
    for (int k = 0; k <= MAGI_MAX_P; ++k) acc[k] = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) {
        const double a = g1[e], gsum = a + g2[e];
#pragma unroll
        for (int p = 0; p < MAGI_MAX_P; ++p) if (p < P) acc[p] = fma(Tq[(size_t)p * n + e], gsum, acc[p]);
        acc[MAGI_MAX_P] = fma(r[e], a, acc[MAGI_MAX_P]);
    }
    block_sum<MAGI_MAX_P + 1>(acc, sh);
    const int p = threadIdx.x;
    const double b1 = 0.9, b2 = 0.999, eps = 1e-7;
    const double b1t = st[TI_B1T] * b1, b2t = st[TI_B2T] * b2;
    const int step = (int)st[TI_STEP];
    __syncthreads();                                  // every thread has read the running powers
    if (p < P) {
        double grad = 0.0;
#pragma unroll
        for (int k = 0; k < MAGI_MAX_P; ++k) if (k == p) grad = acc[k];
        const double m = b1 * st[TI_M + p] + (1.0 - b1) * grad;
        const double v = b2 * st[TI_V + p] + (1.0 - b2) * grad * grad;
        const double alpha = lr * sqrt(1.0 - b2t) / (1.0 - b1t);
        st[TI_M + p] = m; st[TI_V + p] = v;
        st[TI_TH + p] -= alpha * m / (sqrt(v) + eps);
    }
    if (p == 0) {
        st[TI_B1T] = b1t; st[TI_B2T] = b2t; st[TI_STEP] = (double)(step + 1); st[TI_LOSS] = acc[MAGI_MAX_P];
        if (loss_trace) loss_trace[step] = acc[MAGI_MAX_P];
    }
}

template <int DRIFT>
int launch_eval(magi_handle* h, int N, const double* Xh, const double* st, const double* bvec, double* r, double* Tq) {
    constexpr int D = DriftT<DRIFT>::D;
    hipLaunchKernelGGL(k_ti_eval<DRIFT>, dim3((N * D + 255) / 256), dim3(256), 0, h->stream, N, Xh, st, bvec, r, Tq);
    return MAGI_OK;
}

}  // namespace

int magi_theta_init_device(magi_handle* h, int drift, int P, const double* Xhat, const double* mu, int iters, double lr, double* theta, double* loss_trace) {
    const int N = h->dense_N, D = h->dense_D, n = N * D;
    int needD = 0, needP = 0;
#define MAGI_CALL(DR) do { needD = DriftT<DR>::D; needP = DriftT<DR>::P; } while (0)
    MAGI_DRIFT_DISPATCH(drift, MAGI_CALL);
#undef MAGI_CALL
    if (needD != D || needP != P) return magi_fail(h, MAGI_E_BADARG, "drift expects D=" + std::to_string(needD) + ", P=" + std::to_string(needP));
    // work space: Xh | xc | bvec | r | g1 | g2 (n each), Tq (P n), state (TI_COUNT), loss trace (iters)
    const size_t need = (size_t)6 * n + (size_t)P * n + TI_COUNT + (size_t)std::max(iters, 1);
    double* ws = magi_workspace(h, magi_handle::WS_TI, need);
    if (!ws) return MAGI_E_HIP;
    double *dXh = ws, *dxc = ws + n, *dbv = ws + 2 * (size_t)n, *dr = ws + 3 * (size_t)n, *dg1 = ws + 4 * (size_t)n, *dg2 = ws + 5 * (size_t)n,
           *dTq = ws + 6 * (size_t)n, *dst = dTq + (size_t)P * n, *dloss = dst + TI_COUNT;
    std::vector<double> xc((size_t)n), st(TI_COUNT, 0.0);
    for (int d = 0; d < D; ++d)
        for (int i = 0; i < N; ++i) xc[(size_t)d * N + i] = Xhat[(size_t)i * D + d] - mu[d];
    for (int p = 0; p < P; ++p) st[TI_TH + p] = theta[p];
    st[TI_B1T] = 1.0; st[TI_B2T] = 1.0;
    MAGI_HIP_CHECK(h, hipMemcpyAsync(dXh, Xhat, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    MAGI_HIP_CHECK(h, hipMemcpyAsync(dxc, xc.data(), sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    MAGI_HIP_CHECK(h, hipMemcpyAsync(dst, st.data(), sizeof(double) * TI_COUNT, hipMemcpyHostToDevice, h->stream));
    MAGI_HIP_CHECK(h, hipStreamSynchronize(h->stream));              // (xc, st are host temporaries)
    int rc = magi_dense_apply_device(h, 1, 0, 1, dxc, dbv);           // m_d (Xhat - mu)_d, once  (magi_v2.py:139-142)
    if (rc) return rc;
    if (iters > 0) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        MAGI_HIP_CHECK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
#define MAGI_CALL(DR) rc = launch_eval<DR>(h, N, dXh, dst, dbv, dr, dTq)
        MAGI_DRIFT_DISPATCH(drift, MAGI_CALL);
#undef MAGI_CALL
        if (rc == MAGI_OK) rc = magi_dense_apply_device(h, 2, 0, 1, dr, dg1);       // K^-1 toNorm
        if (rc == MAGI_OK) rc = magi_dense_apply_device(h, 2, 1, 1, dr, dg2);       // K^-T toNorm
        if (rc == MAGI_OK) hipLaunchKernelGGL(k_ti_step, dim3(1), dim3(256), 0, h->stream, n, P, dTq, dg1, dg2, dr, dst, lr, loss_trace ? dloss : nullptr);
        hipError_t e = hipStreamEndCapture(h->stream, &graph);
        if (rc == MAGI_OK && e != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, std::string("theta initialiser capture: ") + hipGetErrorString(e));
        if (rc == MAGI_OK && (e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess)
            rc = magi_fail(h, MAGI_E_HIP, std::string("theta initialiser graph: ") + hipGetErrorString(e));
        for (int it = 0; it < iters && rc == MAGI_OK; ++it)
            if ((e = hipGraphLaunch(exec, h->stream)) != hipSuccess) rc = magi_fail(h, MAGI_E_HIP, std::string("theta initialiser step: ") + hipGetErrorString(e));
        (void)hipStreamSynchronize(h->stream);                       // the ONE wait of the loop
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc) return rc;
    }
    MAGI_HIP_CHECK(h, hipMemcpy(st.data(), dst, sizeof(double) * TI_COUNT, hipMemcpyDeviceToHost));
    for (int p = 0; p < P; ++p) theta[p] = st[TI_TH + p];
    if (loss_trace && iters > 0) MAGI_HIP_CHECK(h, hipMemcpy(loss_trace, dloss, sizeof(double) * iters, hipMemcpyDeviceToHost));
    return MAGI_OK;
}
