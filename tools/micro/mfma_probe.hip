// Dev: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, probed with one-hot operands (the ISA guide at hand documents only
// the 16x16x4 form).  Workgroup (la, lb): A = 1 in lane la, B = 1 in lane lb, C = 0; prints for which (la, lb) which result lane is non-zero
// and checks the hypothesis  A[b][i][k] <- lane i + 4 b + 16 k,  B[b][k][j] <- lane j + 4 b + 16 k,  D[b][i][j] -> lane j + 4 b + 16 i
// (= the 16x16x4 layouts with the 16-wide index read as (block, 4)).      hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(double* out) {
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0, c = 0.0;
    asm volatile("s_nop 15\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(c) : "v"(a), "v"(b));
    out[((size_t)la * 64 + lb) * 64 + l] = c;
}
int main() {
    double* d; hipMalloc(&d, sizeof(double) * 64 * 64 * 64);
    hipLaunchKernelGGL(k, dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    int bad = 0, nz = 0, ones = 0, junk = 0;
    for (size_t e = 0; e < h.size(); ++e) { if (h[e] == 1.0) ++ones; else if (h[e] != 0.0) ++junk; }
    printf("entries equal to 1: %d, other non-zero entries: %d\n", ones, junk);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l)
        if (h[((size_t)la * 64 + lb) * 64 + l] == 1.0 && (la < 6 || la == 16 || la == 17 || la == 20) && lb < 24) printf("ONE: A lane %d x B lane %d -> D lane %d\n", la, lb, l);
    for (size_t e = 0; e < h.size(); ++e) if (h[e] != 1.0 && fabs(h[e]) < 1e-300) h[e] = 0.0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const int ia = la & 3, ba = (la >> 2) & 3, ka = la >> 4, jb = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
            const bool expect = ba == bb && ka == kb;
            const int lane_expect = jb + 4 * ba + 16 * ia;
            for (int l = 0; l < 64; ++l) {
                const double v = h[((size_t)la * 64 + lb) * 64 + l];
                const double want = (expect && l == lane_expect) ? 1.0 : 0.0;
                if (v != 0.0) ++nz;
                if (v != want) { if (bad < 20) printf("la %d lb %d lane %d: got %g want %g\n", la, lb, l, v, want); ++bad; }
            }
        }
    printf("non-zero results %d (hypothesis: 256); mismatches against the hypothesis: %d\n", nz, bad);

    return bad != 0;
}
