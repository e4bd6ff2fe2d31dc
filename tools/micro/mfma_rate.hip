// Dev microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 per SIMD with 1..3 waves per SIMD and 1, 2 or 4 accumulator chains
// per wave (the streaming kernels accumulate 8 dependent MFMAs into one register block).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;
// The MFMA is issued through inline assembly with the accumulator tied to its own registers: with the builtin in a loop the compiler
// (ROCm 7.2) keeps the loop-carried accumulators in VGPRs, the MFMA's in AGPRs, and copies all of them both ways in every iteration
// (v_accvgpr_write / _read + s_nop) -- the first version of this file measured those copies (47-50 ns per MFMA, "44 TFLOP/s").
template <int CHAINS>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    d4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the results are read by VALU code the compiler schedules without knowing of the MFMAs)
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678) out[0] = s;
}
template <int CHAINS>
void run(int wgs_per_cu, int iters) {
    double* out; hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * CHAINS * wgs_per_cu;          // one wave of each workgroup per SIMD
    printf("chains %d  waves/SIMD %d: %.1f us  -> %.1f ns per MFMA per SIMD (64 cycles at 2.4 GHz = 26.7 ns), %.1f TFLOP/s\n", CHAINS, wgs_per_cu, ms * 1e3,
           ms * 1e6 / mfma_per_simd, 2048.0 * iters * CHAINS * 4 * grid / (ms * 1e-3) / 1e12);
    hipFree(out);
}
// v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 blocks per instruction (512 flop; accumulator = one double per lane).  Round 4: is its
// rate per flop that of the 16 x 16 x 4 form?  Then products with only 8 live columns (8 chains: xc operands) could fill their instructions.
template <int CHAINS>
__global__ __launch_bounds__(256) void k4(double* out, int iters) {
    double acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = 0.0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    if (s == 12345.678) out[0] = s;
}
template <int CHAINS>
void run4(int wgs_per_cu, int iters) {
    double* out; (void)hipMalloc(&out, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k4<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k4<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * CHAINS * wgs_per_cu;
    printf("4x4x4_4b chains %d  waves/SIMD %d: %.1f us  -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", CHAINS, wgs_per_cu, ms * 1e3,
           ms * 1e6 / mfma_per_simd, 512.0 * iters * CHAINS * 4 * grid / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
// VALU: v_fma_f64 on 8 independent accumulators per lane
__global__ __launch_bounds__(256) void kv(double* out, int iters) {
    double acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = threadIdx.x * 1e-9 * c;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-12;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0.0;
    for (int c = 0; c < 8; ++c) s += acc[c];
    if (s == 12345.678) out[0] = s;
}
void run_valu(int wgs_per_cu, int iters) {
    double* out; (void)hipMalloc(&out, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(kv, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kv, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("VALU fma_f64  waves/SIMD %d: %.1f us -> %.2f ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns), %.1f TFLOP/s\n", wgs_per_cu, ms * 1e3,
           ms * 1e6 / ((double)iters * 8 * wgs_per_cu), 128.0 * iters * 8 * 4 * grid / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    for (int w = 1; w <= 3; ++w) { run<1>(w, 4096); run<2>(w, 2048); run<4>(w, 1024); run<8>(w, 512); run<16>(w, 256); }
    for (int w = 1; w <= 3; ++w) run_valu(w, 65536);
    for (int w = 1; w <= 2; ++w) { run4<1>(w, 8192); run4<4>(w, 2048); run4<16>(w, 512); }
    return 0;
}
