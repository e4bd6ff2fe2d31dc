// Dev microbenchmark: ceiling of a plain streaming read (16 B per lane) for buffers that do / do not fit the
// Infinity Cache -- what k_stream's 5.3-5.9 TB/s should be compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n2, double* out) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
        a0 += v0.x + v0.y; a1 += v1.x + v1.y; a2 += v2.x + v2.y; a3 += v3.x + v3.y;
    }
    for (; i < n2; i += stride) { const double2 v = p[i]; a0 += v.x + v.y; }
    const double s = (a0 + a1) + (a2 + a3);
    if (s == 12345.678) out[0] = s;
}
// blocked variant: each workgroup reads one contiguous 128 KB chunk (k_stream's access shape)
__global__ __launch_bounds__(256) void k_read_blocks(const double2* __restrict__ p, double* out) {
    const double2* q = p + (size_t)blockIdx.x * 8192 + threadIdx.x;     // 8192 double2 = 128 KB
    double a = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) { const double2 v = q[r * 256]; a += v.x + v.y; }
    if (a == 12345.678) out[0] = a;
}
int main() {
    double* out; hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {36, 72, 134, 256, 1024, 4096}) {
        const size_t bytes = mb << 20, n2 = bytes / 16;
        double2* p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes);
        for (int grid : {1024, 2048, 4096}) {
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, p, n2, out);
            hipEventRecord(e0);
            const int reps = 20;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, p, n2, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MB grid %4d: %8.2f us  %6.2f TB/s\n", mb, grid, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
        }
        {
            const int nblk = (int)(bytes / 131072);
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read_blocks, dim3(nblk), dim3(256), 0, 0, p, out);
            hipEventRecord(e0);
            const int reps = 20;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read_blocks, dim3(nblk), dim3(256), 0, 0, p, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MB blocks %5d: %8.2f us  %6.2f TB/s\n", mb, nblk, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
        }
        hipFree(p);
    }
    return 0;
}
