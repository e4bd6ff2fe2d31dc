// Dev microbenchmark: how the SHAPE and the in-flight DEPTH of a wave's loads change the streaming rate of 128 KB row-major
// 128 x 128 fp64 blocks (one block per 256-thread workgroup, wave w = rows 32 w .. 32 w + 31).
//   shape 0: one instruction = one row (1 KB contiguous)               -- k_stream / k_read_tiles
//   shape 1: one instruction = 4 rows x 32 columns (4 x 256 B)          -- k_stream_mc (fragment-shaped for the column-type MFMA)
//   depth : instructions in flight per wave before the first is consumed (32 = everything)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SHAPE, int DEPTH>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, double* out) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lj = lane >> 4;
    const double2* q = p + (size_t)blockIdx.x * 8192;
    auto addr = [&](int k) -> const double2* {
        if (SHAPE == 0) return q + (size_t)(32 * wave + k) * 64 + lane;
        const int s = k >> 2, qq = k & 3;                       // step s = (chunk s >> 2, group s & 3), instruction qq = 4-row group
        return q + (size_t)(32 * wave + 16 * (s >> 2) + 4 * qq + lj) * 64 + 16 * (s & 3) + li;
    };
    double2 r[DEPTH];
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) r[k] = *addr(k);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const double2 v = r[k % DEPTH];
        a += v.x + v.y;
        if (k + DEPTH < 32) r[k % DEPTH] = *addr(k + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (a == 12345.678) out[0] = a;
}
template <int SHAPE, int DEPTH>
void run(const double2* p, double* out, int nblk, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_read<SHAPE, DEPTH>), dim3(nblk), dim3(256), 0, 0, p, out);
    hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read<SHAPE, DEPTH>), dim3(nblk), dim3(256), 0, 0, p, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s depth %2d: %7.2f us  %5.2f TB/s\n", name, DEPTH, ms * 1e3 / reps, nblk * 131072.0 / (ms * 1e-3 / reps) / 1e12);
}
// alternating direction: does an XCD's L2 keep the tail of one launch for the head of the next?  (544 blocks = 9 MB per XCD, L2 = 4 MB)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read_dir(const double2* __restrict__ p, double* out, int rev) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double2* q = p + (size_t)blockIdx.x * 8192;
    auto addr = [&](int k) -> const double2* { const int kk = rev ? 31 - k : k; return q + (size_t)(32 * wave + kk) * 64 + lane; };
    double2 r[DEPTH];
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) r[k] = *addr(k);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const double2 v = r[k % DEPTH];
        a += v.x + v.y;
        if (k + DEPTH < 32) r[k % DEPTH] = *addr(k + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (a == 12345.678) out[0] = a;
}
template <int DEPTH>
void run_dir(const double2* p, double* out, int nblk, int alternate) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL((k_read_dir<DEPTH>), dim3(nblk), dim3(256), 0, 0, p, out, alternate ? (w & 1) : 0);
    (void)hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read_dir<DEPTH>), dim3(nblk), dim3(256), 0, 0, p, out, alternate ? (r & 1) : 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s depth %2d: %7.2f us  %5.2f TB/s\n", alternate ? "alternating direction" : "same direction", DEPTH, ms * 1e3 / reps, nblk * 131072.0 / (ms * 1e-3 / reps) / 1e12);
}
int main() {
    const int nblk = 544;
    double2* p; double* out;
    hipMalloc(&p, (size_t)nblk * 131072); hipMemset(p, 0, (size_t)nblk * 131072); hipMalloc(&out, 8);
    run<0, 32>(p, out, nblk, "row per instruction");
    run<0, 16>(p, out, nblk, "row per instruction");
    run<0, 8>(p, out, nblk, "row per instruction");
    run<0, 4>(p, out, nblk, "row per instruction");
    run<1, 32>(p, out, nblk, "4 rows x 256 B per instr");
    run<1, 20>(p, out, nblk, "4 rows x 256 B per instr");
    run<1, 16>(p, out, nblk, "4 rows x 256 B per instr");
    run<1, 12>(p, out, nblk, "4 rows x 256 B per instr");
    run<1, 8>(p, out, nblk, "4 rows x 256 B per instr");
    run<1, 4>(p, out, nblk, "4 rows x 256 B per instr");
    run_dir<32>(p, out, nblk, 0); run_dir<32>(p, out, nblk, 1);
    run_dir<8>(p, out, nblk, 0); run_dir<8>(p, out, nblk, 1);
    run_dir<4>(p, out, nblk, 0); run_dir<4>(p, out, nblk, 1);
    return 0;
}
