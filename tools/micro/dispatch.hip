// Dev microbenchmark: when do the workgroups of a 545-workgroup launch START, and how long does the launch take around them?
// Each workgroup reads its 128 KB block (16 rows in flight per wave, as k_stream), stamps its entry and its end with the 100 MHz
// counter; the host prints the spread of the entries, the workgroup lives and the event-timed kernel duration.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dispatch.hip -o tools/micro/dispatch && tools/micro/dispatch
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int LDS_KB, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void k_disp(const double2* __restrict__ p, double* out, unsigned long long* stamps, int rev) {
    __shared__ double pad[LDS_KB * 128 + 256];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const double2* q = p + (size_t)blockIdx.x * 8192;
    auto addr = [&](int k) -> const double2* { const int kk = rev ? 31 - k : k; return q + (size_t)(32 * wave + kk) * 64 + lane; };
    constexpr int DEPTH = 16;
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
    if (LDS_KB > 0) { pad[t] = a; __syncthreads(); a += pad[(t + 1) & 255]; }
    if (a == 12345.678) out[0] = a;
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (t == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}
template <int LDS_KB, int OCC>
void run(const double2* p, double* out, unsigned long long* st, int nblk, const char* name) {
    if (nblk > 1090) { printf("grid larger than the buffer\n"); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL((k_disp<LDS_KB, OCC>), dim3(nblk), dim3(256), 0, 0, p, out, st, w & 1);
    hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_disp<LDS_KB, OCC>), dim3(nblk), dim3(256), 0, 0, p, out, st, r & 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * nblk);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * nblk, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[2 * b]); t1 = std::max(t1, h[2 * b + 1]); }
    std::vector<double> ent(nblk), life(nblk);
    for (int b = 0; b < nblk; ++b) { ent[b] = (h[2 * b] - t0) * 0.01; life[b] = (h[2 * b + 1] - h[2 * b]) * 0.01; }
    std::sort(ent.begin(), ent.end()); std::sort(life.begin(), life.end());
    printf("%-34s kernel %6.2f us (events / launch) | first entry -> last end %6.2f us | entries: median %5.2f  90%% %5.2f  max %5.2f us | lives: median %5.2f  max %5.2f us\n",
           name, ms * 1e3 / reps, (t1 - t0) * 0.01, ent[nblk / 2], ent[nblk * 9 / 10], ent[nblk - 1], life[nblk / 2], life[nblk - 1]);
}
int main() {
    const int nblk = 545, max_blk = 1090;          // (every run below reads blocks [0, its grid size) of p: allocate for the largest)
    double2* p; double* out; unsigned long long* st;
    hipMalloc(&p, (size_t)max_blk * 131072); hipMemset(p, 0, (size_t)max_blk * 131072);
    hipMalloc(&out, 64); hipMalloc(&st, sizeof(unsigned long long) * 2 * 4096);
    run<0, 3>(p, out, st, nblk, "545 wgs, no LDS, 3 waves/SIMD");
    run<40, 3>(p, out, st, nblk, "545 wgs, 40 KB LDS, 3 waves/SIMD");
    run<0, 3>(p, out, st, 256, "256 wgs, no LDS");
    run<0, 3>(p, out, st, 512, "512 wgs, no LDS");
    run<0, 3>(p, out, st, 768, "768 wgs, no LDS");
    run<0, 3>(p, out, st, max_blk, "1090 wgs, no LDS");
    return 0;
}
