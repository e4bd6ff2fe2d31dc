// Dev micro-benchmark: is a CU-masked stream (hipExtStreamCreateWithCUMask) honoured on this box, which CUs does a mask bit name,
// and does a 158 KB-LDS workgroup on a second stream START while a masked stream keeps the other CUs full?  (look-ahead
// of the blocked Cholesky -- the diagonal-block kernel cannot co-reside with the rank-k update's workgroups, so it needs CUs of its own.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/cumask tools/micro/cumask.hip && tools/micro/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned hw_where() {       // (xcc << 16) | (se << 8) | cu  (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13])
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
    return (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
}

// 2 workgroups per CU by its LDS (33 KB static would allow 4; 250 VGPRs in the real kernel make it 2: emulated with 66 KB)
__global__ __launch_bounds__(256) void busy(long long cycles, unsigned* where, unsigned long long* t_first) {
    __shared__ double pad[66 * 128];
    pad[threadIdx.x] = 1.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        where[blockIdx.x] = hw_where();
        if (blockIdx.x == 0) { *t_first = __builtin_amdgcn_s_memrealtime(); __threadfence_system(); }
    }
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (pad[threadIdx.x] == 7.0) where[0] = 0;
}

__global__ __launch_bounds__(256) void big(unsigned* where, unsigned long long* t_start) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = 1.0;
    if (threadIdx.x == 0) { where[blockIdx.x] = hw_where(); t_start[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); }
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 100000) __builtin_amdgcn_s_sleep(8);      // ~ the diagonal kernel's 75 us at 100 MHz x ... (memtime ticks)
    if (lds[threadIdx.x] == 7.0) where[0] = 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s: %d CUs\n", p.name, ncu);
    unsigned *where, *wbig; unsigned long long *tf, *tb;
    CK(hipHostMalloc(&where, 1 << 20, hipHostMallocCoherent)); CK(hipHostMalloc(&wbig, 4096, hipHostMallocCoherent)); CK(hipHostMalloc(&tf, 64, hipHostMallocCoherent)); CK(hipHostMalloc(&tb, 4096, hipHostMallocCoherent));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(big), hipFuncAttributeMaxDynamicSharedMemorySize, 158480));
    const int words = (ncu + 31) / 32;
    auto run_busy = [&](hipStream_t s, int wgs, long long cyc, float* ms, std::set<unsigned>* used) -> int {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, s, cyc, where, tf);      // warm
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(busy, dim3(wgs), dim3(256), 0, s, cyc, where, tf);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(ms, e0, e1));
        if (used) { used->clear(); for (int i = 0; i < wgs; ++i) used->insert(where[i]); }
        return 0;
    };
    // ---- 1. all CUs against half of them ----
    hipStream_t s_all; CK(hipStreamCreateWithFlags(&s_all, hipStreamNonBlocking));
    std::vector<uint32_t> half(words, 0u);
    for (int c = 0; c < ncu / 2; ++c) half[c / 32] |= 1u << (c % 32);
    hipStream_t s_half;
    hipError_t em = hipExtStreamCreateWithCUMask(&s_half, (uint32_t)words, half.data());
    printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(em));
    if (em != hipSuccess) return 2;
    float ms_all = 0, ms_half = 0; std::set<unsigned> u_all, u_half;
    if (run_busy(s_all, 8 * ncu, 2000, &ms_all, &u_all)) return 1;       // memtime runs at 100 MHz: 2000 ticks = 20 us
    if (run_busy(s_half, 8 * ncu, 2000, &ms_half, &u_half)) return 1;
    printf("busy x %d workgroups of 20 us: all CUs %.3f ms on %zu distinct (xcc,se,cu); lower-half mask %.3f ms on %zu distinct\n", 8 * ncu, ms_all, u_all.size(), ms_half, u_half.size());
    { int per_xcc[16] = {0}; for (unsigned w : u_half) per_xcc[(w >> 16) & 15]++; printf("  lower-half mask: CUs used per XCC:"); for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]); printf("\n"); }
    // ---- 2. which CUs do bits 0..7 name?  mask = all minus bits {0..7} ----
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<uint32_t> m(words, 0u);
        for (int c = 0; c < ncu; ++c) m[c / 32] |= 1u << (c % 32);
        if (variant == 0) for (int c = 0; c < 8; ++c) m[c / 32] &= ~(1u << (c % 32));                      // bits 0..7
        else for (int x = 0; x < 8; ++x) { int c = x * (ncu / 8); m[c / 32] &= ~(1u << (c % 32)); }        // bits 0, 32, 64, ...
        hipStream_t s_m; CK(hipExtStreamCreateWithCUMask(&s_m, (uint32_t)words, m.data()));
        float ms = 0; std::set<unsigned> used;
        if (run_busy(s_m, 16 * ncu, 2000, &ms, &used)) return 1;
        int per_xcc[16] = {0}; for (unsigned w : used) per_xcc[(w >> 16) & 15]++;
        printf("mask without %s: %.3f ms, %zu distinct CUs; per XCC:", variant == 0 ? "bits 0..7" : "bits 0,32,64,..", ms, used.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("\n   missing:");
        for (unsigned w : u_all) if (!used.count(w)) printf(" (xcc %u se %u cu %u)", (w >> 16) & 15, (w >> 8) & 7, w & 31);
        printf("\n");
        // ---- 3. a 158 KB-LDS workgroup x 4 on an unmasked stream while the masked stream keeps its CUs full for ~3 ms ----
        hipStream_t s_hi; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 4; ++i) tb[i] = 0;
            *tf = 0;
            hipLaunchKernelGGL(busy, dim3(2 * ncu * 30), dim3(256), 0, s_m, 10000LL, where, tf);      // 30 waves of 100 us
            // wait until the busy kernel runs, then launch the big workgroups
            { long spins = 0; while (*(volatile unsigned long long*)tf == 0 && ++spins < 2000000000L) {} }
            hipLaunchKernelGGL(big, dim3(4), dim3(256), 158480, s_hi, wbig, tb);
            CK(hipStreamSynchronize(s_hi));
            unsigned long long t_big_done_host = 0; (void)t_big_done_host;
            CK(hipStreamSynchronize(s_m));
            printf("   rep %d: big workgroups started %.1f %.1f %.1f %.1f us after the busy kernel's first workgroup (busy kernel ~3000 us); on",
                   rep, (tb[0] - *tf) / 100.0, (tb[1] - *tf) / 100.0, (tb[2] - *tf) / 100.0, (tb[3] - *tf) / 100.0);
            for (int i = 0; i < 4; ++i) printf(" (xcc %u se %u cu %u)", (wbig[i] >> 16) & 15, (wbig[i] >> 8) & 7, wbig[i] & 31);
            printf("\n");
        }
        // control: the same with the busy kernel on ALL CUs
        if (variant == 1) {
            for (int i = 0; i < 4; ++i) tb[i] = 0;
            *tf = 0;
            hipLaunchKernelGGL(busy, dim3(2 * ncu * 30), dim3(256), 0, s_all, 10000LL, where, tf);
            { long spins = 0; while (*(volatile unsigned long long*)tf == 0 && ++spins < 2000000000L) {} }
            hipLaunchKernelGGL(big, dim3(4), dim3(256), 158480, s_hi, wbig, tb);
            CK(hipStreamSynchronize(s_hi)); CK(hipStreamSynchronize(s_all));
            printf("   control (busy on all CUs): big workgroups started %.1f %.1f %.1f %.1f us after the busy kernel's first workgroup\n",
                   (tb[0] - *tf) / 100.0, (tb[1] - *tf) / 100.0, (tb[2] - *tf) / 100.0, (tb[3] - *tf) / 100.0);
        }
    }
    return 0;
}
