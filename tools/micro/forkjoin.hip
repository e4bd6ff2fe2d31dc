// Dev microbenchmark: does a hipGraph with a fork/join per slot  [P -> (S || D) -> P ...]  beat the serial chain
// [S, P+D] on this runtime?  Kernels spin on the 100 MHz realtime counter to stand in for the real ones.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(int ticks) {                       // ticks of 10 ns
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) { __builtin_amdgcn_s_sleep(1); }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, ef[64], ej[64];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 64; ++i) { CK(hipEventCreateWithFlags(&ef[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming)); }
    const int S = 1300, D = 600, P = 250, slots = 32;      // 13 us, 6 us, 2.5 us of device work
    for (int mode = 0; mode < 3; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < slots; ++k) {
            if (mode == 0) {            // serial: S, then P and D in one kernel's time (P + D)
                hipLaunchKernelGGL(k_spin, dim3(544), dim3(256), 0, s1, S);
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, P + D);
            } else if (mode == 1) {     // serial three kernels
                hipLaunchKernelGGL(k_spin, dim3(544), dim3(256), 0, s1, S);
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, P);
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, s1, D);
            } else {                    // fork/join: P -> (S || D) -> join
                hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s1, P);
                CK(hipEventRecord(ef[k], s1));
                CK(hipStreamWaitEvent(s2, ef[k], 0));
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(256), 0, s2, D);
                CK(hipEventRecord(ej[k], s2));
                hipLaunchKernelGGL(k_spin, dim3(544), dim3(256), 0, s1, S);
                CK(hipStreamWaitEvent(s1, ej[k], 0));
            }
        }
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s1));
        CK(hipStreamSynchronize(s1));
        CK(hipEventRecord(e0, s1));
        const int reps = 20;
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s1));
        CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const char* names[3] = {"serial [S, P+D]", "serial [S, P, D]", "fork/join P->(S||D)"};
        printf("%-22s %.2f us per slot (device work: S %.1f  P %.1f  D %.1f)\n", names[mode], ms * 1e3 / (reps * slots), S / 100.0, P / 100.0, D / 100.0);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
