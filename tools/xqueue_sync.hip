// What does a dependency between two HIP streams cost on this stack?  (Round 4: the pipelined training loop pays two per vector step.)
//   hipcc -O3 --offload-arch=gfx950 -o xqueue_sync tools/xqueue_sync.hip && ./xqueue_sync
// Ping-pong of a ~3 us kernel between streams A and B, N iterations of [A: kernel, signal; B: wait, kernel, signal; A: wait], against the
// same 2 N kernels on one stream.  Signal / wait variants: events with different creation flags, hipStreamWriteValue32 / hipStreamWaitValue32.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(float *p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    float *buf; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const int N = 2000, W = 150;
    auto run1 = [&]() { for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, buf, W); CK(hipStreamSynchronize(A)); };
    run1();
    double t0 = now(); run1(); double one = (now() - t0) / (2 * N) * 1e6;
    printf("one stream: %.2f us per kernel\n", one);
    struct V { const char *name; unsigned flags; } vs[] = {{"events: DisableTiming", hipEventDisableTiming},
        {"events: DisableTiming | DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
        {"events: DisableTiming | ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
        {"events: Default (timing on)", hipEventDefault}};
    for (auto &v : vs) {
        hipEvent_t ea[2], eb[2];
        for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&ea[i], v.flags)); CK(hipEventCreateWithFlags(&eb[i], v.flags)); }
        auto run = [&]() {
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, buf, W);
                CK(hipEventRecord(ea[i & 1], A)); CK(hipStreamWaitEvent(B, ea[i & 1], 0));
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, B, buf + 512, W);
                CK(hipEventRecord(eb[i & 1], B)); CK(hipStreamWaitEvent(A, eb[i & 1], 0));
            }
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        };
        run();
        t0 = now(); run(); double us = (now() - t0) / (2 * N) * 1e6;
        printf("%-50s %.2f us per kernel + dependency  (dependency = %.2f us)\n", v.name, us, us - one);
        // the same records and waits but already satisfied: B's chain alone [kernel, record, wait(old event of A)]
        CK(hipEventRecord(ea[0], A)); CK(hipStreamSynchronize(A));
        auto runs = [&]() { for (int i = 0; i < 2 * N; ++i) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, B, buf, W); CK(hipEventRecord(eb[i & 1], B)); CK(hipStreamWaitEvent(B, ea[0], 0)); } CK(hipStreamSynchronize(B)); };
        runs(); t0 = now(); runs(); us = (now() - t0) / (2 * N) * 1e6;
        printf("%-50s %.2f us per [kernel, record, satisfied wait]  (+%.2f us)\n", "", us, us - one);
    }
    int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (can) {
        uint64_t *f0, *f1;                                    // signal memory: 8 bytes per allocation
        CK(hipExtMallocWithFlags((void **)&f0, 8, hipMallocSignalMemory)); CK(hipExtMallocWithFlags((void **)&f1, 8, hipMallocSignalMemory));
        CK(hipMemset(f0, 0, 8)); CK(hipMemset(f1, 0, 8));
        uint32_t *flag = (uint32_t *)f0, *flagb = (uint32_t *)f1;
        auto run = [&](uint32_t base) {
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, buf, W);
                CK(hipStreamWriteValue32(A, flag, base + 2 * i + 1, 0)); CK(hipStreamWaitValue32(B, flag, base + 2 * i + 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, B, buf + 512, W);
                CK(hipStreamWriteValue32(B, flagb, base + 2 * i + 2, 0)); CK(hipStreamWaitValue32(A, flagb, base + 2 * i + 2, hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
            CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        };
        run(0);
        t0 = now(); run(2 * N + 2); double us = (now() - t0) / (2 * N) * 1e6;
        printf("%-50s %.2f us per kernel + dependency  (dependency = %.2f us)\n", "hipStreamWriteValue32 / hipStreamWaitValue32", us, us - one);
    }
    return 0;
}
