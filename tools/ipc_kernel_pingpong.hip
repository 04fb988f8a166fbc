// Do kernels of TWO PROCESSES on one device run concurrently, and do they see each other's stores through IPC-mapped memory while running?
// (Round 4: the rehearsal vehicle of a peer-mapped gradient exchange on a one-GPU box.)  fork() BEFORE any HIP call; the parent allocates
// fine-grained device memory, exports it (hipIpcGetMemHandle) over a pipe, the child maps it.  Kernel A: store payload, release, set flagA,
// then wait (bounded) for flagB and check B's payload.  Kernel B: wait (bounded) for flagA, check A's payload, store its own, set flagB.
//   hipcc -O3 --offload-arch=gfx950 -o ipc_pp tools/ipc_kernel_pingpong.hip && ./ipc_pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); _exit(3); } } while (0)
struct Res { unsigned long long waited; int ok; int timed_out; };
__global__ void side(unsigned *buf, int me, Res *res)      // buf[0] = flagA, buf[32] = flagB, payload A at 1024.., payload B at 2048..
{
    unsigned *myflag = buf + (me ? 32 : 0), *other = buf + (me ? 0 : 32);
    unsigned *mypay = buf + (me ? 2048 : 1024), *otherpay = buf + (me ? 1024 : 2048);
    const int t = threadIdx.x;
    unsigned long long spins = 0; int to = 0;
    if (me == 1) {                                          // B waits first
        if (t == 0) { while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0xA11CEu) { if (++spins > (1ull << 22)) { to = 1; break; } __builtin_amdgcn_s_sleep(16); } }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    mypay[t] = 1000u * (me + 1) + t;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (t == 0) __hip_atomic_store(myflag, me ? 0xB0Bu : 0xA11CEu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (me == 0) {
        if (t == 0) { while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0xB0Bu) { if (++spins > (1ull << 22)) { to = 1; break; } __builtin_amdgcn_s_sleep(16); } }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    const unsigned got = otherpay[t];
    const int good = got == 1000u * ((1 - me) + 1) + t;
    int all = __syncthreads_and(good);
    if (t == 0) { res->waited = spins; res->ok = all; res->timed_out = to; }
}
int main()
{
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 2;
    pid_t pid = fork();
    if (pid == 0) {                                          // child = side B
        hipIpcMemHandle_t h;
        if (read(p2c[0], &h, sizeof h) != (ssize_t)sizeof h) _exit(4);
        unsigned *buf = nullptr;
        CK(hipIpcOpenMemHandle((void **)&buf, h, hipIpcMemLazyEnablePeerAccess));
        Res *res; CK(hipHostMalloc((void **)&res, sizeof(Res))); memset(res, 0, sizeof(Res));
        char go = 1; if (write(c2p[1], &go, 1) != 1) _exit(4);
        hipLaunchKernelGGL(side, dim3(1), dim3(256), 0, 0, buf, 1, res);
        CK(hipDeviceSynchronize());
        printf("B: ok=%d timed_out=%d spins=%llu\n", res->ok, res->timed_out, res->waited);
        fflush(stdout);
        CK(hipIpcCloseMemHandle(buf));
        _exit(res->ok && !res->timed_out ? 0 : 5);
    }
    unsigned *buf = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&buf, 1 << 16, hipDeviceMallocFinegrained);
    printf("fine-grained alloc: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) CK(hipMalloc((void **)&buf, 1 << 16));
    CK(hipMemset(buf, 0, 1 << 16));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    CK(hipIpcGetMemHandle(&h, buf));
    if (write(p2c[1], &h, sizeof h) != (ssize_t)sizeof h) return 4;
    char go = 0; if (read(c2p[0], &go, 1) != 1) return 4;
    Res *res; CK(hipHostMalloc((void **)&res, sizeof(Res))); memset(res, 0, sizeof(Res));
    usleep(200000);                                          // let B's kernel start waiting first
    hipLaunchKernelGGL(side, dim3(1), dim3(256), 0, 0, buf, 0, res);
    CK(hipDeviceSynchronize());
    printf("A: ok=%d timed_out=%d spins=%llu\n", res->ok, res->timed_out, res->waited);
    int st = 0; waitpid(pid, &st, 0);
    printf("child exit %d\n", WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return 0;
}
