// What does the CODE SIZE of a kernel cost per launch on gfx950 (round 6)?  Every launch of the training step runs a
// different kernel, i.e. starts with a cold instruction cache (64 KB per two CUs).  Here NK distinct kernels (template
// copies: same code, different addresses) of STEPS straight-line dependent FMAs each (8 bytes per instruction, executed
// ONCE per wave - no loop) are replayed round-robin from one hipGraph, 256 workgroups x 256 threads per launch; the time per
// node against the kernel's code size gives the fetch cost of cold straight-line code.  "warm": the same single kernel
// replayed back to back (its code stays in the instruction cache).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/icache_probe.bin tools/probe/icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int STEPS, int ID>
__global__ __launch_bounds__(256) void chain(float* p, float a, float b) {
    float x = p[threadIdx.x & 63] + (float)ID;
#pragma unroll
    for (int i = 0; i < STEPS; ++i) x = __builtin_fmaf(x, a, b + (float)(i & 7));     // one v_fmac / v_fma per step, never a loop
    if (x == 12345.678f) p[0] = x;
}

template <int STEPS, int... IDS>
static void run(const char* label, float* p, bool warm) {
    hipStream_t s; CK(hipStreamCreate(&s));
    using Fn = void (*)(float*, float, float);
    Fn fns[] = {chain<STEPS, IDS>...};
    const int nk = sizeof(fns) / sizeof(fns[0]);
    const int nodes = 192;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(fns[warm ? 0 : i % nk], dim3(256), dim3(256), 0, s, p, 1.0001f, 0.5f);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-6s %5d straight-line FMAs (~%3d KB of code) x %2d distinct kernels: %7.2f us per launch\n", label, STEPS, STEPS * 8 / 1024 + 1,
           warm ? 1 : nk, best * 1e3 / nodes);
    CK(hipStreamDestroy(s));
}

#define IDS16 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
int main() {
    float* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
    run<16, IDS16>("cold", p, false);   run<16, IDS16>("warm", p, true);
    run<256, IDS16>("cold", p, false);  run<256, IDS16>("warm", p, true);
    run<1024, IDS16>("cold", p, false); run<1024, IDS16>("warm", p, true);
    run<2048, IDS16>("cold", p, false); run<2048, IDS16>("warm", p, true);
    run<4096, IDS16>("cold", p, false); run<4096, IDS16>("warm", p, true);
    run<8192, IDS16>("cold", p, false); run<8192, IDS16>("warm", p, true);
    return 0;
}
