// Cost of an in-kernel grid barrier on gfx950 (round 6: is a persistent multi-phase kernel cheaper than a launch per
// phase?).  A persistent grid of G workgroups x 256 threads runs ITERS phases; in each phase every workgroup writes a
// payload, crosses the barrier and reads the payload of a workgroup that lives on ANOTHER XCD (values checked).
//   mode 0  barrier only (no payload)
//   mode 1  payload through ordinary stores / loads + __threadfence() on both sides of the barrier (L2 write-back + invalidate)
//   mode 2  payload through agent-scope (sc1) stores / loads, no fence
// Also: the same number of EMPTY kernel launches of the same grid replayed from one hipGraph, for the comparison.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/grid_barrier_probe.bin tools/probe/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Bar { unsigned count; unsigned pad0[31]; unsigned gen; unsigned pad1[31]; };

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned G, unsigned& my_gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned old = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == G - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) __builtin_amdgcn_s_sleep(1);
        }
    }
    my_gen++;
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(Bar* bar, float* data, int words, int iters, unsigned* errs, long long* cyc) {
    const unsigned G = gridDim.x, wg = blockIdx.x;
    unsigned my_gen = __hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    my_gen = __builtin_amdgcn_readfirstlane(my_gen);
    unsigned bad = 0;
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        float* mine = data + ((size_t)(it & 1) * G + wg) * words;
        if (MODE == 1) {
            for (int i = threadIdx.x; i < words; i += 256) mine[i] = (float)(it * 131 + wg + i);
            __threadfence();
        } else if (MODE == 2) {
            for (int i = threadIdx.x; i < words; i += 256)
                __hip_atomic_store(mine + i, (float)(it * 131 + wg + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_barrier(bar, G, my_gen);
        const unsigned src = (wg + G / 2 + 1) % G;
        const float* theirs = data + ((size_t)(it & 1) * G + src) * words;
        if (MODE == 1) {
            __threadfence();
            for (int i = threadIdx.x; i < words; i += 256) bad += theirs[i] != (float)(it * 131 + src + i);
        } else if (MODE == 2) {
            for (int i = threadIdx.x; i < words; i += 256)
                bad += __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (float)(it * 131 + src + i);
        }
    }
    long long t1 = wall_clock64();
    if (bad) atomicAdd(errs, bad);
    if (threadIdx.x == 0) cyc[wg] = t1 - t0;
}

__global__ __launch_bounds__(256) void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 9999) p[0] = 1.f; }

__global__ __launch_bounds__(256) void tiny_kernel(float* data, int words, int it) {
    float* mine = data + ((size_t)(it & 1) * gridDim.x + blockIdx.x) * words;
    const unsigned src = (blockIdx.x + gridDim.x / 2 + 1) % gridDim.x;
    const float* theirs = data + ((size_t)((it + 1) & 1) * gridDim.x + src) * words;
    for (int i = threadIdx.x; i < words; i += 256) mine[i] = theirs[i] + 1.f;
}

template <int MODE>
static void run(int G, int words, int iters) {
    Bar* bar; float* data; unsigned* errs; long long* cyc;
    CK(hipMalloc(&bar, sizeof(Bar))); CK(hipMemset(bar, 0, sizeof(Bar)));
    CK(hipMalloc(&data, sizeof(float) * 2 * (size_t)G * (words > 0 ? words : 1)));
    CK(hipMalloc(&errs, 4)); CK(hipMemset(errs, 0, 4));
    CK(hipMalloc(&cyc, 8 * G));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(G), dim3(256), 0, 0, bar, data, words, iters, errs, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h; CK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
        if (rep == 2) printf("mode %d  G=%4d  payload %6d B/wg  %5d phases: %8.2f us total, %6.3f us per phase, errors %u\n", MODE, G, words * 4, iters, ms * 1e3, ms * 1e3 / iters, h);
    }
    CK(hipFree(bar)); CK(hipFree(data)); CK(hipFree(errs)); CK(hipFree(cyc));
}

static void run_graph(int G, int words, int iters, bool tiny) {
    float* data; CK(hipMalloc(&data, sizeof(float) * 2 * (size_t)G * (words > 0 ? words : 1)));
    CK(hipMemset(data, 0, sizeof(float) * 2 * (size_t)G * (words > 0 ? words : 1)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int it = 0; it < iters; ++it) {
        if (tiny) hipLaunchKernelGGL(tiny_kernel, dim3(G), dim3(256), 0, s, data, words, it);
        else hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(256), 0, s, data);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("graph of %5d %s kernels  G=%4d payload %6d B/wg: %8.2f us total, %6.3f us per node\n", iters, tiny ? "copy " : "empty", G, words * 4, ms * 1e3, ms * 1e3 / iters);
    }
    CK(hipFree(data));
}

int main() {
    const int iters = 200;
    for (int G : {256, 512, 1024}) {
        run<0>(G, 0, iters);
        for (int words : {256, 4096}) { run<1>(G, words, iters); run<2>(G, words, iters); }
    }
    for (int G : {256, 512, 1024}) { run_graph(G, 0, iters, false); for (int words : {256, 4096}) run_graph(G, words, iters, true); }
    return 0;
}
