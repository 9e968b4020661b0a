// Round 5 ("bisect the aggregation leg"): times kg_agg_reduce at the C5a shape (3 x 512 -> 512 planes, T = 256,
// V = W = 25, 64 samples: the bench's `roofline_agg` leg) against ANY build of the library - compiled against that build's own
// include/kgan_hip.h (KgAggArgs only ever grew at its end), the library loaded with dlopen:
//   hipcc -O2 -I <tree>/include -o agg_c5a_<tag> tools/probe/agg_c5a_driver.cpp -ldl ;  ./agg_c5a_<tag> <path to libkgan_hip.so>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "kgan_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s libkgan_hip.so [reps]\n", argv[0]); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
    typedef int (*fn_t)(const KgAggArgs*, void*);
    fn_t reduce = (fn_t)dlsym(h, "kg_agg_reduce");
    typedef const char* (*err_t)(void);
    err_t lasterr = (err_t)dlsym(h, "kg_last_error");
    if (!reduce) { printf("no kg_agg_reduce\n"); return 1; }
    const int N = 64, C = 512, K = 3, V = 25, W = 25, T = 256;
    const long L = (long)T * V;
    // plane tensors (channels, N, T, V): y has K*C channels, out C
    float *y, *out, *a;
    const size_t ny = (size_t)K * C * N * L, no = (size_t)C * N * T * W;
    hipMalloc(&y, ny * 4 + 256); hipMalloc(&out, no * 4 + 256); hipMalloc(&a, K * V * W * 4);
    std::vector<float> hy(1 << 20);
    unsigned st = 12345u;
    for (auto& v : hy) { st = st * 1664525u + 1013904223u; v = (float)(st >> 8) / 16777216.f - 0.5f; }
    for (size_t off = 0; off < ny; off += hy.size()) hipMemcpy(y + off, hy.data(), (ny - off < hy.size() ? ny - off : hy.size()) * 4, hipMemcpyHostToDevice);
    hipMemcpy(a, hy.data(), K * V * W * 4, hipMemcpyHostToDevice);
    KgAggArgs g;
    memset(&g, 0, sizeof g);
    g.N = N; g.C = C; g.K = K; g.V = V; g.W = W; g.T = T; g.rep = 1;
    g.a = a;
    g.x = y; g.x_sN = L; g.x_sC = (long)N * L;
    g.out = out; g.o_sN = (long)T * W; g.o_sC = (long)N * T * W;
    int rc = reduce(&g, nullptr);
    hipDeviceSynchronize();
    if (rc != 0) { printf("kg_agg_reduce rc %d: %s\n", rc, lasterr ? lasterr() : "?"); return 1; }
    const int reps = argc > 2 ? atoi(argv[2]) : 30;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) reduce(&g, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= reps; sum += ms; if (ms < best) best = ms;
    }
    std::vector<float> ho(64);
    hipMemcpy(ho.data(), out, 64 * 4, hipMemcpyDeviceToHost);
    double chk = 0; for (float v : ho) chk += v;
    const double bytes = (double)(ny + no) * 4;
    printf("%s: kg_agg_reduce C5a  best %.4f ms  mean %.4f ms  %.0f GB/s (best)  chk %.6f\n", argv[1], best, sum / 5, bytes / best / 1e6, chk);
    return 0;
}
