// Effective shader clock under a saturated fp32 matrix pipe (GPU box): every wave issues independent
// v_mfma_f32_32x32x2_f32 back to back and stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around the loop.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/probe/clock_probe.hip && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void burn(unsigned long long* out, int iters, float seed, const float* rnd) {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // operands: constants (rnd == nullptr: low toggle rate) or 8 + 8 N(0,1) values per lane, cycled (the clock the chip
    // sustains depends on the data: DVFS follows power)
    float av[8], bv[8];
    for (int i = 0; i < 8; ++i) {
        av[i] = rnd ? rnd[(i * 256 + threadIdx.x) + 4096 * (blockIdx.x & 7)] : seed + threadIdx.x * 1e-3f;
        bv[i] = rnd ? rnd[((8 + i) * 256 + threadIdx.x) + 4096 * (blockIdx.x & 7)] : seed * 0.5f + 1e-3f;
    }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u * 4 + i) & 7], bv[(u + i * 3) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = out + 4 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        o[0] = c1 - c0; o[1] = r1 - r0; o[2] = (unsigned long long)(s == 12345.f);
    }
}

int main() {
    unsigned long long* d;
    const int maxwg = 256 * 8;
    hipMalloc(&d, maxwg * 4 * 4 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(maxwg * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float* rnd;
    {
        std::vector<float> hr(8 * 4096);
        unsigned long long st = 88172645463325252ull;
        for (auto& v : hr) {        // sum of 12 uniforms - 6: ~N(0,1)
            float acc = 0.f;
            for (int k = 0; k < 12; ++k) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; acc += (float)((st >> 11) & 0xffffff) / 16777216.f; }
            v = acc - 6.f;
        }
        hipMalloc(&rnd, hr.size() * 4);
        hipMemcpy(rnd, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    }
    printf("%8s %6s %6s | %10s %10s %9s %9s %9s\n", "iters", "wg/CU", "reps", "wall us", "cyc/mfma", "clk GHz", "TF/s", "of 157.3");
    for (int data = 0; data < 2; ++data) {
    printf("operands: %s\n", data ? "N(0,1) per lane" : "constants");
    const float* rp = data ? rnd : nullptr;
    for (int wgcu : {1, 2}) {
        for (int iters : {64, 256, 1024, 16384}) {
            const int grid = 256 * wgcu;
            for (int reps : {1, 50}) {
                if (reps > 1 && iters > 1024) continue;
                for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(burn, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, rp);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(burn, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f, rp);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h.data(), d, grid * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                std::vector<double> clk, cpm;
                for (int i = 0; i < grid * 4; ++i) {
                    const double cyc = (double)h[4 * i], rt = (double)h[4 * i + 1] * 10.0;   // ns
                    clk.push_back(cyc / rt);
                    cpm.push_back(cyc * 1.0 / (16.0 * iters) / wgcu);      // per MFMA of the SIMD (wgcu waves share it)
                }
                std::sort(clk.begin(), clk.end()); std::sort(cpm.begin(), cpm.end());
                const double us = ms * 1e3 / reps;
                const double flop = (double)grid * 4 * 16.0 * iters * 4096.0;
                printf("%8d %6d %6d | %10.1f %10.1f %9.3f %9.1f %9.3f\n", iters, wgcu, reps, us, cpm[cpm.size() / 2],
                       clk[clk.size() / 2], flop / us * 1e-6, flop / us * 1e-6 / 157.3);
            }
        }
    }
    }
    return 0;
}
