// Probe (GPU box): a weight-STATIONARY tap GEMM for the small-M layers of the discriminator (D0 / D1: M <= 96,
// K <= 224).  Every wave keeps ALL weights of its TM row tiles in registers (A operands of v_mfma_f32_32x32x2_f32:
// TM * K/2 VGPRs), walks 32-column groups of the launch and only streams the feature operand: no LDS, no barrier,
// one global load per TM MFMAs, each operand register reloaded for the next group right after its last use.
// Shape of the timing run = disc block 1 tail (bench.py roofline leg): out[64, cols] = W_t (3 taps x 64 ch) * z
// + W_r (32 ch) * x, cols = N*64*11.  Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}

// z: (64, ncols + 2*PADC) with PADC leading zeros, x: (32, ncols); w: (64 rows, 224) row-major; V = frame width
template <int TM, int CZ, int CX>
__global__ __launch_bounds__(256, 1) void ws_kernel(const float* __restrict__ z, const float* __restrict__ x,
                                                     const float* __restrict__ w, float* __restrict__ out, int ncols,
                                                     long z_sC, long x_sC, int V, int ngroups) {
    constexpr int KS = (3 * CZ + CX) / 2;
    const int lane = threadIdx.x & 63, kh = lane >> 5, l31 = lane & 31;
    const int gwave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = gridDim.x * 4;
    // ---- weights -> registers: wr[q][i] = W[32 i + l31][2 q + kh]
    float wr[KS][TM];
    {
        const __amdgpu_buffer_rsrc_t wd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(w), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < KS; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                wr[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wd, (unsigned)(((32 * i + l31) * (2 * KS) + 2 * q + kh) * 4), 0, 0));
    }
    const __amdgpu_buffer_rsrc_t zd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(z), 0, (int)(z_sC * CZ * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x), 0, (int)(x_sC * CX * 4), 0x00020000);
    const unsigned zrow = (unsigned)(z_sC * 4), xrow = (unsigned)(x_sC * 4);
    float b[KS];
    // lane's byte offset of column j (tap d reads column j + (d-1) V of the padded z)
    auto issue = [&](int q, int g) {
        const int col = g * 32 + l31;
        unsigned off;
        if (q < 3 * CZ / 2) {
            const int d = q / (CZ / 2), c = 2 * (q % (CZ / 2)) + kh;
            off = (unsigned)c * zrow + (unsigned)((col + (d - 1) * V + 32) * 4);      // 32 = leading pad
            b[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zd, g < ngroups ? off : 0x80000000u, 0, 0));
        } else {
            const int c = 2 * (q - 3 * CZ / 2) + kh;
            off = (unsigned)c * xrow + (unsigned)(col * 4);
            b[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xd, g < ngroups ? off : 0x80000000u, 0, 0));
        }
    };
    int g = gwave;
#pragma unroll
    for (int q = 0; q < KS; ++q) issue(q, g);
    for (; g < ngroups; g += nwaves) {
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[q][i], b[q], acc[i], 0, 0, 0);
            issue(q, g + nwaves);          // the register is free: fetch the next group's operand
        }
        const int col = g * 32 + l31;
        if (col < ncols) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float v = acc[i][r];
                    out[(long)m * ncols + col] = v > 0.f ? v : 0.2f * v;
                }
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64;
    const int T = 64, V = 11, CZ = 64, CX = 32, M = 64, K = 3 * CZ + CX;
    const int ncols = N * T * V;
    const long zs = ncols + 64, xs = ncols;
    std::vector<float> hz((size_t)CZ * zs, 0.f), hx((size_t)CX * xs), hw((size_t)M * K);
    srand(1);
    for (int c = 0; c < CZ; ++c) for (int j = 0; j < ncols; ++j) hz[c * zs + 32 + j] = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 4000.f;
    float *z, *x, *w, *out;
    hipMalloc(&z, hz.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&out, (size_t)M * ncols * 4);
    hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int ngroups = (ncols + 31) / 32;
    for (int wgs : {256, 512, 352, 704}) {
        if (wgs * 4 > ngroups) { /* fewer waves than groups needed */ }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 3; ++it)
            hipLaunchKernelGGL((ws_kernel<2, 64, 32>), dim3(wgs), dim3(256), 0, 0, z, x, w, out, ncols, zs, xs, V, ngroups);
        hipEventRecord(e0);
        const int reps = 50;
        for (int it = 0; it < reps; ++it)
            hipLaunchKernelGGL((ws_kernel<2, 64, 32>), dim3(wgs), dim3(256), 0, 0, z, x, w, out, ncols, zs, xs, V, ngroups);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, fl = 2.0 * M * K * ncols;
        printf("N=%d cols=%d wgs=%d: %.1f us  %.1f TF/s (%.2f of 157.3)\n", N, ncols, wgs, us, fl / us / 1e6, fl / us / 1e6 / 157.3);
    }
    // check a few outputs
    std::vector<float> ho((size_t)M * ncols);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 200; ++t) {
        const int m = rand() % M, j = rand() % ncols;
        double s = 0;
        for (int d = 0; d < 3; ++d) for (int c = 0; c < CZ; ++c) s += (double)hw[m * K + d * CZ + c] * hz[c * zs + 32 + j + (d - 1) * V];
        for (int c = 0; c < CX; ++c) s += (double)hw[m * K + 3 * CZ + c] * hx[c * xs + j];
        s = s > 0 ? s : 0.2 * s;
        maxerr = fmax(maxerr, fabs(s - ho[(size_t)m * ncols + j]));
    }
    printf("max abs err (200 samples): %.3e\n", maxerr);
    return 0;
}
