// Probe 2 (GPU box): the D1-tail tap GEMM with the FEATURE operand streamed global -> LDS by the DMA form of the
// buffer loads (buffer_load_dwordx4 ... lds: no VGPR round trip, 1 KB per wave instruction) into a double-buffered
// image of the tile's 96 channel rows x (64 columns + halo), all weights resident in LDS, persistent workgroups
// (one per CU) that walk the column tiles.  Both MFMA operands are ds_read_b32; the three temporal taps read the same
// image at shifted columns.  out[64, cols] = lrelu(W_t (3 taps x 64 ch) * z + W_r (32 ch) * x), cols = N*64*11.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}

constexpr int CZ = 64, CX = 32, M = 64, K = 3 * CZ + CX, KS = K / 2;
constexpr int TC = 64;              // columns per tile
constexpr int HALO = 16;            // image starts HALO columns before the tile (>= V, multiple of 4)
constexpr int IW = TC + 2 * HALO;   // image width in floats (96)
constexpr int SEG = IW / 4;         // 16-byte segments per image row (24)
constexpr int WP = M + 1;           // weight pitch in LDS: Wl[k][m]
constexpr int IMG = (CZ + CX) * IW; // floats per image

__global__ __launch_bounds__(256, 2) void dma_kernel(const float* __restrict__ z, const float* __restrict__ x,
                                                      const float* __restrict__ w, float* __restrict__ out, int ncols,
                                                      long z_sC, long x_sC, int V, int ntiles) {
    extern __shared__ float lds[];
    float* const Wl = lds;                       // [K][WP] - only while the weights are moved to registers
    float* const Im = lds;                       // [2][CZ + CX][IW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
    const int ri = wave >> 1, cg = wave & 1;     // this wave: row tile ri, column group cg of the 64-column tile
    // ---- weights -> LDS (once): W[m][k] row-major in global -> Wl[k][m]
    for (int e = tid; e < M * K; e += 256) {
        const int m = e / K, k = e - m * K;
        Wl[k * WP + m] = w[e];
    }
    __syncthreads();
    float wr[KS];                                // A operands of this wave's row tile: W[32 ri + l31][2 q + kh]
#pragma unroll
    for (int q = 0; q < KS; ++q) wr[q] = Wl[(2 * q + kh) * WP + 32 * ri + l31];
    __syncthreads();
    // ---- DMA plan of one image: flat 16-byte segments s = 64 j + lane, j = wave, wave + 4, ...
    const __amdgpu_buffer_rsrc_t zd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(z), 0, (int)(z_sC * CZ * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x), 0, (int)(x_sC * CX * 4), 0x00020000);
    constexpr int NZ = CZ * SEG / 64, NX = CX * SEG / 64;       // wave instructions per image: 24 (z) + 12 (x)
    unsigned zoff[NZ / 4], xoff[NX / 4];
#pragma unroll
    for (int i = 0; i < NZ / 4; ++i) {
        const int s = 64 * (wave + 4 * i) + lane, row = s / SEG, sg = s - row * SEG;
        zoff[i] = (unsigned)(((long)row * z_sC + 32 - HALO + 4 * sg) * 4);       // z has 32 leading pad floats
    }
#pragma unroll
    for (int i = 0; i < NX / 4; ++i) {
        const int s = 64 * (wave + 4 * i) + lane, row = s / SEG, sg = s - row * SEG;
        // x has no leading pad: columns before 0 / beyond the row end are never used by the MFMAs (tap shift 0)
        xoff[i] = (unsigned)(((long)row * x_sC + 4 * sg) * 4) - (unsigned)(HALO * 4);
    }
    auto dma = [&](int tile, int b) {
        float* const im = Im + b * IMG;
        const unsigned tb = (unsigned)(tile * TC * 4);
        const bool live = tile < ntiles;
#pragma unroll
        for (int i = 0; i < NZ / 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(zd, LDSP(im + 256 * (wave + 4 * i)), 16, live ? zoff[i] + tb : 0x80000000u, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NX / 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xd, LDSP(im + CZ * IW + 256 * (wave + 4 * i)), 16,
                                                     (live && tile * TC + 0 >= 0) ? xoff[i] + tb : 0x80000000u, 0, 0, 0);
    };
    int tile = blockIdx.x, b = 0;
    dma(tile, 0);
    for (; tile < ntiles; tile += gridDim.x, b ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // image b complete; nobody reads image b^1 any more
        dma(tile + gridDim.x, b ^ 1);
        const float* const im = Im + b * IMG + HALO + 32 * cg + l31 + kh * IW;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // B operands: LDS reads run CH k-steps ahead of the MFMAs (two register chunks), pinned with sched_barrier -
        // left alone hipcc waits for every ds_read right before its MFMA
        constexpr int CH = 8;
        auto bread = [&](int q) -> float {
            if (q < 3 * (CZ / 2)) {
                const int d = q / (CZ / 2), c2 = q % (CZ / 2);
                return im[2 * c2 * IW + (d - 1) * V];
            }
            return im[(CZ + 2 * (q - 3 * (CZ / 2))) * IW];
        };
        float bq[2][CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) bq[0][i] = bread(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < KS / CH; ++c) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if ((c + 1) * CH + i < KS) bq[(c + 1) & 1][i] = bread((c + 1) * CH + i);
#pragma unroll
            for (int i = 0; i < CH; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[c * CH + i], bq[c & 1][i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int col = tile * TC + 32 * cg + l31;
        if (col < ncols) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * ri + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float v = acc[r];
                out[(long)m * ncols + col] = v > 0.f ? v : 0.2f * v;
            }
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64;
    const int T = 64, V = 11;
    const int ncols = N * T * V;
    const long zs = ncols + 64, xs = ncols + 64;        // x over-allocated by a tail so that halo reads stay in range
    std::vector<float> hz((size_t)CZ * zs, 0.f), hx((size_t)CX * xs, 0.f), hw((size_t)M * K);
    srand(1);
    for (int c = 0; c < CZ; ++c) for (int j = 0; j < ncols; ++j) hz[c * zs + 32 + j] = (rand() % 2001 - 1000) / 1000.f;
    for (int c = 0; c < CX; ++c) for (int j = 0; j < ncols; ++j) hx[c * xs + j] = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 4000.f;
    float *z, *x, *w, *out;
    hipMalloc(&z, hz.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&out, (size_t)M * ncols * 4);
    hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int ntiles = (ncols + TC - 1) / TC;
    const size_t smem = (size_t)(2 * IMG > K * WP ? 2 * IMG : K * WP) * sizeof(float);
    hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    printf("LDS %zu bytes, %d tiles\n", smem, ntiles);
    for (int wgs : {256, 512, 704, 768}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 3; ++it)
            hipLaunchKernelGGL(dma_kernel, dim3(wgs), dim3(256), smem, 0, z, x, w, out, ncols, zs, xs, V, ntiles);
        hipEventRecord(e0);
        const int reps = 50;
        for (int it = 0; it < reps; ++it)
            hipLaunchKernelGGL(dma_kernel, dim3(wgs), dim3(256), smem, 0, z, x, w, out, ncols, zs, xs, V, ntiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, fl = 2.0 * M * K * ncols;
        printf("N=%d cols=%d wgs=%d: %.1f us  %.1f TF/s (%.2f of 157.3)  %s\n", N, ncols, wgs, us, fl / us / 1e6, fl / us / 1e6 / 157.3,
               hipGetErrorString(hipGetLastError()));
    }
    std::vector<float> ho((size_t)M * ncols);
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 400; ++t) {
        const int m = rand() % M, j = t < 64 ? t : (t < 128 ? ncols - 1 - (t - 64) : rand() % ncols);
        double s = 0;
        for (int d = 0; d < 3; ++d) for (int c = 0; c < CZ; ++c) s += (double)hw[m * K + d * CZ + c] * hz[c * zs + 32 + j + (d - 1) * V];
        for (int c = 0; c < CX; ++c) s += (double)hw[m * K + 3 * CZ + c] * hx[c * xs + j];
        s = s > 0 ? s : 0.2 * s;
        maxerr = fmax(maxerr, fabs(s - ho[(size_t)m * ncols + j]));
    }
    printf("max abs err (400 samples): %.3e\n", maxerr);
    return 0;
}
