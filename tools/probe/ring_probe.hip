// Probe (GPU box): is a persistent, LDS-ring form of the D-block-1 tail contraction faster than kg_conv's one-tile-per-workgroup
// form?  out[m, j] = sum_d sum_c Wt(d,m,c) z[c, j + (d-1) V] + sum_c Wr(m,c) x[c, j]   (M = 64, K = 3*64 + 32 = 224; the frame
// boundaries of the real conv are ignored: z is one long row with a halo of V columns).
//   * one workgroup of 8 waves per CU walks 64 x 128 tiles;
//   * the 57 KB weight matrix stays in LDS for the whole launch;
//   * the feature operand arrives through LDS-DMA (buffer_load_dword ... lds) into a ring of 4 stages of 32 channels x 192
//     positions, issued three stages ahead, counted s_waitcnt vmcnt(N), one s_barrier per stage;
//   * the three temporal taps read the SAME staged rows at shifted positions (the direct kernel fetches them three times).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/ring_probe tools/probe/ring_probe.hip && /tmp/ring_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int M = 64, CZ = 64, CX = 32, K = 3 * CZ + CX, BN = 128, SWP = 192, NSTAGE = 4;
constexpr int STAGE_FLOATS = 32 * SWP;

__device__ __forceinline__ void dma_dword(unsigned lds_byte, unsigned voff, v4i rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

__device__ __forceinline__ v4i make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    v4i r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)u);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}

// z: [CZ][zstride] (zstride = ncols + 2V, column j of the problem at index j + V), x: [CX][ncols], w: [K][M] (k-major), out: [M][ncols]
__global__ __launch_bounds__(512, 1) void ring_kernel(const float* __restrict__ z, const float* __restrict__ x, const float* __restrict__ w,
                                                      float* __restrict__ out, int ncols, int ntiles, int zstride, int V, unsigned long long* clk) {
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
    extern __shared__ float lds[];
    float* Wres = lds;                      // [K][M]
    float* ring = lds + K * M;              // NSTAGE x [32][SWP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int rh = wave & 1, cq = wave >> 1;
    for (int i = tid; i < K * M; i += 512) Wres[i] = w[i];
    __syncthreads();
    const v4i rz = make_rsrc(z, (unsigned)((size_t)CZ * zstride * 4));
    const v4i rx = make_rsrc(x, (unsigned)((size_t)CX * ncols * 4));
    const int G = gridDim.x, b = blockIdx.x;
    const int my_tiles = b < ntiles ? (ntiles - b + G - 1) / G : 0;
    const int total = my_tiles * 3;
    const unsigned ring_base = (unsigned)(size_t)(ring - lds) * 4u;      // dynamic LDS starts at byte 0 of the allocation
    const int slab_w = BN + 2 * V;

    auto issue = [&](int g) {
        const int it = g / 3, s = g - it * 3;
        const int c0 = (b + it * G) * BN;
        const unsigned slot = ring_base + (unsigned)(g & (NSTAGE - 1)) * (STAGE_FLOATS * 4u);
        if (s < 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 4 + i;
                const unsigned soff = (unsigned)(s * 32 + row) * (unsigned)zstride * 4u;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    const int p = pc * 64 + lane;
                    const unsigned voff = p < slab_w ? (unsigned)(c0 + p) * 4u : 0x80000000u;
                    dma_dword(slot + (unsigned)(row * SWP + pc * 64) * 4u, voff, rz, soff);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 4 + i;
                const unsigned soff = (unsigned)row * (unsigned)ncols * 4u;
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    dma_dword(slot + (unsigned)(row * SWP + pc * 64) * 4u, (unsigned)(c0 + pc * 64 + lane) * 4u, rx, soff);
            }
        }
    };

    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (total > 0) issue(0);
    if (total > 1) issue(1);
    if (total > 2) issue(2);
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
        const int it = g / 3, s = g - it * 3;
        // stage g has landed when at most the DMAs of the two younger stages are outstanding (12 + 8 at least)
#ifndef NOWAIT
        if (g + 2 < total) wait_vm<20>();
        else wait_vm<0>();
#endif
#ifndef NOBARRIER
        wg_barrier();
#endif
#ifndef NODMA
        if (g + 3 < total) issue(g + 3);
#endif
        const float* slab = ring + (g & (NSTAGE - 1)) * STAGE_FLOATS;
        const float* bcol = slab + h * SWP + cq * 32 + l32;
        const float* arow = Wres + rh * 32 + l32 + h * M;
#ifndef NPF
#define NPF 4
#endif
        // k-steps of this stage: z stage 48 (tap d = i / 16), x stage 16; operands prefetched NPF steps ahead
        const int nsteps = s < 2 ? 48 : 16;
        const float* ap0 = arow + (s < 2 ? s * 32 : 3 * CZ) * M;
        auto lda = [&](int i) { const int d = i >> 4, q = i & 15; return ap0[(d * CZ + 2 * q) * M]; };
        auto ldb = [&](int i) { const int d = i >> 4, q = i & 15; return bcol[2 * q * SWP + d * V]; };
        if (s < 2) {
            float av[NPF], bv[NPF];
#pragma unroll
            for (int i = 0; i < NPF; ++i) { av[i] = lda(i); bv[i] = ldb(i); }
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                const float a_ = av[i % NPF], b_ = bv[i % NPF];
                if (i + NPF < 48) { av[i % NPF] = lda(i + NPF); bv[i % NPF] = ldb(i + NPF); }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, acc, 0, 0, 0);
#ifdef PIN
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // 2 DS reads
#endif
            }
        } else {
            float av[NPF], bv[NPF];
#pragma unroll
            for (int i = 0; i < NPF; ++i) { av[i] = lda(i); bv[i] = ldb(i); }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float a_ = av[i % NPF], b_ = bv[i % NPF];
                if (i + NPF < 16) { av[i % NPF] = lda(i + NPF); bv[i % NPF] = ldb(i + NPF); }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, acc, 0, 0, 0);
#ifdef PIN
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#endif
            }
            (void)nsteps;
            // epilogue: LeakyReLU, C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*h
            const int c0 = (b + it * G) * BN;
            float* op = out + (size_t)(rh * 32 + 4 * h) * ncols + c0 + cq * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r];
#ifdef NOSTORE
                if (v == 123.456f)
#endif
                op[(size_t)((r & 3) + 8 * (r >> 2)) * ncols] = v > 0.f ? v : 0.2f * v;
                acc[r] = 0.f;
            }
        }
    }
    if (tid == 0 && clk) {
        clk[4 * b] = __builtin_amdgcn_s_memtime() - ck0;
        clk[4 * b + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        clk[4 * b + 2] = rt_entry;
        clk[4 * b + 3] = __builtin_amdgcn_s_memrealtime();
    }
}


// v2: the stage's DMA issue and the PREVIOUS tile's stores are spread between the MFMAs of a stage (one per few MFMAs,
// pinned), so that neither is a phase in which both waves of a SIMD leave the matrix pipe idle.
__global__ __launch_bounds__(512, 1) void ring_kernel_v2(const float* __restrict__ z, const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ out, int ncols, int ntiles, int zstride, int V, unsigned long long* clk) {
    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
    extern __shared__ float lds[];
    float* Wres = lds;
    float* ring = lds + K * M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int rh = wave & 1, cq = wave >> 1;
    const v4i rz = make_rsrc(z, (unsigned)((size_t)CZ * zstride * 4));
    const v4i rx = make_rsrc(x, (unsigned)((size_t)CX * ncols * 4));
    const v4i rw = make_rsrc(w, (unsigned)(K * M * 4));
    const int G = gridDim.x, b = blockIdx.x;
    const int my_tiles = b < ntiles ? (ntiles - b + G - 1) / G : 0;
    const unsigned ring_base = (unsigned)(K * M) * 4u;
    const int slab_w = BN + 2 * V;
    // the weights through LDS-DMA too: K*M/64 = 224 pieces of 64 floats, 28 per wave
#pragma unroll 4
    for (int i = 0; i < K * M / 64 / 8; ++i) {
        const int piece = wave * (K * M / 64 / 8) + i;
        dma_dword((unsigned)piece * 256u, (unsigned)(piece * 64 + lane) * 4u, rw, 0u);
    }
    // one DMA of stage type S (0, 1: z channels S*32.., 2: x) for the tile whose first column is c0, index e in [0, 12) / [0, 8)
    auto dma_one = [&](int S, int e, int c0, unsigned slot) {
        if (S < 2) {
            const int i = e / 3, pc = e - i * 3;
            const int row = wave * 4 + i;
            const int p = pc * 64 + lane;
            const unsigned voff = p < slab_w ? (unsigned)(c0 + p) * 4u : 0x80000000u;
            dma_dword(slot + (unsigned)(row * SWP + pc * 64) * 4u, voff, rz, (unsigned)(S * 32 + row) * (unsigned)zstride * 4u);
        } else {
            const int i = e >> 1, pc = e & 1;
            const int row = wave * 4 + i;
            dma_dword(slot + (unsigned)(row * SWP + pc * 64) * 4u, (unsigned)(c0 + pc * 64 + lane) * 4u, rx, (unsigned)row * (unsigned)ncols * 4u);
        }
    };
    auto slot_of = [&](int g) { return ring_base + (unsigned)(g & (NSTAGE - 1)) * (STAGE_FLOATS * 4u); };
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    f16v acc, accp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accp[r] = 0.f; }
    // prologue: the three stages of the first tile
    if (my_tiles > 0) {
        const int c0 = b * BN;
#pragma unroll
        for (int e = 0; e < 12; ++e) dma_one(0, e, c0, slot_of(0));
#pragma unroll
        for (int e = 0; e < 12; ++e) dma_one(1, e, c0, slot_of(1));
#pragma unroll
        for (int e = 0; e < 8; ++e) dma_one(2, e, c0, slot_of(2));
    }
    const float* arow = Wres + rh * 32 + l32 + h * M;
    float* op_prev = out;
#pragma unroll 1
    for (int it = 0; it < my_tiles; ++it) {
        const int c0 = (b + it * G) * BN;
        const int c0n = c0 + G * BN;
        const bool more = it + 1 < my_tiles;
        const bool storep = it > 0;
        float* const op = out + (size_t)(rh * 32 + 4 * h) * ncols + c0 + cq * 32 + l32;
#pragma unroll
        for (int S = 0; S < 3; ++S) {
            const int g = it * 3 + S;
            // stage g has landed when at most the DMAs of the two younger stages are outstanding
#ifndef NOWAIT
            if (more) wait_vm<20>();
            else wait_vm<0>();
#endif
#ifndef NOBARRIER
            wg_barrier();
#endif
            const float* slab = ring + (g & (NSTAGE - 1)) * STAGE_FLOATS;
            const float* bcol = slab + h * SWP + cq * 32 + l32;
            const float* ap0 = arow + (S < 2 ? S * 32 : 3 * CZ) * M;
            constexpr int NS = 48;
            const int nsteps = S < 2 ? 48 : 16;
            const int ndma = S < 2 ? 12 : 8;
            const unsigned nslot = slot_of(g + 3);
            auto lda = [&](int i) { const int d = i >> 4, q = i & 15; return ap0[(d * CZ + 2 * q) * M]; };
            auto ldb = [&](int i) { const int d = i >> 4, q = i & 15; return bcol[2 * q * SWP + d * V]; };
            float av[NPF], bv[NPF];
#pragma unroll
            for (int i = 0; i < NPF; ++i) { av[i] = lda(i); bv[i] = ldb(i); }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (i < nsteps) {
                    const float a_ = av[i % NPF], b_ = bv[i % NPF];
                    if (i + NPF < nsteps) { av[i % NPF] = lda(i + NPF); bv[i % NPF] = ldb(i + NPF); }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, acc, 0, 0, 0);
                    // one DMA of the stage three ahead (same type, next tile) per MFMA until they are out
                    if (i < ndma && more) dma_one(S, i, c0n, nslot);
                    // the previous tile's stores ride along the first stage
                    if (S == 0 && i >= 16 && i < 32 && storep) {
                        const int r = i - 16;
                        const float v = accp[r];
                        op_prev[(size_t)((r & 3) + 8 * (r >> 2)) * ncols] = v > 0.f ? v : 0.2f * v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        accp = acc;
        op_prev = op;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
    if (my_tiles > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = accp[r];
            op_prev[(size_t)((r & 3) + 8 * (r >> 2)) * ncols] = v > 0.f ? v : 0.2f * v;
        }
    }
    if (lane == 0 && clk) {     // (all waves: the slowest one defines the end)
        atomicMax(&clk[4 * b], __builtin_amdgcn_s_memtime() - ck0);
        atomicMax(&clk[4 * b + 1], __builtin_amdgcn_s_memrealtime() - rt0);
        if (tid == 0) clk[4 * b + 2] = rt_entry;
        atomicMax(&clk[4 * b + 3], __builtin_amdgcn_s_memrealtime());
    }
}
#ifdef V2
#define ring_kernel ring_kernel_v2
#endif

int main(int argc, char** argv) {
    const int V = 11;
    const int ncols = argc > 1 ? atoi(argv[1]) : 131072;
    const int ntiles = ncols / BN;
    const int zstride = ncols + 2 * V;
    std::vector<float> hz((size_t)CZ * zstride), hx((size_t)CX * ncols), hw((size_t)K * M);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xffffff) / 8388608.f - 1.f; };
    for (auto& v : hz) v = rnd();
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    float *dz, *dx, *dw, *dout;
    hipMalloc(&dz, hz.size() * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&dout, (size_t)M * ncols * 4);
    hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const size_t lds_bytes = (size_t)(K * M + NSTAGE * STAGE_FLOATS) * 4;
    hipFuncSetAttribute((const void*)ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    unsigned long long* dclk;
    hipMalloc(&dclk, 4 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 128}) {
        hipMemset(dout, 0, (size_t)M * ncols * 4);
        hipMemset(dclk, 0, 4 * 256 * 8);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ring_kernel, dim3(grid), dim3(512), lds_bytes, 0, dz, dx, dw, dout, ncols, ntiles, zstride, V, dclk);
        hipDeviceSynchronize();
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) { printf("launch error %s\n", hipGetErrorString(err)); return 1; }
        const int reps = 20;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(ring_kernel, dim3(grid), dim3(512), lds_bytes, 0, dz, dx, dw, dout, ncols, ntiles, zstride, V, dclk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, fl = 2.0 * M * K * ncols;
        // check a few hundred outputs against a host sum
        std::vector<float> ho((size_t)M * ncols);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int t = 0; t < 600; ++t) {
            const int m = (t * 7) % M;
            const int j = (int)(((long)t * 104729 + (t % 3 == 0 ? ncols - 1 - t : 0)) % ncols);
            double acc = 0;
            for (int d = 0; d < 3; ++d)
                for (int c = 0; c < CZ; ++c) acc += (double)hw[(size_t)(d * CZ + c) * M + m] * hz[(size_t)c * zstride + j + d * V];
            for (int c = 0; c < CX; ++c) acc += (double)hw[(size_t)(3 * CZ + c) * M + m] * hx[(size_t)c * ncols + j];
            const double ref = acc > 0 ? acc : 0.2 * acc;
            maxerr = fmax(maxerr, fabs(ref - ho[(size_t)m * ncols + j]));
            maxref = fmax(maxref, fabs(ref));
        }
        hipMemset(dclk, 0, 4 * 256 * 8);
        hipLaunchKernelGGL(ring_kernel, dim3(grid), dim3(512), lds_bytes, 0, dz, dx, dw, dout, ncols, ntiles, zstride, V, dclk);
        hipDeviceSynchronize();
        unsigned long long hc[1024];
        hipMemcpy(hc, dclk, sizeof(hc), hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0, bmin = 1e30, bmax = 0;
        unsigned long long t0 = ~0ull, t1 = 0, smax = 0;
        for (int i = 0; i < grid; ++i) {
            cyc += (double)hc[4 * i]; rt += (double)hc[4 * i + 1];
            bmin = fmin(bmin, (double)hc[4 * i + 1]); bmax = fmax(bmax, (double)hc[4 * i + 1]);
            if (hc[4 * i + 2] < t0) t0 = hc[4 * i + 2];
            if (hc[4 * i + 2] > smax) smax = hc[4 * i + 2];
            if (hc[4 * i + 3] > t1) t1 = hc[4 * i + 3];
        }
        printf("clock %.2f GHz body avg %.1f min %.1f max %.1f us, last start +%.1f, span %.1f | ", cyc / rt * 0.1, rt / grid * 0.01, bmin * 0.01, bmax * 0.01,
               (double)(smax - t0) * 0.01, (double)(t1 - t0) * 0.01);
        printf("ring probe: ncols %d grid %d: %.2f us  %.1f TF/s (%.3f of 157.3)  max err %.2e (max |ref| %.2f)\n", ncols, grid, us, fl / us / 1e6,
               fl / us / 1e6 / 157.3, maxerr, maxref);
    }
    return 0;
}
