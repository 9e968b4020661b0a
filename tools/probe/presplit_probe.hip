// Probe (GPU box, round 5): what would the D-block-1 tail contraction cost if its operands ARRIVED pre-split and k-interleaved?
//   out[m, j] = lrelu( sum_d sum_c Wt(d,m,c) z[c, j + (d-1) V] + sum_c Wr(m,c) x[c, j] )   (M = 64, K = 3*64 + 32 = 224; the toy
//   problem of ring_probe.hip: z is one long row with a halo of V columns, frame boundaries ignored)
// Round 5's ablations of the bf16-split loop say the fp32 tap GEMM is bound, right behind the fp32 matrix
// pipe, by the VALU work of splitting every operand in every wave and by its 4-byte loads.  Here neither is in the loop:
//   * every fp32 operand is stored as three bf16 planes h / m / l (x = h + m + l to 2^-27, round to nearest at each level),
//     features k-interleaved [term][C/8][position][8]: a lane's eight consecutive channels of one MFMA are ONE 16-byte load;
//   * the weights are stored fragment-ready [slice][term][octet][row][8] and staged through LDS with 16-byte loads / reads;
//   * six v_mfma_f32_32x32x16_bf16 (hh, hm, mh, mm, hl, lh) per 16 channels and 32 x 32 tile: fp32-accurate.
// A workgroup = 4 waves, 64 rows x 128 columns (a wave: 64 x 32, two accumulators), one tile per workgroup as kg_conv_kernel.
// The pack kernels (fp32 -> bf16 triples) are NOT timed: in a product the producing kernel's epilogue would write them.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/presplit_probe tools/probe/presplit_probe.hip && /tmp/presplit_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr int M = 64, CZ = 64, CX = 32, BN = 128, NSL = 7;      // slices: z chunk 0 taps 0..2, z chunk 1 taps 0..2, x

__device__ __forceinline__ unsigned cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt2(x0, x1);
    x0 -= __uint_as_float(h << 16); x1 -= __uint_as_float(h & 0xffff0000u);
    m = cvt2(x0, x1);
    x0 -= __uint_as_float(m << 16); x1 -= __uint_as_float(m & 0xffff0000u);
    l = cvt2(x0, x1);
}

// features: src [C][stride] fp32 -> dst [3][C/8][stride][8] bf16 (as u4 per (term, octet, position))
__global__ void pack_features(const float* src, u4* dst, int C, long stride) {
    const long pos = (long)blockIdx.x * 256 + threadIdx.x;
    const int kg = blockIdx.y;
    if (pos >= stride) return;
    u4 h, m, l;
    for (int p = 0; p < 4; ++p) {
        unsigned a, b, c;
        split_pair(src[(long)(8 * kg + 2 * p) * stride + pos], src[(long)(8 * kg + 2 * p + 1) * stride + pos], a, b, c);
        h[p] = a; m[p] = b; l[p] = c;
    }
    const long per = (long)(C / 8) * stride;
    dst[0 * per + (long)kg * stride + pos] = h;
    dst[1 * per + (long)kg * stride + pos] = m;
    dst[2 * per + (long)kg * stride + pos] = l;
}
// weights: w [K][M] (k-major, k = d*CZ + c for the taps, 3*CZ + c for the residual) -> [slice][term][octet 0..3][m] u4
__global__ void pack_weights(const float* w, u4* dst) {
    const int sl = blockIdx.x, oc = threadIdx.x >> 6, m = threadIdx.x & 63;      // 256 threads: (octet, row)
    const int k0 = sl < 6 ? (sl % 3) * CZ + (sl / 3) * 32 : 3 * CZ;               // first contraction index of the slice
    u4 h, md, l;
    for (int p = 0; p < 4; ++p) {
        unsigned a, b, c;
        split_pair(w[(long)(k0 + 8 * oc + 2 * p) * M + m], w[(long)(k0 + 8 * oc + 2 * p + 1) * M + m], a, b, c);
        h[p] = a; md[p] = b; l[p] = c;
    }
    u4* o = dst + (long)sl * 3 * 4 * M;
    o[(0 * 4 + oc) * M + m] = h;
    o[(1 * 4 + oc) * M + m] = md;
    o[(2 * 4 + oc) * M + m] = l;
}

__device__ __forceinline__ f16v mfma(const u4& a, const u4& b, const f16v& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// zp: [3][CZ/8][zstride] u4, xp: [3][CX/8][ncols] u4, wp: [NSL][3][4][M] u4, out [M][ncols]
__global__ __launch_bounds__(256) void presplit_kernel(const u4* __restrict__ zp, const u4* __restrict__ xp, const u4* __restrict__ wp,
                                                        float* __restrict__ out, int ncols, long zstride, int V) {
    __shared__ u4 Wf[2][3][4][M];                   // 24 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, l32 = lane & 31;
    const int col = blockIdx.x * BN + wave * 32 + l32;
    const bool valid = col < ncols;
    const long zper = (long)(CZ / 8) * zstride, xper = (long)(CX / 8) * ncols;
    f16v acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    u4 wreg[3];                                     // this thread's three 16-byte pieces of a slice's weight block (768 pieces)
    u4 breg[2][2][3];                               // [buffer][K16 group s][term]
    auto fetch = [&](int sl, u4 (&b)[2][3]) {
        const u4* ws = wp + (long)sl * 3 * 4 * M;
#pragma unroll
        for (int i = 0; i < 3; ++i) wreg[i] = ws[tid + 256 * i];
        const int cj = valid ? col : 0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (sl < 6) {
                    const int d = sl % 3, kg = (sl / 3) * 4 + 2 * s + kh;
                    b[s][t] = zp[t * zper + (long)kg * zstride + cj + d * V];
                } else {
                    const int kg = 2 * s + kh;
                    b[s][t] = xp[t * xper + (long)kg * ncols + cj];
                }
            }
    };
    auto stash = [&](int buf) {
        u4* f = &Wf[buf][0][0][0];
#pragma unroll
        for (int i = 0; i < 3; ++i) f[tid + 256 * i] = wreg[i];
    };
    fetch(0, breg[0]);
    stash(0);
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        const int buf = sl & 1;
        if (sl + 1 < NSL) fetch(sl + 1, breg[buf ^ 1]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u4 bh = breg[buf][s][0], bm = breg[buf][s][1], bl = breg[buf][s][2];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const u4 ah = Wf[buf][0][2 * s + kh][tm * 32 + l32], am = Wf[buf][1][2 * s + kh][tm * 32 + l32],
                         al = Wf[buf][2][2 * s + kh][tm * 32 + l32];
                f16v t = acc[tm];
                t = mfma(al, bh, t);
                t = mfma(ah, bl, t);
                t = mfma(am, bm, t);
                t = mfma(am, bh, t);
                t = mfma(ah, bm, t);
                t = mfma(ah, bh, t);
                acc[tm] = t;
            }
        }
        if (sl + 1 < NSL) {
            stash(buf ^ 1);
            __syncthreads();
        }
    }
    if (!valid) return;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float v = acc[tm][r];
            out[(size_t)row * ncols + col] = v > 0.f ? v : 0.2f * v;
        }
}

int main(int argc, char** argv) {
    const int V = 11;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int ncols : {45056, 135168}) {
        const long zstride = ncols + 2 * V;
        const int K = 3 * CZ + CX;
        std::vector<float> hz((size_t)CZ * zstride), hx((size_t)CX * ncols), hw((size_t)K * M);
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xffffff) / 8388608.f - 1.f; };
        for (auto& v : hz) v = rnd();
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd() * 0.1f;
        float *dz, *dx, *dw, *dout;
        u4 *zp, *xp, *wpk;
        hipMalloc(&dz, hz.size() * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&dout, (size_t)M * ncols * 4);
        hipMalloc(&zp, (size_t)3 * (CZ / 8) * zstride * 16); hipMalloc(&xp, (size_t)3 * (CX / 8) * ncols * 16); hipMalloc(&wpk, (size_t)NSL * 3 * 4 * M * 16);
        hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(pack_features, dim3((zstride + 255) / 256, CZ / 8), dim3(256), 0, 0, dz, zp, CZ, zstride);
        hipLaunchKernelGGL(pack_features, dim3((ncols + 255) / 256, CX / 8), dim3(256), 0, 0, dx, xp, CX, (long)ncols);
        hipLaunchKernelGGL(pack_weights, dim3(NSL), dim3(256), 0, 0, dw, wpk);
        const int grid = (ncols + BN - 1) / BN;
        hipMemset(dout, 0, (size_t)M * ncols * 4);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(presplit_kernel, dim3(grid), dim3(256), 0, 0, zp, xp, wpk, dout, ncols, zstride, V);
        hipDeviceSynchronize();
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) { printf("launch error %s\n", hipGetErrorString(err)); return 1; }
        const int reps = 20;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(presplit_kernel, dim3(grid), dim3(256), 0, 0, zp, xp, wpk, dout, ncols, zstride, V);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, fl = 2.0 * M * K * ncols;
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
        printf("presplit probe: ncols %6d (%4d workgroups): %.2f us  %.1f TF/s fp32-equivalent (%.3f of the 157.3-TF fp32 MFMA peak)  max err %.2e (max |ref| %.2f)\n",
               ncols, grid, us, fl / us / 1e6, fl / us / 1e6 / 157.3, maxerr, maxref);
        hipFree(dz); hipFree(dx); hipFree(dw); hipFree(dout); hipFree(zp); hipFree(xp); hipFree(wpk);
    }
    return 0;
}
