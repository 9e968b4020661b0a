// kg_agg_*: spatial graph aggregation over the V joints of a frame with a tiny (K, V, W) matrix.
// These are the HBM-bound kernels of the path: 2*K*V flop per 4*(K+1) bytes.
//
//  expand : out[k*C+c, (n,t',w)] = sum_v x[c,(n,t'/rep,v)] A[k,v,w]
//  reduce : out[c,(n,t,w)]       = sum_q sum_k sum_v y[k*C+c,(n,t*fold+q,v)] A[k,v,w]
//  outer  : dA[k,v,w]            = sum_{c,n,t'} x[c,(n,t'/rep,v)] y[k*C+c,(n,t',w)]
//
// expand / reduce: one thread owns one frame (V contiguous floats) of one channel and keeps all its
// K*W (expand) / W (reduce) outputs in registers; A sits in LDS zero-padded to a multiple of 4 columns
// and is read as wave-uniform 128-bit broadcasts, so the inner loop is 4 FMAs per LDS read and the
// kernel streams at HBM/L2 speed instead of LDS speed.
// outer: dA_k = X^T Y_k is a GEMM with tiny M = V, N = W and a huge contraction (all frames of all channels).
// Stream-contiguous launches (channel-major planes, rep = 1): 64 frames of x and of the K y channels are copied
// to LDS with full-width 128-bit loads (the frames of one channel are one contiguous stream), double buffered;
// v_mfma_f32_16x16x4_f32 contracts four frames per instruction with both operands read from LDS.  Every
// workgroup walks a strided list of (channel, 64-frame chunk) units, its four waves take every fourth frame
// group, the partial sums meet in LDS and leave as one slab per workgroup; kg_agg_outer_sum adds the slabs in a
// fixed order (deterministic).  (Feeding the MFMAs straight from global memory - 2 or 4 frames of <= 100 bytes
// per load instruction - was measured 2x slower than the VALU kernel: load-issue bound.)
// Other launches (sample-strided layouts, rep > 1): frames staged in LDS element-wise, every thread keeps its
// share of the K*V*W outputs in registers.
//
// Reference ops covered: torch.einsum('nkctv,kvw->nctw') (tgcn.py:66) and its gradients;
// upsample_s + nearest T up-sampling (generator.py:172,185-200) with K=1, A=U.
#include <stdlib.h>

#include "kg_common.h"

namespace {

constexpr int NT = 256;

// A (K,V,W) -> LDS As[k][v][WP], zero padded; tr: the source is stored (K,W,V)
template <int K, int WP>
__device__ __forceinline__ void stage_A(float* As, const float* a, int V, int W, int tr) {
    for (int e = threadIdx.x; e < K * V * WP; e += NT) {
        const int w = e % WP;
        const int kv = e / WP;
        float val = 0.f;
        if (w < W) {
            if (tr) {
                const int k = kv / V, v = kv - k * V;
                val = a[(k * W + w) * V + v];
            } else {
                val = a[kv * W + w];
            }
        }
        As[e] = val;
    }
}

// kg_agg_reduce's optional epilogue (the backward pass of a discriminator block): the residual branch's input gradient
// `res` - given at every r_tstride-th frame and at the vertices r_inv picks - is added where it exists, and the result is
// multiplied by the LeakyReLU derivative of the block input (expressed on that activation's output `mask`):
//   out[n,c,t,w] = ( aggregate + [t % s == 0 and r_inv[w] >= 0] res[n,c,t/s,r_inv[w]] ) * lrelu'(mask[n,c,t,w])
// (a separate pass, kg_scatter_add_act, re-read and re-wrote the whole gradient for this: 0.10 ms per iteration).
// Frame index -> (n, t) by a magic multiply (host: agg_epi).
struct AggEpi { unsigned tmul, tshr, wmul, wshr; };

__device__ __forceinline__ int kg_divm(int x, unsigned mul, unsigned shr) {      // floor(x / d), d > 1 (mul == 0: d == 1)
    return mul ? (int)(__umulhi((unsigned)x, mul) >> shr) : x;
}

// the epilogue's operands of output element (n, c, t, w); iv = vertex of `res` that w reads (r_inv[w], or w, or -1).
// Issued BEFORE the aggregation arithmetic of the element (or, matrix-core form, of a group of elements) and applied
// after it: loaded one by one behind it, the mask / inv -> res latencies were the launch's critical path.
struct EpiVal { float r, m; };

__device__ __forceinline__ EpiVal agg_epi_load(const KgAggArgs& a, int c, int n, int t, int w, int iv) {
    EpiVal e = {0.f, 1.f};
    if (a.res) {
        int tb = t, rem = 0;
        if (a.r_tstride == 2) { tb = t >> 1; rem = t & 1; }
        else if (a.r_tstride > 2) { tb = t / a.r_tstride; rem = t - tb * a.r_tstride; }
        const bool hit = rem == 0 && tb < a.r_T && iv >= 0;
        const float r = a.res[(long)c * a.r_sC + (long)n * a.r_sN + (hit ? tb * a.r_V + iv : 0)];
        e.r = hit ? r : 0.f;
    }
    if (a.mask) e.m = a.mask[(long)c * a.m_sC + (long)n * a.m_sN + (long)t * a.W + w];
    return e;
}

__device__ __forceinline__ float agg_epi_apply(const KgAggArgs& a, float v, const EpiVal& e) {
    return (v + e.r) * (e.m > 0.f ? 1.f : a.slope);
}

// ---------------------------------------------------------------------------------------------
template <int K, int WP>
__global__ __launch_bounds__(NT) void kg_agg_expand_kernel(const KgAggArgs a, const AggEpi) {
    __shared__ __attribute__((aligned(16))) float As[K * 25 * WP];
    const int V = a.V, W = a.W, c = blockIdx.y;
    stage_A<K, WP>(As, a.a, V, W, a.a_transposed);
    __syncthreads();
    const int Tout = a.T * a.rep;
    const int nrows = a.N * Tout;
    const int row = blockIdx.x * NT + threadIdx.x;
    if (row >= nrows) return;
    const int n = row / Tout, tp = row - n * Tout;
    const float* xr = a.x + (long)c * a.x_sC + (long)n * a.x_sN + (long)(tp / a.rep) * V;
    float acc[K][WP];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int w = 0; w < WP; ++w) acc[k][w] = 0.f;
    for (int v = 0; v < V; ++v) {
        const float xv = xr[v];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float4* ar = reinterpret_cast<const float4*>(As + (k * V + v) * WP);
#pragma unroll
            for (int w4 = 0; w4 < WP / 4; ++w4) {
                const float4 av = ar[w4];
                acc[k][4 * w4 + 0] = fmaf(xv, av.x, acc[k][4 * w4 + 0]);
                acc[k][4 * w4 + 1] = fmaf(xv, av.y, acc[k][4 * w4 + 1]);
                acc[k][4 * w4 + 2] = fmaf(xv, av.z, acc[k][4 * w4 + 2]);
                acc[k][4 * w4 + 3] = fmaf(xv, av.w, acc[k][4 * w4 + 3]);
            }
        }
    }
    const long o = (long)n * a.o_sN + (long)tp * W;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float* op = a.out + (long)(k * a.C + c) * a.o_sC + o;
#pragma unroll
        for (int w = 0; w < WP; ++w)
            if (w < W) op[w] = acc[k][w];
    }
}

// ---------------------------------------------------------------------------------------------
template <int K, int WP>
__global__ __launch_bounds__(NT) void kg_agg_reduce_kernel(const KgAggArgs a, const AggEpi) {
    __shared__ __attribute__((aligned(16))) float As[K * 25 * WP];
    const int V = a.V, W = a.W, c = blockIdx.y;
    stage_A<K, WP>(As, a.a, V, W, a.a_transposed);
    __syncthreads();
    const int fold = a.rep;
    const int Tout = a.T;
    const int nrows = a.N * Tout;
    const int row = blockIdx.x * NT + threadIdx.x;
    if (row >= nrows) return;
    const int n = row / Tout, t = row - n * Tout;
    float acc[WP];
#pragma unroll
    for (int w = 0; w < WP; ++w) acc[w] = 0.f;
    const bool epi = a.res || a.mask;       // (uniform)
    EpiVal ev[WP];
    if (epi) {
#pragma unroll
        for (int w = 0; w < WP; ++w)
            ev[w] = w < W ? agg_epi_load(a, c, n, t, w, a.r_inv ? a.r_inv[w] : w) : EpiVal{0.f, 1.f};
    }
    for (int q = 0; q < fold; ++q) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float* yr = a.x + (long)(k * a.C + c) * a.x_sC + (long)n * a.x_sN + (long)(t * fold + q) * V;
            for (int v = 0; v < V; ++v) {
                const float yv = yr[v];
                const float4* ar = reinterpret_cast<const float4*>(As + (k * V + v) * WP);
#pragma unroll
                for (int w4 = 0; w4 < WP / 4; ++w4) {
                    const float4 av = ar[w4];
                    acc[4 * w4 + 0] = fmaf(yv, av.x, acc[4 * w4 + 0]);
                    acc[4 * w4 + 1] = fmaf(yv, av.y, acc[4 * w4 + 1]);
                    acc[4 * w4 + 2] = fmaf(yv, av.z, acc[4 * w4 + 2]);
                    acc[4 * w4 + 3] = fmaf(yv, av.w, acc[4 * w4 + 3]);
                }
            }
        }
    }
    float* op = a.out + (long)c * a.o_sC + (long)n * a.o_sN + (long)t * W;
    if (epi) {
#pragma unroll
        for (int w = 0; w < WP; ++w)
            if (w < W) acc[w] = agg_epi_apply(a, acc[w], ev[w]);
    }
#pragma unroll
    for (int w = 0; w < WP; ++w)
        if (w < W) op[w] = acc[w];
}

// ---------------------------------------------------------------------------------------------
// Stream variants of expand / reduce (channel-major planes: the frames of one channel are ONE contiguous run in
// the input and in the output).  The frame-per-thread kernels above read and write 44..100-byte rows at a stride
// of one row per lane - every load / store instruction touches ~20 cache lines - and reach 1.3-2.8 TB/s.  Here a
// workgroup copies its run of input frames to LDS with 128-bit loads, and a THREAD owns one output element
// (frame, w): consecutive threads write consecutive addresses (fully coalesced stores), the thread's A column
// A[k][:, w] lives in registers, and the input frame is read from LDS (the W threads of a frame read the same
// words: broadcast).
template <int K, int VM>
__global__ __launch_bounds__(NT) void kg_agg_expand_stream_kernel(const KgAggArgs a, int FO, const AggEpi) {
    extern __shared__ __attribute__((aligned(16))) float kg_asm[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int V = a.V, W = a.W, c = blockIdx.y, rep = a.rep;
    const long nrows = (long)a.N * a.T * rep;                 // output frames of one channel
    const long r0 = (long)blockIdx.x * FO;                    // FO is a multiple of 4 * rep
    const int fo = (int)(nrows - r0 < FO ? nrows - r0 : FO);
    const long x0 = r0 / rep;
    const int fx = (int)((r0 + fo - 1) / rep - x0) + 1;
    const int w = tid % W, fslot = tid / W, FPI = NT / W;

    float Areg[K][VM];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VM; ++v)
            Areg[k][v] = v < V ? (a.a_transposed ? a.a[(k * W + w) * V + v] : a.a[(k * V + v) * W + w]) : 0.f;

    // ---- the run of input frames -> LDS
    const long xoff = (long)c * a.x_sC + x0 * V;
    const long xext = (long)(a.C - 1) * a.x_sC + (long)a.N * a.T * V;         // end of the tensor
    long rem = (xext - xoff) * 4;
    rem = rem < 0 ? 0 : (rem > 0x7fffffffL ? 0x7fffffffL : rem);
    const __amdgpu_buffer_rsrc_t xd = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(a.x + xoff), 0, __builtin_amdgcn_readfirstlane((int)rem), 0x00020000);
    const int nfl = fx * V;
    for (int q = tid; 4 * q < nfl; q += NT)
        *reinterpret_cast<f4*>(kg_asm + 4 * q) =
            __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xd, (unsigned)(16 * q), 0, 0));
    __syncthreads();
    if (fslot >= FPI) return;

    float* op = a.out + (long)c * a.o_sC + r0 * W + w;
    const long kstep = (long)a.C * a.o_sC;
    for (int f = fslot; f < fo; f += FPI) {
        const float* xr = kg_asm + (rep == 1 ? f : f / rep) * V;
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
#pragma unroll
        for (int v = 0; v < VM; ++v) {
            if (v < V) {
                const float xv = xr[v];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = fmaf(xv, Areg[k][v], acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) op[k * kstep + (long)f * W] = acc[k];
    }
}

template <int K, int VM>
__global__ __launch_bounds__(NT) void kg_agg_reduce_stream_kernel(const KgAggArgs a, int FO, const AggEpi ep) {
    extern __shared__ __attribute__((aligned(16))) float kg_asm[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int V = a.V, W = a.W, c = blockIdx.y, fold = a.rep;
    const long nrows = (long)a.N * a.T;                       // output frames of one channel
    const long r0 = (long)blockIdx.x * FO;                    // FO is a multiple of 4
    const int fo = (int)(nrows - r0 < FO ? nrows - r0 : FO);
    const int w = tid % W, fslot = tid / W, FPI = NT / W;

    float Areg[K][VM];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < VM; ++v)
            Areg[k][v] = v < V ? (a.a_transposed ? a.a[(k * W + w) * V + v] : a.a[(k * V + v) * W + w]) : 0.f;

    // ---- K runs of input frames -> LDS  ys[k][fo * fold * V]
    const int nfl = fo * fold * V;
    const int pitch = FO * fold * V;                          // multiple of 4
    const long yext = (long)(K * a.C - 1) * a.x_sC + nrows * fold * V;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const long yoff = (long)(k * a.C + c) * a.x_sC + r0 * fold * V;
        long rem = (yext - yoff) * 4;
        rem = rem < 0 ? 0 : (rem > 0x7fffffffL ? 0x7fffffffL : rem);
        const __amdgpu_buffer_rsrc_t yd = __builtin_amdgcn_make_buffer_rsrc(
            kg_uniform_ptr(a.x + yoff), 0, __builtin_amdgcn_readfirstlane((int)rem), 0x00020000);
        for (int q = tid; 4 * q < nfl; q += NT)
            *reinterpret_cast<f4*>(kg_asm + k * pitch + 4 * q) =
                __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(yd, (unsigned)(16 * q), 0, 0));
    }
    __syncthreads();
    if (fslot >= FPI) return;

    float* op = a.out + (long)c * a.o_sC + r0 * W + w;
    const bool epi = a.res || a.mask;       // (uniform)
    const int iv = (a.res && a.r_inv) ? a.r_inv[w] : w;
    for (int f = fslot; f < fo; f += FPI) {
        EpiVal ev = {0.f, 1.f};
        if (epi) {
            const int row = (int)(r0 + f);
            const int n = kg_divm(row, ep.tmul, ep.tshr);
            ev = agg_epi_load(a, c, n, row - n * a.T, w, iv);
        }
        float acc = 0.f;
        for (int q = 0; q < fold; ++q) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float* yr = kg_asm + k * pitch + (f * fold + q) * V;
#pragma unroll
                for (int v = 0; v < VM; ++v)
                    if (v < V) acc = fmaf(yr[v], Areg[k][v], acc);
            }
        }
        if (epi) acc = agg_epi_apply(a, acc, ev);
        op[(long)f * W] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA form of expand / reduce (K = 3, rep = 1, channel-major planes).  Per frame the aggregation is a
// [1 x KI*V] x [KI*V x KO*W] product - 2*K*V*W flop for 4*(K+1)*V bytes, ~19 flop/B at V = W = 25 - and on the vector
// ALUs the kernels above are VALU / LDS-issue bound long before HBM (C5a: 1.0-1.9 TB/s).  Here a workgroup copies 128
// frames of its KI input planes to LDS with 128-bit loads, each wave contracts 32 frames on the matrix cores
// (v_mfma_f32_32x32x2_f32: i = frame, j = output column w, the adjacency - B operand - lives in registers), the
// result tile is transposed through LDS and leaves in 128-bit coalesced stores.
//   reduce: KI = 3, KO = 1 (contraction over (k, v));   expand: KI = 1, KO = 3 (three output planes).
// KS = k-steps of two contraction indices each the instantiation provides (>= ceil(KI*V / 2)).
constexpr int AG_F = 128;          // frames per sub-tile (4 waves x 32 frames); a tile is SUB of them

// SUB (runtime, 1..8) sub-tiles per tile: narrow frames (V = 5: 20 bytes) would otherwise give tiles of a few KB
// with two barriers each; SUB * V <= 32 keeps the staging at four 128-bit loads per thread and plane.
// EPI: the instantiation carries the residual / mask epilogue of the adjoint pass (reduce only).  As a run-time branch inside
// ONE instantiation (round 3, 5985f48) the epilogue's registers counted for every launch: the plain C5a reduction went
// from 0.733 to 0.836 ms (bisected in round 5 with tools/probe/agg_c5a_driver.cpp, profiles/r05_agg_bisect.log).
// RES / MSK: which of the two epilogue operands the launch has (RES or MSK <=> EPI).
template <int KI, int KO, int KS, bool RES = false, bool MSK = false>
__global__ __launch_bounds__(NT) void kg_agg_mfma_kernel(const KgAggArgs a, int ntiles, int tiles_per_c, int SUB, const AggEpi ep) {
    constexpr bool EPI = RES || MSK;
    extern __shared__ __attribute__((aligned(16))) float kg_gsm[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = a.V, W = a.W;
    const int Lc = KI * V, ksteps = (Lc + 1) / 2;
    const long nrows = (long)a.N * a.T;
    const int F = AG_F * SUB;                           // frames per tile
    float* const lin = kg_gsm;                          // [KI][F * V]
    float* const lout = kg_gsm + KI * F * V;            // [KO][F * W]
    const int kh = lane >> 5, l31 = lane & 31;
    __shared__ int inv_l[RES ? 32 : 1];                 // reduce epilogue: vertex of `res` that w reads (or -1)
    if (RES && tid < 32) inv_l[tid] = (a.r_inv && tid < W) ? a.r_inv[tid] : tid;

    // Contraction index 2 s + kh = k1 * V + v of k-step s, walked without divisions.
    // B operand: lane (k = kh, j = l31) holds A[k1][v][w = j] (expand: of output plane ko), zero beyond Lc / W;
    // A operand: lane (i = l31 -> frame 32 wave + l31 of the sub-tile, k = kh) reads lin[k1][frame * V + v].
    float breg[KS][KO];
    int aoff[KS];
    {
        int k1 = 0, v = kh;
        while (v >= V) { v -= V; ++k1; }
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const bool in = 2 * s_ + kh < Lc;
#pragma unroll
            for (int ko = 0; ko < KO; ++ko) {
                // unconditional load from a clamped index + select: a guarded load would make hipcc branch around
                // each of the KS*KO loads and pay their latencies one after the other
                const int kk = KI == 1 ? ko : k1;
                const bool ok = in && l31 < W;
                const int idx = ok ? (a.a_transposed ? (kk * W + l31) * V + v : (kk * V + v) * W + l31) : 0;
                const float val = a.a[idx];
                breg[s_][ko] = ok ? val : 0.f;
            }
            aoff[s_] = in ? (KI == 1 ? 0 : k1) * (F * V) + (32 * wave + l31) * V + v : 0;
            v += 2;
            while (v >= V) { v -= V; ++k1; }
        }
    }

    constexpr int INLP = 4;                               // 128-bit loads per thread and plane (SUB * V <= 32)
    f4 inreg[KI][INLP];
    const int in_f4 = F * V / 4;                          // per plane (F * V is a multiple of 4)
    auto issue = [&](int t) {
        const bool live = t < ntiles;
        const int tt = live ? t : 0;
        const int c = tt / tiles_per_c;
        const long r0 = (long)(tt - c * tiles_per_c) * F;
        const long left = nrows - r0;                                          // frames of the channel from r0 on
        const long rem = (left < F ? left : F) * V * 4;                        // bytes of this tile that exist
#pragma unroll
        for (int ki = 0; ki < KI; ++ki) {
            const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(
                kg_uniform_ptr(a.x + (long)(ki * a.C + c) * a.x_sC + r0 * V), 0,
                __builtin_amdgcn_readfirstlane((int)rem), 0x00020000);
#pragma unroll
            for (int i = 0; i < INLP; ++i) {
                const int q = tid + NT * i;
                const unsigned off = (live && q < in_f4) ? (unsigned)(16 * q) : 0x80000000u;
                inreg[ki][i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(d, off, 0, 0));
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int ki = 0; ki < KI; ++ki)
#pragma unroll
            for (int i = 0; i < INLP; ++i) {
                const int q = tid + NT * i;
                if (q < in_f4) *reinterpret_cast<f4*>(lin + ki * (F * V) + 4 * q) = inreg[ki][i];
            }
    };

    int t = blockIdx.x;
    issue(t);
    stash();
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
        issue(t + gridDim.x);                              // next tile in flight during the MFMAs
        const int c = t / tiles_per_c;
        const long r0 = (long)(t - c * tiles_per_c) * F;
        for (int sub = 0; sub < SUB; ++sub) {
            kg_f32x16 acc[KO];
#pragma unroll
            for (int ko = 0; ko < KO; ++ko)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ko][r] = 0.f;
            const float* lsub = lin + sub * (AG_F * V);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                if (s_ < ksteps) {
                    const float av = lsub[aoff[s_]];
#pragma unroll
                    for (int ko = 0; ko < KO; ++ko)
                        acc[ko] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, breg[s_][ko], acc[ko], 0, 0, 0);
                }
            }
            // C/D layout: col = l31 (= w), row = (r&3) + 8*(r>>2) + 4*kh (= frame inside the wave's 32) -> lout[ko][frame][w]
            if (l31 < W) {
#pragma unroll
                for (int ko = 0; ko < KO; ++ko)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        lout[ko * (F * W) + (sub * AG_F + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh) * W + l31] = acc[ko][r];
            }
        }
        __syncthreads();                                   // lin is free, lout is complete
        stash();                                           // next tile -> lin
        // coalesced 128-bit stores of the KO planes (frames past the end of the channel are dropped)
        const long left = nrows - r0;
        const int nfl = (int)((left < F ? left : F) * W);
        const int out_f4 = F * W / 4;
        if constexpr (EPI) {                           // reduce with the residual / mask epilogue
            // two 128-bit groups per thread and trip, every load of both (the mask as one 128-bit load where its rows
            // line up with the output's, the residual through the vertex table in LDS) issued before the first use:
            // one by one the four scalar mask loads and the inv -> res chains of a group made the D1 launch of the
            // critic's backward pass 74 us instead of 26 + 36 us for the separate scatter pass
            const bool m4ok = MSK && a.m_sN == (long)a.T * W && (a.m_sC & 3) == 0 && (((unsigned long long)a.mask) & 15ull) == 0;
            for (int q0 = tid; q0 < out_f4; q0 += 2 * NT) {
                f4 v4[2], mk[2];
                float rr[2][4];
                bool live[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int qq = q0 + u * NT;
                    live[u] = qq < out_f4;
                    const int e0 = live[u] ? 4 * qq : 0;
                    v4[u] = *reinterpret_cast<const f4*>(lout + e0);
                    int fr = kg_divm(e0, ep.wmul, ep.wshr), w = e0 - fr * W;           // frame inside the tile, vertex
                    const int row = (int)r0 + fr;
                    int n = kg_divm(row, ep.tmul, ep.tshr), t = row - n * a.T;
                    const long mbase = (long)c * a.m_sC + r0 * W + e0;
                    if (m4ok) mk[u] = *reinterpret_cast<const f4*>(a.mask + (e0 + 4 <= nfl ? mbase : (long)c * a.m_sC));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool in = e0 + e < nfl;                                   // (past the end of the channel: dropped below)
                        float r = 0.f;
                        if constexpr (RES) {
                            int tb = t, rem = 0;
                            if (a.r_tstride == 2) { tb = t >> 1; rem = t & 1; }
                            else if (a.r_tstride > 2) { tb = t / a.r_tstride; rem = t - tb * a.r_tstride; }
                            const int iv = inv_l[w];
                            const bool hit = in && rem == 0 && tb < a.r_T && iv >= 0;
                            r = a.res[(long)c * a.r_sC + (long)n * a.r_sN + (hit ? tb * a.r_V + iv : 0)];
                            r = hit ? r : 0.f;
                        }
                        rr[u][e] = r;
                        if (MSK && (!m4ok || e0 + 4 > nfl))
                            mk[u][e] = in ? a.mask[(long)c * a.m_sC + (long)n * a.m_sN + (long)t * W + w] : 1.f;
                        if (++w == W) { w = 0; if (++t == a.T) { t = 0; ++n; } }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!live[u]) continue;
                    const int e0 = 4 * (q0 + u * NT);
                    float* dst = a.out + (long)c * a.o_sC + r0 * W + e0;
                    f4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (v4[u][e] + rr[u][e]) * ((!MSK || mk[u][e] > 0.f) ? 1.f : a.slope);
                    if (e0 + 4 <= nfl) *reinterpret_cast<f4*>(dst) = o4;
                    else
                        for (int e = 0; e < 4; ++e)
                            if (e0 + e < nfl) dst[e] = o4[e];
                }
            }
        } else {
            for (int q = tid; q < KO * out_f4; q += NT) {
                const int ko = q / out_f4, qq = q - ko * out_f4;
                float* dst = a.out + (long)(ko * a.C + c) * a.o_sC + r0 * W + 4 * qq;
                const f4 v4 = *reinterpret_cast<const f4*>(lout + ko * (F * W) + 4 * qq);
                if (4 * qq + 4 <= nfl) *reinterpret_cast<f4*>(dst) = v4;
                else
                    for (int e = 0; e < 4; ++e)
                        if (4 * qq + e < nfl) dst[e] = v4[e];
            }
        }
        __syncthreads();                                   // lout is free, lin (next tile) is visible
    }
}

// ---------------------------------------------------------------------------------------------
// outer: every workgroup walks a strided list of (channel, row tile) units, keeps its share of
// the K*V*W outputs in registers, and writes one partial slab; kg_agg_outer_sum adds the slabs.
constexpr int OUT_R = 32;        // frames per unit
constexpr int OUT_PER_THREAD = 8;  // ceil(1875 / 256)

template <int K>
__global__ __launch_bounds__(NT) void kg_agg_outer_kernel(const KgAggArgs a, int nunits, int row_tiles) {
    __shared__ float xs[OUT_R * 25];
    __shared__ float ys[K * OUT_R * 25];
    const int tid = threadIdx.x;
    const int V = a.V, W = a.W;
    const int Tp = a.T * a.rep;
    const int nrows = a.N * Tp;
    const int nout = K * V * W;

    int ok[OUT_PER_THREAD], ov[OUT_PER_THREAD], ow[OUT_PER_THREAD];
    float acc[OUT_PER_THREAD];
#pragma unroll
    for (int i = 0; i < OUT_PER_THREAD; ++i) {
        int e = tid + i * NT;
        int ee = e < nout ? e : 0;
        ok[i] = ee / (V * W);
        int rem = ee - ok[i] * V * W;
        ov[i] = rem / W;
        ow[i] = rem - ov[i] * W;
        acc[i] = 0.f;
    }

    for (int u = blockIdx.x; u < nunits; u += gridDim.x) {
        const int c = u / row_tiles;
        const int row0 = (u - c * row_tiles) * OUT_R;
        __syncthreads();
        for (int e = tid; e < OUT_R * V; e += NT) {
            int rr = e / V, v = e - rr * V;
            const int row = row0 + rr;
            const int rc = row < nrows ? row : nrows - 1;      // clamped, unconditional load + select
            const int n = rc / Tp, tp = rc - n * Tp;
            const float val = a.x[(long)c * a.x_sC + (long)n * a.x_sN + (long)(tp / a.rep) * V + v];
            xs[e] = row < nrows ? val : 0.f;
        }
        for (int e = tid; e < K * OUT_R * W; e += NT) {
            int k = e / (OUT_R * W), rem = e - k * (OUT_R * W);
            int rr = rem / W, w = rem - rr * W;
            const int row = row0 + rr;
            const int rc = row < nrows ? row : nrows - 1;
            const int n = rc / Tp, tp = rc - n * Tp;
            const float val = a.y[(long)(k * a.C + c) * a.y_sC + (long)n * a.y_sN + (long)tp * W + w];
            ys[e] = row < nrows ? val : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < OUT_PER_THREAD; ++i) {
            if (tid + i * NT < nout) {
                float s = acc[i];
                const float* xp = xs + ov[i];
                const float* yp = ys + ok[i] * OUT_R * W + ow[i];
#pragma unroll 8
                for (int rr = 0; rr < OUT_R; ++rr) s = fmaf(xp[rr * V], yp[rr * W], s);
                acc[i] = s;
            }
        }
    }
    float* slab = a.ws + (long)blockIdx.x * nout;
#pragma unroll
    for (int i = 0; i < OUT_PER_THREAD; ++i) {
        int e = tid + i * NT;
        if (e < nout) slab[e] = acc[i];
    }
}

typedef float kg_f32x4 __attribute__((ext_vector_type(4)));

// Geometry of the MFMA kernel.  P = frame blocks packed into one 16x16 tile (i = (p, v), j = (p, w): only the
// diagonal blocks p == p' are kept), so one v_mfma_f32_16x16x4_f32 contracts 4 P frames.  F = frames per unit
// (multiple of 16 P): as many as two 128-bit loads per thread and stream, and the LDS budget, allow.
struct OuterGeom {
    int P, F;
};

// TV / TW: 16-wide tiles covering V / W (2 only when V / W > 16, then P = 1)
template <int K, int TV, int TW>
__device__ __forceinline__ void outer_mfma_body(float* const kg_osm, const KgAggArgs& a, const int nunits, const int chunks,
                                                const OuterGeom gm, const int bid, const int nblocks) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int V = a.V, W = a.W;
    const int P = gm.P, F = gm.F;
    const int nrows = a.N * a.T;
    const int xs_f = F * V, ys_f = F * W;                    // floats per unit: x, one y channel
    const int stage = xs_f + K * ys_f;
    const long x_extent = (long)(a.C - 1) * a.x_sC + (long)nrows * V;
    const long y_extent = (long)(K * a.C - 1) * a.y_sC + (long)nrows * W;

    kg_f32x4 acc[K][TV][TW];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < TV; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[k][i][j] = kg_f32x4{0.f, 0.f, 0.f, 0.f};

    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 xr[2], yr[K][2];
    // unit u -> registers (u >= nunits: every offset out of range)
    auto issue = [&](int u) {
        const bool live = u < nunits;
        const int uu = live ? u : 0;
        const int c = uu / chunks;
        const long row0 = (long)(uu - c * chunks) * F;
        auto clampb = [](long fl) { long b = fl * 4; return (int)(b < 0 ? 0 : (b > 0x7fffffffL ? 0x7fffffffL : b)); };
        const long xo = (long)c * a.x_sC + row0 * V;
        const __amdgpu_buffer_rsrc_t xd = __builtin_amdgcn_make_buffer_rsrc(
            kg_uniform_ptr(a.x + xo), 0, __builtin_amdgcn_readfirstlane(clampb(x_extent - xo)), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + NT * i;
            const unsigned off = (live && 4 * q < xs_f) ? (unsigned)(16 * q) : 0x80000000u;
            xr[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xd, off, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long yo = (long)(k * a.C + c) * a.y_sC + row0 * W;
            const __amdgpu_buffer_rsrc_t yd = __builtin_amdgcn_make_buffer_rsrc(
                kg_uniform_ptr(a.y + yo), 0, __builtin_amdgcn_readfirstlane(clampb(y_extent - yo)), 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + NT * i;
                const unsigned off = (live && 4 * q < ys_f) ? (unsigned)(16 * q) : 0x80000000u;
                yr[k][i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(yd, off, 0, 0));
            }
        }
    };
    auto stash = [&](int b) {
        float* base = kg_osm + b * stage;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + NT * i;
            if (4 * q < xs_f) *reinterpret_cast<f4*>(base + 4 * q) = xr[i];
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = tid + NT * i;
                if (4 * q < ys_f) *reinterpret_cast<f4*>(base + xs_f + k * ys_f + 4 * q) = yr[k][i];
            }
    };
    // operand lanes: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]; tile position 16 t + (lane&15) is
    // (frame block p, vertex) = (pos / V, pos % V); the lane's frame in group g is 4 P g + 4 p + k
    const int l15 = lane & 15, l4 = lane >> 4;
    int xo_[TV], yo_[TW];            // LDS offset of the lane's element in group 0, or -1
    int xf_[TV], yf_[TW];            // its frame inside the group
#pragma unroll
    for (int i = 0; i < TV; ++i) {
        const int pos = 16 * i + l15, pp = pos / V;
        xf_[i] = 4 * pp + l4;
        xo_[i] = pp < P ? xf_[i] * V + (pos - pp * V) : -1;
    }
#pragma unroll
    for (int j = 0; j < TW; ++j) {
        const int pos = 16 * j + l15, pp = pos / W;
        yf_[j] = 4 * pp + l4;
        yo_[j] = pp < P ? yf_[j] * W + (pos - pp * W) : -1;
    }
    const int ngroups = F / (4 * P);
    auto compute = [&](int b, int u) {
        const float* base = kg_osm + b * stage;
        const int c = u / chunks;
        const int left = nrows - (u - c * chunks) * F;       // frames of this unit that exist
        for (int g = wave; g < ngroups; g += 4) {
            const int f0 = 4 * P * g;
            float av[TV], bv[K][TW];
#pragma unroll
            for (int i = 0; i < TV; ++i) {
                const bool ok = xo_[i] >= 0 && f0 + xf_[i] < left;            // frames past the end: next channel
                const float t = base[f0 * V + (xo_[i] >= 0 ? xo_[i] : 0)];
                av[i] = ok ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const bool ok = yo_[j] >= 0 && f0 + yf_[j] < left;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float t = base[xs_f + k * ys_f + f0 * W + (yo_[j] >= 0 ? yo_[j] : 0)];
                    bv[k][j] = ok ? t : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int i = 0; i < TV; ++i)
#pragma unroll
                    for (int j = 0; j < TW; ++j)
                        acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[k][j], acc[k][i][j], 0, 0, 0);
        }
    };

    int u = bid;
    issue(u);
    stash(0);
    __syncthreads();
    int b = 0;
    for (; u < nunits; u += nblocks, b ^= 1) {
        issue(u + nblocks);
        compute(b, u);
        stash(b ^ 1);
        __syncthreads();
    }
    // ---- the four waves' partial sums meet in LDS.  C/D layout: col = lane&15, row = 4*(lane>>4) + r
    constexpr int RV = 16 * TV, RW = 16 * TW;
    float* red = kg_osm;                                       // [4][K][RV][RW]
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < TV; ++i)
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[((wave * K + k) * RV + 16 * i + 4 * l4 + r) * RW + 16 * j + l15] = acc[k][i][j][r];
    __syncthreads();
    const int nout = K * V * W;
    float* slab = a.ws + (long)bid * nout;
    for (int e = tid; e < nout; e += NT) {
        const int k = e / (V * W);
        const int rem = e - k * V * W;
        const int v = rem / W, w = rem - v * W;
        float s = 0.f;
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int q = 0; q < 4; ++q) s += red[((q * K + k) * RV + pp * V + v) * RW + pp * W + w];
        slab[e] = s;
    }
}

template <int K, int TV, int TW>
__global__ __launch_bounds__(NT) void kg_agg_outer_mfma_kernel(const KgAggArgs a, int nunits, int chunks, OuterGeom gm) {
    extern __shared__ float kg_osm[];
    outer_mfma_body<K, TV, TW>(kg_osm, a, nunits, chunks, gm, blockIdx.x, gridDim.x);
}

// The adjacency gradients of all blocks of a backward pass (6 in D, 7 in G; 5-20 us each, mostly launch latency) in
// ONE launch: every job keeps its own slab count and geometry, the workgroups of the launch are dealt to the jobs
// by a prefix table.
constexpr int OUTER_MANY_MAX = 8;
struct OuterManyJob { KgAggArgs a; int nunits, chunks, wg_begin, nwg, variant; OuterGeom gm; };
struct OuterMany { int njobs; OuterManyJob job[OUTER_MANY_MAX]; };

__global__ __launch_bounds__(NT) void kg_agg_outer_many_kernel(const OuterMany m) {
    extern __shared__ float kg_osm[];
    kg_kernarg_warm<(int)sizeof(OuterMany)>();       // (the job search reads one line per job, each behind the other)
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.job[ji + 1].wg_begin) ++ji;      // (uniform)
    const OuterManyJob& j = m.job[ji];
    const int bid = blockIdx.x - j.wg_begin;
    switch (j.variant) {                                                            // (uniform) K, TV, TW
        case 0: outer_mfma_body<1, 1, 1>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
        case 1: outer_mfma_body<1, 1, 2>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
        case 2: outer_mfma_body<1, 2, 1>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
        case 3: outer_mfma_body<1, 2, 2>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
        case 4: outer_mfma_body<3, 1, 1>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
        case 5: outer_mfma_body<3, 1, 2>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
        case 6: outer_mfma_body<3, 2, 1>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
        default: outer_mfma_body<3, 2, 2>(kg_osm, j.a, j.nunits, j.chunks, j.gm, bid, j.nwg); break;
    }
}

__global__ __launch_bounds__(256) void kg_agg_outer_sum(const float* ws, float* out, int nout, int slabs) {
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = kg_slab_sum_256(ws, nout, e, e < nout, slabs);
    if (e < nout && threadIdx.x < 64) out[e] = s;
}

__global__ __launch_bounds__(256) void kg_agg_outer_sum_many_kernel(const KgOuterSumJobs js) {
    const KgOuterSumJob& j = js.job[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.nout) return;                 // (uniform) grid.x covers the largest job (<= 30 blocks)
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = kg_slab_sum_256(j.ws, j.nout, e, e < j.nout, j.slabs);
    if (e < j.nout && threadIdx.x < 64) j.out[e] = s;
}

int validate(const KgAggArgs* a, const char* who) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->rep >= 1, "%s: bad dims", who);
    KG_REQUIRE(a->C <= 65535, "%s: C=%d too large", who, a->C);
    KG_REQUIRE(a->K == 1 || a->K == 3, "%s: K=%d (1 or 3)", who, a->K);
    KG_REQUIRE(a->V >= 1 && a->V <= 25 && a->W >= 1 && a->W <= 25, "%s: V=%d W=%d (1..25)", who, a->V, a->W);
    KG_REQUIRE((long)a->N * a->T * a->rep < (1L << 30), "%s: too many frames", who);
    KG_REQUIRE(a->a && a->x && a->out, "%s: null pointer", who);
    return 0;
}

int outer_slabs(const KgAggArgs* a, int* nunits, int* row_tiles) {
    *row_tiles = kg_cdiv((long)a->N * a->T * a->rep, OUT_R);
    long u = (long)a->C * *row_tiles;
    *nunits = (int)u;
    return (int)(u < 512 ? u : 512);
}

// W padded to a multiple of 4: 4, 8, 16 or 28
#define KG_AGG_DISPATCH(KERNEL, grid)                                                                  \
    do {                                                                                               \
        const int wp = a->W <= 4 ? 4 : (a->W <= 8 ? 8 : (a->W <= 16 ? 16 : 28));                       \
        if (a->K == 3) {                                                                               \
            if (wp == 4) hipLaunchKernelGGL((KERNEL<3, 4>), grid, dim3(NT), 0, s, *a, ep);                 \
            else if (wp == 8) hipLaunchKernelGGL((KERNEL<3, 8>), grid, dim3(NT), 0, s, *a, ep);            \
            else if (wp == 16) hipLaunchKernelGGL((KERNEL<3, 16>), grid, dim3(NT), 0, s, *a, ep);          \
            else hipLaunchKernelGGL((KERNEL<3, 28>), grid, dim3(NT), 0, s, *a, ep);                        \
        } else {                                                                                       \
            if (wp == 4) hipLaunchKernelGGL((KERNEL<1, 4>), grid, dim3(NT), 0, s, *a, ep);                 \
            else if (wp == 8) hipLaunchKernelGGL((KERNEL<1, 8>), grid, dim3(NT), 0, s, *a, ep);            \
            else if (wp == 16) hipLaunchKernelGGL((KERNEL<1, 16>), grid, dim3(NT), 0, s, *a, ep);          \
            else hipLaunchKernelGGL((KERNEL<1, 28>), grid, dim3(NT), 0, s, *a, ep);                        \
        }                                                                                              \
    } while (0)

}  // namespace

// frames per workgroup of the stream kernels: ~8 passes of the NT / W frame slots, a multiple of `mult`, inside the
// LDS budget (per_frame floats per output frame)
static int stream_frames(int W, int mult, int per_frame, int lds_floats) {
    const int fpi = NT / W;
    int fo = 8 * fpi;
    const int cap = lds_floats / per_frame;
    if (fo > cap) fo = cap;
    fo = fo / mult * mult;
    return fo < mult ? mult : fo;
}

#define KG_AGG_STREAM_GO(KERNEL, K_)                                                                       \
    do {                                                                                                   \
        if (a->V <= 4)       hipLaunchKernelGGL((KERNEL<K_, 4>), grid, dim3(NT), lds, s, *a, fo, ep);         \
        else if (a->V <= 8)  hipLaunchKernelGGL((KERNEL<K_, 8>), grid, dim3(NT), lds, s, *a, fo, ep);         \
        else if (a->V <= 16) hipLaunchKernelGGL((KERNEL<K_, 16>), grid, dim3(NT), lds, s, *a, fo, ep);        \
        else                 hipLaunchKernelGGL((KERNEL<K_, 25>), grid, dim3(NT), lds, s, *a, fo, ep);        \
    } while (0)

// KG_AGG_STREAM: "0" frame-per-thread kernels only, "1" stream kernels wherever the layout allows (tests),
// unset: stream kernels where they were measured faster (`heuristic`)
static bool agg_stream_wanted(bool heuristic) {
    const int v = kg_env().agg_stream;
    return v < 0 ? heuristic : v == 1;
}

// KG_AGG_MFMA: "0" never, "1" wherever the launch allows (tests), unset: where measured faster (`heuristic`)
static bool agg_mfma_wanted(bool heuristic) {
    const int v = kg_env().agg_mfma;
    return v < 0 ? heuristic : v == 1;
}

// expand (KI = 1, KO = 3) / reduce (KI = 3, KO = 1) on the matrix cores; returns false when the launch is not eligible
template <int KI, int KO>
static bool agg_mfma_launch(const KgAggArgs* a, hipStream_t s, int* rc, const AggEpi& ep) {
    const long nrows = (long)a->N * a->T;
    const int lc = KI * a->V;
    if (a->K != 3 || a->rep != 1 || a->V > 25 || a->W > 25) return false;
    if (a->N > 1 && (a->x_sN != (long)a->T * a->V || a->o_sN != (long)a->T * a->W)) return false;
    // sub-tiles per tile: as many as four 128-bit loads per thread and plane (SUB * V <= 32) and 60 KB of LDS allow,
    // but no more than keeps >= 1024 tiles in the launch
    int sub = 32 / a->V;
    const int per128 = AG_F * (KI * a->V + KO * a->W) * (int)sizeof(float);
    if (sub > 61440 / per128) sub = 61440 / per128;
    if (sub > 8) sub = 8;
    while (sub > 1 && (long)kg_cdiv(nrows, AG_F * sub) * a->C < 1024) --sub;
    if (sub < 1) sub = 1;
    if (const int e = kg_env().agg_mfma_sub) sub = e >= 1 && e * a->V <= 32 && e * per128 <= 61440 ? e : sub;
    const int tiles_per_c = kg_cdiv(nrows, AG_F * sub);
    const long ntiles = (long)tiles_per_c * a->C;
    if (ntiles > (1L << 30)) return false;
    // persistent grid: two workgroups per CU (tools/sweep_agg_train.py, profiles/r03_agg_grid_sweep.log: D1 / D3 adjoint
    // aggregation at 192 samples 46.7 us the pair with 512 workgroups, 51.0 with 1024, 52.3 with 768, 56.7 with 256; the
    // C5a launch 0.852 / 0.861 / 0.938 / 1.246 ms) - with more, the surplus starts when the first ones finish and the
    // launch ends on a ragged second round
    // (round 5: the narrow reduce instantiation - KS = 8, 156 VGPRs: three workgroups fit a CU - takes three per CU: 3.285 ->
    // 3.278 ms per iteration, profiles/r05_epilogue_instantiations_ab.log; 640 is worse, the wider instantiations stay at two)
    const int ks = (lc + 1) / 2;
    int cap = (KI == 3 && ks <= 8) ? 768 : 512;
    if (const int e = kg_env().agg_mfma_grid) cap = e > 0 ? e : cap;      // tuning hook
    const int grid = (int)(ntiles < cap ? ntiles : cap);
    const size_t lds = (size_t)sub * per128;
    const bool hr = KO == 1 && a->res != nullptr, hm = KO == 1 && a->mask != nullptr;
#define KG_AGM_GO(KS_) do { \
        if (KO == 1 && hr && hm) hipLaunchKernelGGL((kg_agg_mfma_kernel<KI, KO, KS_, KO == 1, KO == 1>), dim3(grid), dim3(NT), lds, s, *a, (int)ntiles, tiles_per_c, sub, ep); \
        else if (KO == 1 && hr)  hipLaunchKernelGGL((kg_agg_mfma_kernel<KI, KO, KS_, KO == 1, false>), dim3(grid), dim3(NT), lds, s, *a, (int)ntiles, tiles_per_c, sub, ep); \
        else if (KO == 1 && hm)  hipLaunchKernelGGL((kg_agg_mfma_kernel<KI, KO, KS_, false, KO == 1>), dim3(grid), dim3(NT), lds, s, *a, (int)ntiles, tiles_per_c, sub, ep); \
        else                     hipLaunchKernelGGL((kg_agg_mfma_kernel<KI, KO, KS_, false, false>), dim3(grid), dim3(NT), lds, s, *a, (int)ntiles, tiles_per_c, sub, ep); } while (0)
    if (KI == 3) {
        if (ks <= 8) KG_AGM_GO(8); else if (ks <= 17) KG_AGM_GO(17); else KG_AGM_GO(38);
    } else {
        if (ks <= 3) KG_AGM_GO(3); else if (ks <= 6) KG_AGM_GO(6); else KG_AGM_GO(13);
    }
#undef KG_AGM_GO
    *rc = kg_launch_status(KI == 3 ? "kg_agg_reduce (mfma)" : "kg_agg_expand (mfma)");
    return true;
}

static AggEpi agg_epi(const KgAggArgs* a) {
    auto magic = [](unsigned d, unsigned& mul, unsigned& shr) {      // floor(x / d) == umulhi(x, mul) >> shr, x < 2^31
        mul = 0; shr = 0;
        if (d <= 1) return;
        unsigned lg = 0;
        while ((1u << lg) < d) ++lg;
        const unsigned p = 31 + lg;
        mul = (unsigned)(((1ull << p) + d - 1) / d);
        shr = p - 32;
    };
    AggEpi ep;
    magic((unsigned)a->T, ep.tmul, ep.tshr);
    magic((unsigned)a->W, ep.wmul, ep.wshr);
    return ep;
}

extern "C" int kg_agg_expand(const KgAggArgs* a, void* stream) {
    if (int rc = validate(a, "kg_agg_expand")) return rc;
    KG_REQUIRE(a->res == nullptr && a->mask == nullptr, "kg_agg_expand: the res / mask epilogue belongs to kg_agg_reduce");
    const AggEpi ep = agg_epi(a);
    const long nrows = (long)a->N * a->T * a->rep;
    hipStream_t s = (hipStream_t)stream;
    // measured (profiles/r01_v12_time_agg.log): the matrix-core kernel wins for wide frames (V*W >= 200: 1.7x at
    // V=25/W=11, 3-5x at V=W=25); at V=W=11 the stream kernel below is still ahead for expand
    // (3-channel planes of 64 samples - the generator's last block, D0's input gradient: 12 k frames - take it too:
    // 30.4 -> 9.5 us, tools/time_agg_c3.py; one thread per frame puts 48 workgroups on the chip there)
    if (agg_mfma_wanted(a->V * a->W >= 200 && nrows * a->C >= (1L << 13))) {
        int rc = 0;
        if (agg_mfma_launch<1, 3>(a, s, &rc, ep)) return rc;
    }
    const bool streams = a->N == 1 || (a->x_sN == (long)a->T * a->V && a->o_sN == (long)a->T * a->rep * a->W);
    // measured (profiles/r01_v9_time_agg.log): the stream kernel wins 1.1-2.2x when a frame has >= 4 output columns
    // and the launch is big enough to fill the chip; tiny launches / W < 4 stay with one thread per frame
    if (streams && a->rep <= 64 && agg_stream_wanted(a->W >= 4 && nrows * a->C >= (1L << 17))) {
        const int fo = stream_frames(a->W, 4 * a->rep, a->V, 8192 * a->rep - 2 * a->V * a->rep);
        const size_t lds = (size_t)((fo / a->rep + 2) * a->V + 4) * sizeof(float);
        dim3 grid(kg_cdiv(nrows, fo), a->C);
        if (a->K == 3) KG_AGG_STREAM_GO(kg_agg_expand_stream_kernel, 3);
        else           KG_AGG_STREAM_GO(kg_agg_expand_stream_kernel, 1);
        return kg_launch_status("kg_agg_expand (stream)");
    }
    dim3 grid(kg_cdiv(nrows, NT), a->C);
    KG_AGG_DISPATCH(kg_agg_expand_kernel, grid);
    return kg_launch_status("kg_agg_expand");
}

extern "C" int kg_agg_reduce(const KgAggArgs* a, void* stream) {
    if (int rc = validate(a, "kg_agg_reduce")) return rc;
    KG_REQUIRE((a->res == nullptr && a->mask == nullptr) || a->rep == 1, "kg_agg_reduce: res / mask epilogue with fold=%d", a->rep);
    KG_REQUIRE(a->res == nullptr || (a->r_T > 0 && a->r_V > 0 && a->r_tstride >= 1 && (a->r_inv != nullptr || a->r_V == a->W)),
               "kg_agg_reduce: bad residual geometry (r_T=%d r_V=%d r_tstride=%d)", a->r_T, a->r_V, a->r_tstride);
    const AggEpi ep = agg_epi(a);
    const long nrows = (long)a->N * a->T;
    hipStream_t s = (hipStream_t)stream;
    // reduce contracts K*V values per output: the matrix cores win from V*W >= 50 on (profiles/r01_v12_time_agg.log)
    if (agg_mfma_wanted(a->V * a->W >= 50 && nrows * a->C >= (1L << 13))) {        // (21.6 -> 11.2 us at 3 x 4096 frames)
        int rc = 0;
        if (agg_mfma_launch<3, 1>(a, s, &rc, ep)) return rc;
    }
    const bool streams = a->N == 1 || (a->x_sN == (long)a->T * a->rep * a->V && a->o_sN == (long)a->T * a->W);
    const int per_frame = a->K * a->rep * a->V;
    // the stream reduce pays one LDS read per FMA and only wins when the output frame is much wider than the input
    // frame (coalesced stores dominate): W >= 2 V
    if (streams && 4 * per_frame <= 12288 && agg_stream_wanted(a->W >= 2 * a->V && nrows * a->C >= (1L << 17))) {
        const int fo = stream_frames(a->W, 4, per_frame, 12288);
        const size_t lds = (size_t)(fo * per_frame + 4) * sizeof(float);
        dim3 grid(kg_cdiv(nrows, fo), a->C);
        if (a->K == 3) KG_AGG_STREAM_GO(kg_agg_reduce_stream_kernel, 3);
        else           KG_AGG_STREAM_GO(kg_agg_reduce_stream_kernel, 1);
        return kg_launch_status("kg_agg_reduce (stream)");
    }
    dim3 grid(kg_cdiv(nrows, NT), a->C);
    KG_AGG_DISPATCH(kg_agg_reduce_kernel, grid);
    return kg_launch_status("kg_agg_reduce");
}

// stream-contiguous launch: the frames of one channel follow each other without a gap in x and in y
static bool outer_streams(const KgAggArgs* a) {
    if (a->rep != 1) return false;
    if (a->N > 1 && (a->x_sN != (long)a->T * a->V || a->y_sN != (long)a->T * a->W)) return false;
    const long xe = (long)a->C * a->x_sC, ye = (long)a->K * a->C * a->y_sC;
    return a->x_sC >= 0 && a->y_sC >= 0 && xe < (1L << 40) && ye < (1L << 40);
}

static OuterGeom outer_geom(const KgAggArgs* a) {
    OuterGeom g;
    const int mx = a->V > a->W ? a->V : a->W;
    g.P = mx <= 16 ? 16 / mx : 1;
    const long nrows = (long)a->N * a->T;
    long fmax = 2048 / mx;                                        // two 128-bit loads per thread and stream
    const long lds = 7680 / (a->V + a->K * a->W);                 // 2 x 30 KB of LDS
    if (lds < fmax) fmax = lds;
    const int q = 16 * g.P;
    long f = fmax / q * q;
    if (f < q) f = q;
    const long need = (nrows + q - 1) / q * q;                    // no point in units longer than a channel
    if (f > need) f = need;
    g.F = (int)f;
    return g;
}

static size_t outer_mfma_lds(const KgAggArgs* a, const OuterGeom& g) {
    const int tv = a->V > 16 ? 2 : 1, tw = a->W > 16 ? 2 : 1;
    const size_t stage = (size_t)g.F * (a->V + a->K * a->W);
    const size_t red = (size_t)4 * a->K * 16 * tv * 16 * tw;
    return (2 * stage > red ? 2 * stage : red) * sizeof(float);
}

extern "C" int64_t kg_agg_outer_workspace_bytes(const KgAggArgs* a) {
    if (a == nullptr || a->C <= 0 || a->N <= 0 || a->T <= 0 || a->rep < 1) return -1;
    return (int64_t)512 * a->K * a->V * a->W * (int64_t)sizeof(float);      // at most 512 slabs on either path
}

static int outer_slab_count(const KgAggArgs* a) {
    int nunits, row_tiles;
    int slabs = outer_slabs(a, &nunits, &row_tiles);
    if (outer_streams(a) && kg_env().agg_outer_mfma != 0) {
        const OuterGeom gm = outer_geom(a);
        const long units = (long)a->C * kg_cdiv((long)a->N * a->T, gm.F);
        slabs = (int)(units < 512 ? units : 512);
    }
    return slabs;
}

extern "C" int kg_agg_outer_slabs(const KgAggArgs* a) {
    if (a == nullptr || a->N <= 0 || a->C <= 0 || a->T <= 0 || a->rep < 1 || (a->K != 1 && a->K != 3) || a->V < 1 ||
        a->V > 25 || a->W < 1 || a->W > 25) {
        kg_set_error("kg_agg_outer_slabs: bad arguments");
        return -1;
    }
    return outer_slab_count(a);
}

extern "C" int kg_agg_outer_sum_many(const KgOuterSumJobs* jobs, void* stream) {
    KG_REQUIRE(jobs != nullptr && jobs->njobs >= 1 && jobs->njobs <= KG_OUTER_SUM_MAX_JOBS, "kg_agg_outer_sum_many: njobs");
    int mx = 0;
    for (int i = 0; i < jobs->njobs; ++i) {
        const KgOuterSumJob& j = jobs->job[i];
        KG_REQUIRE(j.ws && j.out && j.nout >= 1 && j.slabs >= 1, "kg_agg_outer_sum_many: job %d is malformed", i);
        if (j.nout > mx) mx = j.nout;
    }
    hipLaunchKernelGGL(kg_agg_outer_sum_many_kernel, dim3(kg_cdiv(mx, 64), jobs->njobs), dim3(256), 0, (hipStream_t)stream, *jobs);
    return kg_launch_status("kg_agg_outer_sum_many");
}

extern "C" int kg_agg_outer(const KgAggArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr, "kg_agg_outer: null args");
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->rep >= 1, "kg_agg_outer: bad dims");
    KG_REQUIRE(a->K == 1 || a->K == 3, "kg_agg_outer: K=%d", a->K);
    KG_REQUIRE(a->V >= 1 && a->V <= 25 && a->W >= 1 && a->W <= 25, "kg_agg_outer: V=%d W=%d", a->V, a->W);
    KG_REQUIRE(a->x && a->y && a->out && a->ws, "kg_agg_outer: null pointer");
    int nunits, row_tiles;
    int slabs = outer_slabs(a, &nunits, &row_tiles);
    const int nout = a->K * a->V * a->W;
    KG_REQUIRE(a->ws_bytes >= (int64_t)512 * nout * 4, "kg_agg_outer: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (outer_streams(a) && kg_env().agg_outer_mfma != 0) {    // KG_AGG_OUTER_MFMA=0: element-wise kernel only (tests)
        const OuterGeom gm = outer_geom(a);
        const int chunks = kg_cdiv((long)a->N * a->T, gm.F);
        const long units = (long)a->C * chunks;
        slabs = (int)(units < 512 ? units : 512);
        const size_t lds = outer_mfma_lds(a, gm);
        KG_REQUIRE(lds <= 65536, "kg_agg_outer: LDS budget exceeded (%ld bytes)", (long)lds);
        const int tv = a->V > 16 ? 2 : 1, tw = a->W > 16 ? 2 : 1;
#define KG_OUTER_GO(K_, TV_, TW_) \
    hipLaunchKernelGGL((kg_agg_outer_mfma_kernel<K_, TV_, TW_>), dim3(slabs), dim3(NT), lds, s, *a, (int)units, chunks, gm)
        if (a->K == 3) {
            if (tv == 1) { if (tw == 1) KG_OUTER_GO(3, 1, 1); else KG_OUTER_GO(3, 1, 2); }
            else         { if (tw == 1) KG_OUTER_GO(3, 2, 1); else KG_OUTER_GO(3, 2, 2); }
        } else {
            if (tv == 1) { if (tw == 1) KG_OUTER_GO(1, 1, 1); else KG_OUTER_GO(1, 1, 2); }
            else         { if (tw == 1) KG_OUTER_GO(1, 2, 1); else KG_OUTER_GO(1, 2, 2); }
        }
#undef KG_OUTER_GO
        if (int rc = kg_launch_status("kg_agg_outer (mfma)")) return rc;
        if (a->defer_sum) return 0;
        hipLaunchKernelGGL(kg_agg_outer_sum, dim3(kg_cdiv(nout, 64)), dim3(256), 0, s, a->ws, a->out, nout, slabs);
        return kg_launch_status("kg_agg_outer_sum");
    }
    if (a->K == 3) hipLaunchKernelGGL(kg_agg_outer_kernel<3>, dim3(slabs), dim3(NT), 0, s, *a, nunits, row_tiles);
    else           hipLaunchKernelGGL(kg_agg_outer_kernel<1>, dim3(slabs), dim3(NT), 0, s, *a, nunits, row_tiles);
    if (int rc = kg_launch_status("kg_agg_outer")) return rc;
    if (a->defer_sum) return 0;
    hipLaunchKernelGGL(kg_agg_outer_sum, dim3(kg_cdiv(nout, 64)), dim3(256), 0, s, a->ws, a->out, nout, slabs);
    return kg_launch_status("kg_agg_outer_sum");
}

// Several adjacency gradients at once: the jobs that take the matrix-core kernel share ONE launch, the others are
// launched one by one, and all slab sums finish in one kg_agg_outer_sum_many launch per 16 jobs.  Every job brings its
// own workspace (kg_agg_outer_workspace_bytes) and destination.
extern "C" int kg_agg_outer_many(const KgAggArgs* jobs, int32_t njobs, void* stream) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1, "kg_agg_outer_many: no jobs");
    hipStream_t s = (hipStream_t)stream;
    OuterMany m;
    m.njobs = 0;
    int wgs = 0;
    size_t lds = 0;
    auto flush = [&]() -> int {
        if (m.njobs == 0) return 0;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)kg_agg_outer_many_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            attr_set = true;
        }
        hipLaunchKernelGGL(kg_agg_outer_many_kernel, dim3(wgs), dim3(NT), lds, s, m);
        m.njobs = 0;
        wgs = 0;
        lds = 0;
        return kg_launch_status("kg_agg_outer_many");
    };
    KgOuterSumJobs sums;
    sums.njobs = 0;
    auto flush_sums = [&]() -> int {
        if (sums.njobs == 0) return 0;
        if (int rc = flush()) return rc;                       // the slabs of these jobs must have been enqueued
        const int rc = kg_agg_outer_sum_many(&sums, stream);
        sums.njobs = 0;
        return rc;
    };
    // ONE budget of workgroups for the whole launch - two per CU, what is resident at once (60 KB of LDS each) - dealt to
    // the jobs in proportion to their units.  With up to 512 workgroups per JOB (rounds 1-2) the six-block launch of the
    // critic's backward pass had 2560: five dispatch rounds, a slab written and summed per workgroup
    // (profiles/r03_outer_budget.log: 3.751 -> 3.717 ms per iteration with 512, 3.714 with 768, 3.733 with 1536).
    const int budget = kg_env().agg_outer_budget > 0 ? kg_env().agg_outer_budget : 512;
    double total_units = 0.0;
    if (budget > 0)
        for (int i = 0; i < njobs; ++i) {
            const KgAggArgs* a = &jobs[i];
            if (a->N > 0 && a->C > 0 && a->T > 0 && a->V >= 1 && a->V <= 25 && a->W >= 1 && a->W <= 25 && outer_streams(a) &&
                kg_env().agg_outer_mfma != 0)
                total_units += (double)a->C * kg_cdiv((long)a->N * a->T, outer_geom(a).F);
        }
    for (int i = 0; i < njobs; ++i) {
        const KgAggArgs* a = &jobs[i];
        KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->rep >= 1, "kg_agg_outer_many: job %d bad dims", i);
        KG_REQUIRE(a->K == 1 || a->K == 3, "kg_agg_outer_many: job %d K=%d", i, a->K);
        KG_REQUIRE(a->V >= 1 && a->V <= 25 && a->W >= 1 && a->W <= 25, "kg_agg_outer_many: job %d V=%d W=%d", i, a->V, a->W);
        KG_REQUIRE(a->x && a->y && a->out && a->ws, "kg_agg_outer_many: job %d null pointer", i);
        const int nout = a->K * a->V * a->W;
        KG_REQUIRE(a->ws_bytes >= (int64_t)512 * nout * 4, "kg_agg_outer_many: job %d workspace too small", i);
        for (int k = 0; k < i; ++k) KG_REQUIRE(jobs[k].out != a->out, "kg_agg_outer_many: jobs %d and %d share a destination", k, i);
        int slabs;
        if (outer_streams(a) && kg_env().agg_outer_mfma != 0) {
            const OuterGeom gm = outer_geom(a);
            const int chunks = kg_cdiv((long)a->N * a->T, gm.F);
            const long units = (long)a->C * chunks;
            slabs = (int)(units < 512 ? units : 512);
            if (budget > 0 && total_units > 0.0) {
                const long want = (long)(budget * ((double)units / total_units) + 0.5);
                slabs = (int)(want < 1 ? 1 : (want < slabs ? want : slabs));
            }
            const size_t l = outer_mfma_lds(a, gm);
            KG_REQUIRE(l <= 65536, "kg_agg_outer_many: LDS budget exceeded (%ld bytes)", (long)l);
            OuterManyJob& j = m.job[m.njobs++];
            j.a = *a;
            j.nunits = (int)units; j.chunks = chunks; j.gm = gm;
            j.wg_begin = wgs; j.nwg = slabs;
            j.variant = (a->K == 3 ? 4 : 0) + (a->V > 16 ? 2 : 0) + (a->W > 16 ? 1 : 0);
            wgs += slabs;
            if (l > lds) lds = l;
            if (m.njobs == OUTER_MANY_MAX)
                if (int rc = flush()) return rc;
        } else {
            KgAggArgs one = *a;
            one.defer_sum = 1;
            slabs = outer_slab_count(a);
            if (int rc = kg_agg_outer(&one, stream)) return rc;
        }
        KgOuterSumJob& r = sums.job[sums.njobs++];
        r.ws = a->ws; r.out = a->out; r.nout = nout; r.slabs = slabs;
        if (sums.njobs == KG_OUTER_SUM_MAX_JOBS)
            if (int rc = flush_sums()) return rc;
    }
    return flush_sums();
}
