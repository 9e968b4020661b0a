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
// outer: frames staged in LDS with coalesced loads; every thread keeps its share of the K*V*W outputs in
// registers over a strided list of (channel, frame tile) units; one partial slab per workgroup, fixed-order
// reduction (deterministic).  (An MFMA formulation - dA_k = X^T Y_k with V, W padded to 32 - was measured
// 2x slower: its operand loads are 100-byte rows, i.e. bound by VMEM instruction issue, not bytes.)
//
// Reference ops covered: torch.einsum('nkctv,kvw->nctw') (tgcn.py:66) and its gradients;
// upsample_s + nearest T up-sampling (generator.py:172,185-200) with K=1, A=U.
#include "kg_common.h"

namespace {

constexpr int NT = 256;

// A (K,V,W) -> LDS As[k][v][WP], zero padded
template <int K, int WP>
__device__ __forceinline__ void stage_A(float* As, const float* a, int V, int W) {
    for (int e = threadIdx.x; e < K * V * WP; e += NT) {
        const int w = e % WP;
        const int kv = e / WP;
        As[e] = w < W ? a[kv * W + w] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
template <int K, int WP>
__global__ __launch_bounds__(NT) void kg_agg_expand_kernel(const KgAggArgs a) {
    __shared__ __attribute__((aligned(16))) float As[K * 25 * WP];
    const int V = a.V, W = a.W, c = blockIdx.y;
    stage_A<K, WP>(As, a.a, V, W);
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
__global__ __launch_bounds__(NT) void kg_agg_reduce_kernel(const KgAggArgs a) {
    __shared__ __attribute__((aligned(16))) float As[K * 25 * WP];
    const int V = a.V, W = a.W, c = blockIdx.y;
    stage_A<K, WP>(As, a.a, V, W);
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
#pragma unroll
    for (int w = 0; w < WP; ++w)
        if (w < W) op[w] = acc[w];
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

__global__ __launch_bounds__(256) void kg_agg_outer_sum(const float* ws, float* out, int nout, int slabs) {
    const int e = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = kg_slab_sum_256(ws, nout, e, e < nout, slabs);
    if (e < nout && threadIdx.x < 64) out[e] = s;
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
            if (wp == 4) hipLaunchKernelGGL((KERNEL<3, 4>), grid, dim3(NT), 0, s, *a);                 \
            else if (wp == 8) hipLaunchKernelGGL((KERNEL<3, 8>), grid, dim3(NT), 0, s, *a);            \
            else if (wp == 16) hipLaunchKernelGGL((KERNEL<3, 16>), grid, dim3(NT), 0, s, *a);          \
            else hipLaunchKernelGGL((KERNEL<3, 28>), grid, dim3(NT), 0, s, *a);                        \
        } else {                                                                                       \
            if (wp == 4) hipLaunchKernelGGL((KERNEL<1, 4>), grid, dim3(NT), 0, s, *a);                 \
            else if (wp == 8) hipLaunchKernelGGL((KERNEL<1, 8>), grid, dim3(NT), 0, s, *a);            \
            else if (wp == 16) hipLaunchKernelGGL((KERNEL<1, 16>), grid, dim3(NT), 0, s, *a);          \
            else hipLaunchKernelGGL((KERNEL<1, 28>), grid, dim3(NT), 0, s, *a);                        \
        }                                                                                              \
    } while (0)

}  // namespace

extern "C" int kg_agg_expand(const KgAggArgs* a, void* stream) {
    if (int rc = validate(a, "kg_agg_expand")) return rc;
    const long nrows = (long)a->N * a->T * a->rep;
    dim3 grid(kg_cdiv(nrows, NT), a->C);
    hipStream_t s = (hipStream_t)stream;
    KG_AGG_DISPATCH(kg_agg_expand_kernel, grid);
    return kg_launch_status("kg_agg_expand");
}

extern "C" int kg_agg_reduce(const KgAggArgs* a, void* stream) {
    if (int rc = validate(a, "kg_agg_reduce")) return rc;
    const long nrows = (long)a->N * a->T;
    dim3 grid(kg_cdiv(nrows, NT), a->C);
    hipStream_t s = (hipStream_t)stream;
    KG_AGG_DISPATCH(kg_agg_reduce_kernel, grid);
    return kg_launch_status("kg_agg_reduce");
}

extern "C" int64_t kg_agg_outer_workspace_bytes(const KgAggArgs* a) {
    if (a == nullptr || a->C <= 0 || a->N <= 0 || a->T <= 0 || a->rep < 1) return -1;
    int nunits, row_tiles;
    int slabs = outer_slabs(a, &nunits, &row_tiles);
    return (int64_t)slabs * a->K * a->V * a->W * (int64_t)sizeof(float);
}

extern "C" int kg_agg_outer(const KgAggArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr, "kg_agg_outer: null args");
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->rep >= 1, "kg_agg_outer: bad dims");
    KG_REQUIRE(a->K == 1 || a->K == 3, "kg_agg_outer: K=%d", a->K);
    KG_REQUIRE(a->V >= 1 && a->V <= 25 && a->W >= 1 && a->W <= 25, "kg_agg_outer: V=%d W=%d", a->V, a->W);
    KG_REQUIRE(a->x && a->y && a->out && a->ws, "kg_agg_outer: null pointer");
    int nunits, row_tiles;
    const int slabs = outer_slabs(a, &nunits, &row_tiles);
    const int nout = a->K * a->V * a->W;
    KG_REQUIRE(a->ws_bytes >= (int64_t)slabs * nout * 4, "kg_agg_outer: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (a->K == 3) hipLaunchKernelGGL(kg_agg_outer_kernel<3>, dim3(slabs), dim3(NT), 0, s, *a, nunits, row_tiles);
    else           hipLaunchKernelGGL(kg_agg_outer_kernel<1>, dim3(slabs), dim3(NT), 0, s, *a, nunits, row_tiles);
    if (int rc = kg_launch_status("kg_agg_outer")) return rc;
    hipLaunchKernelGGL(kg_agg_outer_sum, dim3(kg_cdiv(nout, 64)), dim3(256), 0, s, a->ws, a->out, nout, slabs);
    return kg_launch_status("kg_agg_outer_sum");
}
