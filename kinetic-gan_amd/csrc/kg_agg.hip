// kg_agg_*: spatial graph aggregation over the V joints of a frame with a tiny (K, V, W) matrix
// staged in LDS (<= 3*25*25 floats).  These are the HBM-bound kernels of the path: 2*K*V flop per
// 4*(K+1) bytes.  Frames ("rows" of V contiguous floats) stream from HBM with
// coalesced loads, are staged in LDS, and every thread produces output elements of one frame.
//
//  expand : out[k*C+c, (n,t',w)] = sum_v x[c,(n,t'/rep,v)] A[k,v,w]
//  reduce : out[c,(n,t,w)]       = sum_q sum_k sum_v y[k*C+c,(n,t*fold+q,v)] A[k,v,w]
//  outer  : dA[k,v,w]            = sum_{c,n,t'} x[c,(n,t'/rep,v)] y[k*C+c,(n,t',w)]
//
// Reference ops covered: torch.einsum('nkctv,kvw->nctw') (tgcn.py:66) and its gradients;
// upsample_s + nearest T up-sampling (generator.py:172,185-200) with K=1, A=U.
#include "kg_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXA = 3 * 25 * 25;   // K*V*W ceiling (NTU level 0)
constexpr int MAXROWS = 256;

// ---------------------------------------------------------------------------------------------
// expand: block = (row tile, channel c); thread = one (row, w) and all K partitions.
template <int K>
__global__ __launch_bounds__(NT) void kg_agg_expand_kernel(const KgAggArgs a, int R) {
    __shared__ float As[MAXA];
    __shared__ float xs[MAXROWS * 25];
    const int tid = threadIdx.x;
    const int V = a.V, W = a.W, c = blockIdx.y;
    const int Tout = a.T * a.rep;
    const int nrows = a.N * Tout;
    const int row0 = blockIdx.x * R;
    for (int e = tid; e < K * V * W; e += NT) As[e] = a.a[e];
    for (int e = tid; e < R * V; e += NT) {
        int rr = e / V, v = e - rr * V;
        int row = row0 + rr;
        float val = 0.f;
        if (row < nrows) {
            int n = row / Tout, tp = row - n * Tout;
            val = a.x[(long)c * a.x_sC + (long)n * a.x_sN + (long)(tp / a.rep) * V + v];
        }
        xs[e] = val;
    }
    __syncthreads();
    for (int e = tid; e < R * W; e += NT) {
        int rr = e / W, w = e - rr * W;
        int row = row0 + rr;
        if (row >= nrows) continue;
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = 0.f;
        const float* xr = xs + rr * V;
        for (int v = 0; v < V; ++v) {
            float xv = xr[v];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = fmaf(xv, As[(k * V + v) * W + w], acc[k]);
        }
        int n = row / Tout, tp = row - n * Tout;
        long o = (long)n * a.o_sN + (long)tp * W + w;
#pragma unroll
        for (int k = 0; k < K; ++k) a.out[(long)(k * a.C + c) * a.o_sC + o] = acc[k];
    }
}

// ---------------------------------------------------------------------------------------------
// reduce: block = (row tile of OUTPUT frames, channel c); thread = one (row, w).
template <int K>
__global__ __launch_bounds__(NT) void kg_agg_reduce_kernel(const KgAggArgs a, int R) {
    __shared__ float As[MAXA];
    __shared__ float ys[K * MAXROWS * 25 / 2];   // K * R * V floats, R*V <= 3200 guaranteed by the host
    const int tid = threadIdx.x;
    const int V = a.V, W = a.W, c = blockIdx.y;
    const int fold = a.rep;
    const int Tout = a.T, Tin = a.T * fold;
    const int nrows = a.N * Tout;
    const int row0 = blockIdx.x * R;
    for (int e = tid; e < K * V * W; e += NT) As[e] = a.a[e];

    const int npt = (R * W + NT - 1) / NT;   // outputs per thread (host keeps this <= 4)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};

    for (int q = 0; q < fold; ++q) {
        __syncthreads();
        for (int e = tid; e < K * R * V; e += NT) {
            int k = e / (R * V), rem = e - k * (R * V);
            int rr = rem / V, v = rem - rr * V;
            int row = row0 + rr;
            float val = 0.f;
            if (row < nrows) {
                int n = row / Tout, t = row - n * Tout;
                val = a.x[(long)(k * a.C + c) * a.x_sC + (long)n * a.x_sN + (long)(t * fold + q) * V + v];
            }
            ys[e] = val;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int e = tid + i * NT;
            if (i < npt && e < R * W) {
                int rr = e / W, w = e - rr * W;
                float s = acc[i];
                for (int k = 0; k < K; ++k) {
                    const float* yr = ys + (k * R + rr) * V;
                    const float* ak = As + k * V * W + w;
                    for (int v = 0; v < V; ++v) s = fmaf(yr[v], ak[v * W], s);
                }
                acc[i] = s;
            }
        }
    }
    (void)Tin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int e = tid + i * NT;
        if (i < npt && e < R * W) {
            int rr = e / W, w = e - rr * W;
            int row = row0 + rr;
            if (row < nrows) {
                int n = row / Tout, t = row - n * Tout;
                a.out[(long)c * a.o_sC + (long)n * a.o_sN + (long)t * W + w] = acc[i];
            }
        }
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
            int row = row0 + rr;
            float val = 0.f;
            if (row < nrows) {
                int n = row / Tp, tp = row - n * Tp;
                val = a.x[(long)c * a.x_sC + (long)n * a.x_sN + (long)(tp / a.rep) * V + v];
            }
            xs[e] = val;
        }
        for (int e = tid; e < K * OUT_R * W; e += NT) {
            int k = e / (OUT_R * W), rem = e - k * (OUT_R * W);
            int rr = rem / W, w = rem - rr * W;
            int row = row0 + rr;
            float val = 0.f;
            if (row < nrows) {
                int n = row / Tp, tp = row - n * Tp;
                val = a.y[(long)(k * a.C + c) * a.y_sC + (long)n * a.y_sN + (long)tp * W + w];
            }
            ys[e] = val;
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

}  // namespace

extern "C" int kg_agg_expand(const KgAggArgs* a, void* stream) {
    if (int rc = validate(a, "kg_agg_expand")) return rc;
    int R = NT / a->W;                       // one output per thread
    if (R > MAXROWS) R = MAXROWS;
    const long nrows = (long)a->N * a->T * a->rep;
    dim3 grid(kg_cdiv(nrows, R), a->C);
    hipStream_t s = (hipStream_t)stream;
    if (a->K == 3) hipLaunchKernelGGL(kg_agg_expand_kernel<3>, grid, dim3(NT), 0, s, *a, R);
    else           hipLaunchKernelGGL(kg_agg_expand_kernel<1>, grid, dim3(NT), 0, s, *a, R);
    return kg_launch_status("kg_agg_expand");
}

extern "C" int kg_agg_reduce(const KgAggArgs* a, void* stream) {
    if (int rc = validate(a, "kg_agg_reduce")) return rc;
    int R = NT / a->W;                       // one output per thread
    if (R > 128) R = 128;                    // ys holds K*R*V <= K*3200 floats (V <= 25)
    const long nrows = (long)a->N * a->T;
    dim3 grid(kg_cdiv(nrows, R), a->C);
    hipStream_t s = (hipStream_t)stream;
    if (a->K == 3) hipLaunchKernelGGL(kg_agg_reduce_kernel<3>, grid, dim3(NT), 0, s, *a, R);
    else           hipLaunchKernelGGL(kg_agg_reduce_kernel<1>, grid, dim3(NT), 0, s, *a, R);
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
