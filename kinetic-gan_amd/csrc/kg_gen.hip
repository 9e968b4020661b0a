// Generator st_gcn block, contract-first on the COARSE grid (generator.py:168-182).
//
// The reference up-samples first (upsample_s + nearest T, generator.py:169-172) and then runs the block's 1x1 convs on
// the fine grid.  A 1x1 conv is per column, so it commutes with both up-samplings:
//
//     gcn:   z[c,(n,t',w)] = sum_k sum_v (W_k (x U))[c,(n,t'/rep,v)] A_k[v,w]
//                          = sum_k sum_vc (W_k x)[c,(n,t'/rep,vc)] (U A_k)[vc,w]                 B_k := U A_k  (Vc x V)
//     res:   r[c,(n,t',w)] = (W_r (x U))[c,...] + b_r = sum_vc (W_r x)[c,(n,t'/rep,vc)] U[vc,w] + b_r
//
// so ONE channel contraction [W_gcn; W_r] x on the block's INPUT grid (kg_conv, 1/2 .. 1/5 of the columns) is followed
// by kg_gen_expand, which applies B_k / U and the frame repeat and writes z and r at the output resolution: the
// up-sampled input x U, the 3*C_out-plane conv output at the fine resolution and the separate residual conv launch
// of the op-by-op form never exist.  kg_gen_fold is the adjoint (gradients of z and r back to the coarse conv output,
// plus the frame-folded gz the adjacency gradient needs) and kg_gen_adj_finish turns the per-block (K, V, Vc) outer
// products into edge_importance gradients (d A_k = U^T d B_k, d importance = A * d A) for all blocks in one launch.
//
// These are HBM / latency bound streaming kernels (the generator is 5 % of the iteration's flops); the adjacency
// product U A_k is formed in LDS by every workgroup (<= 3 * 11 * 25 * 25 multiply-adds).
#include "kg_common.h"

namespace {

constexpr int NT = 256;
constexpr int VMAX = 32;       // vertices per level (NTU 25, H36M 16)
constexpr int KMAX = 3;

// B[k][vc][w] = sum_v U[vc][v] A[k][v][w]  (U == NULL: identity, Vc == V) and Us[vc][w] = U (or identity) into LDS
__device__ __forceinline__ void stage_adjacency(const float* a, const float* u, int K, int Vc, int V, float* As, float* Us,
                                                float* Bs) {
    const int tid = threadIdx.x;
    for (int i = tid; i < K * V * V; i += NT) As[i] = a ? a[i] : 0.f;
    for (int i = tid; i < Vc * V; i += NT) Us[i] = u ? u[i] : ((i / V) == (i % V) ? 1.f : 0.f);
    __syncthreads();
    for (int i = tid; i < K * Vc * V; i += NT) {
        const int k = i / (Vc * V), r = i - k * Vc * V, vc = r / V, w = r - vc * V;
        float s = 0.f;
        for (int v = 0; v < V; ++v) s = fmaf(Us[vc * V + v], As[(k * V + v) * V + w], s);
        Bs[i] = s;
    }
    __syncthreads();
}

// One thread per (channel, coarse frame, output vertex): its value is written to the `rep` fine frames of the coarse one.
__global__ __launch_bounds__(NT) void kg_gen_expand_kernel(const KgGenArgs a) {
    __shared__ float As[KMAX * VMAX * VMAX], Us[VMAX * VMAX], Bs[KMAX * VMAX * VMAX];
    stage_adjacency(a.a, a.u, a.z ? a.K : 0, a.Vc, a.V, As, Us, Bs);
    const int V = a.V, Vc = a.Vc;
    const long per_c = (long)a.N * a.Tc * V;                  // (frame, vertex) items of one channel
    const int Cz = a.z ? a.C : 0, Cr = a.r ? a.Cr : 0;
    const long total = (long)(Cz + Cr) * per_c;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i / per_c);
        const long rem = i - (long)c * per_c;
        const int f = (int)(rem / V), w = (int)(rem - (long)f * V);
        const int n = f / a.Tc, tc = f - n * a.Tc;
        float s;
        float* op;
        if (c < Cz) {
            s = 0.f;
            for (int k = 0; k < a.K; ++k) {
                const float* yp = a.y + (long)(k * a.C + c) * a.y_sC + (long)n * a.y_sN + (long)tc * Vc;
                const float* bp = Bs + (k * Vc) * V + w;
                for (int vc = 0; vc < Vc; ++vc) s = fmaf(yp[vc], bp[vc * V], s);
            }
            op = a.z + (long)c * a.z_sC + (long)n * a.z_sN + (long)(tc * a.rep) * V + w;
        } else {
            const int cr = c - Cz;
            const float* rp = a.rs + (long)cr * a.rs_sC + (long)n * a.rs_sN + (long)tc * Vc;
            s = a.rbias ? a.rbias[cr] : 0.f;
            for (int vc = 0; vc < Vc; ++vc) s = fmaf(rp[vc], Us[vc * V + w], s);
            op = a.r + (long)cr * a.r_sC + (long)n * a.r_sN + (long)(tc * a.rep) * V + w;
        }
        for (int q = 0; q < a.rep; ++q) op[(long)q * V] = s;
    }
}

// Adjoint: one thread per (channel, coarse frame, coarse vertex) computes the K partition outputs of gz (they share the
// loads of the frame's rep * V values) or the residual's; then (optionally) gzf = gz summed over the repeated frames.
__global__ __launch_bounds__(NT) void kg_gen_fold_kernel(const KgGenArgs a) {
    __shared__ float As[KMAX * VMAX * VMAX], Us[VMAX * VMAX], Bs[KMAX * VMAX * VMAX];
    stage_adjacency(a.a, a.u, a.z ? a.K : 0, a.Vc, a.V, As, Us, Bs);
    const int V = a.V, Vc = a.Vc;
    const long per_c = (long)a.N * a.Tc * Vc;
    const int Cz = a.z ? a.C : 0, Cr = a.r ? a.Cr : 0;
    const long tot_y = (long)(Cz + Cr) * per_c;
    const long per_f = (long)a.N * a.Tc * V;
    const long tot_f = a.zf ? (long)a.C * per_f : 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < tot_y + tot_f; i += (long)gridDim.x * NT) {
        if (i < tot_y) {
            const int c = (int)(i / per_c);
            const long rem = i - (long)c * per_c;
            const int f = (int)(rem / Vc), vc = (int)(rem - (long)f * Vc);
            const int n = f / a.Tc, tc = f - n * a.Tc;
            if (c < Cz) {
                const float* gp = a.z + (long)c * a.z_sC + (long)n * a.z_sN + (long)(tc * a.rep) * V;
                float s[KMAX] = {0.f, 0.f, 0.f};
                for (int q = 0; q < a.rep; ++q)
                    for (int w = 0; w < V; ++w) {
                        const float g = gp[q * V + w];
#pragma unroll
                        for (int k = 0; k < KMAX; ++k)
                            if (k < a.K) s[k] = fmaf(g, Bs[(k * Vc + vc) * V + w], s[k]);
                    }
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < a.K) a.y_out[(long)(k * a.C + c) * a.y_sC + (long)n * a.y_sN + (long)tc * Vc + vc] = s[k];
            } else {
                const int cr = c - Cz;
                const float* gp = a.r + (long)cr * a.r_sC + (long)n * a.r_sN + (long)(tc * a.rep) * V;
                float s = 0.f;
                for (int q = 0; q < a.rep; ++q)
                    for (int w = 0; w < V; ++w) s = fmaf(gp[q * V + w], Us[vc * V + w], s);
                a.rs_out[(long)cr * a.rs_sC + (long)n * a.rs_sN + (long)tc * Vc + vc] = s;
            }
        } else {
            const long j = i - tot_y;
            const int c = (int)(j / per_f);
            const long rem = j - (long)c * per_f;
            const int f = (int)(rem / V), w = (int)(rem - (long)f * V);
            const int n = f / a.Tc, tc = f - n * a.Tc;
            const float* gp = a.z + (long)c * a.z_sC + (long)n * a.z_sN + (long)(tc * a.rep) * V + w;
            float s = 0.f;
            for (int q = 0; q < a.rep; ++q) s += gp[(long)q * V];
            a.zf[(long)c * a.zf_sC + (long)n * a.zf_sN + (long)tc * V + w] = s;
        }
    }
}

struct AdjJobs { int njobs; int beg[KG_GEN_ADJ_MAX_JOBS + 1]; KgGenAdjJob job[KG_GEN_ADJ_MAX_JOBS]; };

// dimp[k][v][w] (+)= A[k][v][w] * sum_vc U[vc][v] * dBt[k][w][vc]
__global__ __launch_bounds__(NT) void kg_gen_adj_finish_kernel(const AdjJobs js) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= js.beg[js.njobs]) return;
    int ji = 0;
    while (ji + 1 < js.njobs && i >= js.beg[ji + 1]) ++ji;
    const KgGenAdjJob& j = js.job[ji];
    const int e = i - js.beg[ji];
    const int V = j.V, Vc = j.Vc;
    const int k = e / (V * V), r = e - k * V * V, v = r / V, w = r - v * V;
    float s = 0.f;
    if (k < j.Kd) {          // partitions beyond Kd carry no gradient (single-partition blocks)
        const float* d = j.dbt + ((long)k * V + w) * Vc;
        if (j.u) {
            for (int vc = 0; vc < Vc; ++vc) s = fmaf(j.u[vc * V + v], d[vc], s);
        } else {
            s = d[v];
        }
    }
    const float val = (j.a ? j.a[e] : 1.f) * s;
    j.out[e] = j.accumulate ? j.out[e] + val : val;
}

int validate(const KgGenArgs* a, const char* who, bool fold) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->Tc > 0 && a->Vc > 0 && a->V > 0 && a->rep >= 1, "%s: bad dims", who);
    KG_REQUIRE(a->V <= VMAX && a->Vc <= VMAX, "%s: V=%d / Vc=%d exceed %d vertices", who, a->V, a->Vc, VMAX);
    KG_REQUIRE(a->u != nullptr || a->Vc == a->V, "%s: Vc=%d != V=%d without an up-sampling matrix", who, a->Vc, a->V);
    KG_REQUIRE(a->z != nullptr || a->r != nullptr, "%s: neither the gcn nor the residual branch is given", who);
    if (a->z) {
        KG_REQUIRE(a->C > 0 && a->K >= 1 && a->K <= KMAX && a->a != nullptr, "%s: gcn branch: C=%d K=%d", who, a->C, a->K);
        KG_REQUIRE(fold ? a->y_out != nullptr : a->y != nullptr, "%s: gcn branch: null conv tensor", who);
    }
    if (a->r) {
        KG_REQUIRE(a->Cr > 0, "%s: residual branch: Cr=%d", who, a->Cr);
        KG_REQUIRE(fold ? a->rs_out != nullptr : a->rs != nullptr, "%s: residual branch: null source tensor", who);
    }
    KG_REQUIRE(!a->zf || (fold && a->z), "%s: zf is an output of kg_gen_fold's gcn branch", who);
    KG_REQUIRE((long)(a->C + a->Cr) * a->N * a->Tc * a->rep * a->V < (1L << 40), "%s: too large", who);
    return 0;
}

int grid_for(long items) {
    // every workgroup forms U A_k in LDS first (~1.5 us): at most two workgroups per CU, each striding over the items
    // (first version: 16 per CU - the set-up, repeated 9x per CU, was 3/4 of the G6 launch: 25 us)
    long g = (items + NT - 1) / NT;
    const long cap = 256L * 2;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int kg_gen_expand(const KgGenArgs* a, void* stream) {
    if (int rc = validate(a, "kg_gen_expand", false)) return rc;
    const long items = (long)((a->z ? a->C : 0) + (a->r ? a->Cr : 0)) * a->N * a->Tc * a->V;
    hipLaunchKernelGGL(kg_gen_expand_kernel, dim3(grid_for(items)), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_gen_expand");
}

extern "C" int kg_gen_fold(const KgGenArgs* a, void* stream) {
    if (int rc = validate(a, "kg_gen_fold", true)) return rc;
    const long items = (long)((a->z ? a->C : 0) + (a->r ? a->Cr : 0)) * a->N * a->Tc * a->Vc +
                       (a->zf ? (long)a->C * a->N * a->Tc * a->V : 0);
    hipLaunchKernelGGL(kg_gen_fold_kernel, dim3(grid_for(items)), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_gen_fold");
}

extern "C" int kg_gen_adj_finish(const KgGenAdjJob* jobs, int32_t njobs, void* stream) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= KG_GEN_ADJ_MAX_JOBS, "kg_gen_adj_finish: 1..%d jobs", KG_GEN_ADJ_MAX_JOBS);
    AdjJobs js;
    js.njobs = njobs;
    int tot = 0;
    for (int i = 0; i < njobs; ++i) {
        const KgGenAdjJob& j = jobs[i];
        KG_REQUIRE(j.K >= 1 && j.K <= KMAX && j.Kd >= 0 && j.Kd <= j.K && j.V >= 1 && j.V <= VMAX && j.Vc >= 1 && j.Vc <= VMAX,
                   "kg_gen_adj_finish: job %d dims", i);
        KG_REQUIRE(j.dbt && j.out, "kg_gen_adj_finish: job %d null pointer", i);
        KG_REQUIRE(j.u != nullptr || j.Vc == j.V, "kg_gen_adj_finish: job %d Vc != V without U", i);
        js.beg[i] = tot;
        js.job[i] = j;
        tot += j.K * j.V * j.V;
    }
    js.beg[njobs] = tot;
    hipLaunchKernelGGL(kg_gen_adj_finish_kernel, dim3(kg_cdiv(tot, NT)), dim3(NT), 0, (hipStream_t)stream, js);
    return kg_launch_status("kg_gen_adj_finish");
}
