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

struct GenDivs { FastDiv v, vc, tc, per_out, per_in; };      // per_out = N*Tc*V, per_in = N*Tc*Vc (items of one channel)

// B[k][vc][w] = sum_v U[vc][v] A[k][v][w]  (U == NULL: identity, Vc == V) and Us[vc][w] = U (or identity) into LDS;
// with a precomputed product (a.b, kg_gen_adj_prepare) it is only loaded
__device__ __forceinline__ void stage_adjacency(const KgGenArgs& a, int K, float* As, float* Us, float* Bs) {
    const int tid = threadIdx.x;
    const int V = a.V, Vc = a.Vc;
    for (int i = tid; i < Vc * V; i += NT) Us[i] = a.u ? a.u[i] : ((i / V) == (i % V) ? 1.f : 0.f);
    if (a.b) {
        for (int i = tid; i < K * Vc * V; i += NT) Bs[i] = a.b[i];
        __syncthreads();
        return;
    }
    for (int i = tid; i < K * V * V; i += NT) As[i] = a.a[i];
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
// Lanes run along (frame, vertex): the Vc source values of a frame are read by the V lanes of that frame (L1 broadcast),
// the stores are contiguous runs of V floats.
__global__ __launch_bounds__(NT) void kg_gen_expand_kernel(const KgGenArgs a, const GenDivs dv) {
    __shared__ float As[KMAX * VMAX * VMAX], Us[VMAX * VMAX], Bs[KMAX * VMAX * VMAX];
    stage_adjacency(a, a.z ? a.K : 0, As, Us, Bs);
    const int V = a.V, Vc = a.Vc;
    const unsigned per_c = dv.per_out.d;
    const int Cz = a.z ? a.C : 0, Cr = a.r ? a.Cr : 0;
    const unsigned total = (unsigned)(Cz + Cr) * per_c;
    for (unsigned i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
        unsigned c, rem, f, w, n, tc;
        dv.per_out.divmod(i, c, rem);
        dv.v.divmod(rem, f, w);
        dv.tc.divmod(f, n, tc);
        float s;
        float* op;
        if ((int)c < Cz) {
            s = 0.f;
            for (int k = 0; k < a.K; ++k) {
                const float* yp = a.y + (long)(k * a.C + c) * a.y_sC + (long)n * a.y_sN + tc * Vc;
                const float* bp = Bs + (k * Vc) * V + w;
#pragma unroll 4
                for (int vc = 0; vc < Vc; ++vc) s = fmaf(yp[vc], bp[vc * V], s);
            }
            op = a.z + (long)c * a.z_sC + (long)n * a.z_sN + (tc * a.rep) * V + w;
        } else {
            const int cr = c - Cz;
            const float* rp = a.rs + (long)cr * a.rs_sC + (long)n * a.rs_sN + tc * Vc;
            s = a.rbias ? a.rbias[cr] : 0.f;
#pragma unroll 4
            for (int vc = 0; vc < Vc; ++vc) s = fmaf(rp[vc], Us[vc * V + w], s);
            op = a.r + (long)cr * a.r_sC + (long)n * a.r_sN + (tc * a.rep) * V + w;
        }
        for (int q = 0; q < a.rep; ++q) op[q * V] = s;
    }
}

// Adjoint: one thread per (channel, coarse frame, coarse vertex) computes the K partition outputs of gz (they share the
// loads of the frame's rep * V values) or the residual's; then (optionally) gzf = gz summed over the repeated frames.
__global__ __launch_bounds__(NT) void kg_gen_fold_kernel(const KgGenArgs a, const GenDivs dv) {
    __shared__ float As[KMAX * VMAX * VMAX], Us[VMAX * VMAX], Bs[KMAX * VMAX * VMAX];
    stage_adjacency(a, a.z ? a.K : 0, As, Us, Bs);
    const int V = a.V, Vc = a.Vc;
    const unsigned per_c = dv.per_in.d;
    const int Cz = a.z ? a.C : 0, Cr = a.r ? a.Cr : 0;
    const unsigned tot_y = (unsigned)(Cz + Cr) * per_c;
    const unsigned per_f = dv.per_out.d;
    const unsigned tot_f = a.zf ? (unsigned)a.C * per_f : 0;
    for (unsigned i = blockIdx.x * NT + threadIdx.x; i < tot_y + tot_f; i += gridDim.x * NT) {
        if (i < tot_y) {
            unsigned c, rem, f, vc, n, tc;
            dv.per_in.divmod(i, c, rem);
            dv.vc.divmod(rem, f, vc);
            dv.tc.divmod(f, n, tc);
            if ((int)c < Cz) {
                const float* gp = a.z + (long)c * a.z_sC + (long)n * a.z_sN + (tc * a.rep) * V;
                float s[KMAX] = {0.f, 0.f, 0.f};
                const int len = a.rep * V;               // the repeated frames follow each other
                const float* b0 = Bs + vc * V;
                const int kstep = Vc * V;
                int w = 0;
                for (int e = 0; e < len; ++e) {
                    const float g = gp[e];
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        if (k < a.K) s[k] = fmaf(g, b0[k * kstep + w], s[k]);
                    if (++w == V) w = 0;
                }
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < a.K) a.y_out[(long)(k * a.C + c) * a.y_sC + (long)n * a.y_sN + tc * Vc + vc] = s[k];
            } else {
                const int cr = c - Cz;
                const float* gp = a.r + (long)cr * a.r_sC + (long)n * a.r_sN + (tc * a.rep) * V;
                float s = 0.f;
                const int len = a.rep * V;
                int w = 0;
                for (int e = 0; e < len; ++e) {
                    s = fmaf(gp[e], Us[vc * V + w], s);
                    if (++w == V) w = 0;
                }
                a.rs_out[(long)cr * a.rs_sC + (long)n * a.rs_sN + tc * Vc + vc] = s;
            }
        } else {
            const unsigned j = i - tot_y;
            unsigned c, rem, f, w, n, tc;
            dv.per_out.divmod(j, c, rem);
            dv.v.divmod(rem, f, w);
            dv.tc.divmod(f, n, tc);
            const float* gp = a.z + (long)c * a.z_sC + (long)n * a.z_sN + (tc * a.rep) * V + w;
            float s = 0.f;
            for (int q = 0; q < a.rep; ++q) s += gp[q * V];
            a.zf[(long)c * a.zf_sC + (long)n * a.zf_sN + tc * V + w] = s;
        }
    }
}

// A_eff = A * importance and B = U A_eff of several blocks in one launch (kg_gen_adj_prepare): one workgroup per
// block, the masked adjacency staged in LDS and the product formed from there (a thread-per-element version with the
// V-deep products read straight from global memory took 11 us for 4 k elements: ~75 dependent loads per thread)
struct PrepJobs { int njobs; KgGenPrepJob job[KG_GEN_ADJ_MAX_JOBS]; };

__global__ __launch_bounds__(NT) void kg_gen_adj_prepare_kernel(const PrepJobs js) {
    __shared__ float As[KMAX * VMAX * VMAX], Us[VMAX * VMAX];
    const KgGenPrepJob& j = js.job[blockIdx.x];
    const int tid = threadIdx.x;
    const int V = j.V, Vc = j.Vc, na = j.K * V * V;
    for (int e = tid; e < na; e += NT) {
        const float v = j.a[e] * (j.imp ? j.imp[e] : 1.f);
        As[e] = v;
        j.aeff[e] = v;
    }
    if (j.u)
        for (int e = tid; e < Vc * V; e += NT) Us[e] = j.u[e];
    __syncthreads();
    for (int e = tid; e < j.K * Vc * V; e += NT) {
        const int k = e / (Vc * V), r = e - k * Vc * V, vc = r / V, w = r - vc * V;
        float s = 0.f;
        if (j.u) {
            for (int v = 0; v < V; ++v) s = fmaf(Us[vc * V + v], As[(k * V + v) * V + w], s);
        } else {
            s = As[(k * V + vc) * V + w];
        }
        j.b[e] = s;
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

// ---- backward of a block's tail: out = act(BN_t(u) + BN_r(r) + w_noise * noise)  (generator.py:142,160,176,179-182) ------
// kg_gen_tail_stats: per channel, over (n, t, v), with gp = g * act'(out) formed on the fly (never stored):
//     s_g = sum gp,  s_u = sum gp (u - mean_t),  s_r = sum gp (r - mean_r),  s_n = sum gp * noise
// A workgroup takes a 4096-element chunk of one channel; the last workgroup of a channel to arrive (ticket counter, left
// at zero) adds the partials in chunk order and writes the BatchNorm-backward coefficients
//     coef[0..2] = (a_t, b_t, c_t),  coef[3..5] = (a_r, b_r, c_r):   d/du = a_t gp + b_t u + c_t,  d/dr likewise
// and ADDS the parameter gradients (d gamma = s * rstd, d beta = s_g, d w_noise = s_n) into the given buffers.
// kg_gen_tail_apply: du / dr from those coefficients in one pass (a branch without BatchNorm gets gp itself).
// Replaces kg_act_bwd + kg_bn_bwd_many + kg_rowsum (noise) + two kg_affine_act launches per block.
constexpr int TAIL_CHUNK = 2048;

struct TailPlan { int P; FastDiv l; };

__device__ __forceinline__ float tail_gp(const KgGenTailArgs& a, float g, float o) {
    return g * kg_dact_from_out(o, a.act, a.slope);
}

__global__ __launch_bounds__(NT) void kg_gen_tail_stats_kernel(const KgGenTailArgs a, const TailPlan pl) {
    __shared__ float red[4][NT / 64];
    __shared__ int last;
    const int P = pl.P;
    const int c = blockIdx.x / P, p = blockIdx.x - c * P;
    const int tid = threadIdx.x;
    const int L = a.T * a.V;
    const int ncols = a.N * L;
    const int jbeg = p * TAIL_CHUNK;
    const bool bn_t = a.u != nullptr, bn_r = a.r != nullptr && a.mean_r != nullptr, nz = a.noise != nullptr;
    const float mt = bn_t ? a.mean_t[c] : 0.f, mr = bn_r ? a.mean_r[c] : 0.f;
    constexpr int PER = TAIL_CHUNK / NT;
    // every load of the thread is issued before the first use (one by one the loop runs at memory latency)
    float gv[PER], ov[PER], uv[PER], rv[PER], nv_[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int j = jbeg + tid + i * NT;
        const bool ok = j < ncols;
        const unsigned jj = ok ? (unsigned)j : 0u;
        unsigned n, r;
        pl.l.divmod(jj, n, r);
        gv[i] = ok ? a.g[(long)c * a.g_sC + (long)n * a.g_sN + r] : 0.f;
        ov[i] = ok ? a.out[(long)c * a.o_sC + (long)n * a.o_sN + r] : 0.f;
        uv[i] = (ok && bn_t) ? a.u[(long)c * a.u_sC + (long)n * a.u_sN + r] : mt;
        rv[i] = (ok && bn_r) ? a.r[(long)c * a.r_sC + (long)n * a.r_sN + r] : mr;
        nv_[i] = (ok && nz) ? a.noise[(long)n * L + r] : 0.f;
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float gp = tail_gp(a, gv[i], ov[i]);
        s[0] += gp;
        s[1] = fmaf(gp, uv[i] - mt, s[1]);
        s[2] = fmaf(gp, rv[i] - mr, s[2]);
        s[3] = fmaf(gp, nv_[i], s[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = s[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) red[q][tid >> 6] = v;
    }
    __syncthreads();
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    if (P > 1) {                        // (uniform)
        float* const part = a.ws + ((long)c * P) * 4;
        if (tid == 0) {
            for (int q = 0; q < 4; ++q) {
                float tq = 0.f;
                for (int w = 0; w < NT / 64; ++w) tq += red[q][w];
                __hip_atomic_store(part + p * 4 + q, tq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int tk = __hip_atomic_fetch_add(a.counters + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (tk == P - 1);
        }
        __syncthreads();
        if (!last) return;
        // the last arriver: ALL its threads fetch partials (chunk k by thread k mod NT), then a fixed-shape tree adds
        // them - deterministic, and not a chain of P dependent loads in one lane (50 chunks: ~12 us)
        for (int k = tid; k < P; k += NT)
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] += __hip_atomic_load(part + k * 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                    // red is read by thread 0 above
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = t[q];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((tid & 63) == 0) red[q][tid >> 6] = v;
        }
        __syncthreads();
    }
    // (a channel with ONE chunk - the generator's first blocks - has its sums in `red` already: no partial record, no
    // ticket, no second trip through memory)
    if (tid != 0) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        t[q] = 0.f;
        for (int w = 0; w < NT / 64; ++w) t[q] += red[q][w];
    }
    const float inv_n = 1.f / (float)ncols;
    float at = 1.f, bt = 0.f, ct = 0.f, ar = 1.f, br = 0.f, cr = 0.f;
    if (bn_t) {
        const float rstd = a.rstd_t[c], q = t[1] * rstd;
        at = (a.gamma_t ? a.gamma_t[c] : 1.f) * rstd;
        bt = -at * rstd * q * inv_n;
        ct = -at * t[0] * inv_n - bt * mt;
        if (a.dgamma_t) a.dgamma_t[c] += q;
        if (a.dbeta_t) a.dbeta_t[c] += t[0];
    }
    if (bn_r) {
        const float rstd = a.rstd_r[c], q = t[2] * rstd;
        ar = (a.gamma_r ? a.gamma_r[c] : 1.f) * rstd;
        br = -ar * rstd * q * inv_n;
        cr = -ar * t[0] * inv_n - br * mr;
        if (a.dgamma_r) a.dgamma_r[c] += q;
        if (a.dbeta_r) a.dbeta_r[c] += t[0];
    }
    if (a.noise && a.dnw) a.dnw[c] += t[3];
    a.coef[0 * a.C + c] = at; a.coef[1 * a.C + c] = bt; a.coef[2 * a.C + c] = ct;
    a.coef[3 * a.C + c] = ar; a.coef[4 * a.C + c] = br; a.coef[5 * a.C + c] = cr;
    if (P > 1) a.counters[c] = 0;
}

__global__ __launch_bounds__(NT) void kg_gen_tail_apply_kernel(const KgGenTailArgs a, const FastDiv ld) {
    // grid (column tiles over (n, t, v), channel); four elements per thread, loads before stores
    const int c = blockIdx.y;
    const int L = a.T * a.V, ncols = a.N * L;
    const bool bn_t = a.u != nullptr, bn_r = a.r != nullptr && a.mean_r != nullptr;
    const float at = a.coef[0 * a.C + c], bt = a.coef[1 * a.C + c], ct = a.coef[2 * a.C + c];
    const float ar = a.coef[3 * a.C + c], br = a.coef[4 * a.C + c], cr = a.coef[5 * a.C + c];
    float gv[4], ov[4], uv[4], rv[4];
    unsigned nn[4], rr[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = (blockIdx.x * 4 + q) * NT + threadIdx.x;
        ok[q] = j < ncols;
        ld.divmod(ok[q] ? (unsigned)j : 0u, nn[q], rr[q]);
        gv[q] = ok[q] ? a.g[(long)c * a.g_sC + (long)nn[q] * a.g_sN + rr[q]] : 0.f;
        ov[q] = ok[q] ? a.out[(long)c * a.o_sC + (long)nn[q] * a.o_sN + rr[q]] : 0.f;
        uv[q] = (ok[q] && bn_t) ? a.u[(long)c * a.u_sC + (long)nn[q] * a.u_sN + rr[q]] : 0.f;
        rv[q] = (ok[q] && bn_r) ? a.r[(long)c * a.r_sC + (long)nn[q] * a.r_sN + rr[q]] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!ok[q]) continue;
        const float gp = tail_gp(a, gv[q], ov[q]);
        a.du[(long)c * a.du_sC + (long)nn[q] * a.du_sN + rr[q]] = bn_t ? fmaf(at, gp, fmaf(bt, uv[q], ct)) : gp;
        if (a.dr) a.dr[(long)c * a.dr_sC + (long)nn[q] * a.dr_sN + rr[q]] = bn_r ? fmaf(ar, gp, fmaf(br, rv[q], cr)) : gp;
    }
}

int validate_tail(const KgGenTailArgs* a, const char* who) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0 && a->C <= 65535 && (long)a->N * a->T * a->V < (1L << 31), "%s: bad dims", who);
    KG_REQUIRE(a->g && a->out && a->coef, "%s: null pointer", who);
    KG_REQUIRE(a->u == nullptr || (a->mean_t && a->rstd_t), "%s: BatchNorm on the tcn branch needs its statistics", who);
    KG_REQUIRE(a->mean_r == nullptr || (a->r && a->rstd_r), "%s: BatchNorm on the residual branch needs r and its statistics", who);
    return 0;
}

int validate(const KgGenArgs* a, const char* who, bool fold) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->Tc > 0 && a->Vc > 0 && a->V > 0 && a->rep >= 1, "%s: bad dims", who);
    KG_REQUIRE(a->V <= VMAX && a->Vc <= VMAX, "%s: V=%d / Vc=%d exceed %d vertices", who, a->V, a->Vc, VMAX);
    KG_REQUIRE(a->u != nullptr || a->Vc == a->V, "%s: Vc=%d != V=%d without an up-sampling matrix", who, a->Vc, a->V);
    KG_REQUIRE(a->z != nullptr || a->r != nullptr, "%s: neither the gcn nor the residual branch is given", who);
    if (a->z) {
        KG_REQUIRE(a->C > 0 && a->K >= 1 && a->K <= KMAX, "%s: gcn branch: C=%d K=%d", who, a->C, a->K);
        KG_REQUIRE(fold ? a->y_out != nullptr : a->y != nullptr, "%s: gcn branch: null conv tensor", who);
    }
    if (a->r) {
        KG_REQUIRE(a->Cr > 0, "%s: residual branch: Cr=%d", who, a->Cr);
        KG_REQUIRE(fold ? a->rs_out != nullptr : a->rs != nullptr, "%s: residual branch: null source tensor", who);
    }
    KG_REQUIRE(!a->zf || (fold && a->z), "%s: zf is an output of kg_gen_fold's gcn branch", who);
    KG_REQUIRE((long)(2 * a->C + a->Cr) * a->N * a->Tc * (a->V > a->Vc ? a->V : a->Vc) < (1L << 31), "%s: too many elements for 32-bit item indices", who);
    KG_REQUIRE(a->z == nullptr || a->a != nullptr || a->b != nullptr, "%s: gcn branch needs the adjacency (a) or the product U A (b)", who);
    return 0;
}

int grid_for(const KgGenArgs* a, long items) {
    // with a precomputed U A_k (a->b) a workgroup's set-up is a ~1 k-float copy: up to eight workgroups per CU; without
    // it every workgroup forms the product in LDS first (~1.5 us): two per CU, each striding over the items
    long g = (items + NT - 1) / NT;
    const long cap = 256L * (a->b ? 8 : 2);
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

GenDivs divs_of(const KgGenArgs* a) {
    GenDivs d;
    d.v = FastDiv::make((unsigned)a->V);
    d.vc = FastDiv::make((unsigned)a->Vc);
    d.tc = FastDiv::make((unsigned)a->Tc);
    d.per_out = FastDiv::make((unsigned)((long)a->N * a->Tc * a->V));
    d.per_in = FastDiv::make((unsigned)((long)a->N * a->Tc * a->Vc));
    return d;
}

}  // namespace

extern "C" int kg_gen_expand(const KgGenArgs* a, void* stream) {
    if (int rc = validate(a, "kg_gen_expand", false)) return rc;
    const long items = (long)((a->z ? a->C : 0) + (a->r ? a->Cr : 0)) * a->N * a->Tc * a->V;
    hipLaunchKernelGGL(kg_gen_expand_kernel, dim3(grid_for(a, items)), dim3(NT), 0, (hipStream_t)stream, *a, divs_of(a));
    return kg_launch_status("kg_gen_expand");
}

extern "C" int kg_gen_fold(const KgGenArgs* a, void* stream) {
    if (int rc = validate(a, "kg_gen_fold", true)) return rc;
    const long items = (long)((a->z ? a->C : 0) + (a->r ? a->Cr : 0)) * a->N * a->Tc * a->Vc +
                       (a->zf ? (long)a->C * a->N * a->Tc * a->V : 0);
    hipLaunchKernelGGL(kg_gen_fold_kernel, dim3(grid_for(a, items)), dim3(NT), 0, (hipStream_t)stream, *a, divs_of(a));
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

extern "C" int kg_gen_adj_prepare(const KgGenPrepJob* jobs, int32_t njobs, void* stream) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= KG_GEN_ADJ_MAX_JOBS, "kg_gen_adj_prepare: 1..%d jobs", KG_GEN_ADJ_MAX_JOBS);
    PrepJobs js;
    js.njobs = njobs;
    for (int i = 0; i < njobs; ++i) {
        const KgGenPrepJob& j = jobs[i];
        KG_REQUIRE(j.K >= 1 && j.K <= KMAX && j.V >= 1 && j.V <= VMAX && j.Vc >= 1 && j.Vc <= VMAX, "kg_gen_adj_prepare: job %d dims", i);
        KG_REQUIRE(j.a && j.aeff && j.b, "kg_gen_adj_prepare: job %d null pointer", i);
        KG_REQUIRE(j.u != nullptr || j.Vc == j.V, "kg_gen_adj_prepare: job %d Vc != V without U", i);
        js.job[i] = j;
    }
    hipLaunchKernelGGL(kg_gen_adj_prepare_kernel, dim3(njobs), dim3(NT), 0, (hipStream_t)stream, js);
    return kg_launch_status("kg_gen_adj_prepare");
}

extern "C" int64_t kg_gen_tail_workspace_bytes(const KgGenTailArgs* a) {
    if (validate_tail(a, "kg_gen_tail_workspace_bytes")) return -1;
    const long P = ((long)a->N * a->T * a->V + TAIL_CHUNK - 1) / TAIL_CHUNK;
    return (int64_t)a->C * P * 4 * (int64_t)sizeof(float);
}

extern "C" int kg_gen_tail_stats(const KgGenTailArgs* a, void* stream) {
    if (int rc = validate_tail(a, "kg_gen_tail_stats")) return rc;
    const long P = ((long)a->N * a->T * a->V + TAIL_CHUNK - 1) / TAIL_CHUNK;
    KG_REQUIRE(a->ws && a->ws_bytes >= kg_gen_tail_workspace_bytes(a), "kg_gen_tail_stats: workspace too small");
    KG_REQUIRE(a->counters && a->counters_len >= a->C, "kg_gen_tail_stats: %d zeroed counters needed", a->C);
    KG_REQUIRE((long)a->C * P < (1L << 31), "kg_gen_tail_stats: grid too large");
    TailPlan pl;
    pl.P = (int)P;
    pl.l = FastDiv::make((unsigned)(a->T * a->V));
    hipLaunchKernelGGL(kg_gen_tail_stats_kernel, dim3((int)(a->C * P)), dim3(NT), 0, (hipStream_t)stream, *a, pl);
    return kg_launch_status("kg_gen_tail_stats");
}

extern "C" int kg_gen_tail_apply(const KgGenTailArgs* a, void* stream) {
    if (int rc = validate_tail(a, "kg_gen_tail_apply")) return rc;
    KG_REQUIRE(a->du, "kg_gen_tail_apply: null du");
    dim3 grid(kg_cdiv((long)a->N * a->T * a->V, 4 * NT), a->C);
    hipLaunchKernelGGL(kg_gen_tail_apply_kernel, grid, dim3(NT), 0, (hipStream_t)stream, *a, FastDiv::make((unsigned)(a->T * a->V)));
    return kg_launch_status("kg_gen_tail_apply");
}
