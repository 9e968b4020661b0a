// Container-level fusions around the discriminator's six blocks (discriminator.py:52-74 and the
// critic step of kinetic-gan.py:94-114,137-155).  Each kernel replaces a run of small stock launches:
//
//   kg_head_fwd / kg_head_bwd / kg_head_wgrad   global average pool + Linear(latent, 1) (discriminator.py:68-72), the
//                                               top gradient of the backward pass with the last block's LeakyReLU
//                                               derivative applied, and the Linear's weight / bias gradients
//   kg_label_bias_fwd / _bwd                    the label channels of block 0 (discriminator.py:57-60: class embedding
//                                               broadcast over (t, v), concatenated in front of x) as a per-sample bias
//                                               of the gcn output and its three gradients
//   kg_mix3                                     [real | fake | alpha real + (1 - alpha) fake] (kinetic-gan.py:97-99) as
//                                               ONE (3n, C, T, V) tensor: the critic runs D on all three at once
//   kg_masked_adj_fwd / _bwd                    (A[lvl] * edge_importance)[kept columns] of all blocks, packed
//                                               (discriminator.py:63-64) and d edge_importance
//
// All HBM / latency bound, a few hundred KB each; deterministic (no atomics: every output element has one owner).
#include "kg_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum, result valid in thread 0 (red: NT / 64 floats of LDS)
__device__ __forceinline__ float block_sum(float s, float* red) {
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// ---- head -------------------------------------------------------------------------------------------------------
// v[n] = b + sum_c w[c] * mean_{t,v} h[n,c,t,v]: one workgroup per sample
__global__ __launch_bounds__(NT) void kg_head_fwd_kernel(const KgHeadArgs a) {
    __shared__ float red[NT / 64];
    const int n = blockIdx.x, L = a.T * a.V;
    const float* base = a.h + (long)n * a.h_sN;
    float s = 0.f;
    for (int e = threadIdx.x; e < a.C * L; e += NT) {
        const int c = e / L, r = e - c * L;
        s = fmaf(a.w[c], base[(long)c * a.h_sC + r], s);
    }
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) a.v[n] = t / (float)L + (a.b ? a.b[0] : 0.f);
}

// g[n,c,t,v] = gv[n] * w[c] / (T V) * (masked ? lrelu'(h[n,c,t,v]) : 1)
__global__ __launch_bounds__(NT) void kg_head_bwd_kernel(const KgHeadArgs a) {
    const int L = a.T * a.V;
    const long total = (long)a.N * a.C * L;
    const float inv = 1.f / (float)L;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        // channel-major item order: (c, n, r)
        const int c = (int)(i / ((long)a.N * L));
        const long rem = i - (long)c * a.N * L;
        const int n = (int)(rem / L), r = (int)(rem - (long)n * L);
        float v = a.gv[n] * a.w[c] * inv;
        if (a.masked) v *= a.h[(long)n * a.h_sN + (long)c * a.h_sC + r] > 0.f ? 1.f : a.slope;
        a.g[(long)n * a.g_sN + (long)c * a.g_sC + r] = v;
    }
}

// dw[c] (+)= sum_n gv[n] * mean_{t,v} x[n,c,t,v] (one workgroup per channel); db (+)= sum_n gv[n] (workgroup C)
__global__ __launch_bounds__(NT) void kg_head_wgrad_kernel(const KgHeadArgs a) {
    __shared__ float red[NT / 64];
    const int c = blockIdx.x, L = a.T * a.V;
    float s = 0.f;
    if (c < a.C) {
        for (int e = threadIdx.x; e < a.N * L; e += NT) {
            const int n = e / L, r = e - n * L;
            s = fmaf(a.gv[n], a.h[(long)n * a.h_sN + (long)c * a.h_sC + r], s);
        }
        const float t = block_sum(s, red);
        if (threadIdx.x == 0) a.dw[c] = (a.accumulate ? a.dw[c] : 0.f) + t / (float)L;
    } else {
        for (int n = threadIdx.x; n < a.N; n += NT) s += a.gv[n];
        const float t = block_sum(s, red);
        if (threadIdx.x == 0 && a.db) a.db[0] = (a.accumulate ? a.db[0] : 0.f) + t;
    }
}

// ---- label bias ---------------------------------------------------------------------------------------------------
constexpr int LB_MAXKW = 3 * 32, LB_MAXKC = 3 * 64, LB_MAXCW = 64 * 32, LB_MAXW = 12288;

// S[k][w] = sum_v A[k][v][w] into LDS (K * W <= LB_MAXKW).  The adjacency goes through LDS first (coalesced, every load
// of a thread in flight at once): summing straight from global memory is a chain of V dependent loads per thread (12 us).
__device__ __forceinline__ void colsums(const KgLabelBiasArgs& a, float* S, float* scratch) {
    const int n = a.K * a.V * a.W;
    for (int i = threadIdx.x; i < n; i += NT) scratch[i] = a.ak[i];
    __syncthreads();
    for (int i = threadIdx.x; i < a.K * a.W; i += NT) {
        const int k = i / a.W, w = i - k * a.W;
        float s = 0.f;
        for (int v = 0; v < a.V; ++v) s += scratch[(k * a.V + v) * a.W + w];
        S[i] = s;
    }
    __syncthreads();
}

// the label columns of the gcn weight, Wl[(k*C + c)*J + j] = Wc(k,c,j), into LDS with coalesced row reads (a thread-per-
// (k,c) loop over j straight from global memory is a chain of J dependent L2 round trips: 22 us per launch)
__device__ __forceinline__ void stage_wc(const KgLabelBiasArgs& a, float* Wl) {
    // a wave takes four rows (k, c) at a time: their loads are all in flight before the first LDS store
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = a.K * a.C;
    for (int r0 = wave * 4; r0 < rows; r0 += (NT / 64) * 4) {
        for (int j0 = 0; j0 < a.J; j0 += 64) {
            const int j = j0 + lane;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kc = r0 + q;
                const int k = kc / a.C, c = kc - k * a.C;              // (wave-uniform)
                v[q] = (kc < rows && j < a.J) ? a.w[(long)k * a.w_sK + (long)c * a.w_sC + j] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + q < rows && j < a.J) Wl[(r0 + q) * a.J + j] = v[q];
        }
    }
}

// forward, phase 1 - one workgroup per CLASS: table[l,c,w] = sum_k S[k,w] * P_l[k,c],  P_l[k,c] = sum_j Wc(k,c,j) * E[l,j]
// (the bias depends on the sample only through its class: L classes instead of N = 3n samples stage the weight)
__global__ __launch_bounds__(NT) void kg_label_bias_table_kernel(const KgLabelBiasArgs a, float* table) {
    __shared__ float S[LB_MAXKW], P[LB_MAXKC], El[512], Wl[LB_MAXW];
    const int l = blockIdx.x;
    colsums(a, S, Wl);             // (Wl is free until the weights are staged)
    stage_wc(a, Wl);
    for (int j = threadIdx.x; j < a.J; j += NT) El[j] = a.emb[(long)l * a.J + j];
    __syncthreads();
    for (int i = threadIdx.x; i < a.K * a.C; i += NT) {
        const float* wp = Wl + i * a.J;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // four chains: the LDS reads of a step overlap
        int j = 0;
        for (; j + 3 < a.J; j += 4) {
            s0 = fmaf(wp[j], El[j], s0);
            s1 = fmaf(wp[j + 1], El[j + 1], s1);
            s2 = fmaf(wp[j + 2], El[j + 2], s2);
            s3 = fmaf(wp[j + 3], El[j + 3], s3);
        }
        for (; j < a.J; ++j) s0 = fmaf(wp[j], El[j], s0);
        P[i] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.C * a.W; i += NT) {
        const int c = i / a.W, w = i - c * a.W;
        float s = 0.f;
        for (int k = 0; k < a.K; ++k) s = fmaf(S[k * a.W + w], P[k * a.C + c], s);
        table[(long)l * a.C * a.W + i] = s;
    }
}

// phase 2 - zl[n] = table[label_n]
__global__ __launch_bounds__(NT) void kg_label_bias_lookup_kernel(const KgLabelBiasArgs a, const float* table) {
    const int CW = a.C * a.W;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= a.N * CW) return;
    const int n = i / CW, e = i - n * CW;
    // a label outside [0, L) poisons its sample with NaN instead of reading past the table (the reference's nn.Embedding
    // raises, discriminator.py:57; there is no way to raise from a kernel, and the host must not synchronise here)
    const long lab = a.labels[n];
    a.zl[i] = (lab >= 0 && lab < a.L) ? table[lab * CW + e] : __builtin_nanf("");
}

// backward, phase 0 - one workgroup per (sample, channel): gzl[n,c,w] = sum_t gz[n,c,t,w] (the T*W run is contiguous;
// lane = (t mod TQ, w) so that a wave reads consecutive addresses), into ws behind the per-class records
__global__ __launch_bounds__(NT) void kg_label_bias_bwd0_kernel(const KgLabelBiasArgs a, float* gzl) {
    __shared__ float red[NT];
    const int n = blockIdx.x / a.C, c = blockIdx.x - n * a.C;
    const int W = a.W, TQ = NT / W;                      // frames summed side by side
    const int tid = threadIdx.x;
    const int tq = tid / W, w = tid - tq * W;
    const float* gp = a.gz + (long)n * a.gz_sN + (long)c * a.gz_sC;
    float s = 0.f;
    if (tq < TQ)
        for (int t = tq; t < a.T; t += TQ) s += gp[t * W + w];
    red[tid] = s;
    __syncthreads();
    if (tid < W) {
        float t = 0.f;
        for (int q = 0; q < TQ; ++q) t += red[q * W + tid];
        gzl[((long)n * a.C + c) * W + tid] = t;
    }
}

// phase 1 - one workgroup per CLASS l (its samples are visited in index order: deterministic):
//   dT[c,w] = sum_{n: label_n = l} gzl[n,c,w];  Q[l,k,c] = sum_w dT[c,w] S[k,w];  R[l,k,w] = sum_c dT[c,w] P_l[k,c]
//   dE[l,j] (+)= sum_{k,c} Wc(k,c,j) Q[l,k,c]
__global__ __launch_bounds__(NT) void kg_label_bias_bwd1_kernel(const KgLabelBiasArgs a, const float* gzl) {
    __shared__ float S[LB_MAXKW], P[LB_MAXKC], Q[LB_MAXKC], dT[LB_MAXCW], El[512], Wl[LB_MAXW];
    const int l = blockIdx.x;
    colsums(a, S, Wl);
    stage_wc(a, Wl);
    for (int j = threadIdx.x; j < a.J; j += NT) El[j] = a.emb[(long)l * a.J + j];
    const int CW = a.C * a.W;
    // the labels go through LDS, a chunk at a time: read from global memory inside the sample loop every sample was a
    // (uniform) load of its own in front of the branch - 128 dependent latencies, 28 us for a 45 KB problem
    __shared__ int Lb[1024];
    constexpr int DQ = LB_MAXCW / NT;
    float dacc[DQ];
#pragma unroll
    for (int q = 0; q < DQ; ++q) dacc[q] = 0.f;
    for (int n0 = 0; n0 < a.N; n0 += 1024) {
        const int nn = a.N - n0 < 1024 ? a.N - n0 : 1024;
        __syncthreads();
        for (int i = threadIdx.x; i < nn; i += NT) Lb[i] = (int)a.labels[n0 + i];
        __syncthreads();
        for (int n = 0; n < nn; ++n) {
            if (Lb[n] != l) continue;                                   // (uniform across the workgroup)
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                const int i = threadIdx.x + q * NT;
                if (i < CW) dacc[q] += gzl[(long)(n0 + n) * CW + i];      // samples in index order: deterministic
            }
        }
    }
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
        const int i = threadIdx.x + q * NT;
        if (i < CW) dT[i] = dacc[q];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.K * a.C; i += NT) {
        const float* wp = Wl + i * a.J;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // four chains: the LDS reads of a step overlap
        int j = 0;
        for (; j + 3 < a.J; j += 4) {
            s0 = fmaf(wp[j], El[j], s0);
            s1 = fmaf(wp[j + 1], El[j + 1], s1);
            s2 = fmaf(wp[j + 2], El[j + 2], s2);
            s3 = fmaf(wp[j + 3], El[j + 3], s3);
        }
        for (; j < a.J; ++j) s0 = fmaf(wp[j], El[j], s0);
        P[i] = (s0 + s1) + (s2 + s3);
    }
    float* Qg = a.ws + (long)l * (a.K * a.C + a.K * a.W);
    for (int i = threadIdx.x; i < a.K * a.C; i += NT) {
        const int k = i / a.C, c = i - k * a.C;
        float s = 0.f;
        for (int w = 0; w < a.W; ++w) s = fmaf(dT[c * a.W + w], S[k * a.W + w], s);
        Q[i] = s;
        Qg[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.K * a.W; i += NT) {
        const int k = i / a.W, w = i - k * a.W;
        float s = 0.f;
        for (int c = 0; c < a.C; ++c) s = fmaf(dT[c * a.W + w], P[k * a.C + c], s);
        Qg[a.K * a.C + i] = s;
    }
    if (a.demb)
        for (int j = threadIdx.x; j < a.J; j += NT) {
            float s = 0.f;
#pragma unroll 8
            for (int i = 0; i < a.K * a.C; ++i) s = fmaf(Wl[i * a.J + j], Q[i], s);
            float* d = a.demb + (long)l * a.J + j;
            *d = (a.accumulate ? *d : 0.f) + s;
        }
}

// phase 2 - thread per output: dWc(k,c,j) (+)= sum_l E[l,j] Q[l,k,c];  dak[k,v,w] (+)= sum_l R[l,k,w] for every v
__global__ __launch_bounds__(NT) void kg_label_bias_bwd2_kernel(const KgLabelBiasArgs a) {
    const int i = blockIdx.x * NT + threadIdx.x;
    const int nw = a.K * a.C * a.J, na = a.K * a.V * a.W;
    const int per = a.K * a.C + a.K * a.W;
    if (i < nw) {
        if (!a.dw) return;
        const int kc = i / a.J, j = i - kc * a.J;
        float s = 0.f;
#pragma unroll 8
        for (int l = 0; l < a.L; ++l) s = fmaf(a.emb[(long)l * a.J + j], a.ws[(long)l * per + kc], s);
        const int k = kc / a.C, c = kc - k * a.C;
        float* d = a.dw + (long)k * a.w_sK + (long)c * a.w_sC + j;
        *d = (a.accumulate ? *d : 0.f) + s;
    } else if (i < nw + na) {
        if (!a.dak) return;
        const int e = i - nw;
        const int k = e / (a.V * a.W), w = e % a.W;
        float s = 0.f;
#pragma unroll 8
        for (int l = 0; l < a.L; ++l) s += a.ws[(long)l * per + a.K * a.C + k * a.W + w];
        a.dak[e] = (a.dak_accumulate ? a.dak[e] : 0.f) + s;
    }
}

// ---- critic input ------------------------------------------------------------------------------------------------
// out[0:n] = real, out[n:2n] = fake, out[2n:3n] = alpha real + (1 - alpha) fake   (kinetic-gan.py:97-99,146-148)
__global__ __launch_bounds__(NT) void kg_mix3_kernel(const KgMixArgs a) {
    const int L = a.T * a.V;
    const long per = (long)a.C * L;
    const long total = (long)a.N * per;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int n = (int)(i / per);
        const long rem = i - (long)n * per;
        const int c = (int)(rem / L), r = (int)(rem - (long)c * L);
        const float x = a.real[(long)n * a.r_sN + (long)c * a.r_sC + r];
        const float f = a.fake[(long)n * a.f_sN + (long)c * a.f_sC + r];
        const float al = a.alpha[n];
        float* o = a.out + (long)c * a.o_sC + r;
        o[(long)n * a.o_sN] = x;
        o[(long)(n + a.N) * a.o_sN] = f;
        o[(long)(n + 2 * a.N) * a.o_sN] = al * x + (1.f - al) * f;
    }
}

// ---- masked, kept-column adjacencies -----------------------------------------------------------------------------
// fwd: ak[i] = A[s] * imp[s], s = sel ? sel[i] : i;   bwd: dimp[s] (+)= g[i] * A[s]   (sel is injective)
__global__ __launch_bounds__(NT) void kg_masked_adj_kernel(const KgMaskedAdjArgs a, const int backward) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= a.n) return;
    const long s = a.sel ? a.sel[i] : i;
    if (!backward) {
        a.ak[i] = a.a[s] * (a.imp ? a.imp[s] : 1.f);
    } else {
        const float v = a.g[i] * a.a[s];
        a.dimp[s] = (a.accumulate ? a.dimp[s] : 0.f) + v;
    }
}

int validate_head(const KgHeadArgs* a, const char* who) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0, "%s: bad dims", who);
    KG_REQUIRE((long)a->N * a->C * a->T * a->V < (1L << 31), "%s: too large", who);
    return 0;
}

int validate_lb(const KgLabelBiasArgs* a, const char* who) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->L > 0 && a->J > 0 && a->K >= 1 && a->K <= 3 && a->C > 0 && a->V > 0 && a->W > 0, "%s: bad dims", who);
    KG_REQUIRE(a->K * a->W <= LB_MAXKW && a->K * a->C <= LB_MAXKC && a->C * a->W <= LB_MAXCW && a->J <= 512 &&
               a->K * a->C * a->J <= LB_MAXW && a->W <= NT && a->K * a->V * a->W <= LB_MAXW,
               "%s: K=%d C=%d W=%d J=%d exceed the kernel's LDS tables", who, a->K, a->C, a->W, a->J);
    KG_REQUIRE(a->labels && a->emb && a->w && a->ak, "%s: null pointer", who);
    return 0;
}

}  // namespace

extern "C" int kg_head_fwd(const KgHeadArgs* a, void* stream) {
    if (int rc = validate_head(a, "kg_head_fwd")) return rc;
    KG_REQUIRE(a->h && a->w && a->v, "kg_head_fwd: null pointer");
    hipLaunchKernelGGL(kg_head_fwd_kernel, dim3(a->N), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_head_fwd");
}

extern "C" int kg_head_bwd(const KgHeadArgs* a, void* stream) {
    if (int rc = validate_head(a, "kg_head_bwd")) return rc;
    KG_REQUIRE(a->gv && a->w && a->g && (!a->masked || a->h), "kg_head_bwd: null pointer");
    const long total = (long)a->N * a->C * a->T * a->V;
    long grid = (total + NT - 1) / NT;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(kg_head_bwd_kernel, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_head_bwd");
}

extern "C" int kg_head_wgrad(const KgHeadArgs* a, void* stream) {
    if (int rc = validate_head(a, "kg_head_wgrad")) return rc;
    KG_REQUIRE(a->gv && a->h && a->dw, "kg_head_wgrad: null pointer");
    hipLaunchKernelGGL(kg_head_wgrad_kernel, dim3(a->C + 1), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_head_wgrad");
}

extern "C" int kg_label_bias_fwd(const KgLabelBiasArgs* a, void* stream) {
    if (int rc = validate_lb(a, "kg_label_bias_fwd")) return rc;
    KG_REQUIRE(a->zl, "kg_label_bias_fwd: null zl");
    KG_REQUIRE(a->ws && a->ws_bytes >= (int64_t)a->L * a->C * a->W * (int64_t)sizeof(float),
               "kg_label_bias_fwd: workspace too small (kg_label_bias_workspace_bytes)");
    hipLaunchKernelGGL(kg_label_bias_table_kernel, dim3(a->L), dim3(NT), 0, (hipStream_t)stream, *a, a->ws);
    if (int rc = kg_launch_status("kg_label_bias_fwd (class table)")) return rc;
    hipLaunchKernelGGL(kg_label_bias_lookup_kernel, dim3(kg_cdiv((long)a->N * a->C * a->W, NT)), dim3(NT), 0, (hipStream_t)stream,
                       *a, (const float*)a->ws);
    return kg_launch_status("kg_label_bias_fwd (lookup)");
}

extern "C" int64_t kg_label_bias_workspace_bytes(const KgLabelBiasArgs* a) {
    if (validate_lb(a, "kg_label_bias_workspace_bytes")) return -1;
    // backward: per-class records (Q, R) + the frame-summed gradient (N, C, W); forward: the class table (L, C, W)
    const int64_t bwd = (int64_t)a->L * (a->K * a->C + a->K * a->W) + (int64_t)a->N * a->C * a->W;
    const int64_t fwd = (int64_t)a->L * a->C * a->W;
    return (bwd > fwd ? bwd : fwd) * (int64_t)sizeof(float);
}

extern "C" int kg_label_bias_bwd(const KgLabelBiasArgs* a, void* stream) {
    if (int rc = validate_lb(a, "kg_label_bias_bwd")) return rc;
    KG_REQUIRE(a->gz && a->T > 0, "kg_label_bias_bwd: null gz");
    KG_REQUIRE(a->ws && a->ws_bytes >= kg_label_bias_workspace_bytes(a), "kg_label_bias_bwd: workspace too small");
    float* gzl = a->ws + (int64_t)a->L * (a->K * a->C + a->K * a->W);
    hipLaunchKernelGGL(kg_label_bias_bwd0_kernel, dim3(a->N * a->C), dim3(NT), 0, (hipStream_t)stream, *a, gzl);
    if (int rc = kg_launch_status("kg_label_bias_bwd (frame sums)")) return rc;
    hipLaunchKernelGGL(kg_label_bias_bwd1_kernel, dim3(a->L), dim3(NT), 0, (hipStream_t)stream, *a, (const float*)gzl);
    if (int rc = kg_launch_status("kg_label_bias_bwd (classes)")) return rc;
    const int items = a->K * a->C * a->J + a->K * a->V * a->W;
    hipLaunchKernelGGL(kg_label_bias_bwd2_kernel, dim3(kg_cdiv(items, NT)), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_label_bias_bwd (finish)");
}

extern "C" int kg_mix3(const KgMixArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr && a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0, "kg_mix3: bad dims");
    KG_REQUIRE(a->real && a->fake && a->alpha && a->out, "kg_mix3: null pointer");
    const long total = (long)a->N * a->C * a->T * a->V;
    long grid = (total + NT - 1) / NT;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(kg_mix3_kernel, dim3((int)grid), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_mix3");
}

extern "C" int kg_masked_adj_fwd(const KgMaskedAdjArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr && a->n > 0 && a->a && a->ak, "kg_masked_adj_fwd: bad args");
    hipLaunchKernelGGL(kg_masked_adj_kernel, dim3(kg_cdiv(a->n, NT)), dim3(NT), 0, (hipStream_t)stream, *a, 0);
    return kg_launch_status("kg_masked_adj_fwd");
}

extern "C" int kg_masked_adj_bwd(const KgMaskedAdjArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr && a->n > 0 && a->a && a->g && a->dimp, "kg_masked_adj_bwd: bad args");
    hipLaunchKernelGGL(kg_masked_adj_kernel, dim3(kg_cdiv(a->n, NT)), dim3(NT), 0, (hipStream_t)stream, *a, 1);
    return kg_launch_status("kg_masked_adj_bwd");
}

