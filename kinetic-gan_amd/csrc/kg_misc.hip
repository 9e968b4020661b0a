// Per-channel reductions, pointwise epilogues, flat-buffer Adam and the library's info/error
// entry points (see include/kgan_hip.h for the reference call sites each one replaces).
#include <stdarg.h>
#include <stdio.h>

#include "kg_common.h"

// ---- error string ---------------------------------------------------------------------------
static thread_local char kg_err_buf[512] = "";

void kg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(kg_err_buf, sizeof(kg_err_buf), fmt, ap);
    va_end(ap);
}

extern "C" const char* kg_last_error(void) { return kg_err_buf; }

// ---- test / tuning switches -------------------------------------------------------------------------
#include <stdlib.h>
static int kg_env_tri(const char* name) {
    const char* e = getenv(name);
    if (e && e[0] == '0') return 0;
    if (e && e[0] == '1') return 1;
    return -1;
}
static int kg_env_int(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}
static KgEnv kg_env_read() {
    KgEnv v;
    v.conv_plan_tile = -1;
    v.conv_plan_split = 0;
    if (const char* e = getenv("KG_CONV_PLAN")) {
        int t = -1, ns = 0;
        if (sscanf(e, "%d,%d", &t, &ns) >= 1) { v.conv_plan_tile = t; v.conv_plan_split = ns; }
    }
    v.conv_kw = kg_env_tri("KG_CONV_KW");
    v.conv_tiny = kg_env_tri("KG_CONV_TINY");
    v.conv_fast = kg_env_tri("KG_CONV_FAST");
    v.conv_many = kg_env_tri("KG_CONV_MANY");
    v.conv_xcd_min = kg_env_int("KG_CONV_XCD_MIN");
    v.agg_stream = kg_env_tri("KG_AGG_STREAM");
    v.agg_mfma = kg_env_tri("KG_AGG_MFMA");
    v.agg_mfma_sub = kg_env_int("KG_AGG_MFMA_SUB");
    v.agg_mfma_grid = kg_env_int("KG_AGG_MFMA_GRID");
    v.agg_outer_mfma = kg_env_tri("KG_AGG_OUTER_MFMA");
    v.agg_outer_budget = kg_env_int("KG_AGG_OUTER_BUDGET");
    v.wgrad_budget = kg_env_int("KG_WGRAD_BUDGET");
    v.wgrad_bigcols = getenv("KG_WGRAD_BIGCOLS") ? kg_env_int("KG_WGRAD_BIGCOLS") : -1;
    v.wgrad_split = getenv("KG_WGRAD_SPLIT") ? kg_env_int("KG_WGRAD_SPLIT") : 0;
    v.aggconv_plan = kg_env_int("KG_AGGCONV_PLAN");
    v.conv_ring = kg_env_tri("KG_CONV_RING");
    v.conv_ring_stagger = kg_env_int("KG_CONV_RING_STAGGER");
    v.conv_ring_tile = getenv("KG_CONV_RING_TILE") ? kg_env_int("KG_CONV_RING_TILE") : -1;
    v.gb_rt = kg_env_int("KG_GB_RT");
    v.conv_bs = kg_env_tri("KG_CONV_BS");
    if (const char* e = getenv("KG_CONV_BS")) if (e[0] == '2') v.conv_bs = 2;      // 2: the round-5 plan rule (bs_auto_rule)
    v.conv_bs_asm = kg_env_tri("KG_CONV_BS_ASM");
    v.conv_plain_epi = kg_env_tri("KG_CONV_PLAIN_EPI");
    v.conv_inkernel = kg_env_tri("KG_CONV_INKERNEL");
    v.conv_inkernel_max = kg_env_int("KG_CONV_INKERNEL_MAX");
    v.conv_bs_tile = getenv("KG_CONV_BS_TILE") ? kg_env_int("KG_CONV_BS_TILE") : -1;
    return v;
}
static KgEnv kg_env_value = kg_env_read();       // library load
const KgEnv& kg_env() { return kg_env_value; }
extern "C" void kg_reload_env(void) { kg_env_value = kg_env_read(); }
extern "C" int kg_abi_version(void) { return KG_ABI_VERSION; }
extern "C" const char* kg_arch(void) { return "gfx950"; }

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// walks the (n, t*V + v) elements of one channel with stride NT without an integer division per element
struct ColWalk {
    int n, r, dn, dr, L;
    __device__ __forceinline__ ColWalk(long j, int L_, int stride) : L(L_) {
        n = (int)(j / L_);
        r = (int)(j - (long)n * L_);
        dn = stride / L_;
        dr = stride - dn * L_;
    }
    __device__ __forceinline__ void next() {
        r += dr;
        n += dn;
        if (r >= L) { r -= L; ++n; }
    }
};

// block-wide sums of two values (THREADS per block); result valid in every thread
template <int THREADS = NT>
__device__ __forceinline__ void block_sum2(float& s0, float& s1, float (*red)[THREADS / 64]) {
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    __syncthreads();                     // red may still be read from a previous call
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { t0 += red[0][w]; t1 += red[1][w]; }
    s0 = t0; s1 = t1;
}

// ---- rowsum: grid (P, C); partial [2][C][P] then a finishing kernel; P == 1 finishes in place ---------------
constexpr int RS_CHUNK = 4096;   // columns per workgroup

__device__ __forceinline__ void rowsum_store(const KgRowsumArgs& a, int idx, float s) {
    a.out[idx] = a.accumulate ? a.out[idx] + s : s;
    if (a.out2) a.out2[idx] = a.accumulate ? a.out2[idx] + s : s;
}

__device__ __forceinline__ void rowsum_part(const KgRowsumArgs& a, const int P, const int p, const int c) {
    __shared__ float red[2][NT / 64];
    const int tid = threadIdx.x;
    const int L = a.T * a.V;
    const long ncols = (long)a.N * L;
    const long jbeg = (long)p * RS_CHUNK;
    const long jend = jbeg + RS_CHUNK < ncols ? jbeg + RS_CHUNK : ncols;
    float s0 = 0.f, s1 = 0.f;
    const float sh = a.shift ? a.shift[c] : 0.f;
    const float* xp = a.x + (long)c * a.x_sC;
    const float* yp = a.y ? a.y + (long)c * a.y_sC : nullptr;
    ColWalk w(jbeg + tid, L, NT);
    for (long j = jbeg + tid; j < jend; j += NT, w.next()) {
        const float xv = xp[(long)w.n * a.x_sN + w.r];
        s0 += xv;
        if (a.want_second) {
            if (yp) s1 = fmaf(xv, yp[(long)w.n * a.y_sN + w.r] - sh, s1);
            else    s1 = fmaf(xv - sh, xv - sh, s1);
        }
    }
    block_sum2(s0, s1, red);
    if (tid == 0) {
        if (P == 1) {
            if (a.want_second == 2) {
                rowsum_store(a, c, s1);                     // product row only, (1, C)
            } else {
                rowsum_store(a, c, s0);
                if (a.want_second) rowsum_store(a, a.C + c, s1);
            }
        } else {
            a.ws[((long)0 * a.C + c) * P + p] = s0;
            a.ws[((long)1 * a.C + c) * P + p] = s1;
        }
    }
}

__global__ __launch_bounds__(NT) void kg_rowsum_kernel(const KgRowsumArgs a, int P) {
    rowsum_part(a, P, blockIdx.x, blockIdx.y);
}

__device__ __forceinline__ void rowsum_finish_row(const KgRowsumArgs& a, const int P, const int idx) {
    // one wave per (which, c); idx = which*C + c
    const int nrow = a.want_second == 1 ? 2 : 1;
    if (idx >= nrow * a.C) return;
    const long src = a.want_second == 2 ? (long)a.C + idx : idx;     // product row only: partial row 1 -> out row 0
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += 64) s += a.ws[src * P + p];
    s = wave_sum(s);
    if (threadIdx.x == 0) rowsum_store(a, idx, s);
}

__global__ __launch_bounds__(64) void kg_rowsum_finish(const KgRowsumArgs a, int P) {
    rowsum_finish_row(a, P, blockIdx.x);
}

// Several per-channel reductions (the bias gradients of all convs of a backward pass, 6-25 of them, each a few
// microseconds) in one launch + one finishing launch.
constexpr int RS_MANY_MAX = 24;
struct RowsumMany { int njobs; int beg[RS_MANY_MAX + 1]; int P[RS_MANY_MAX]; KgRowsumArgs job[RS_MANY_MAX]; };

__global__ __launch_bounds__(NT) void kg_rowsum_many_kernel(const RowsumMany m) {
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.beg[ji + 1]) ++ji;      // (uniform)
    const int local = blockIdx.x - m.beg[ji];
    const int P = m.P[ji];
    rowsum_part(m.job[ji], P, local % P, local / P);
}

__global__ __launch_bounds__(64) void kg_rowsum_many_finish(const RowsumMany m) {
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.beg[ji + 1]) ++ji;      // (uniform); beg: finish rows here
    rowsum_finish_row(m.job[ji], m.P[ji], blockIdx.x - m.beg[ji]);
}

// ---- BatchNorm2d statistics + coefficients, one workgroup per channel ------------------------------------------
// BT threads per channel: 1024 when a channel has many elements (the generator's last blocks: 3-32 channels of
// 100 k elements each), 256 otherwise
template <int BT>
__global__ __launch_bounds__(BT) void kg_bn_fwd_kernel(const KgBnArgs a) {
    constexpr int NT = BT;
    __shared__ float red[2][NT / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int L = a.T * a.V;
    const long ncols = (long)a.N * L;
    float mean, var;
    if (a.training) {
        const float* xp = a.x + (long)c * a.x_sC;
        float s = 0.f, dummy = 0.f;
        {
            ColWalk w(tid, L, NT);
            for (long j = tid; j < ncols; j += NT, w.next()) s += xp[(long)w.n * a.x_sN + w.r];
        }
        block_sum2<NT>(s, dummy, red);
        mean = s / (float)ncols;
        float q = 0.f;
        {
            ColWalk w(tid, L, NT);
            for (long j = tid; j < ncols; j += NT, w.next()) {
                const float d = xp[(long)w.n * a.x_sN + w.r] - mean;
                q = fmaf(d, d, q);
            }
        }
        block_sum2<NT>(q, dummy, red);
        var = q / (float)ncols;
    } else {
        mean = a.running_mean[c];
        var = a.running_var[c];
    }
    if (tid == 0) {
        const float rstd = 1.f / sqrtf(var + a.eps);
        const float scale = a.gamma ? a.gamma[c] * rstd : rstd;
        const float shift = (a.beta ? a.beta[c] : 0.f) - mean * scale;
        a.coef[0 * a.C + c] = scale;
        a.coef[1 * a.C + c] = shift;
        a.coef[2 * a.C + c] = mean;
        a.coef[3 * a.C + c] = rstd;
        if (a.training && a.running_mean) {
            // momentum < 0: torch's momentum=None, the cumulative moving average with factor 1 / (batches seen so far,
            // this one included) - formed from the LIVE counter, which a trailing one-thread launch increments (all
            // channels' workgroups read the same, not yet incremented, value)
            const float m = a.momentum >= 0.f ? a.momentum
                                              : 1.f / (float)((a.num_batches_tracked ? *a.num_batches_tracked : 0) + 1);
            const float unb = var * ((float)ncols / (float)(ncols > 1 ? ncols - 1 : 1));
            a.running_mean[c] = (1.f - m) * a.running_mean[c] + m * mean;
            a.running_var[c] = (1.f - m) * a.running_var[c] + m * unb;
        }
        if (a.training && a.num_batches_tracked && c == 0 && a.momentum >= 0.f) *a.num_batches_tracked += 1;
    }
}

__global__ void kg_bn_count_kernel(int64_t* num_batches_tracked) { *num_batches_tracked += 1; }

// ---- BatchNorm2d statistics of SEVERAL layers / stacked batches in one launch ------------------------------------
// The generator's paired synthesis (two batches stacked along N, separate statistics) has up to two BatchNorm layers
// per block: four kg_bn_fwd launches with one workgroup per channel - and the last blocks have 3 channels of 100 k
// elements each, i.e. three busy CUs.  Here a workgroup takes a chunk of BN_CHUNK elements of one (layer, channel,
// batch): its element count, mean and centred sum of squares (the values stay in registers between the two passes).
// The last workgroup of a (layer, channel) to arrive - a ticket counter, left at zero again - merges the partials in
// chunk order (Chan et al., deterministic, as accurate as the two-pass form) and writes the coefficients and the
// running-statistics updates of all stacked batches, in batch order.
constexpr int BN_CHUNK = 4096;
constexpr int BN_MANY_MAX = 4;
constexpr int BN_PART_LDS = 2048;      // partials the merging workgroup stages in LDS
struct BnMany { int njobs; int beg[BN_MANY_MAX + 1]; int P[BN_MANY_MAX]; int cbeg[BN_MANY_MAX]; long wbeg[BN_MANY_MAX]; KgBnJob job[BN_MANY_MAX]; float* ws; int* counters; };

__global__ __launch_bounds__(NT) void kg_bn_fwd_many_kernel(const BnMany m) {
    __shared__ float red[2][NT / 64];
    __shared__ int last;
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.beg[ji + 1]) ++ji;      // (uniform)
    const KgBnArgs& a = m.job[ji].a;
    const int G = m.job[ji].groups, P = m.P[ji];
    int local = blockIdx.x - m.beg[ji];
    const int p = local % P;
    local /= P;
    const int q = local % G, c = local / G;
    const int tid = threadIdx.x;
    const int L = a.T * a.V;
    const long ncols = (long)a.N * L;
    const long jbeg = (long)p * BN_CHUNK;
    const long jend = jbeg + BN_CHUNK < ncols ? jbeg + BN_CHUNK : ncols;
    const float* xp = a.x + (long)c * a.x_sC + (long)q * a.N * a.x_sN;
    constexpr int PER = BN_CHUNK / NT;
    float v[PER];
    float s = 0.f, dummy = 0.f;
    {
        ColWalk w(jbeg + tid, L, NT);
#pragma unroll
        for (int i = 0; i < PER; ++i, w.next()) {
            const long j = jbeg + tid + (long)i * NT;
            v[i] = j < jend ? xp[(long)w.n * a.x_sN + w.r] : 0.f;
            s += v[i];
        }
    }
    block_sum2<NT>(s, dummy, red);
    const float cnt = (float)(jend - jbeg);
    const float mloc = s / cnt;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const long j = jbeg + tid + (long)i * NT;
        const float d = v[i] - mloc;
        if (j < jend) q2 = fmaf(d, d, q2);
    }
    block_sum2<NT>(q2, dummy, red);
    float* const part = m.ws + m.wbeg[ji] + ((long)c * G * P) * 2;     // [G][P][2] of this channel
    if (tid == 0) {
        // agent-scope (write-through) stores, acknowledged before the ticket is drawn; the merging workgroup reads
        // them with agent-scope loads.  (A __threadfence() here writes back / invalidates the whole L2 per workgroup:
        // measured 32 us for a 1024-workgroup launch.)
        __hip_atomic_store(part + ((long)q * P + p) * 2 + 0, mloc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + ((long)q * P + p) * 2 + 1, q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(m.counters + m.cbeg[ji] + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == G * P - 1);
    }
    __syncthreads();
    if (!last) return;
    // the last arriver: ALL its threads fetch the partials into LDS (one lane reading them one after the other is a
    // chain of 2 G P dependent agent-scope loads: ~8 us for the generator's last blocks), then lane 0 merges in order
    __shared__ float pl[BN_PART_LDS];
    const int npart = G * P * 2;
    const bool staged = npart <= BN_PART_LDS;
    if (staged) {
        for (int i = tid; i < npart; i += NT) pl[i] = __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    if (tid != 0) return;
    for (int g = 0; g < G; ++g) {
        float n_tot = 0.f, mean = 0.f, M2 = 0.f;
        for (int k = 0; k < P; ++k) {
            const long kb = (long)k * BN_CHUNK;
            const float nb = (float)((kb + BN_CHUNK < ncols ? kb + BN_CHUNK : ncols) - kb);
            const float mk = staged ? pl[(g * P + k) * 2 + 0]
                                    : __hip_atomic_load(part + ((long)g * P + k) * 2 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float qk = staged ? pl[(g * P + k) * 2 + 1]
                                    : __hip_atomic_load(part + ((long)g * P + k) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float n_new = n_tot + nb;
            const float delta = mk - mean;
            mean += delta * (nb / n_new);
            M2 += qk + delta * delta * (n_tot * nb / n_new);
            n_tot = n_new;
        }
        const float var = M2 / n_tot;
        const float rstd = 1.f / sqrtf(var + a.eps);
        const float scale = a.gamma ? a.gamma[c] * rstd : rstd;
        const float shift = (a.beta ? a.beta[c] : 0.f) - mean * scale;
        float* coef = a.coef + (long)g * 4 * a.C;
        coef[0 * a.C + c] = scale;
        coef[1 * a.C + c] = shift;
        coef[2 * a.C + c] = mean;
        coef[3 * a.C + c] = rstd;
        if (a.running_mean) {
            const float mo = a.momentum;
            const float unb = var * (n_tot / (n_tot > 1.f ? n_tot - 1.f : 1.f));
            a.running_mean[c] = (1.f - mo) * a.running_mean[c] + mo * mean;
            a.running_var[c] = (1.f - mo) * a.running_var[c] + mo * unb;
        }
    }
    if (a.num_batches_tracked && c == 0) *a.num_batches_tracked += G;
    m.counters[m.cbeg[ji] + c] = 0;
}

template <int BT>
__global__ __launch_bounds__(BT) void kg_bn_bwd_kernel(const KgBnArgs a) {
    constexpr int NT = BT;
    __shared__ float red[2][NT / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int L = a.T * a.V;
    const long ncols = (long)a.N * L;
    const float mean = a.mean[c], rstd = a.rstd[c];
    const float* xp = a.x + (long)c * a.x_sC;
    const float* gp = a.g + (long)c * a.g_sC;
    float s0 = 0.f, s1 = 0.f;
    ColWalk w(tid, L, NT);
    for (long j = tid; j < ncols; j += NT, w.next()) {
        const float gv = gp[(long)w.n * a.g_sN + w.r];
        s0 += gv;
        s1 = fmaf(gv, xp[(long)w.n * a.x_sN + w.r] - mean, s1);
    }
    block_sum2<NT>(s0, s1, red);
    if (tid == 0) {
        const float q = s1 * rstd;                                   // sum g * xhat
        const float ga = (a.gamma ? a.gamma[c] : 1.f) * rstd;
        float b = 0.f, cc = 0.f;
        if (a.training) {
            b = -ga * rstd * q / (float)ncols;
            cc = -ga * s0 / (float)ncols - b * mean;
        }
        a.coef[0 * a.C + c] = ga;
        a.coef[1 * a.C + c] = b;
        a.coef[2 * a.C + c] = cc;
        a.coef[3 * a.C + c] = q;
        a.coef[4 * a.C + c] = s0;
    }
}

// The backward sums of several BatchNorm layers (the two of a generator block share the incoming gradient) in one
// launch, a 4096-element chunk of one (layer, channel) per workgroup and the last workgroup of a channel to arrive
// (ticket counter) adding the partials in chunk order - kg_bn_bwd puts ONE workgroup on every channel, i.e. 3 to 32
// workgroups on the chip for the generator's last blocks (15-23 us per launch).
struct BnBwdMany { int njobs; int beg[BN_MANY_MAX + 1]; int P[BN_MANY_MAX]; int cbeg[BN_MANY_MAX]; long wbeg[BN_MANY_MAX]; KgBnArgs job[BN_MANY_MAX]; float* ws; int* counters; };

__global__ __launch_bounds__(NT) void kg_bn_bwd_many_kernel(const BnBwdMany m) {
    __shared__ float red[2][NT / 64];
    __shared__ int last;
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.beg[ji + 1]) ++ji;      // (uniform)
    const KgBnArgs& a = m.job[ji];
    const int P = m.P[ji];
    const int local = blockIdx.x - m.beg[ji];
    const int p = local % P, c = local / P;
    const int tid = threadIdx.x;
    const int L = a.T * a.V;
    const long ncols = (long)a.N * L;
    const long jbeg = (long)p * BN_CHUNK;
    const long jend = jbeg + BN_CHUNK < ncols ? jbeg + BN_CHUNK : ncols;
    const float mean = a.mean[c], rstd = a.rstd[c];
    const float* xp = a.x + (long)c * a.x_sC;
    const float* gp = a.g + (long)c * a.g_sC;
    float s0 = 0.f, s1 = 0.f;
    {
        // all 2 x 16 loads of the thread in flight before the first use (one by one the loop ran at memory latency)
        constexpr int PER = BN_CHUNK / NT;
        float gv[PER], xv[PER];
        ColWalk w(jbeg + tid, L, NT);
#pragma unroll
        for (int i = 0; i < PER; ++i, w.next()) {
            const bool ok = jbeg + tid + (long)i * NT < jend;
            gv[i] = ok ? gp[(long)w.n * a.g_sN + w.r] : 0.f;
            xv[i] = ok ? xp[(long)w.n * a.x_sN + w.r] : mean;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            s0 += gv[i];
            s1 = fmaf(gv[i], xv[i] - mean, s1);
        }
    }
    block_sum2<NT>(s0, s1, red);
    float* const part = m.ws + m.wbeg[ji] + (long)c * P * 2;
    if (tid == 0) {
        __hip_atomic_store(part + p * 2 + 0, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + p * 2 + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(m.counters + m.cbeg[ji] + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == P - 1);
    }
    __syncthreads();
    if (!last || tid != 0) return;
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < P; ++k) {
        t0 += __hip_atomic_load(part + k * 2 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t1 += __hip_atomic_load(part + k * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float q = t1 * rstd;                                   // sum g * xhat
    const float ga = (a.gamma ? a.gamma[c] : 1.f) * rstd;
    float b = 0.f, cc = 0.f;
    if (a.training) {
        b = -ga * rstd * q / (float)ncols;
        cc = -ga * t0 / (float)ncols - b * mean;
    }
    a.coef[0 * a.C + c] = ga;
    a.coef[1 * a.C + c] = b;
    a.coef[2 * a.C + c] = cc;
    a.coef[3 * a.C + c] = q;
    a.coef[4 * a.C + c] = t0;
    m.counters[m.cbeg[ji] + c] = 0;
}

// ---- pointwise -------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void kg_act_bwd_kernel(const KgEltArgs a) {
    const int c = blockIdx.y;
    const int L = a.T * a.V;
    const long j = (long)blockIdx.x * NT + threadIdx.x;
    if (j >= (long)a.N * L) return;
    int n = (int)(j / L);
    int r = (int)(j - (long)n * L);
    float g = a.x[(long)c * a.x_sC + (long)n * a.x_sN + r];
    float o = a.r[(long)c * a.r_sC + (long)n * a.r_sN + r];
    a.out[(long)c * a.o_sC + (long)n * a.o_sN + r] = g * kg_dact_from_out(o, a.act, a.slope);
}

__global__ __launch_bounds__(NT) void kg_affine_act_kernel(const KgEltArgs a) {
    const int c = blockIdx.y;
    const int L = a.T * a.V;
    const long j = (long)blockIdx.x * NT + threadIdx.x;
    if (j >= (long)a.N * L) return;
    int n = (int)(j / L);
    int r = (int)(j - (long)n * L);
    float v = a.x[(long)c * a.x_sC + (long)n * a.x_sN + r];
    const long cg = c + (a.groups > 1 ? (long)(n / (a.N / a.groups)) * a.coef_gs : 0L);     // this batch's coefficients
    if (a.sx) v *= a.sx[cg];
    if (a.bx) v += a.bx[cg];
    if (a.r) {
        float rv = a.r[(long)c * a.r_sC + (long)n * a.r_sN + r];
        if (a.sr) rv *= a.sr[cg];
        v += rv;
    }
    if (a.br) v += a.br[cg];
    if (a.noise && a.nw) v = fmaf(a.nw[c], a.noise[(long)n * L + r], v);
    a.out[(long)c * a.o_sC + (long)n * a.o_sN + r] = kg_act(v, a.act, a.slope);
}

// ---- Adam -------------------------------------------------------------------------------------
// torch.optim.Adam (no amsgrad, no weight decay):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                         float step_size, float rs_bc2) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= step_size * (m / (sqrtf(v) * rs_bc2 + eps));
}

// four parameters per thread in 128-bit accesses (VEC) when the buffers are 16-byte aligned; scalar tail / fallback.
// FUSED (kg_adam_step_fused) with zero: the consumed gradient is cleared (the next backward pass accumulates into a clean
// bucket without a fill launch).  (Advancing *step inside the launch - last workgroup to finish, ticket counter - was
// measured and dropped: ~900 same-address atomics cost 15 us per launch, the `step += 1` launch it replaced 2 us.)
template <bool VEC, bool FUSED>
__global__ __launch_bounds__(NT) void kg_adam_kernel(float* p, float* g, float* m, float* v, long n,
                                                     float lr, float b1, float b2, float eps,
                                                     const int32_t* step, float gscale, int zero) {
    const float t = (float)(*step);
    const float step_size = lr / (1.f - powf(b1, t));
    const float rs_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if constexpr (VEC) {
        const long e = 4 * i;
        if (e + 3 < n) {
            float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            adam_one(pp.x, gg.x * gscale, mm.x, vv.x, b1, b2, eps, step_size, rs_bc2);
            adam_one(pp.y, gg.y * gscale, mm.y, vv.y, b1, b2, eps, step_size, rs_bc2);
            adam_one(pp.z, gg.z * gscale, mm.z, vv.z, b1, b2, eps, step_size, rs_bc2);
            adam_one(pp.w, gg.w * gscale, mm.w, vv.w, b1, b2, eps, step_size, rs_bc2);
            reinterpret_cast<float4*>(p)[i] = pp;
            reinterpret_cast<float4*>(m)[i] = mm;
            reinterpret_cast<float4*>(v)[i] = vv;
            if (FUSED && zero) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (long k = e; k < n; ++k) {
                adam_one(p[k], g[k] * gscale, m[k], v[k], b1, b2, eps, step_size, rs_bc2);
                if (FUSED && zero) g[k] = 0.f;
            }
        }
    } else {
        if (i < n) {
            adam_one(p[i], g[i] * gscale, m[i], v[i], b1, b2, eps, step_size, rs_bc2);
            if (FUSED && zero) g[i] = 0.f;
        }
    }
}

int validate_elt(const KgEltArgs* a, const char* who) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0, "%s: bad dims", who);
    KG_REQUIRE(a->C <= 65535, "%s: C=%d too large", who, a->C);
    KG_REQUIRE(a->x && a->out, "%s: null pointer", who);
    KG_REQUIRE(a->act >= KG_ACT_NONE && a->act <= KG_ACT_TANH, "%s: act=%d", who, a->act);
    return 0;
}

int rowsum_parts(const KgRowsumArgs* a) { return kg_cdiv((long)a->N * a->T * a->V, RS_CHUNK); }

}  // namespace

extern "C" int64_t kg_rowsum_workspace_bytes(const KgRowsumArgs* a) {
    if (a == nullptr || a->N <= 0 || a->C <= 0 || a->T <= 0 || a->V <= 0) return -1;
    return (int64_t)2 * a->C * rowsum_parts(a) * (int64_t)sizeof(float);
}

extern "C" int64_t kg_rowsum_many_workspace_bytes(const KgRowsumArgs* jobs, int32_t njobs) {
    if (jobs == nullptr || njobs < 1) return -1;
    int64_t total = 0;
    for (int i = 0; i < njobs; ++i) {
        const int64_t b = kg_rowsum_workspace_bytes(&jobs[i]);
        if (b < 0) return -1;
        total += b;
    }
    return total;
}

extern "C" int kg_rowsum_many(const KgRowsumArgs* jobs, int32_t njobs, float* ws, int64_t ws_bytes, void* stream) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1, "kg_rowsum_many: no jobs");
    hipStream_t s = (hipStream_t)stream;
    int64_t off = 0;
    for (int base = 0; base < njobs; base += RS_MANY_MAX) {
        const int n = njobs - base < RS_MANY_MAX ? njobs - base : RS_MANY_MAX;
        RowsumMany m, f;
        m.njobs = n;
        f.njobs = 0;
        m.beg[0] = 0;
        f.beg[0] = 0;
        for (int i = 0; i < n; ++i) {
            const KgRowsumArgs* a = &jobs[base + i];
            KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0 && a->C <= 65535, "kg_rowsum_many: job %d bad dims", base + i);
            KG_REQUIRE(a->x && a->out, "kg_rowsum_many: job %d null pointer", base + i);
            for (int k = 0; k < base + i; ++k)
                KG_REQUIRE(jobs[k].out != a->out && (a->out2 == nullptr || (jobs[k].out2 != a->out2 && jobs[k].out != a->out2)),
                           "kg_rowsum_many: jobs %d and %d write the same destination", k, base + i);
            const int P = rowsum_parts(a);
            const int64_t bytes = (int64_t)2 * a->C * P * (int64_t)sizeof(float);
            KG_REQUIRE(ws != nullptr && off + bytes <= ws_bytes, "kg_rowsum_many: workspace too small");
            m.job[i] = *a;
            m.job[i].ws = ws + off / (int64_t)sizeof(float);
            m.job[i].ws_bytes = bytes;
            off += bytes;
            m.P[i] = P;
            m.beg[i + 1] = m.beg[i] + P * a->C;
            if (P > 1) {
                f.job[f.njobs] = m.job[i];
                f.P[f.njobs] = P;
                f.beg[f.njobs + 1] = f.beg[f.njobs] + (a->want_second ? 2 : 1) * a->C;
                ++f.njobs;
            }
        }
        hipLaunchKernelGGL(kg_rowsum_many_kernel, dim3(m.beg[n]), dim3(NT), 0, s, m);
        if (int rc = kg_launch_status("kg_rowsum_many")) return rc;
        if (f.njobs > 0) {
            hipLaunchKernelGGL(kg_rowsum_many_finish, dim3(f.beg[f.njobs]), dim3(64), 0, s, f);
            if (int rc = kg_launch_status("kg_rowsum_many_finish")) return rc;
        }
    }
    return 0;
}

extern "C" int kg_rowsum(const KgRowsumArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr, "kg_rowsum: null args");
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0, "kg_rowsum: bad dims");
    KG_REQUIRE(a->C <= 65535, "kg_rowsum: C too large");
    KG_REQUIRE(a->x && a->out && a->ws, "kg_rowsum: null pointer");
    const int P = rowsum_parts(a);
    KG_REQUIRE(a->ws_bytes >= (int64_t)2 * a->C * P * 4, "kg_rowsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(kg_rowsum_kernel, dim3(P, a->C), dim3(NT), 0, s, *a, P);
    if (int rc = kg_launch_status("kg_rowsum")) return rc;
    if (P == 1) return 0;                                   // small inputs: the first kernel wrote the result
    hipLaunchKernelGGL(kg_rowsum_finish, dim3((a->want_second ? 2 : 1) * a->C), dim3(64), 0, s, *a, P);
    return kg_launch_status("kg_rowsum_finish");
}

static int validate_bn(const KgBnArgs* a, const char* who) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0, "%s: bad dims", who);
    KG_REQUIRE(a->C <= 65535, "%s: C=%d too large", who, a->C);
    KG_REQUIRE(a->x && a->coef, "%s: null pointer", who);
    return 0;
}

extern "C" int kg_bn_fwd(const KgBnArgs* a, void* stream) {
    if (int rc = validate_bn(a, "kg_bn_fwd")) return rc;
    KG_REQUIRE(a->training || (a->running_mean && a->running_var), "kg_bn_fwd: eval mode needs running statistics");
    KG_REQUIRE((a->running_mean == nullptr) == (a->running_var == nullptr), "kg_bn_fwd: running_mean / running_var");
    if ((long)a->N * a->T * a->V >= 16384 && a->training)
        hipLaunchKernelGGL(kg_bn_fwd_kernel<1024>, dim3(a->C), dim3(1024), 0, (hipStream_t)stream, *a);
    else
        hipLaunchKernelGGL(kg_bn_fwd_kernel<256>, dim3(a->C), dim3(256), 0, (hipStream_t)stream, *a);
    if (a->training && a->momentum < 0.f && a->num_batches_tracked)
        hipLaunchKernelGGL(kg_bn_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a->num_batches_tracked);
    return kg_launch_status("kg_bn_fwd");
}

static int bn_many_layout(const KgBnJob* jobs, int32_t njobs, BnMany* m, int64_t* ws_floats, int* ncounters) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= BN_MANY_MAX, "kg_bn_fwd_many: 1..%d jobs", BN_MANY_MAX);
    long wg = 0, wsf = 0;
    int cb = 0;
    for (int i = 0; i < njobs; ++i) {
        const KgBnArgs* a = &jobs[i].a;
        if (int rc = validate_bn(a, "kg_bn_fwd_many")) return rc;
        KG_REQUIRE(a->training, "kg_bn_fwd_many: training-mode statistics only (eval mode: kg_bn_fwd)");
        KG_REQUIRE(a->momentum >= 0.f, "kg_bn_fwd_many: cumulative moving average (momentum < 0) is kg_bn_fwd's");
        KG_REQUIRE(jobs[i].groups >= 1 && jobs[i].groups <= 8, "kg_bn_fwd_many: job %d groups=%d", i, jobs[i].groups);
        KG_REQUIRE((a->running_mean == nullptr) == (a->running_var == nullptr), "kg_bn_fwd_many: running_mean / running_var");
        const int P = kg_cdiv((long)a->N * a->T * a->V, BN_CHUNK);
        if (m) { m->beg[i] = (int)wg; m->P[i] = P; m->cbeg[i] = cb; m->wbeg[i] = wsf; m->job[i] = jobs[i]; }
        wg += (long)a->C * jobs[i].groups * P;
        wsf += (long)a->C * jobs[i].groups * P * 2;
        cb += a->C;
    }
    KG_REQUIRE(wg < (1L << 31), "kg_bn_fwd_many: grid too large");
    if (m) { m->beg[njobs] = (int)wg; m->njobs = njobs; }
    *ws_floats = wsf;
    *ncounters = cb;
    return 0;
}

extern "C" int64_t kg_bn_fwd_many_workspace_bytes(const KgBnJob* jobs, int32_t njobs) {
    int64_t wsf = 0;
    int nc = 0;
    if (bn_many_layout(jobs, njobs, nullptr, &wsf, &nc)) return -1;
    return wsf * (int64_t)sizeof(float);
}

extern "C" int kg_bn_fwd_many(const KgBnJob* jobs, int32_t njobs, float* ws, int64_t ws_bytes, int32_t* counters,
                              int32_t counters_len, void* stream) {
    BnMany m;
    int64_t wsf = 0;
    int nc = 0;
    if (int rc = bn_many_layout(jobs, njobs, &m, &wsf, &nc)) return rc;
    KG_REQUIRE(ws != nullptr && ws_bytes >= wsf * (int64_t)sizeof(float), "kg_bn_fwd_many: workspace too small");
    KG_REQUIRE(counters != nullptr && counters_len >= nc, "kg_bn_fwd_many: %d zeroed counters needed", nc);
    m.ws = ws;
    m.counters = counters;
    hipLaunchKernelGGL(kg_bn_fwd_many_kernel, dim3(m.beg[njobs]), dim3(NT), 0, (hipStream_t)stream, m);
    return kg_launch_status("kg_bn_fwd_many");
}

extern "C" int kg_bn_bwd(const KgBnArgs* a, void* stream) {
    if (int rc = validate_bn(a, "kg_bn_bwd")) return rc;
    KG_REQUIRE(a->g && a->mean && a->rstd, "kg_bn_bwd: null pointer");
    if ((long)a->N * a->T * a->V >= 16384)
        hipLaunchKernelGGL(kg_bn_bwd_kernel<1024>, dim3(a->C), dim3(1024), 0, (hipStream_t)stream, *a);
    else
        hipLaunchKernelGGL(kg_bn_bwd_kernel<256>, dim3(a->C), dim3(256), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_bn_bwd");
}

extern "C" int64_t kg_bn_bwd_many_workspace_bytes(const KgBnArgs* jobs, int32_t njobs) {
    if (jobs == nullptr || njobs < 1 || njobs > BN_MANY_MAX) { kg_set_error("kg_bn_bwd_many: 1..%d jobs", BN_MANY_MAX); return -1; }
    int64_t f = 0;
    for (int i = 0; i < njobs; ++i) {
        if (validate_bn(&jobs[i], "kg_bn_bwd_many")) return -1;
        f += (int64_t)jobs[i].C * kg_cdiv((long)jobs[i].N * jobs[i].T * jobs[i].V, BN_CHUNK) * 2;
    }
    return f * (int64_t)sizeof(float);
}

extern "C" int kg_bn_bwd_many(const KgBnArgs* jobs, int32_t njobs, float* ws, int64_t ws_bytes, int32_t* counters,
                              int32_t counters_len, void* stream) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1 && njobs <= BN_MANY_MAX, "kg_bn_bwd_many: 1..%d jobs", BN_MANY_MAX);
    BnBwdMany m;
    long wg = 0, wsf = 0;
    int cb = 0;
    for (int i = 0; i < njobs; ++i) {
        const KgBnArgs* a = &jobs[i];
        if (int rc = validate_bn(a, "kg_bn_bwd_many")) return rc;
        KG_REQUIRE(a->g && a->mean && a->rstd, "kg_bn_bwd_many: job %d null pointer", i);
        const int P = kg_cdiv((long)a->N * a->T * a->V, BN_CHUNK);
        m.beg[i] = (int)wg; m.P[i] = P; m.cbeg[i] = cb; m.wbeg[i] = wsf; m.job[i] = *a;
        wg += (long)a->C * P;
        wsf += (long)a->C * P * 2;
        cb += a->C;
    }
    KG_REQUIRE(wg < (1L << 31), "kg_bn_bwd_many: grid too large");
    KG_REQUIRE(ws != nullptr && ws_bytes >= wsf * (int64_t)sizeof(float), "kg_bn_bwd_many: workspace too small");
    KG_REQUIRE(counters != nullptr && counters_len >= cb, "kg_bn_bwd_many: %d zeroed counters needed", cb);
    m.beg[njobs] = (int)wg;
    m.njobs = njobs;
    m.ws = ws;
    m.counters = counters;
    hipLaunchKernelGGL(kg_bn_bwd_many_kernel, dim3((int)wg), dim3(NT), 0, (hipStream_t)stream, m);
    return kg_launch_status("kg_bn_bwd_many");
}

extern "C" int kg_act_bwd(const KgEltArgs* a, void* stream) {
    if (int rc = validate_elt(a, "kg_act_bwd")) return rc;
    KG_REQUIRE(a->r != nullptr, "kg_act_bwd: null ref");
    dim3 grid(kg_cdiv((long)a->N * a->T * a->V, NT), a->C);
    hipLaunchKernelGGL(kg_act_bwd_kernel, grid, dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_act_bwd");
}

extern "C" int kg_affine_act(const KgEltArgs* a, void* stream) {
    if (int rc = validate_elt(a, "kg_affine_act")) return rc;
    KG_REQUIRE(a->groups <= 1 || (a->N % a->groups == 0 && a->coef_gs >= a->C), "kg_affine_act: groups=%d N=%d coef_gs=%ld",
               a->groups, a->N, (long)a->coef_gs);
    dim3 grid(kg_cdiv((long)a->N * a->T * a->V, NT), a->C);
    hipLaunchKernelGGL(kg_affine_act_kernel, grid, dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_affine_act");
}

// ---- WGAN-GP gradient penalty (kinetic-gan.py:112-113) ------------------------------------------------------------
// gp = mean_n (|g_n|_2 - 1)^2 over the per-sample gradients g (N, C, T, V); backward d gp / d g = (2/N)(1 - 1/|g_n|) g_n
// (0 where |g_n| = 0, torch.norm's convention), scaled by the upstream gradient read from device memory.
namespace {

__global__ __launch_bounds__(NT) void kg_gp_norm_kernel(const KgGpArgs a) {
    __shared__ float red[NT / 64];
    const int n = blockIdx.x;
    const int L = a.T * a.V;
    const long total = (long)a.C * L;
    const float* base = a.g + (long)n * a.g_sN;
    float s = 0.f;
    // eight loads of a thread in flight before the first use (one by one the 19 trips of a 4800-element sample were a
    // chain of memory latencies: 17 us for 1.2 MB)
    constexpr int PER = 8;
    for (long e0 = threadIdx.x; e0 < total; e0 += (long)NT * PER) {
        float v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const long e = e0 + (long)u * NT;
            const bool ok = e < total;
            const int c = ok ? (int)(e / L) : 0, r = ok ? (int)(e - (long)c * L) : 0;
            const float x = base[(long)c * a.g_sC + r];
            v[u] = ok ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) s = fmaf(v[u], v[u], s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        a.nrm[n] = sqrtf(t);
    }
}

__global__ __launch_bounds__(NT) void kg_gp_mean_kernel(const KgGpArgs a) {
    __shared__ float red[NT / 64];
    float s = 0.f;
    for (int n = threadIdx.x; n < a.N; n += NT) {
        const float d = a.nrm[n] - 1.f;
        s = fmaf(d, d, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < NT / 64; ++i) t += red[i];
        a.gp[0] = t / (float)a.N;
    }
}

__global__ __launch_bounds__(NT) void kg_gp_bwd_kernel(const KgGpArgs a) {
    const int L = a.T * a.V;
    const int n = blockIdx.z, c = blockIdx.y;
    const int r = blockIdx.x * NT + threadIdx.x;
    if (r >= L) return;
    const float nr = a.nrm[n];
    const float coef = nr > 0.f ? (2.f / (float)a.N) * (1.f - 1.f / nr) * a.gout[0] : 0.f;
    a.out[(long)n * a.o_sN + (long)c * a.o_sC + r] = coef * a.g[(long)n * a.g_sN + (long)c * a.g_sC + r];
}

int validate_gp(const KgGpArgs* a, const char* who) {
    KG_REQUIRE(a != nullptr, "%s: null args", who);
    KG_REQUIRE(a->N > 0 && a->C > 0 && a->T > 0 && a->V > 0 && a->N <= 65535 && a->C <= 65535, "%s: bad dims", who);
    KG_REQUIRE(a->g && a->nrm, "%s: null pointer", who);
    return 0;
}

}  // namespace

extern "C" int kg_gp_fwd(const KgGpArgs* a, void* stream) {
    if (int rc = validate_gp(a, "kg_gp_fwd")) return rc;
    KG_REQUIRE(a->gp != nullptr, "kg_gp_fwd: null gp");
    hipLaunchKernelGGL(kg_gp_norm_kernel, dim3(a->N), dim3(NT), 0, (hipStream_t)stream, *a);
    if (int rc = kg_launch_status("kg_gp_fwd (norms)")) return rc;
    hipLaunchKernelGGL(kg_gp_mean_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_gp_fwd (mean)");
}

extern "C" int kg_gp_bwd(const KgGpArgs* a, void* stream) {
    if (int rc = validate_gp(a, "kg_gp_bwd")) return rc;
    KG_REQUIRE(a->gout && a->out, "kg_gp_bwd: null pointer");
    hipLaunchKernelGGL(kg_gp_bwd_kernel, dim3(kg_cdiv((long)a->T * a->V, NT), a->C, a->N), dim3(NT), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_gp_bwd");
}

extern "C" int kg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                            float b2, float eps, const int32_t* step, float grad_scale, void* stream) {
    KG_REQUIRE(p && g && m && v && step, "kg_adam_step: null pointer");
    KG_REQUIRE(n > 0, "kg_adam_step: n=%ld", (long)n);
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL((kg_adam_kernel<true, false>), dim3(kg_cdiv(kg_cdiv(n, 4), NT)), dim3(NT), 0, (hipStream_t)stream, p,
                           const_cast<float*>(g), m, v, (long)n, lr, b1, b2, eps, step, grad_scale, 0);
    else
        hipLaunchKernelGGL((kg_adam_kernel<false, false>), dim3(kg_cdiv(n, NT)), dim3(NT), 0, (hipStream_t)stream, p,
                           const_cast<float*>(g), m, v, (long)n, lr, b1, b2, eps, step, grad_scale, 0);
    return kg_launch_status("kg_adam_step");
}

extern "C" int kg_adam_step_fused(float* p, float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                                  const int32_t* step, float grad_scale, int32_t zero_grad, void* stream) {
    KG_REQUIRE(p && g && m && v && step, "kg_adam_step_fused: null pointer");
    KG_REQUIRE(n > 0, "kg_adam_step_fused: n=%ld", (long)n);
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL((kg_adam_kernel<true, true>), dim3(kg_cdiv(kg_cdiv(n, 4), NT)), dim3(NT), 0, (hipStream_t)stream, p, g, m,
                           v, (long)n, lr, b1, b2, eps, step, grad_scale, (int)zero_grad);
    else
        hipLaunchKernelGGL((kg_adam_kernel<false, true>), dim3(kg_cdiv(n, NT)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v,
                           (long)n, lr, b1, b2, eps, step, grad_scale, (int)zero_grad);
    return kg_launch_status("kg_adam_step_fused");
}


// ---- measured peaks (kgan_hip.h, ABI v7) ---------------------------------------------------------------------------------
namespace {
constexpr int PEAK_WGS = 256 * 4;           // four workgroups of four waves per CU: four waves per SIMD

__global__ __launch_bounds__(256) void kg_peak_mfma_kernel(float* sink, int iters) {
    kg_f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // N(0,1)-like operands per lane (the clock the chip sustains follows the data's toggle rate): a hash of the thread
    // index, sum of four uniforms
    float av[8], bv[8];
    unsigned st = 0x9e3779b9u * (threadIdx.x + 1u) + 0x85ebca6bu * (blockIdx.x + 1u);
    auto rnd = [&]() {
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            st ^= st << 13; st ^= st >> 17; st ^= st << 5;
            u += (float)(st >> 8) * (1.f / 16777216.f);
        }
        return (u - 2.f) * 1.7320508f;
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) { av[i] = rnd(); bv[i] = rnd() * 0.01f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u * 4 + i) & 7], bv[(u + i * 3) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[threadIdx.x & 63] = s;        // (never true for these operands; keeps the loop)
}

__global__ __launch_bounds__(256) void kg_peak_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int kg_peak_mfma_f32(float* sink, int32_t iters, double* flops, void* stream) {
    KG_REQUIRE(sink != nullptr && iters > 0, "kg_peak_mfma_f32: bad arguments");
    if (flops) *flops = (double)PEAK_WGS * 4.0 * (double)iters * 16.0 * (2.0 * 32 * 32 * 2);
    hipLaunchKernelGGL(kg_peak_mfma_kernel, dim3(PEAK_WGS), dim3(256), 0, (hipStream_t)stream, sink, (int)iters);
    return kg_launch_status("kg_peak_mfma_f32");
}

extern "C" int kg_peak_copy(const float* src, float* dst, int64_t n, void* stream) {
    KG_REQUIRE(src != nullptr && dst != nullptr && n > 0 && n % 4 == 0, "kg_peak_copy: bad arguments");
    KG_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "kg_peak_copy: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(kg_peak_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, (long)(n / 4));
    return kg_launch_status("kg_peak_copy");
}
