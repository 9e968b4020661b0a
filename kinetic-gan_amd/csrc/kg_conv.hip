// kg_conv: the channel contraction of the st_gcn blocks as a "tap GEMM" on the fp32 matrix cores.
//
//   out[m, j] = act( sum_g sum_d sum_c W_g(d,m,c) * X_g[c, src_g(j,d)] + bias + add )
//
// M = output channels, columns j = (n, t, v) of the whole batch, K-slices = (group, tap, 32 input
// channels).  v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// Work decomposition.  A workgroup owns BM output channels x (32 * NW) columns; every wave owns 32
// columns and ALL BM rows.  The B operand of the 32x32x2 MFMA is (k = lane>>5, j = lane&31), i.e. two
// 128-byte rows of the feature matrix, so each lane loads its own operand element straight from
// HBM/L2 into a register (coalesced buffer load) with the tap's time shift / stride / vertex gather
// folded into the address - the feature tile never goes through LDS.  The conv's zero padding,
// dropped vertices and ragged channel / column tails are out-of-range buffer offsets, for which the
// hardware returns 0 (no guarded loads: hipcc branches around those and serialises their latencies).
// Only the weight tile Ws[k][m] (shared by the NW waves) is staged in LDS, double buffered.
// Software pipeline: the loads of slice s+1 (weights and features) are in flight while the MFMAs of
// slice s run; one barrier per slice.  Everything that needs an integer division or a kernel-argument
// read (per-group geometry, per-thread weight / column offsets for the three taps) is computed once
// per K-slice group and kept in registers: the slice loop touches no scalar memory.
// Skinny problems (few columns, deep K: the 512-channel blocks at T<=16, V<=5) are split along K
// across workgroups into partial slabs that a second kernel sums in a fixed order together with
// bias / residual add / activation (deterministic, no atomics).
//
// Reference ops covered: tgcn.py:61, discriminator.py:99-105,115-120,130-136,139-142,
// generator.py:134-140,154-159,176,182 and their backward-data passes (transposed mode).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "kg_common.h"

// Optional instrumentation build (-DKG_CONV_TIMING, tools/time_conv.py): wave 0 of every workgroup records
// s_memtime at four points of each K-slice and writes the accumulated segment lengths to the END of a.ws.
#ifdef KG_CONV_TIMING
#define KG_STAMP(i)                                                  \
    do {                                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
        if ((i) > 0) kg_seg[(i)-1] += t_ - kg_last;                  \
        kg_last = t_;                                                \
    } while (0)
#define KG_STAMP_FLUSH()                                                                           \
    do {                                                                                           \
        if (threadIdx.x == 0 && a.ws) {                                                            \
            unsigned long long* o_ = (unsigned long long*)((char*)a.ws + a.ws_bytes - (1 << 20)) + \
                                     (blockIdx.x + gridDim.x * blockIdx.y) * 8;                    \
            for (int q_ = 0; q_ < 6; ++q_) o_[q_] = kg_seg[q_];                                    \
            o_[6] = (unsigned long long)(s_end - s_beg);                                           \
        }                                                                                          \
    } while (0)
#define KG_STAMP_DECL() unsigned long long kg_seg[6] = {0, 0, 0, 0, 0, 0}, kg_last = 0
#else
#define KG_STAMP(i) do {} while (0)
#define KG_STAMP_FLUSH() do {} while (0)
#define KG_STAMP_DECL() do {} while (0)
#endif

namespace {

constexpr int BK = 32;

struct ColInfo {
    int n, to, vo;
    bool valid;
};

__device__ __forceinline__ ColInfo decode_col(int j, int ncols, int T_out, int V_out) {
    ColInfo c;
    c.valid = j < ncols;
    int jj = c.valid ? j : 0;
    int L = T_out * V_out;
    c.n = jj / L;
    int r = jj - c.n * L;
    c.to = r / V_out;
    c.vo = r - c.to * V_out;
    return c;
}

__host__ __device__ inline int slices_of(const KgConvGroup& g) { return g.taps * ((g.Cin + BK - 1) / BK); }

struct Split {
    int nsplit;          // workgroups along K
    int per;             // slices per split
};

constexpr unsigned W_RANGE = 0x40000000u;   // weight descriptor: 1 GiB; valid offsets are below it
constexpr unsigned X_RANGE = 0x80000000u;   // feature descriptor: 2 GiB (validated on the host)
constexpr unsigned W_OOB = 0x40000000u;     // adding one or two of these to a valid offset stays out of range
constexpr unsigned X_OOB = 0x80000000u;

template <int BM, int NW>
__global__ __launch_bounds__(64 * NW) void kg_conv_kernel(const KgConvArgs a, const Split sp) {
    constexpr int NT = 64 * NW;
    constexpr int BN = 32 * NW;
    constexpr int TM = BM / 32;
    constexpr int WREG = BK * BM / NT;       // weight elements each thread stages per slice
    constexpr int BREG = BK / 2;             // B fragments per slice (one per k-step of 2)
    static_assert((BK * BM) % NT == 0 && NT % BK == 0 && NT % BM == 0, "tile/thread mismatch");

    __shared__ float Ws[2][BK][BM + 1];      // +1: the k-fastest staging pattern writes a column of Ws per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ncols = a.N * a.T_out * a.V_out;
    const int j0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;
    const int kh = lane >> 5;                // which of the two k rows of an MFMA step this lane feeds

    const ColInfo xc = decode_col(j0 + wave * 32 + (lane & 31), ncols, a.T_out, a.V_out);

    const int s_total = slices_of(a.g[0]) + (a.ngroups > 1 ? slices_of(a.g[1]) : 0);
    const int s_beg = blockIdx.z * sp.per;
    const int s_end = min(s_total, s_beg + sp.per);

    kg_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    KG_STAMP_DECL();
    float wreg[WREG];
    float b0[BREG], b1[BREG];

    // ---- per-group state, all in registers -------------------------------------------------------------
    const float* gx = nullptr;      // wave-uniform copies of the group's geometry
    const float* gw = nullptr;
    long g_xsC = 0, g_wsT = 0;
    int g_Cin = 0, g_taps = 1, g_cchunks = 1, g_chanblock = 0;
    unsigned g_wsi4 = 0;
    bool kf = true;                 // weight staging pattern: k fastest (forward layouts) or m fastest (transposed)
    unsigned woff[WREG];            // per thread: byte offset of the m-part of its i-th weight element (or W_OOB)
    unsigned xoffB[3];              // per thread: byte offset of its column's source for tap 0..2 (or X_OOB)
    auto setup_group = [&](int gi) {
        const KgConvGroup& g = a.g[gi];
        gx = g.x; gw = g.w; g_xsC = g.x_sC; g_wsT = g.w_sT;
        g_Cin = g.Cin; g_taps = g.taps; g_cchunks = (g.Cin + BK - 1) / BK;
        g_chanblock = g.tap_mode == KG_TAP_CHANBLOCK ? g.Cin : 0;
        g_wsi4 = (unsigned)g.w_sI * 4u;
        kf = g.w_sI <= g.w_sO;
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int m = kf ? tid / BK + i * (NT / BK) : tid % BM;
            const int mm = m0 + m;
            const int mb = mm / g.w_MB;
            const unsigned off = (unsigned)(mb * g.w_sMB + (mm - mb * g.w_MB) * g.w_sO) * 4u;
            woff[i] = mm < a.M ? off : W_OOB;
        }
        const int vi = g.vmap ? (xc.valid ? g.vmap[xc.vo] : -1) : xc.vo;
        const int pad = (g.tap_mode == KG_TAP_TIME) ? (g.taps - 1) / 2 : 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int shift = (g.tap_mode == KG_TAP_TIME) ? d - pad : 0;
            int ti;
            bool ok = xc.valid && vi >= 0 && d < g.taps;
            if (!g.transposed) {
                ti = xc.to * g.t_stride + shift;
            } else {
                const int num = xc.to - shift;
                ok = ok && num >= 0 && (num % g.t_stride) == 0;
                ti = num / g.t_stride;
            }
            ok = ok && ti >= 0 && ti < g.T_in;
            const long off = (long)kh * g.x_sC + (long)xc.n * g.x_sN + (long)ti * g.V_in + vi;
            xoffB[d] = ok ? (unsigned)(off * 4) : X_OOB;
        }
    };

    // global -> registers for slice (d, c0) of the current group: no scalar-memory reads, no divisions
    auto fetch = [&](auto kfc, float (&breg)[BREG], int d, int c0) {
        constexpr bool KF = decltype(kfc)::value;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            kg_uniform_ptr(gw + (long)d * g_wsT), 0, (int)W_RANGE, 0x00020000);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            kg_uniform_ptr(gx + (long)(d * g_chanblock + c0) * g_xsC), 0, (int)X_RANGE, 0x00020000);
        KG_STAMP(1);
        if constexpr (KF) {
            const int cc = c0 + tid % BK;
            const unsigned kterm = cc < g_Cin ? (unsigned)cc * g_wsi4 : W_OOB;
#pragma unroll
            for (int i = 0; i < WREG; ++i)
                wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, woff[i] + kterm, 0, 0));
        } else {
            const int k0 = c0 + tid / BM;
            const unsigned base = woff[0] + (unsigned)k0 * g_wsi4;
            const unsigned step = (unsigned)(NT / BM) * g_wsi4;
            const int nvalid = (g_Cin - k0 + (NT / BM) - 1) / (NT / BM);    // elements i < nvalid are inside Cin
#pragma unroll
            for (int i = 0; i < WREG; ++i)
                wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wr, i < nvalid ? base + i * step : W_OOB, 0, 0));
        }
        KG_STAMP(2);
        const unsigned base = d == 0 ? xoffB[0] : (d == 1 ? xoffB[1] : xoffB[2]);
        const unsigned step = (unsigned)(2 * g_xsC * 4);
        const int nvalid = (g_Cin - c0 - kh + 1) / 2;      // fragments i < nvalid have their channel inside Cin
#ifdef KG_CONV_HALFLOADS   // experiment (wrong results): issue only every 4th feature load, to see what the loads cost
#pragma unroll
        for (int i = 0; i < BREG; i += 4) {
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                xr, i < nvalid ? base + i * step : X_OOB, 0, 0));
            breg[i + 1] = breg[i]; breg[i + 2] = breg[i]; breg[i + 3] = breg[i];
        }
#else
#pragma unroll
        for (int i = 0; i < BREG; ++i)
            breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                xr, i < nvalid ? base + i * step : X_OOB, 0, 0));
#endif
    };
    // weight registers -> LDS buffer b
    auto stash = [&](auto kfc, int b) {
        constexpr bool KF = decltype(kfc)::value;
        if constexpr (KF) {
            float* p = &Ws[b][tid % BK][tid / BK];
#pragma unroll
            for (int i = 0; i < WREG; ++i) p[i * (NT / BK)] = wreg[i];
        } else {
            float* p = &Ws[b][tid / BM][tid % BM];
#pragma unroll
            for (int i = 0; i < WREG; ++i) p[i * (NT / BM) * (BM + 1)] = wreg[i];
        }
    };
    auto fetch_any = [&](float (&breg)[BREG], int d, int c0) {
        if (kf) fetch(std::true_type{}, breg, d, c0);
        else fetch(std::false_type{}, breg, d, c0);
    };
    auto stash_any = [&](int b) {
        if (kf) stash(std::true_type{}, b);
        else stash(std::false_type{}, b);
    };

    if (s_beg < s_end) {
        // locate the first slice: (group, tap, channel chunk); afterwards the triple is advanced incrementally
        int gi = 0, sl = s_beg;
        const int s0 = slices_of(a.g[0]);
        if (sl >= s0) { gi = 1; sl -= s0; }
        setup_group(gi);
        int d = sl / g_cchunks;
        int cch = sl - d * g_cchunks;
        fetch_any(b0, d, cch * BK);
        stash_any(0);
        __syncthreads();
        // one pipeline step: MFMAs of the current slice (weights in Ws[b], features in `cur`) while the next
        // slice's weights / features are loaded into registers (`nxt`)
        auto step = [&](float (&cur)[BREG], float (&nxt)[BREG], int b, bool more) {
            KG_STAMP(0);
            if (more) {
                if (++cch == g_cchunks) {
                    cch = 0;
                    if (++d == g_taps) {
                        d = 0;
                        setup_group(++gi);
                    }
                }
                fetch_any(nxt, d, cch * BK);
            }
            KG_STAMP(3);
            // keep the issue order loads -> MFMAs -> (wait + LDS writes): without the fences hipcc hoists the
            // LDS writes (and their vmcnt waits) above the MFMA loop and the load latency is exposed again
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);      // waves in their MFMA phase win issue arbitration over staging waves
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float av[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = Ws[b][kk + kh][i * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], cur[kk / 2], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            KG_STAMP(4);
            if (more) stash_any(b ^ 1);
            KG_STAMP(5);
            __syncthreads();
            KG_STAMP(6);
        };
        for (int s = s_beg; s < s_end; s += 2) {
            step(b0, b1, 0, s + 1 < s_end);
            if (s + 1 < s_end) step(b1, b0, 1, s + 2 < s_end);
        }
    }
    KG_STAMP_FLUSH();

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool partial = sp.nsplit > 1;
    float* slab = partial ? a.ws + (long)blockIdx.z * a.M * ncols : nullptr;
    const int j = j0 + wave * 32 + (lane & 31);
    if (xc.valid) {
        const long ooff = (long)xc.n * a.o_sN + (long)xc.to * a.V_out + xc.vo;
        const long aoff = a.add ? (long)xc.n * a.a_sN + (long)(xc.to * a.a_tstride) * a.V_out + xc.vo : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m >= a.M) continue;
                float v = acc[i][r];
                if (partial) {
                    slab[(long)m * ncols + j] = v;
                } else {
                    if (a.bias0) v += a.bias0[m];
                    if (a.bias1) v += a.bias1[m];
                    if (a.add) v += a.add[(long)m * a.a_sC + aoff];
                    a.out[(long)m * a.o_sC + ooff] = kg_act(v, a.act, a.slope);
                }
            }
        }
    }
}

// sum of the K-split slabs + bias + residual add + activation
__global__ __launch_bounds__(256) void kg_conv_splitk_epilogue(const KgConvArgs a, int nsplit) {
    const int ncols = a.N * a.T_out * a.V_out;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.y;
    if (j >= ncols) return;
    const long per = (long)a.M * ncols;
    const float* p = a.ws + (long)m * ncols + j;
    float v = 0.f;
#pragma unroll 4
    for (int k = 0; k < nsplit; ++k) v += p[(long)k * per];
    const ColInfo oc = decode_col(j, ncols, a.T_out, a.V_out);
    if (a.bias0) v += a.bias0[m];
    if (a.bias1) v += a.bias1[m];
    if (a.add) v += a.add[(long)m * a.a_sC + (long)oc.n * a.a_sN + (long)(oc.to * a.a_tstride) * a.V_out + oc.vo];
    a.out[(long)m * a.o_sC + (long)oc.n * a.o_sN + (long)oc.to * a.V_out + oc.vo] = kg_act(v, a.act, a.slope);
}

enum Tile { T128x128, T64x128, T32x128, T64x64, T32x64, NTILES };
const int kTileBM[NTILES] = {128, 64, 32, 64, 32};
const int kTileBN[NTILES] = {128, 128, 128, 64, 64};

struct Plan {
    Tile tile;
    Split sp;
};

Plan make_plan(const KgConvArgs* a) {
    const long ncols = (long)a->N * a->T_out * a->V_out;
    const int M = a->M;
    int s_total = slices_of(a->g[0]) + (a->ngroups > 1 ? slices_of(a->g[1]) : 0);
    auto count = [&](Tile t) { return (long)kg_cdiv(M, kTileBM[t]) * kg_cdiv(ncols, kTileBN[t]); };
    Plan p;
    // Measured on MI355X (tools/tune_conv.py, profiles/r01_*_tune_conv.log): the 32-row tile wins whenever the
    // bigger tiles cannot give every CU ~2.5 workgroups - with few resident waves the staging phase of one
    // wave has no other wave's MFMA phase to hide under.
    const long full = 600;
    if (M > 64 && count(T128x128) >= full)     p.tile = T128x128;
    else if (M > 32 && count(T64x128) >= full) p.tile = T64x128;
    else if (count(T32x128) >= full / 2)       p.tile = T32x128;
    else                                       p.tile = M > 32 ? T32x128 : T32x64;
    // tuning hook (tools/tune_conv.py): KG_CONV_PLAN="<tile 0..4>,<nsplit>" forces the plan
    int forced_split = 0;
    if (const char* env = getenv("KG_CONV_PLAN")) {
        int t = -1, ns = 0;
        if (sscanf(env, "%d,%d", &t, &ns) >= 1 && t >= 0 && t < NTILES) {
            p.tile = (Tile)t;
            forced_split = ns;
        }
    }
    const long wgs = count(p.tile);
    int nsplit = 1;
    if (forced_split > 0) {
        nsplit = forced_split > s_total ? s_total : forced_split;
    } else if (wgs < 400 && s_total >= 4) {
        nsplit = (int)((700 + wgs - 1) / wgs);               // aim at ~3 workgroups per CU
        if (nsplit > s_total / 2) nsplit = s_total / 2;      // at least two slices per split
        if (nsplit > 8) nsplit = 8;
        if (nsplit < 1) nsplit = 1;
    }
    p.sp.per = kg_cdiv(s_total, nsplit);
    p.sp.nsplit = kg_cdiv(s_total, p.sp.per);
    return p;
}

template <int BM, int NW>
int launch(const KgConvArgs* a, const Plan& p, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    dim3 grid(kg_cdiv(ncols, 32 * NW), kg_cdiv(a->M, BM), p.sp.nsplit);
    hipLaunchKernelGGL((kg_conv_kernel<BM, NW>), grid, dim3(64 * NW), 0, s, *a, p.sp);
    if (int rc = kg_launch_status("kg_conv")) return rc;
    if (p.sp.nsplit > 1) {
        dim3 g2(kg_cdiv(ncols, 256), a->M);
        hipLaunchKernelGGL(kg_conv_splitk_epilogue, g2, dim3(256), 0, s, *a, p.sp.nsplit);
        return kg_launch_status("kg_conv_splitk_epilogue");
    }
    return 0;
}

int validate(const KgConvArgs* a) {
    KG_REQUIRE(a != nullptr, "kg_conv: null args");
    KG_REQUIRE(a->N > 0 && a->M > 0 && a->T_out > 0 && a->V_out > 0, "kg_conv: bad dims N=%d M=%d T=%d V=%d",
               a->N, a->M, a->T_out, a->V_out);
    KG_REQUIRE((long)a->N * a->T_out * a->V_out < (1L << 31), "kg_conv: too many columns");
    KG_REQUIRE(a->M <= 65535, "kg_conv: M=%d too large", a->M);
    KG_REQUIRE(a->ngroups >= 1 && a->ngroups <= 2, "kg_conv: ngroups=%d", a->ngroups);
    KG_REQUIRE(a->act >= KG_ACT_NONE && a->act <= KG_ACT_TANH, "kg_conv: act=%d", a->act);
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        KG_REQUIRE(g.Cin > 0 && g.T_in > 0 && g.V_in > 0, "kg_conv: group %d bad input dims", i);
        KG_REQUIRE(g.taps == 1 || g.taps == 3, "kg_conv: group %d taps=%d (1 or 3)", i, g.taps);
        KG_REQUIRE(g.tap_mode == KG_TAP_TIME || g.tap_mode == KG_TAP_CHANBLOCK, "kg_conv: group %d tap_mode", i);
        KG_REQUIRE(g.t_stride >= 1, "kg_conv: group %d t_stride=%d", i, g.t_stride);
        KG_REQUIRE(g.w_MB >= 1, "kg_conv: group %d w_MB=%d", i, g.w_MB);
        KG_REQUIRE(g.vmap != nullptr || g.V_in == a->V_out, "kg_conv: group %d V_in=%d != V_out=%d without vmap",
                   i, g.V_in, a->V_out);
        // 32-bit byte offsets inside one K-slice (buffer-load addressing): 32 channels + one column offset < 2 GiB
        const long xspan = 32L * g.x_sC + (long)(a->N - 1) * g.x_sN + (long)g.T_in * g.V_in;
        KG_REQUIRE(g.x_sC >= 0 && g.x_sN >= 0 && xspan < (1L << 29),
                   "kg_conv: group %d feature tensor too large for 32-bit slice offsets (span %ld elements)", i, xspan);
        const long wspan = (long)(a->M / g.w_MB) * g.w_sMB + (long)(g.w_MB < a->M ? g.w_MB : a->M) * g.w_sO +
                           (long)g.Cin * g.w_sI;
        KG_REQUIRE(g.w_sO >= 0 && g.w_sI >= 0 && g.w_sMB >= 0 && g.w_sT >= 0 && wspan < (1L << 28),
                   "kg_conv: group %d weight tensor too large (span %ld elements)", i, wspan);
    }
    return 0;
}

int64_t ws_bytes(const KgConvArgs* a, const Plan& p) {
    int64_t n = p.sp.nsplit > 1 ? (int64_t)p.sp.nsplit * a->M * a->N * a->T_out * a->V_out * (int64_t)sizeof(float) : 0;
#ifdef KG_CONV_TIMING
    n += 2 << 20;
#endif
    return n;
}

}  // namespace

extern "C" int64_t kg_conv_workspace_bytes(const KgConvArgs* a) {
    if (validate(a) != 0) return -1;
    return ws_bytes(a, make_plan(a));
}

extern "C" int kg_conv(const KgConvArgs* a, void* stream) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(a->out != nullptr, "kg_conv: null out");
    for (int i = 0; i < a->ngroups; ++i) KG_REQUIRE(a->g[i].x && a->g[i].w, "kg_conv: group %d null pointer", i);
    Plan p = make_plan(a);
    const int64_t need = ws_bytes(a, p);
    KG_REQUIRE(need == 0 || (a->ws != nullptr && a->ws_bytes >= need), "kg_conv: workspace %ld < %ld bytes",
               (long)a->ws_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    switch (p.tile) {
        case T128x128: return launch<128, 4>(a, p, s);
        case T64x128:  return launch<64, 4>(a, p, s);
        case T32x128:  return launch<32, 4>(a, p, s);
        case T64x64:   return launch<64, 2>(a, p, s);
        default:       return launch<32, 2>(a, p, s);
    }
}
