// kg_conv: the channel contraction of the st_gcn blocks as a "tap GEMM" on the fp32 matrix cores.
//
//   out[m, j] = act( sum_g sum_d sum_c W_g(d,m,c) * X_g[c, src_g(j,d)] + bias + add )
//
// M = output channels, columns j = (n, t, v) of the whole batch, K-slices = (group, tap, 32 input
// channels).  v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): A = weights staged in LDS as
// Ws[k][m], B = feature tile staged as Xs[k][j] with the tap's time shift / stride / vertex gather
// folded into the global-load address and zero-filled outside the frame range (the conv's zero
// padding).  Software pipeline: while the MFMAs of slice s run out of one LDS buffer, the global
// loads of slice s+1 are in flight into registers and are written to the other buffer afterwards
// (one barrier per slice).  Skinny problems (few columns, deep K: the 512-channel blocks at
// T<=16, V<=5) are split along K across workgroups into partial slabs that a second kernel sums
// in a fixed order together with bias / residual add / activation (deterministic, no atomics).
//
// Reference ops covered: tgcn.py:61, discriminator.py:99-105,115-120,130-136,139-142,
// generator.py:134-140,154-159,176,182 and their backward-data passes (transposed mode).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "kg_common.h"

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

// element offset (without the channel term) of the source of output column `c` for tap d, or -1
__device__ __forceinline__ long src_offset(const KgConvGroup& g, const ColInfo& c, int d, int vi) {
    if (!c.valid || vi < 0) return -1;
    int shift = (g.tap_mode == KG_TAP_TIME) ? d - (g.taps - 1) / 2 : 0;
    int ti;
    if (!g.transposed) {
        ti = c.to * g.t_stride + shift;
    } else {
        int num = c.to - shift;
        if (num < 0 || (num % g.t_stride) != 0) return -1;
        ti = num / g.t_stride;
    }
    if (ti < 0 || ti >= g.T_in) return -1;
    return (long)c.n * g.x_sN + (long)ti * g.V_in + vi;
}

// make a pointer provably wave-uniform for the compiler (else every buffer op gets a waterfall loop)
__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}

__host__ __device__ inline int slices_of(const KgConvGroup& g) { return g.taps * ((g.Cin + BK - 1) / BK); }

struct Split {
    int nsplit;          // workgroups along K
    int per;             // slices per split
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void kg_conv_kernel(const KgConvArgs a, const Split sp) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int KSTEP = NT / BN;           // k-rows covered by one pass of the block over Xs
    constexpr int XREG = BK / KSTEP;         // feature elements each thread stages per slice
    constexpr int WREG = BK * BM / NT;       // weight elements each thread stages per slice
    static_assert(NT % BN == 0 && BK % KSTEP == 0 && (BK * BM) % NT == 0, "tile/thread mismatch");
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    __shared__ float Ws[2][BK][BM + 1];   // +1: the k-fastest staging pattern writes a column of Ws per wave
    __shared__ float Xs[2][BK][BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ncols = a.N * a.T_out * a.V_out;
    const int j0 = blockIdx.x * BN;
    const int m0 = blockIdx.y * BM;

    // the one feature column this thread stages
    const int xj = tid % BN;
    const int xk0 = tid / BN;
    const ColInfo xc = decode_col(j0 + xj, ncols, a.T_out, a.V_out);

    // slice range of this workgroup
    const int s_total = slices_of(a.g[0]) + (a.ngroups > 1 ? slices_of(a.g[1]) : 0);
    const int s_beg = blockIdx.z * sp.per;
    const int s_end = min(s_total, s_beg + sp.per);

    kg_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < TN; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    float wreg[WREG], xreg[XREG];

    // Staging uses raw buffer loads: the descriptor base is wave-uniform (tensor base + tap / channel-chunk
    // term, SALU math), each element is a 32-bit byte offset, and an element that must read as zero (padding
    // frame, dropped vertex, channel or row beyond the tensor) gets an out-of-range offset, for which the
    // hardware returns 0 - no guarded loads (hipcc would branch around each one and serialise their
    // latencies), no clamps, no selects.
    constexpr unsigned W_RANGE = 0x40000000u;   // weight descriptor: 1 GiB; valid offsets are below it
    constexpr unsigned X_RANGE = 0x80000000u;   // feature descriptor: 2 GiB (validated on the host)
    constexpr unsigned W_OOB = 0x40000000u;     // adding one or two of these to a valid offset stays out of range
    constexpr unsigned X_OOB = 0x80000000u;

    // ---- per-group, per-thread staging state (integer divisions happen here, not per slice) -------------
    unsigned woff[WREG];            // byte offset of the m-part of this thread's i-th weight element (or W_OOB)
    int vi = -1;                    // source vertex of this thread's feature column
    bool kf = true;                 // weight staging pattern: k fastest (forward layouts) or m fastest (transposed)
    auto setup_group = [&](int gi) {
        const KgConvGroup& g = a.g[gi];
        kf = g.w_sI <= g.w_sO;
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int m = kf ? tid / BK + i * (NT / BK) : tid % BM;
            const int mm = m0 + m;
            const int mb = mm / g.w_MB;
            const unsigned off = (unsigned)(mb * g.w_sMB + (mm - mb * g.w_MB) * g.w_sO) * 4u;
            woff[i] = mm < a.M ? off : W_OOB;
        }
        vi = g.vmap ? (xc.valid ? g.vmap[xc.vo] : -1) : xc.vo;
    };

    // global -> registers for slice (gi, d, c0)
    auto fetch = [&](auto kfc, int gi, int d, int c0) {
        constexpr bool KF = decltype(kfc)::value;
        const KgConvGroup& g = a.g[gi];
        const long xoff = src_offset(g, xc, d, vi);
        const int choff = (g.tap_mode == KG_TAP_CHANBLOCK) ? d * g.Cin : 0;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(g.w + (long)d * g.w_sT), 0, (int)W_RANGE, 0x00020000);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            uniform_ptr(g.x + (long)(choff + c0) * g.x_sC), 0, (int)X_RANGE, 0x00020000);
        const unsigned wsi4 = (unsigned)g.w_sI * 4u;
        if constexpr (KF) {
            const int cc = c0 + tid % BK;
            const unsigned kterm = cc < g.Cin ? (unsigned)cc * wsi4 : W_OOB;
#pragma unroll
            for (int i = 0; i < WREG; ++i)
                wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, woff[i] + kterm, 0, 0));
        } else {
            const int k0 = c0 + tid / BM;
            const unsigned base = woff[0] + (unsigned)k0 * wsi4;
            const unsigned step = (unsigned)(NT / BM) * wsi4;
            const int nvalid = (g.Cin - k0 + (NT / BM) - 1) / (NT / BM);    // elements i < nvalid are inside Cin
#pragma unroll
            for (int i = 0; i < WREG; ++i)
                wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wr, i < nvalid ? base + i * step : W_OOB, 0, 0));
        }
        {
            const unsigned base = xoff >= 0 ? (unsigned)(((long)xk0 * g.x_sC + xoff) * 4) : X_OOB;
            const unsigned step = (unsigned)(KSTEP * g.x_sC * 4);
            const int nvalid = (g.Cin - c0 - xk0 + KSTEP - 1) / KSTEP;
#pragma unroll
            for (int i = 0; i < XREG; ++i)
                xreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    xr, i < nvalid ? base + i * step : X_OOB, 0, 0));
        }
    };
    // registers -> LDS buffer b
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
        float* q = &Xs[b][xk0][xj];
#pragma unroll
        for (int i = 0; i < XREG; ++i) q[i * KSTEP * BN] = xreg[i];
    };
    auto fetch_any = [&](int gi, int d, int c0) {
        if (kf) fetch(std::true_type{}, gi, d, c0);
        else fetch(std::false_type{}, gi, d, c0);
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
        int cchunks = (a.g[gi].Cin + BK - 1) / BK;
        int d = sl / cchunks;
        int cch = sl - d * cchunks;
        setup_group(gi);
        fetch_any(gi, d, cch * BK);
        stash_any(0);
        __syncthreads();
        for (int s = s_beg; s < s_end; ++s) {
            const int b = (s - s_beg) & 1;
            const bool more = s + 1 < s_end;
            if (more) {
                if (++cch == cchunks) {
                    cch = 0;
                    if (++d == a.g[gi].taps) {
                        d = 0;
                        ++gi;
                        cchunks = (a.g[gi].Cin + BK - 1) / BK;
                        setup_group(gi);
                    }
                }
                fetch_any(gi, d, cch * BK);
            }
            // keep the issue order loads -> MFMAs -> (wait + LDS writes): without the fences hipcc hoists the
            // LDS writes (and their vmcnt waits) above the MFMA loop and the load latency is exposed again
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int kr = kk + (lane >> 5);
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = Ws[b][kr][wm * (BM / WM) + i * 32 + (lane & 31)];
#pragma unroll
                for (int k = 0; k < TN; ++k) bv[k] = Xs[b][kr][wn * (BN / WN) + k * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int k = 0; k < TN; ++k)
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[k], acc[i][k], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) stash_any(b ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool partial = sp.nsplit > 1;
    float* slab = partial ? a.ws + (long)blockIdx.z * a.M * ncols : nullptr;
#pragma unroll
    for (int k = 0; k < TN; ++k) {
        const int j = j0 + wn * (BN / WN) + k * 32 + (lane & 31);
        const ColInfo oc = decode_col(j, ncols, a.T_out, a.V_out);
        if (!oc.valid) continue;
        const long ooff = (long)oc.n * a.o_sN + (long)oc.to * a.V_out + oc.vo;
        const long aoff = a.add ? (long)oc.n * a.a_sN + (long)(oc.to * a.a_tstride) * a.V_out + oc.vo : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= a.M) continue;
                float v = acc[i][k][r];
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

enum Tile { T128x128, T64x128, T32x128, T64x64, T32x64 };

struct Plan {
    Tile tile;
    int bm, bn;
    Split sp;
};

Plan make_plan(const KgConvArgs* a) {
    const long ncols = (long)a->N * a->T_out * a->V_out;
    const int M = a->M;
    int s_total = slices_of(a->g[0]) + (a->ngroups > 1 ? slices_of(a->g[1]) : 0);
    auto count = [&](int bm, int bn) { return (long)kg_cdiv(M, bm) * kg_cdiv(ncols, bn); };
    Plan p;
    // Largest tile that still gives every CU ~2.5 workgroups (measured on MI355X, tools/tune_conv.py: with
    // fewer resident waves the staging VALU work and the MFMA phases of a workgroup do not overlap).
    const long full = 600;
    if (M > 64 && count(128, 128) >= full)      { p.tile = T128x128; p.bm = 128; p.bn = 128; }
    else if (M > 32 && count(64, 128) >= full)  { p.tile = T64x128;  p.bm = 64;  p.bn = 128; }
    else if (count(32, 128) >= full || M <= 32) { p.tile = T32x128;  p.bm = 32;  p.bn = 128; }
    else                                        { p.tile = T64x64;   p.bm = 64;  p.bn = 64;  }
    if (p.tile == T32x128 && count(32, 128) < full / 2 && M <= 32) { p.tile = T32x64; p.bm = 32; p.bn = 64; }
    // tuning hook (tools/tune_conv.py): KG_CONV_PLAN="<tile 0..4>,<nsplit>" forces the plan
    int forced_split = 0;
    if (const char* env = getenv("KG_CONV_PLAN")) {
        int t = -1, ns = 0;
        if (sscanf(env, "%d,%d", &t, &ns) >= 1 && t >= 0 && t <= 4) {
            static const int bms[5] = {128, 64, 32, 64, 32}, bns[5] = {128, 128, 128, 64, 64};
            p.tile = (Tile)t; p.bm = bms[t]; p.bn = bns[t];
            forced_split = ns;
        }
    }
    const long wgs = count(p.bm, p.bn);
    int nsplit = 1;
    if (forced_split > 0) {
        nsplit = forced_split > s_total ? s_total : forced_split;
    } else if (wgs < full && s_total >= 4) {
        nsplit = (int)((1000 + wgs - 1) / wgs);              // aim at ~4 workgroups per CU
        if (nsplit > s_total / 2) nsplit = s_total / 2;      // at least two slices per split
        if (nsplit > 16) nsplit = 16;
        if (nsplit < 1) nsplit = 1;
    }
    p.sp.per = kg_cdiv(s_total, nsplit);
    p.sp.nsplit = kg_cdiv(s_total, p.sp.per);
    return p;
}

template <int BM, int BN, int WM, int WN>
int launch(const KgConvArgs* a, const Plan& p, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    dim3 grid(kg_cdiv(ncols, BN), kg_cdiv(a->M, BM), p.sp.nsplit);
    hipLaunchKernelGGL((kg_conv_kernel<BM, BN, WM, WN>), grid, dim3(64 * WM * WN), 0, s, *a, p.sp);
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
        const long xspan = 32L * (g.x_sC > 0 ? g.x_sC : -g.x_sC) + (long)(a->N - 1) * (g.x_sN > 0 ? g.x_sN : -g.x_sN) +
                           (long)g.T_in * g.V_in;
        KG_REQUIRE(g.x_sC >= 0 && g.x_sN >= 0 && xspan < (1L << 29),
                   "kg_conv: group %d feature tensor too large for 32-bit slice offsets (span %ld elements)", i, xspan);
        const long wspan = (long)(a->M / g.w_MB) * g.w_sMB + (long)(g.w_MB < a->M ? g.w_MB : a->M) * g.w_sO +
                           (long)g.Cin * g.w_sI;
        KG_REQUIRE(g.w_sO >= 0 && g.w_sI >= 0 && g.w_sMB >= 0 && g.w_sT >= 0 && wspan < (1L << 28),
                   "kg_conv: group %d weight tensor too large (span %ld elements)", i, wspan);
    }
    return 0;
}

}  // namespace

extern "C" int64_t kg_conv_workspace_bytes(const KgConvArgs* a) {
    if (validate(a) != 0) return -1;
    Plan p = make_plan(a);
    if (p.sp.nsplit <= 1) return 0;
    return (int64_t)p.sp.nsplit * a->M * a->N * a->T_out * a->V_out * (int64_t)sizeof(float);
}

extern "C" int kg_conv(const KgConvArgs* a, void* stream) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(a->out != nullptr, "kg_conv: null out");
    for (int i = 0; i < a->ngroups; ++i) KG_REQUIRE(a->g[i].x && a->g[i].w, "kg_conv: group %d null pointer", i);
    Plan p = make_plan(a);
    if (p.sp.nsplit > 1) {
        const int64_t need = (int64_t)p.sp.nsplit * a->M * a->N * a->T_out * a->V_out * (int64_t)sizeof(float);
        KG_REQUIRE(a->ws != nullptr && a->ws_bytes >= need, "kg_conv: workspace %ld < %ld bytes", (long)a->ws_bytes,
                   (long)need);
    }
    hipStream_t s = (hipStream_t)stream;
    switch (p.tile) {
        case T128x128: return launch<128, 128, 2, 2>(a, p, s);
        case T64x128:  return launch<64, 128, 2, 2>(a, p, s);
        case T32x128:  return launch<32, 128, 1, 4>(a, p, s);
        case T64x64:   return launch<64, 64, 2, 2>(a, p, s);
        default:       return launch<32, 64, 1, 2>(a, p, s);
    }
}
