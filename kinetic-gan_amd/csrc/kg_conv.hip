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

    __shared__ float Ws[2][BK][BM];
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

    // global -> registers for slice s
    auto fetch = [&](int s) {
        int gi = 0, sl = s;
        const int s0 = slices_of(a.g[0]);
        if (sl >= s0) { gi = 1; sl -= s0; }
        const KgConvGroup& g = a.g[gi];
        const int cchunks = (g.Cin + BK - 1) / BK;
        const int d = sl / cchunks;
        const int c0 = (sl - d * cchunks) * BK;
        const int vi = g.vmap ? (xc.valid ? g.vmap[xc.vo] : -1) : xc.vo;
        const long xoff = src_offset(g, xc, d, vi);
        const int choff = (g.tap_mode == KG_TAP_CHANBLOCK) ? d * g.Cin : 0;
        const float* wtap = g.w + (long)d * g.w_sT;
        const bool w_k_fast = g.w_sI <= g.w_sO;
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int e = tid + i * NT;
            int m, k;
            if (w_k_fast) { m = e / BK; k = e - m * BK; }
            else          { k = e / BM; m = e - k * BM; }
            const int mm = m0 + m, cc = c0 + k;
            float v = 0.f;
            if (mm < a.M && cc < g.Cin) {
                const int mb = mm / g.w_MB;
                v = wtap[(long)mb * g.w_sMB + (long)(mm - mb * g.w_MB) * g.w_sO + (long)cc * g.w_sI];
            }
            wreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < XREG; ++i) {
            const int cc = c0 + xk0 + i * KSTEP;
            float v = 0.f;
            if (xoff >= 0 && cc < g.Cin) v = g.x[(long)(choff + cc) * g.x_sC + xoff];
            xreg[i] = v;
        }
        return w_k_fast;
    };
    // registers -> LDS buffer b
    auto stash = [&](int b, bool w_k_fast) {
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int e = tid + i * NT;
            int m, k;
            if (w_k_fast) { m = e / BK; k = e - m * BK; }
            else          { k = e / BM; m = e - k * BM; }
            Ws[b][k][m] = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < XREG; ++i) Xs[b][xk0 + i * KSTEP][xj] = xreg[i];
    };

    if (s_beg < s_end) {
        bool kf = fetch(s_beg);
        stash(0, kf);
        __syncthreads();
        for (int s = s_beg; s < s_end; ++s) {
            const int b = (s - s_beg) & 1;
            const bool more = s + 1 < s_end;
            if (more) kf = fetch(s + 1);
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
            if (more) stash(b ^ 1, kf);
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
    const long full = 256;     // one workgroup per CU
    if (M > 64 && count(128, 128) >= full)      { p.tile = T128x128; p.bm = 128; p.bn = 128; }
    else if (M > 32 && count(64, 128) >= full)  { p.tile = T64x128;  p.bm = 64;  p.bn = 128; }
    else if (M <= 32 && count(32, 128) >= full) { p.tile = T32x128;  p.bm = 32;  p.bn = 128; }
    else if (M > 32)                            { p.tile = T64x64;   p.bm = 64;  p.bn = 64;  }
    else                                        { p.tile = T32x64;   p.bm = 32;  p.bn = 64;  }
    const long wgs = count(p.bm, p.bn);
    int nsplit = 1;
    if (wgs < full && s_total >= 4) {
        nsplit = (int)((2 * full + wgs - 1) / wgs);          // aim at ~2 workgroups per CU
        if (nsplit > s_total / 2) nsplit = s_total / 2;      // at least two slices per split
        if (nsplit > 32) nsplit = 32;
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
