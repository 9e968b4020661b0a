// kg_conv: the channel contraction of the st_gcn blocks as a "tap GEMM" on the fp32 matrix cores.
//
//   out[m, j] = act( sum_g sum_d sum_c W_g(d,m,c) * X_g[c, src_g(j,d)] + bias + add )
//
// M = output channels, columns j = (n, t, v) of the whole batch, K-slices = (group, tap, 16
// input channels).  v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): A = weights staged in
// LDS as Ws[k][m], B = feature tile staged as Xs[k][j] with the tap's time shift / stride /
// vertex gather folded into the global-load address and zero-filled outside the frame range
// (the conv's zero padding).  Bias, residual add and activation run on the accumulators.
//
// Reference ops covered: tgcn.py:61, discriminator.py:99-105,115-120,130-136,139-142,
// generator.py:134-140,154-159,176,182 and their backward-data passes (transposed mode).
#include "kg_common.h"

namespace {

constexpr int BK = 16;

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

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void kg_conv_kernel(const KgConvArgs a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int KSTEP = NT / BN;          // k-rows covered by one pass of the block over Xs
    static_assert(NT % BN == 0 && BK % KSTEP == 0, "tile/thread mismatch");
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    __shared__ float Ws[BK][BM];
    __shared__ float Xs[BK][BN];

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

    kg_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < TN; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    for (int gi = 0; gi < a.ngroups; ++gi) {
        const KgConvGroup& g = a.g[gi];
        const int vi = g.vmap ? (xc.valid ? g.vmap[xc.vo] : -1) : xc.vo;
        const bool w_k_fast = g.w_sI <= g.w_sO;     // which weight index is closer to contiguous
        for (int d = 0; d < g.taps; ++d) {
            const long xoff = src_offset(g, xc, d, vi);
            const int choff = (g.tap_mode == KG_TAP_CHANBLOCK) ? d * g.Cin : 0;
            const float* wtap = g.w + (long)d * g.w_sT;
            for (int c0 = 0; c0 < g.Cin; c0 += BK) {
                // ---- stage the weight tile Ws[k][m] = W(d, m0+m, c0+k)
                for (int e = tid; e < BK * BM; e += NT) {
                    int m, k;
                    if (w_k_fast) { m = e / BK; k = e - m * BK; }
                    else          { k = e / BM; m = e - k * BM; }
                    int mm = m0 + m, cc = c0 + k;
                    float v = 0.f;
                    if (mm < a.M && cc < g.Cin) {
                        int mb = mm / g.w_MB;
                        v = wtap[(long)mb * g.w_sMB + (long)(mm - mb * g.w_MB) * g.w_sO + (long)cc * g.w_sI];
                    }
                    Ws[k][m] = v;
                }
                // ---- stage the feature tile Xs[k][j]
#pragma unroll
                for (int kk = 0; kk < BK; kk += KSTEP) {
                    int k = kk + xk0;
                    int cc = c0 + k;
                    float v = 0.f;
                    if (xoff >= 0 && cc < g.Cin) v = g.x[(long)(choff + cc) * g.x_sC + xoff];
                    Xs[k][xj] = v;
                }
                __syncthreads();
                // ---- 32x32x2 MFMA over the 16-deep slice
#pragma unroll
                for (int kk = 0; kk < BK; kk += 2) {
                    const int kr = kk + (lane >> 5);
                    float av[TM], bv[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[i] = Ws[kr][wm * (BM / WM) + i * 32 + (lane & 31)];
#pragma unroll
                    for (int k = 0; k < TN; ++k) bv[k] = Xs[kr][wn * (BN / WN) + k * 32 + (lane & 31)];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int k = 0; k < TN; ++k)
                            acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[k], acc[i][k], 0, 0, 0);
                }
                __syncthreads();
            }
        }
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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
                if (a.bias0) v += a.bias0[m];
                if (a.bias1) v += a.bias1[m];
                if (a.add) v += a.add[(long)m * a.a_sC + aoff];
                a.out[(long)m * a.o_sC + ooff] = kg_act(v, a.act, a.slope);
            }
        }
    }
}

struct TileCfg { int bm, bn; };

template <int BM, int BN, int WM, int WN>
int launch(const KgConvArgs* a, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    dim3 grid(kg_cdiv(ncols, BN), kg_cdiv(a->M, BM));
    hipLaunchKernelGGL((kg_conv_kernel<BM, BN, WM, WN>), grid, dim3(64 * WM * WN), 0, s, *a);
    return kg_launch_status("kg_conv");
}

}  // namespace

extern "C" int kg_conv(const KgConvArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr, "kg_conv: null args");
    KG_REQUIRE(a->N > 0 && a->M > 0 && a->T_out > 0 && a->V_out > 0, "kg_conv: bad dims N=%d M=%d T=%d V=%d",
               a->N, a->M, a->T_out, a->V_out);
    KG_REQUIRE((long)a->N * a->T_out * a->V_out < (1L << 31), "kg_conv: too many columns");
    KG_REQUIRE(a->out != nullptr, "kg_conv: null out");
    KG_REQUIRE(a->ngroups >= 1 && a->ngroups <= 2, "kg_conv: ngroups=%d", a->ngroups);
    KG_REQUIRE(a->act >= KG_ACT_NONE && a->act <= KG_ACT_TANH, "kg_conv: act=%d", a->act);
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        KG_REQUIRE(g.x && g.w, "kg_conv: group %d null pointer", i);
        KG_REQUIRE(g.Cin > 0 && g.T_in > 0 && g.V_in > 0, "kg_conv: group %d bad input dims", i);
        KG_REQUIRE(g.taps == 1 || g.taps == 3, "kg_conv: group %d taps=%d (1 or 3)", i, g.taps);
        KG_REQUIRE(g.tap_mode == KG_TAP_TIME || g.tap_mode == KG_TAP_CHANBLOCK, "kg_conv: group %d tap_mode", i);
        KG_REQUIRE(g.t_stride >= 1, "kg_conv: group %d t_stride=%d", i, g.t_stride);
        KG_REQUIRE(g.w_MB >= 1, "kg_conv: group %d w_MB=%d", i, g.w_MB);
        KG_REQUIRE(g.vmap != nullptr || g.V_in == a->V_out, "kg_conv: group %d V_in=%d != V_out=%d without vmap",
                   i, g.V_in, a->V_out);
    }
    hipStream_t s = (hipStream_t)stream;
    const long ncols = (long)a->N * a->T_out * a->V_out;
    const int M = a->M;
    auto count = [&](int bm, int bn) { return (long)kg_cdiv(M, bm) * kg_cdiv(ncols, bn); };
    const long want = 384;   // ~1.5 workgroups per CU before we prefer a bigger tile
    if (M > 64 && count(128, 128) >= want) return launch<128, 128, 2, 2>(a, s);
    if (M > 32 && count(64, 128) >= want) return launch<64, 128, 2, 2>(a, s);
    if (M <= 32 && count(32, 128) >= want) return launch<32, 128, 1, 4>(a, s);
    if (M > 32 && count(64, 64) >= want) return launch<64, 64, 2, 2>(a, s);
    if (M <= 32 && count(32, 64) >= want / 2) return launch<32, 64, 1, 2>(a, s);
    if (M > 32 && count(64, 32) >= want / 2) return launch<64, 32, 2, 1>(a, s);
    return launch<32, 32, 1, 1>(a, s);
}
