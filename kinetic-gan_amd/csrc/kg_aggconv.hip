// kg_aggconv: the spatial graph aggregation FUSED into the channel contraction that consumes it - the "gcn" half of a
// discriminator st_gcn block in aggregate-first order (tgcn.py:61-66 computes conv1x1 then einsum; sum_k (W_k x) A_k ==
// sum_k W_k (x A_k)):
//
//   out[m, (n,t,w)] = sum_k sum_c W(k,m,c) * xa_k[c, (n,t,w)] (+ add),      xa_k[c, (n,t,w)] = sum_v x[c, (n,t,v)] A[k,v,w]
//
// Unfused this is kg_agg_expand (3*Cin planes written to HBM) followed by kg_conv reading them back.  Here the
// K*Cin aggregated planes never exist in HBM: a workgroup owns BM output channels x 128 columns (n,t,w); per slice of
// 16 input channels it stages, ONCE for the three partitions, the contiguous run of source frames its columns read
// (coalesced rows of x -> LDS) and the three weight tiles; every lane then forms the B operand of its MFMA column on
// the fly from LDS with the (at most 1 / 4 / 1) non-zeros of its column of A_0 / A_1 / A_2 - the adjacency is ~4 %
// dense with a fixed pattern, passed as a neighbour table, values read from the live A_eff = A * importance.
// v_mfma_f32_32x32x2_f32 (exact fp32) does only the dense channel contraction.  Against the unfused pair the feature
// operand is fetched once per 3 taps (1/3 of the vector-memory loads per MFMA), there is one barrier per 24*TM MFMAs
// instead of one per 16*TM, and one launch + one HBM round trip of 3*Cin planes disappear.
// Optionally the aggregated planes are ALSO written out (xa): the weight gradient of the gcn conv needs them, and the
// lanes hold them anyway.
#include <stdlib.h>

#include <type_traits>

#include "kg_common.h"

namespace {

constexpr int DK = 16;         // input channels per slice
constexpr int NW = 4;          // column waves: 32 columns each
constexpr int BN = 32 * NW;    // columns per workgroup
constexpr int PMAX = 4;        // width of the neighbour table
constexpr unsigned OOB = 0x80000000u;

struct AcPlan {
    int spanp;                 // floats per staged source row (frames of one column tile x V, padded to 4)
    int tapmask;               // bit k: partition k has a non-zero in some kept column
    int xa_store;
};

// P0/P1/P2: most non-zeros per column of A_0/A_1/A_2 (table entries beyond are not read); XE: 64-float pieces per row.
// KS: k-split INSIDE the workgroup.  KS = 2: eight waves, wave (cw, kg) owns the 32 columns of column-wave cw and the
// k-steps q = kg (mod 2) of every slice; the two partial accumulators of a column-wave are summed through LDS at the
// end.  The whole 64- (or 32-) row tile then sees every staged feature / weight element from ONE staging pass while
// twice the waves share the MFMA work: at the batch sizes of BASELINE.json the launches have too few tiles for big
// tiles with four waves (2-3 workgroups per CU is all there is), and with small tiles every source row is staged and
// aggregated once per 32 output channels.
// ADD = false: the lean epilogue - no `add` operand (block 0's per-sample label bias) AND rows that fill whole tiles (the
// host checks): neither the operand path nor the per-store row guards are compiled in (round 5: the same change took
// 3-6 % + 2-4 % off every kg_conv launch).  ADD = true: the general epilogue (operand optional, rows guarded).
template <int BM, int XE, int KS, bool XA, bool ADD, int P0, int P1, int P2>
__global__ __launch_bounds__(64 * NW * KS) void kg_aggconv_kernel(const KgAggConvArgs a, const AcPlan pl) {
    constexpr int NT = 64 * NW * KS;
    constexpr int TM = BM / 32;
    constexpr int WPITCH = BM + 1;
    constexpr int MPT = NT / DK;                // weight rows covered per staging pass
    constexpr int WI = BM / MPT;                // weight loads per thread, tap and slice
    constexpr int WL = 3 * WI;
    constexpr int RPW = DK / (NW * KS);         // source rows staged per wave
    constexpr int QS = DK / 2 / KS;             // k-steps per wave, tap and slice
    static_assert(BM % MPT == 0 && DK % (NW * KS) == 0, "tile / thread mismatch");
    extern __shared__ float kg_acsm[];
    const int SP = pl.spanp;
    float* const Xs = kg_acsm;                               // [2][DK][SP]
    float* const Ws = kg_acsm + 2 * DK * SP;                 // [2][3][DK][WPITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cw = wave & (NW - 1), kg = wave / NW;          // column wave, k group
    const int kh = lane >> 5;
    const int ncols = a.N * a.T * a.W;
    const int ctile = blockIdx.x, rtile = blockIdx.y;
    const int m0 = rtile * BM;
    const int j = ctile * BN + cw * 32 + (lane & 31);
    const bool valid = j < ncols;
    const int jj = valid ? j : 0;
    // (round 4: quotients by float reciprocal, exact for the < 2^22 columns / frames of any launch this kernel is chosen
    // for - the prologue had ~10 integer divisions of ~40 instructions each per thread)
    const bool small = ncols < (1 << 22);                    // (uniform)
    auto divw = [&](int x, int d, int& q, int& r) {
        if (small) {
            q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
            r = x - q * d;
            if (r < 0) { --q; r += d; }
            if (r >= d) { ++q; r -= d; }
        } else {
            q = x / d; r = x - q * d;
        }
    };
    int f, wv;
    divw(jj, a.W, f, wv);                                    // global frame index (n*T + t) and kept vertex
    const int f_lo = (ctile * BN) / a.W;                     // (uniform) first frame of the tile
    const int nframes = a.N * a.T;

    // ---- this lane's column of the three partitions: LDS position of each source vertex and its weight.  Filled BEHIND
    // the first slice's loads (round 4): neighbour index -> adjacency value are two dependent global loads per entry,
    // which used to sit in front of the first feature / weight fetch - three memory round trips before the first MFMA
    constexpr int PK[3] = {P0, P1, P2};
    int src[3][PMAX];
    float av[3][PMAX];
    int n_f, t_f;
    divw(f, a.T, n_f, t_f);                                  // sample and frame of this lane's column

    // ---- staging maps.  x: wave w stages rows w, w+4, ..; lane covers positions e = lane + 64 i of a row
    unsigned xoff[XE];
#pragma unroll
    for (int i = 0; i < XE; ++i) {
        const int e = lane + 64 * i;
        int fq, ve, n, t;
        divw(e, a.V, fq, ve);
        const int fe = f_lo + fq;
        const bool ok = e < SP && fe < nframes;
        divw(ok ? fe : 0, a.T, n, t);
        xoff[i] = ok ? (unsigned)(((long)n * a.x_sN + (long)t * a.V + ve) * 4) : OOB;
    }
    // weights: thread -> (c = tid % 16, m = tid / 16 + MPT i) for each partition
    const int wc = tid & (DK - 1), wm = tid >> 4;

    kg_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float xreg[RPW][XE];
    float wreg[WL];
    const int nslices = (a.Cin + DK - 1) / DK;

    auto fetch = [&](int s) {
        const int c0 = s * DK;
        const bool live = s < nslices;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            kg_uniform_ptr(a.x + (long)c0 * a.x_sC), 0, (int)0x80000000u, 0x00020000);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int row = wave + NW * KS * r;
            const bool rl = live && c0 + row < a.Cin;
            const unsigned rb = (unsigned)((long)row * a.x_sC * 4);
#pragma unroll
            for (int i = 0; i < XE; ++i)
                xreg[r][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    xr, (rl && xoff[i] != OOB) ? rb + xoff[i] : OOB, 0, 0));
        }
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            kg_uniform_ptr(a.w), 0, (int)0x40000000u, 0x00020000);
        const bool cl = live && c0 + wc < a.Cin;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                const int m = m0 + wm + MPT * i;
                const bool ok = cl && m < a.M && k < a.K;
                const unsigned off = (unsigned)(((long)k * a.w_sT + (long)m * a.w_sO + (long)(c0 + wc) * a.w_sI) * 4);
                wreg[k * WI + i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wr, ok ? off : 0x40000000u, 0, 0));
            }
    };
    auto stash = [&](int b) {
        float* xs = Xs + b * DK * SP;
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int i = 0; i < XE; ++i)
                if (lane + 64 * i < SP) xs[(wave + NW * KS * r) * SP + lane + 64 * i] = xreg[r][i];
        float* ws = Ws + b * 3 * DK * WPITCH;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < WI; ++i) ws[(k * DK + wc) * WPITCH + wm + MPT * i] = wreg[k * WI + i];
    };
    // the aggregated planes as a side product (first row tile only): xa[k*Cin + c, column j].  Stores of lanes without
    // a column / rows beyond Cin use an out-of-range buffer offset (dropped by the hardware): no branch in the loop.
    const bool xa_on = XA && rtile == 0;                           // (uniform)
    unsigned xa_col = OOB;
    if (XA && valid) {
        xa_col = (unsigned)(((long)n_f * a.xa_sN + (long)t_f * a.W + wv) * 4);
    }
    // The k-step loops below are branch-free straight-line code (rows beyond Cin are zero in LDS: their loads were out
    // of range) so that the LDS reads of later k-steps can be scheduled under the MFMAs of earlier ones.
    // ALL: all three partitions carry non-zeros (every launch the fused path is chosen for) - one straight-line
    // region per slice; otherwise a partition without non-zeros is skipped by a (uniform) branch.
    auto compute = [&](int b, int s, auto all_taps) {
        constexpr bool ALL = decltype(all_taps)::value;
        // this wave's k-steps of the slice: q = kg + KS * qq (rows 2q + kh of the staged tiles)
        const float* xs = Xs + b * DK * SP + (2 * kg + kh) * SP;
        const float* ws = Ws + b * 3 * DK * WPITCH + (2 * kg + kh) * WPITCH + (lane & 31);
        const int c0 = s * DK;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool tap_on = ALL || ((pl.tapmask >> k) & 1);    // (uniform) partition with any non-zero
            if (!ALL && !tap_on && !(xa_on && k < a.K)) continue;
            __amdgpu_buffer_rsrc_t xar;
            if (XA) xar = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.xa + (long)(k * a.Cin + c0) * a.xa_sC), 0,
                                                            (int)0x80000000u, 0x00020000);
            float bvs[QS];
#pragma unroll
            for (int qq = 0; qq < QS; ++qq) {
                const float* xrow = xs + 2 * KS * qq * SP;
                float bv = av[k][0] * xrow[src[k][0]];
#pragma unroll
                for (int p = 1; p < PMAX; ++p)
                    if (p < PK[k]) bv = fmaf(av[k][p], xrow[src[k][p]], bv);
                bvs[qq] = bv;
            }
            if (XA && xa_on) {
#pragma unroll
                for (int qq = 0; qq < QS; ++qq) {
                    const int crel = 2 * kg + kh + 2 * KS * qq;
                    const unsigned off = (c0 + crel < a.Cin && xa_col != OOB) ? (unsigned)((long)crel * a.xa_sC * 4) + xa_col : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bvs[qq]), xar, off, 0, 0);
                }
            }
            if (ALL || tap_on) {
#pragma unroll
                for (int qq = 0; qq < QS; ++qq)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ws[(k * DK + 2 * KS * qq) * WPITCH + i * 32], bvs[qq], acc[i], 0, 0, 0);
            }
        }
    };

    fetch(0);
    {
        int nb[3][PMAX];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int p = 0; p < PMAX; ++p)
                nb[k][p] = (p < PK[k] && k < a.K) ? a.nbr[(k * a.W + wv) * PMAX + p] : -1;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int p = 0; p < PMAX; ++p) {
                src[k][p] = 0;
                av[k][p] = 0.f;
                if (p < PK[k] && k < a.K) {
                    const int v = nb[k][p];
                    const bool ok = valid && v >= 0;
                    const int vv = ok ? v : 0;
                    const float val = a.a_transposed ? a.a[((long)k * a.W + wv) * a.V + vv] : a.a[((long)k * a.V + vv) * a.W + wv];
                    av[k][p] = ok ? val : 0.f;
                    src[k][p] = (f - f_lo) * a.V + vv;
                }
            }
    }
    stash(0);
    __syncthreads();
    if (pl.tapmask == 7) {
        for (int s = 0; s < nslices; ++s) {
            const int b = s & 1;
            fetch(s + 1);                      // dead slice after the last one: every offset out of range
            compute(b, s, std::true_type{});
            stash(b ^ 1);
            __syncthreads();
        }
    } else {
        for (int s = 0; s < nslices; ++s) {
            const int b = s & 1;
            fetch(s + 1);
            compute(b, s, std::false_type{});
            stash(b ^ 1);
            __syncthreads();
        }
    }

    // ---- the k groups' partial tiles are summed through LDS (the staging buffers are free after the last barrier)
    if constexpr (KS == 2) {
        float* red = kg_acsm + (cw * TM * 16) * 64 + lane;      // [cw][i][r][lane]
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(i * 16 + r) * 64] = acc[i][r];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] += red[(i * 16 + r) * 64];
    }
    // ---- epilogue.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (!valid) return;
    const int n = n_f, t = t_f;
    float* op = a.out + (long)n * a.o_sN + (long)t * a.W + wv;
    const float* ap = (ADD && a.add) ? a.add + (long)n * a.a_sN + (long)(t * a.a_tstride) * a.W + wv : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (!ADD || m < a.M) {
                float v = acc[i][r];
                if (ADD && ap) v += ap[(long)m * a.a_sC];
                op[(long)m * a.o_sC] = v;
            }
        }
}

// Tiny-channel form (discriminator block 0: 3 data channels x 3 partitions = 9 aggregated values per column, 32 output
// rows): the 32 x 128 MFMA tile contracts over a 16-channel slice of which 3 are real, and the launch - 0.08 GFLOP - took
// 43 us at 192 samples.  Here a thread owns ONE output column: it aggregates its K * Cin values from the (L1-resident)
// source frame with the column's neighbour list, multiplies them with the weights broadcast from LDS and writes the M
// outputs (and the aggregated planes, if asked): a streaming VALU kernel bound by its 17 MB of output stores.
constexpr int AT_MAXM = 64, AT_K = 3, AT_C = 4;      // <= 3 partitions x <= 4 input channels

// (every private array is indexed with compile-time constants only - loops over K / Cin are unrolled to their maxima
// and predicated: a runtime index would put the arrays in scratch memory)
template <int MT>
__global__ __launch_bounds__(256, 4) void kg_aggconv_tiny_kernel(const KgAggConvArgs a) {
    __shared__ float Wl[AT_K * AT_C][MT];
    __shared__ float Av[AT_K * 32 * PMAX];
    __shared__ int Nb[AT_K * 32 * PMAX];
    const int tid = threadIdx.x;
    for (int e = tid; e < AT_K * AT_C * MT; e += 256) {
        const int kc = e / MT, m = e - kc * MT;
        const int k = kc / AT_C, c = kc - k * AT_C;
        Wl[kc][m] = (m < a.M && k < a.K && c < a.Cin) ? a.w[(long)k * a.w_sT + (long)m * a.w_sO + (long)c * a.w_sI] : 0.f;
    }
    // the column's neighbour list and adjacency values come from LDS: read per column from global memory they are two
    // more levels of dependent loads in front of the feature loads
    for (int e = tid; e < a.K * a.W * PMAX; e += 256) {
        const int kw = e / PMAX, p = e - kw * PMAX;
        const int k = kw / a.W, w = kw - k * a.W;
        int v = p < a.pcount[k] ? a.nbr[e] : -1;
        float av = 0.f;
        if (v >= 0) av = a.a_transposed ? a.a[((long)k * a.W + w) * a.V + v] : a.a[((long)k * a.V + v) * a.W + w];
        Nb[e] = v < 0 ? 0 : v;
        Av[e] = av;                                    // (absent neighbour: weight 0 on vertex 0)
    }
    __syncthreads();
    const int ncols = a.N * a.T * a.W;
    const int j = blockIdx.x * 256 + tid;
    if (j >= ncols) return;
    const int f = j / a.W, wv = j - f * a.W;
    const int n = f / a.T, t = f - n * a.T;
    float xa[AT_K][AT_C];
#pragma unroll
    for (int k = 0; k < AT_K; ++k)
#pragma unroll
        for (int c = 0; c < AT_C; ++c) xa[k][c] = 0.f;
    const float* xp = a.x + (long)n * a.x_sN + (long)t * a.V;
#pragma unroll
    for (int k = 0; k < AT_K; ++k) {
        if (k < a.K) {
#pragma unroll
            for (int p = 0; p < PMAX; ++p) {
                const int e = (k * a.W + wv) * PMAX + p;
                const int v = Nb[e];
                const float av = Av[e];
#pragma unroll
                for (int c = 0; c < AT_C; ++c)
                    if (c < a.Cin) xa[k][c] = fmaf(av, xp[(long)c * a.x_sC + v], xa[k][c]);
            }
        }
    }
    if (a.xa && blockIdx.y == 0) {
        float* xo = a.xa + (long)n * a.xa_sN + (long)t * a.W + wv;
#pragma unroll
        for (int k = 0; k < AT_K; ++k)
#pragma unroll
            for (int c = 0; c < AT_C; ++c)
                if (k < a.K && c < a.Cin) xo[(long)(k * a.Cin + c) * a.xa_sC] = xa[k][c];
    }
    float* op = a.out + (long)n * a.o_sN + (long)t * a.W + wv;
    const float* ap = a.add ? a.add + (long)n * a.a_sN + (long)(t * a.a_tstride) * a.W + wv : nullptr;
    // eight output rows per thread, the row groups side by side in grid.y (a thread that walks all M rows is a serial
    // chain of 4-8 load / multiply / store rounds: with one workgroup per CU at 64 samples the launch took 13 us)
    {
        const int m0 = blockIdx.y * 8;
        float acc[8], addv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc[q] = 0.f;
            addv[q] = (ap && m0 + q < a.M) ? ap[(long)(m0 + q) * a.a_sC] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < AT_K; ++k)
#pragma unroll
            for (int c = 0; c < AT_C; ++c) {
                const float v = xa[k][c];                    // (0 beyond K / Cin, and so are the weights)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(Wl[k * AT_C + c][m0 + q], v, acc[q]);
            }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (m0 + q < a.M) op[(long)(m0 + q) * a.o_sC] = acc[q] + addv[q];
    }
}

bool tiny_form(const KgAggConvArgs* a) {
    return a->K <= AT_K && a->Cin <= AT_C && a->M <= AT_MAXM && a->W <= 32 && kg_env().aggconv_plan == 0;
}

int validate(const KgAggConvArgs* a) {
    KG_REQUIRE(a != nullptr, "kg_aggconv: null args");
    KG_REQUIRE(a->N > 0 && a->Cin > 0 && a->M > 0 && a->T > 0 && a->V > 0 && a->W > 0, "kg_aggconv: bad dims");
    KG_REQUIRE(a->K >= 1 && a->K <= 3, "kg_aggconv: K=%d (1..3)", a->K);
    KG_REQUIRE((long)a->N * a->T * a->W < (1L << 31) && (long)a->N * a->T < (1L << 30), "kg_aggconv: too many columns");
    KG_REQUIRE(a->x && a->a && a->nbr && a->w && a->out, "kg_aggconv: null pointer");
    for (int k = 0; k < 3; ++k)
        KG_REQUIRE(a->pcount[k] >= 0 && a->pcount[k] <= PMAX, "kg_aggconv: pcount[%d]=%d", k, a->pcount[k]);
    KG_REQUIRE(a->pcount[0] <= 1 && a->pcount[2] <= 1,
               "kg_aggconv: adjacency pattern (%d, %d, %d non-zeros per column) is not supported by the fused kernel",
               a->pcount[0], a->pcount[1], a->pcount[2]);
    const long xspan = 16L * a->x_sC + (long)(a->N - 1) * a->x_sN + (long)a->T * a->V;
    KG_REQUIRE(a->x_sC >= 0 && a->x_sN >= 0 && xspan < (1L << 29), "kg_aggconv: x too large for 32-bit slice offsets");
    const long wspan = 3L * a->w_sT + (long)a->M * a->w_sO + (long)a->Cin * a->w_sI;
    KG_REQUIRE(a->w_sT >= 0 && a->w_sO >= 0 && a->w_sI >= 0 && wspan < (1L << 28), "kg_aggconv: weight tensor too large");
    return 0;
}

// floats of one staged source row: the frames a 128-column tile can touch
int span_of(const KgAggConvArgs* a) { return ((BN - 1) / a->W + 2) * a->V; }

template <int BM, int XE, int KS, bool XA>
int launch_xa(const KgAggConvArgs* a, const AcPlan& pl, hipStream_t s) {
    const int ncols = a->N * a->T * a->W;
    dim3 grid(kg_cdiv(ncols, BN), kg_cdiv(a->M, BM));
    size_t lds = (size_t)(2 * DK * pl.spanp + 2 * 3 * DK * (BM + 1)) * sizeof(float);
    if (KS == 2 && lds < (size_t)NW * (BM / 32) * 16 * 64 * sizeof(float)) lds = (size_t)NW * (BM / 32) * 16 * 64 * sizeof(float);
    auto kern0 = kg_aggconv_kernel<BM, XE, KS, XA, false, 1, 4, 1>;
    auto kern1 = kg_aggconv_kernel<BM, XE, KS, XA, true, 1, 4, 1>;
    static unsigned long long attr_mask = 0;
    if (kg_first_on_device(attr_mask)) {
        KG_SET_DYN_LDS(kern0, 96 * 1024);
        KG_SET_DYN_LDS(kern1, 96 * 1024);
    }
    if (a->add || a->M % BM != 0) hipLaunchKernelGGL(kern1, grid, dim3(64 * NW * KS), lds, s, *a, pl);
    else                          hipLaunchKernelGGL(kern0, grid, dim3(64 * NW * KS), lds, s, *a, pl);
    return kg_launch_status("kg_aggconv");
}

template <int BM, int XE, int KS>
int launch(const KgAggConvArgs* a, const AcPlan& pl, hipStream_t s) {
    return pl.xa_store ? launch_xa<BM, XE, KS, true>(a, pl, s) : launch_xa<BM, XE, KS, false>(a, pl, s);
}

}  // namespace

extern "C" int kg_aggconv_supported(const KgAggConvArgs* a) {
    if (validate(a) != 0) return 0;
    if (tiny_form(a)) return 1;
    return span_of(a) <= 384 ? 1 : 0;
}

extern "C" int kg_aggconv(const KgAggConvArgs* a, void* stream) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(a->xa == nullptr || (a->xa_sC > 0), "kg_aggconv: xa strides");
    if (tiny_form(a)) {
        const int ncols = a->N * a->T * a->W;
        const dim3 grid(kg_cdiv(ncols, 256), kg_cdiv(a->M, 8));
        if (a->M <= 32) hipLaunchKernelGGL(kg_aggconv_tiny_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, *a);
        else            hipLaunchKernelGGL(kg_aggconv_tiny_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, *a);
        return kg_launch_status("kg_aggconv (tiny)");
    }
    const int span = span_of(a);
    KG_REQUIRE(span <= 384, "kg_aggconv: source span %d floats per tile > 384 (use kg_agg_expand + kg_conv)", span);
    KG_REQUIRE(a->xa == nullptr || (a->xa_sC > 0), "kg_aggconv: xa strides");
    AcPlan pl;
    pl.spanp = (span + 3) / 4 * 4;
    pl.tapmask = 0;
    for (int k = 0; k < a->K; ++k)
        if (a->pcount[k] > 0) pl.tapmask |= 1 << k;
    pl.xa_store = a->xa != nullptr;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = pl.spanp > 192;
    // tuning hook (tools/time_aggconv.py): KG_AGGCONV_PLAN = "<BM><KS>", e.g. "642"
    const int envplan = kg_env().aggconv_plan;
    // 64-row tiles (every source row staged and aggregated once per 64 output channels) where they still leave every
    // CU two or more workgroups; measured on MI355X (profiles/r02_time_aggconv.log)
    const long ctiles = kg_cdiv((long)a->N * a->T * a->W, BN);
    int bm = (a->M >= 64 && a->M <= 128 && ctiles * kg_cdiv(a->M, 64) >= 600) ? 64 : 32, ks = 1;
    // the critic's 3n launches (D2 / D3 at 192 samples, tools/time_aggconv.py KG_AGGCONV_SWEEP=1): eight waves that
    // split the channel slices in two share one staged source span per 64 rows - 10-12 % faster than the above
    if (a->M >= 128 && ctiles * kg_cdiv(a->M, 64) >= 900) { bm = 64; ks = 2; }
    if (envplan > 0) {
        bm = envplan / 10;
        ks = envplan % 10;
    }
    if (bm == 64 && ks == 2) return wide ? launch<64, 6, 2>(a, pl, s) : launch<64, 3, 2>(a, pl, s);
    if (bm == 64)            return wide ? launch<64, 6, 1>(a, pl, s) : launch<64, 3, 1>(a, pl, s);
    if (ks == 2)             return wide ? launch<32, 6, 2>(a, pl, s) : launch<32, 3, 2>(a, pl, s);
    return wide ? launch<32, 6, 1>(a, pl, s) : launch<32, 3, 1>(a, pl, s);
}
