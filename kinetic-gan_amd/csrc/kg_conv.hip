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

__host__ __device__ inline int slices_of(const KgConvGroup& g, int bk = BK) { return g.taps * ((g.Cin + bk - 1) / bk); }

struct Split {
    int nsplit;          // workgroups along K
    int per;             // slices per split
};

constexpr unsigned W_RANGE = 0x40000000u;   // weight descriptor: 1 GiB; valid offsets are below it
constexpr unsigned X_RANGE = 0x80000000u;   // feature descriptor: 2 GiB (validated on the host)
constexpr unsigned W_OOB = 0x40000000u;     // adding one or two of these to a valid offset stays out of range
constexpr unsigned X_OOB = 0x80000000u;

constexpr int PADF = 32;                    // floats the 128-bit path may read in front of a channel row

typedef float kg_f4 __attribute__((ext_vector_type(4)));
typedef int kg_i4 __attribute__((ext_vector_type(4)));

// per K-slice-group state: everything that costs a kernel-argument read or an integer division is computed once,
// before the slice loop, for both groups; the loop selects between the two copies with v_cndmask / s_cselect
template <int WREG>
struct GroupState {
    unsigned woff[WREG];            // per thread: byte offset of the m-part of its i-th weight element (or W_OOB)
    unsigned xoff[3];               // per thread: byte offset of its column(s)' source for tap 0..2 (or X_OOB)
    unsigned tapmask;               // 128-bit path: 4 validity bits per tap for the lane's four columns
    const float* x;                 // wave-uniform geometry
    const float* w;
    long xsC, wsT, extent;
    int Cin, taps, cchunks, chanblock;
    unsigned wsi4;
};

// One kernel, two operand-load flavours (XV = feature elements per load):
//  XV = 1  general: any stride / vertex gather / layout; a wave owns 32 columns, K-slices are 32 channels deep.
//  XV = 4  "column-contiguous" launches (every group: stride 1, no vertex gather, same (T, V) in and out,
//          channel-major features whose samples follow each other without a gap).  The source of output column
//          j for tap d is then j + shift_d * V, so a lane fetches FOUR consecutive columns of a channel row
//          with one buffer_load_dwordx4.  A wave owns 128 columns split into four INTERLEAVED 32-column MFMA
//          tiles (tile q = columns 4j+q): component q of the lane's float4 is directly its B operand for tile q.
//          Frames that a temporal tap shifts out of [0, T) are zeroed by a 4-bit per-tap lane mask right before
//          the MFMA; reads that a negative shift moves in front of a row stay inside the allocation
//          (x_lead >= 32 floats), reads behind the tensor are out of the descriptor's range.  Slices are 16 deep.
// KF: weight staging pattern - k fastest (forward layouts) or m fastest (transposed).
//
// The slice loop is STRAIGHT-LINE code: slices are processed in pairs (ping-pong register sets), the slice after
// the last one is a "dead" slice whose loads all use out-of-range offsets (zeros -> its MFMAs add nothing).  With
// branches around the loads hipcc's s_waitcnt bookkeeping merges states at the joins and waits for the loads it
// has just issued, which serialises the pipeline (measured: 2x slower).
template <int BM, int NW, int XV, bool KF>
__global__ __launch_bounds__(64 * NW) void kg_conv_kernel(const KgConvArgs a, const Split sp) {
    constexpr int NT = 64 * NW;
    constexpr int TM = BM / 32;
    constexpr int DK = XV == 4 ? 16 : 32;        // slice depth
    constexpr int WREG = DK * BM / NT;           // weight elements each thread stages per slice
    constexpr int BREG = DK / 2;                 // B fragments per slice (one per k-step of 2)
    constexpr int BN = 32 * XV * NW;
    static_assert((DK * BM) % NT == 0 && NT % DK == 0 && NT % BM == 0, "tile/thread mismatch");
    using BT = typename std::conditional<XV == 4, kg_f4, float>::type;

    __shared__ float Ws[2][DK][BM + 1];      // +1: the k-fastest staging pattern writes a column of Ws per wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ncols = a.N * a.T_out * a.V_out;
    const int L = a.T_out * a.V_out;
    const int m0 = blockIdx.y * BM;
    const int kh = lane >> 5;                // which of the two k rows of an MFMA step this lane feeds
    const int col0 = blockIdx.x * BN + wave * (32 * XV) + XV * (lane & 31);   // this lane's first column

    const int s_total = slices_of(a.g[0], DK) + (a.ngroups > 1 ? slices_of(a.g[1], DK) : 0);
    const int s_beg = blockIdx.z * sp.per;
    const int s_end = min(s_total, s_beg + sp.per);
    const int ns = s_end - s_beg;

    kg_f32x16 acc[TM][XV];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < XV; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;

    // ---- this lane's column(s)
    const ColInfo xc = decode_col(col0, ncols, a.T_out, a.V_out);     // XV == 1: the column; XV == 4: the first one
    int tq[XV];
    unsigned colmask = 0;
#pragma unroll
    for (int q = 0; q < XV; ++q) {
        const int c = col0 + q;
        tq[q] = (c % L) / a.V_out;
        colmask |= (c < ncols ? 1u : 0u) << q;
    }

    // ---- per-group state for both groups (no kernel-argument reads or divisions after this point)
    GroupState<WREG> g0, g1;
    auto setup = [&](GroupState<WREG>& gs, const KgConvGroup& g) {
        gs.x = g.x; gs.w = g.w; gs.xsC = g.x_sC; gs.wsT = g.w_sT;
        gs.Cin = g.Cin; gs.taps = g.taps; gs.cchunks = (g.Cin + DK - 1) / DK;
        gs.chanblock = g.tap_mode == KG_TAP_CHANBLOCK ? g.Cin : 0;
        gs.extent = (long)(g.Cin * (g.tap_mode == KG_TAP_CHANBLOCK ? g.taps : 1) - 1) * g.x_sC + (long)ncols;
        gs.wsi4 = (unsigned)g.w_sI * 4u;
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int m = KF ? tid / DK + i * (NT / DK) : tid % BM;
            const int mm = m0 + m;
            const int mb = mm / g.w_MB;
            const unsigned off = (unsigned)(mb * g.w_sMB + (mm - mb * g.w_MB) * g.w_sO) * 4u;
            gs.woff[i] = mm < a.M ? off : W_OOB;
        }
        const int pad = (g.tap_mode == KG_TAP_TIME) ? (g.taps - 1) / 2 : 0;
        gs.tapmask = 0;
        if constexpr (XV == 1) {
            const int vi = g.vmap ? (xc.valid ? g.vmap[xc.vo] : -1) : xc.vo;
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
                gs.xoff[d] = ok ? (unsigned)(off * 4) : X_OOB;
            }
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                int shift = (g.tap_mode == KG_TAP_TIME) ? d - pad : 0;
                if (g.transposed) shift = -shift;
                unsigned mk = 0;
#pragma unroll
                for (int q = 0; q < XV; ++q) {
                    const int ti = tq[q] + shift;
                    mk |= ((((colmask >> q) & 1u) != 0 && ti >= 0 && ti < g.T_in && d < g.taps) ? 1u : 0u) << q;
                }
                gs.tapmask |= mk << (4 * d);
                const long off = (long)kh * g.x_sC + (long)col0 + (long)shift * g.V_in + PADF;
                gs.xoff[d] = (mk != 0) ? (unsigned)(off * 4) : X_OOB;
            }
        }
    };
    setup(g0, a.g[0]);
    setup(g1, a.g[a.ngroups > 1 ? 1 : 0]);

    // ---- slice iterator: (gi, d, cch) of the next slice to fetch, f = slices fetched so far
    int gi = 0, d = 0, cch = 0, f = 0;
    {
        int sl = s_beg;
        const int s0 = slices_of(a.g[0], DK);
        if (sl >= s0) { gi = 1; sl -= s0; }
        const int cc = gi ? g1.cchunks : g0.cchunks;
        d = sl / cc;
        cch = sl - d * cc;
    }

    float wreg[WREG];
    BT b0[BREG], b1[BREG];
    unsigned mk0 = 0, mk1 = 0;

    // global -> registers for the next slice (dead slices: every offset out of range), then advance the iterator
    auto fetch = [&](BT (&breg)[BREG], unsigned& mk) {
        const bool live = f < ns;
        const bool g1sel = gi != 0;
        const float* gx = g1sel ? g1.x : g0.x;
        const float* gw = g1sel ? g1.w : g0.w;
        const long xsC = g1sel ? g1.xsC : g0.xsC;
        const long wsT = g1sel ? g1.wsT : g0.wsT;
        const int Cin = g1sel ? g1.Cin : g0.Cin;
        const int taps = g1sel ? g1.taps : g0.taps;
        const int cchunks = g1sel ? g1.cchunks : g0.cchunks;
        const int chanblock = g1sel ? g1.chanblock : g0.chanblock;
        const unsigned wsi4 = g1sel ? g1.wsi4 : g0.wsi4;
        const int c0 = cch * DK;
        const long chan = (long)(d * chanblock + c0);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            kg_uniform_ptr(gw + (long)d * wsT), 0, (int)W_RANGE, 0x00020000);
        __amdgpu_buffer_rsrc_t xr;
        if constexpr (XV == 4) {
            long remain = ((g1sel ? g1.extent : g0.extent) - chan * xsC + PADF) * 4;   // bytes up to the tensor's end
            if (remain > 0x7fffffffL) remain = 0x7fffffffL;
            xr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gx + chan * xsC - PADF), 0,
                                                   __builtin_amdgcn_readfirstlane((int)remain), 0x00020000);
        } else {
            xr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gx + chan * xsC), 0, (int)X_RANGE, 0x00020000);
        }
        if constexpr (KF) {
            const int cc = c0 + tid % DK;
            const unsigned kterm = (live && cc < Cin) ? (unsigned)cc * wsi4 : W_OOB;
#pragma unroll
            for (int i = 0; i < WREG; ++i)
                wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wr, (g1sel ? g1.woff[i] : g0.woff[i]) + kterm, 0, 0));
        } else {
            const int k0 = c0 + tid / BM;
            const unsigned base = (g1sel ? g1.woff[0] : g0.woff[0]) + (unsigned)k0 * wsi4;
            const unsigned step = (unsigned)(NT / BM) * wsi4;
            const int nvalid = live ? (Cin - k0 + (NT / BM) - 1) / (NT / BM) : 0;    // elements i < nvalid are inside Cin
#pragma unroll
            for (int i = 0; i < WREG; ++i)
                wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wr, i < nvalid ? base + i * step : W_OOB, 0, 0));
        }
        const unsigned xo0 = g1sel ? g1.xoff[0] : g0.xoff[0];
        const unsigned xo1 = g1sel ? g1.xoff[1] : g0.xoff[1];
        const unsigned xo2 = g1sel ? g1.xoff[2] : g0.xoff[2];
        const unsigned base = d == 0 ? xo0 : (d == 1 ? xo1 : xo2);
        const unsigned step = (unsigned)(2 * xsC * 4);
        const int nvalid = live ? (Cin - c0 - kh + 1) / 2 : 0;      // fragments i < nvalid have their channel inside Cin
#pragma unroll
        for (int i = 0; i < BREG; ++i) {
            const unsigned off = i < nvalid ? base + i * step : X_OOB;
            if constexpr (XV == 4) breg[i] = __builtin_bit_cast(kg_f4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
            else breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
        }
        mk = ((g1sel ? g1.tapmask : g0.tapmask) >> (4 * d)) & 15u;
        // advance (scalar)
        ++f;
        if (++cch == cchunks) {
            cch = 0;
            if (++d == taps) {
                d = 0;
                if (gi + 1 < a.ngroups) ++gi;
            }
        }
    };
    // weight registers -> LDS buffer b
    auto stash = [&](int b) {
        if constexpr (KF) {
            float* p = &Ws[b][tid % DK][tid / DK];
#pragma unroll
            for (int i = 0; i < WREG; ++i) p[i * (NT / DK)] = wreg[i];
        } else {
            float* p = &Ws[b][tid / BM][tid % BM];
#pragma unroll
            for (int i = 0; i < WREG; ++i) p[i * (NT / BM) * (BM + 1)] = wreg[i];
        }
    };
    auto mfma_slice = [&](const BT (&cur)[BREG], unsigned mk, int b) {
#pragma unroll
        for (int kk = 0; kk < DK; kk += 2) {
            float av[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = Ws[b][kk + kh][i * 32 + (lane & 31)];
            if constexpr (XV == 4) {
                const kg_f4 bv = cur[kk / 2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float bq = ((mk >> q) & 1u) ? bv[q] : 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bq, acc[i][q], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], cur[kk / 2], acc[i][0], 0, 0, 0);
            }
        }
    };

    if (ns > 0) {
        // an odd slice count is made even by running the first slice through the second register set before the
        // pair loop; both entry paths reach the loop with the same pending-load picture (hipcc's waits stay exact)
        if (ns & 1) {
            fetch(b1, mk1);
            stash(1);
            __syncthreads();
            fetch(b0, mk0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_slice(b1, mk1, 1);
            __builtin_amdgcn_sched_barrier(0);
            stash(0);
            __syncthreads();
        } else {
            fetch(b0, mk0);
            stash(0);
            __syncthreads();
        }
        const int npairs = ns / 2;
        for (int p = 0; p < npairs; ++p) {
            // the sched_barriers pin the issue order loads -> MFMAs -> (wait + LDS writes)
            fetch(b1, mk1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_slice(b0, mk0, 0);
            __builtin_amdgcn_sched_barrier(0);
            stash(1);
            __syncthreads();
            fetch(b0, mk0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_slice(b1, mk1, 1);
            __builtin_amdgcn_sched_barrier(0);
            stash(0);
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool partial = sp.nsplit > 1;
    if constexpr (XV == 4) {
        // the lane holds, for every row, four consecutive columns -> 128-bit stores
        if (colmask != 0) {
            const bool full4 = colmask == 15u;
            float* obase = partial ? a.ws + (long)blockIdx.z * a.M * ncols : a.out;
            const long orow = partial ? (long)ncols : a.o_sC;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (m >= a.M) continue;
                    kg_f4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                    if (!partial) {
                        float bsum = 0.f;
                        if (a.bias0) bsum += a.bias0[m];
                        if (a.bias1) bsum += a.bias1[m];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float t = v[q] + bsum;
                            if (a.add && ((colmask >> q) & 1u)) t += a.add[(long)m * a.a_sC + col0 + q];
                            v[q] = kg_act(t, a.act, a.slope);
                        }
                    }
                    float* op = obase + (long)m * orow + col0;
                    if (full4) {
                        *reinterpret_cast<kg_f4*>(op) = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if ((colmask >> q) & 1u) op[q] = v[q];
                    }
                }
            }
        }
    } else {
        float* slab = partial ? a.ws + (long)blockIdx.z * a.M * ncols : nullptr;
        if (xc.valid) {
            const long ooff = (long)xc.n * a.o_sN + (long)xc.to * a.V_out + xc.vo;
            const long aoff = a.add ? (long)xc.n * a.a_sN + (long)(xc.to * a.a_tstride) * a.V_out + xc.vo : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (m >= a.M) continue;
                    float v = acc[i][0][r];
                    if (partial) {
                        slab[(long)m * ncols + col0] = v;
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

enum Tile { T128x128, T64x128, T32x128, T64x64, T32x64, X32x256, X64x256, NTILES };
const int kTileBM[NTILES] = {128, 64, 32, 64, 32, 32, 64};
const int kTileBN[NTILES] = {128, 128, 128, 64, 64, 256, 256};

// can the 128-bit kernel run this launch?  (see kg_conv_x4_kernel)
bool x4_eligible(const KgConvArgs* a) {
    const long L = (long)a->T_out * a->V_out;
    if (a->o_sN != L && a->N > 1) return false;
    if (a->add && ((a->a_sN != L && a->N > 1) || a->a_tstride != 1)) return false;
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        if (g.vmap || g.t_stride != 1 || g.T_in != a->T_out || g.V_in != a->V_out) return false;
        if (g.x_sN != L && a->N > 1) return false;
        if (g.x_lead < PADF) return false;
        const long extent = (long)(g.Cin * (g.tap_mode == KG_TAP_CHANBLOCK ? g.taps : 1)) * g.x_sC;
        if (extent >= (1L << 29)) return false;
    }
    return true;
}

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
    const char* x4env = getenv("KG_CONV_X4");          // "0" disables the 128-bit kernel (tests / tuning)
    const bool x4_ok = x4_eligible(a) && !(x4env && x4env[0] == '0');
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
        if (sscanf(env, "%d,%d", &t, &ns) >= 1 && t >= 0 && t < NTILES && (t < X32x256 || x4_ok)) {
            p.tile = (Tile)t;
            forced_split = ns;
        }
    }
    // The 128-bit tiles (X32x256 / X64x256) are only taken when forced through KG_CONV_PLAN: at the batch sizes of
    // BASELINE.json they put 4x fewer waves on the chip, and these launches are bound by bytes in flight
    // (memory latency), not by load-instruction issue - measured equal or slower (profiles/r01_v5_tune_conv.log).
    if (p.tile >= X32x256) s_total = slices_of(a->g[0], 16) + (a->ngroups > 1 ? slices_of(a->g[1], 16) : 0);
    const long wgs = count(p.tile);
    int nsplit = 1;
    if (forced_split > 0) {
        nsplit = forced_split > s_total ? s_total : forced_split;
    } else if (p.tile >= X32x256) {
        if (wgs * 2 < 1024 && s_total >= 4) {                 // fewer waves than SIMDs: split K
            nsplit = (int)((1536 + wgs * 2 - 1) / (wgs * 2));
            if (nsplit > s_total / 2) nsplit = s_total / 2;
            if (nsplit > 8) nsplit = 8;
            if (nsplit < 1) nsplit = 1;
        }
    } else if (wgs < 400 && s_total >= 8) {
        nsplit = (int)((512 + wgs - 1) / wgs);               // aim at ~2 workgroups per CU
        if (nsplit > s_total / 2) nsplit = s_total / 2;      // at least two slices per split
        if (nsplit > 8) nsplit = 8;
        if (nsplit < 1) nsplit = 1;
    }
    p.sp.per = kg_cdiv(s_total, nsplit);
    p.sp.nsplit = kg_cdiv(s_total, p.sp.per);
    return p;
}

template <int BM, int NW, int XV>
int launch(const KgConvArgs* a, const Plan& p, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    dim3 grid(kg_cdiv(ncols, 32 * XV * NW), kg_cdiv(a->M, BM), p.sp.nsplit);
    if (a->g[0].w_sI <= a->g[0].w_sO)
        hipLaunchKernelGGL((kg_conv_kernel<BM, NW, XV, true>), grid, dim3(64 * NW), 0, s, *a, p.sp);
    else
        hipLaunchKernelGGL((kg_conv_kernel<BM, NW, XV, false>), grid, dim3(64 * NW), 0, s, *a, p.sp);
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
    KG_REQUIRE(a->ngroups == 1 || (a->g[0].w_sI <= a->g[0].w_sO) == (a->g[1].w_sI <= a->g[1].w_sO),
               "kg_conv: both K-slice groups must store their weights in the same orientation");
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        KG_REQUIRE(g.Cin > 0 && g.T_in > 0 && g.V_in > 0, "kg_conv: group %d bad input dims", i);
        KG_REQUIRE(g.x_lead >= 0, "kg_conv: group %d x_lead=%d", i, g.x_lead);
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

extern "C" int kg_conv_plan_info(const KgConvArgs* a, int32_t* tile, int32_t* nsplit) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(tile && nsplit, "kg_conv_plan_info: null output");
    Plan p = make_plan(a);
    *tile = (int32_t)p.tile;
    *nsplit = p.sp.nsplit;
    return 0;
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
        case T128x128: return launch<128, 4, 1>(a, p, s);
        case T64x128:  return launch<64, 4, 1>(a, p, s);
        case T32x128:  return launch<32, 4, 1>(a, p, s);
        case T64x64:   return launch<64, 2, 1>(a, p, s);
        case T32x64:   return launch<32, 2, 1>(a, p, s);
        case X32x256:  return launch<32, 2, 4>(a, p, s);
        default:       return launch<64, 2, 4>(a, p, s);
    }
}
