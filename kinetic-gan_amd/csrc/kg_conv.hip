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

#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kg_common.h"

// Optional instrumentation build (-DKG_CONV_TIMING, tools/time_conv.py): lane 0 of every workgroup records the
// constant-rate (100 MHz) s_memrealtime at kernel entry, before and after the K-slice loop and after its stores,
// plus HW_ID, and writes the record to the END of a.ws (the last MiB).
#ifdef KG_CONV_TIMING
#define KG_STAMP(i) do { kg_t[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define KG_STAMP_FLUSH()                                                                           \
    do {                                                                                           \
        __builtin_amdgcn_s_waitcnt(0);                                                             \
        kg_t[3] = __builtin_amdgcn_s_memrealtime();                                                \
        if (threadIdx.x == 0 && a.ws) {                                                            \
            unsigned long long* o_ = (unsigned long long*)((char*)a.ws + a.ws_bytes - (1 << 20)) + \
                                     (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16; \
            for (int q_ = 0; q_ < 4; ++q_) o_[q_] = kg_t[q_];                                      \
            for (int q_ = 0; q_ < 4; ++q_) o_[8 + q_] = kg_seg[q_];                                \
            o_[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                   \
            o_[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));                  \
            o_[6] = (unsigned long long)(s_end - s_beg);                                           \
        }                                                                                          \
    } while (0)
#define KG_STAMP_DECL() unsigned long long kg_t[4] = {0, 0, 0, 0}, kg_seg[4] = {0, 0, 0, 0}, kg_last = 0
#define KG_SEG(i)                                                    \
    do {                                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
        if ((i) >= 0) kg_seg[(i) < 0 ? 0 : (i)] += t_ - kg_last;     \
        kg_last = t_;                                                \
    } while (0)
#else
#define KG_SEG(i) do {} while (0)
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

// the same for fewer than 2^22 columns (every launch of the training step) without integer divisions: quotient by float
// reciprocal (exact to +-1 below 2^22, then corrected), ~10 instructions per division instead of ~40
__device__ __forceinline__ void kg_divmod_small(int x, int d, int& q, int& r) {
    q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
    r = x - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}
__device__ __forceinline__ ColInfo decode_col_fast(int j, int ncols, int T_out, int V_out) {
    if (ncols >= (1 << 22)) return decode_col(j, ncols, T_out, V_out);       // (uniform)
    ColInfo c;
    c.valid = j < ncols;
    const int jj = c.valid ? j : 0;
    int rr;
    kg_divmod_small(jj, T_out * V_out, c.n, rr);
    kg_divmod_small(rr, V_out, c.to, c.vo);
    return c;
}

// num / s and num % s for the temporal strides that occur (1, 2) without an integer division
__device__ __forceinline__ void divmod_stride(int num, int s, int& q, int& r) {
    if (s == 1) {
        q = num; r = 0;
    } else if (s == 2) {
        q = num >> 1; r = num & 1;
    } else {
        q = num / s; r = num - q * s;
    }
}

__host__ __device__ inline int slices_of(const KgConvGroup& g, int bk = BK) { return g.taps * ((g.Cin + bk - 1) / bk); }

struct Split {
    int nsplit;          // workgroups along K
    int per;             // slices per split
    int xcd;             // 1: 1-D grid with the XCD-aware tile map (kg_tile_of_block), 0: grid (column tile, row tile)
    int inkernel;        // 1 (nsplit > 1 only): the last workgroup of a tile to arrive sums the slabs and runs the epilogue
                         // (ticket counters KgConvArgs.sync); 0: the separate kg_conv_splitk_epilogue launch does
};


constexpr unsigned W_RANGE = 0x40000000u;   // weight descriptor: 1 GiB; valid offsets are below it
constexpr unsigned X_RANGE = 0x80000000u;   // feature descriptor: 2 GiB (validated on the host)
constexpr unsigned W_OOB = 0x40000000u;     // adding one or two of these to a valid offset stays out of range
constexpr unsigned X_OOB = 0x80000000u;


// XCD-aware workgroup -> tile map.  Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2), in
// launch order.  With the plain (column tile, row tile) grid the row tiles of one column tile - which read the SAME
// feature columns - sit gridDim.x apart: they run in different dispatch rounds and each sweep over the row tiles
// streams the whole feature tensor from HBM again (C5a: 13 GB fetched for 2.5 GB of operands).  Here column tiles
// are dealt to XCDs round-robin and, inside an XCD, the row tiles of a column tile are consecutive: they run at
// the same time on the same XCD and share its L2.  Returns false for the padding workgroups (column tiles are
// padded to a multiple of 8).
// Only launches that run in several dispatch rounds are regrouped (round 3: from 1500 tiles - the critic's 192-sample
// launches of D1 / D2 read their feature columns once instead of once per row tile, 214 -> ~130 MB on the D1 tail, and
// run 6-10 % faster, profiles/r03_ab_xcd.log; every workgroup of the bs=64 launches is resident at once, there the plain
// order measured 1-2 % faster), and only with enough column tiles to keep every XCD busy.
constexpr long KG_XCD_MIN_TILES = 1500;
__host__ __device__ inline bool kg_xcd_grouped(int ctiles, int rtiles, long min_tiles = KG_XCD_MIN_TILES) {
    return ctiles >= 64 && rtiles >= 2 && (long)ctiles * rtiles >= min_tiles;
}

struct Blk { int x, y, z; };       // workgroup coordinates inside ONE problem's grid (blockIdx, or derived from it: kg_conv_many)

__device__ __forceinline__ bool kg_tile_of_block(const Blk& blk, bool grouped, int ctiles, int rtiles, int& ct, int& rt) {
    if (!grouped) {
        ct = blk.x;
        rt = blk.y;
        return true;
    }
    const int L = blk.x;
    const int xcd = L & 7, slot = L >> 3;
    const int cgrp = slot / rtiles;
    rt = slot - cgrp * rtiles;
    ct = cgrp * 8 + xcd;
    return ct < ctiles;
}

// epilogue of a 32-column wave tile.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// bias_lds[BM]: bias0 + bias1 of the workgroup's rows (staged before the slice loop: the epilogue has no dependent
// global loads besides the residual, whose 16*TM loads are issued together before the first use).
// output frame stride (0 and 1 both mean "contiguous frames")
__host__ __device__ __forceinline__ int kg_ots(const KgConvArgs& a) { return a.o_tstride > 1 ? a.o_tstride : 1; }

// PART: only the accumulator registers whose bit is set in `regmask` hold finished values (wave-level K-split: after the
// partial tiles have met in LDS every wave finishes and stores 16 / KW registers = that many row pairs of the tile)
// PLAIN: the launch has neither an `add` operand nor a `mask` and its rows fill whole tiles (M a multiple of the tile's rows;
// the host checks) - those operand paths and the per-store row guards are not compiled in (the guards alone: 2-4 % per
// launch, tools/exp_conv.py on a -DKG_EXP_FULLM build: 250 -> 241 us / 527 -> 516 us over the 13 shapes).  As
// run-time branches they cost every launch 3-6 % (round 5, tools/exp_conv.py on a -DKG_CONV_PLAIN_EPI build: the 13 shapes
// at 192 samples 551 -> 531 us; the same pattern that had cost kg_agg_reduce 12 %, profiles/r05_agg_bisect.log).
template <int TM, bool PART = false, bool PLAIN = false, bool DEV = false>
__device__ __forceinline__ void store_tile(const KgConvArgs& a, const Split& sp, const kg_f32x16 (&acc)[TM],
                                           const ColInfo& xc, int col0, int m0, int kh, int ncols,
                                           const float* bias_lds, int bz, unsigned regmask = 0xffffu) {
    if (!xc.valid) return;
#define KG_REG_ON(r_) (!PART || ((regmask >> (r_)) & 1u))
    const int mrem = PLAIN ? (1 << 20) : a.M - m0 - 4 * kh;     // row (r, i) exists iff i*32 + (r&3) + 8*(r>>2) < mrem
    if (sp.nsplit > 1) {
        float* slab = a.ws + (long)bz * a.M * ncols + (long)(m0 + 4 * kh) * ncols + col0;
        if constexpr (DEV) {     // read by a workgroup of THIS launch, possibly on another XCD: device-scope stores
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                    if (KG_REG_ON(r) && row < mrem)
                        __hip_atomic_store(slab + (long)row * ncols, acc[i][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                if (KG_REG_ON(r) && row < mrem) slab[(long)row * ncols] = acc[i][r];
            }
        return;
    }
    float v[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[i][r] = acc[i][r] + bias_lds[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
    if (!PLAIN && a.add) {
        const float* ap = a.add + (long)(m0 + 4 * kh) * a.a_sC + (long)xc.n * a.a_sN +
                          (long)(xc.to * a.a_tstride) * a.V_out + xc.vo;
        float rv[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                // clamped to the tile's first row (always < M) - NOT to this lane's base row m0 + 4*kh, which lies
                // beyond M for the upper half-wave of a tile with <= 4 valid rows (M = 2, 3: the generator's image
                // channels) and read up to four channel rows past the end of `add`: no branch between the loads
                rv[i][r] = KG_REG_ON(r) ? ap[(long)(row < mrem ? row : -4 * kh) * a.a_sC] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[i][r] += rv[i][r];
    }
    float* op = a.out + (long)(m0 + 4 * kh) * a.o_sC + (long)xc.n * a.o_sN + (long)xc.to * kg_ots(a) * a.V_out + xc.vo;
    if (!PLAIN && a.mask) {       // LeakyReLU derivative on the consumer's activation output, all loads issued together
        const float* mp = a.mask + (long)(m0 + 4 * kh) * a.m_sC + (long)xc.n * a.m_sN + (long)xc.to * a.V_out + xc.vo;
        float mv[TM][16];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                mv[i][r] = KG_REG_ON(r) ? mp[(long)(row < mrem ? row : -4 * kh) * a.m_sC] : 0.f;
            }
        const float sl = a.slope;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                if (KG_REG_ON(r) && row < mrem) op[(long)row * a.o_sC] = kg_act(v[i][r], a.act, sl) * (mv[i][r] > 0.f ? 1.f : sl);
            }
        return;
    }
    auto emit = [&](auto fn) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                if (KG_REG_ON(r) && row < mrem) op[(long)row * a.o_sC] = fn(v[i][r]);
            }
    };
    const float slope = a.slope;
    if (a.act == KG_ACT_LRELU)     emit([slope](float t) { return t > 0.f ? t : t * slope; });
    else if (a.act == KG_ACT_TANH) emit([](float t) { return tanhf(t); });
    else                           emit([](float t) { return t; });
}

// In-kernel completion of a K-split tile (Split::inkernel).  Every workgroup has written its partial tile to its slab with
// device-scope stores; it waits for them, takes a ticket of the tile's counter, and the LAST of the tile's nsplit workgroups
// to arrive re-reads all slabs (its own included) in split order - the order kg_conv_splitk_epilogue sums in - and runs the
// ordinary epilogue on the sums: deterministic, no second launch.  The counter is left at zero.  Loads: 16 accumulator
// registers x up to 4 slabs in flight per lane.
template <int TM, bool PART, bool PLAIN>
__device__ __forceinline__ void finish_split(const KgConvArgs& a, const Split& sp, kg_f32x16 (&acc)[TM], const ColInfo& xc,
                                             int col0, int m0, int kh, int ncols, const float* bias_lds, unsigned regmask,
                                             int tile_id, int* flag_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tk = __hip_atomic_fetch_add(a.sync + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = tk == sp.nsplit - 1;
        if (last) __hip_atomic_store(a.sync + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag_lds = last;
    }
    __syncthreads();
    if (!*flag_lds) return;
    if (xc.valid) {
        const int mrem = PLAIN ? (1 << 20) : a.M - m0 - 4 * kh;
        const long per = (long)a.M * ncols;
        const float* slab = a.ws + (long)(m0 + 4 * kh) * ncols + col0;
#ifndef KG_INK_KB
#define KG_INK_KB 4
#endif
        constexpr int KB = KG_INK_KB;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float s16[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) s16[r] = 0.f;
            for (int k0 = 0; k0 < sp.nsplit; k0 += KB) {
                float v[KB][16];
#pragma unroll
                for (int k = 0; k < KB; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // (no lane-dependent branch between the loads: a row beyond M reads the tile's first row, a slab
                        // beyond the last reads the last, and the value is dropped below)
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2);
                        const int kk = k0 + k < sp.nsplit ? k0 + k : sp.nsplit - 1;
                        v[k][r] = 0.f;
                        if (!PART || ((regmask >> r) & 1u))          // (uniform per wave)
                            v[k][r] = __hip_atomic_load(slab + (long)kk * per + (long)(row < mrem ? row : -4 * kh) * ncols,
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int k = 0; k < KB; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s16[r] += k0 + k < sp.nsplit ? v[k][r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = s16[r];
        }
    }
    const Split one{1, 0, sp.xcd, 0};
    store_tile<TM, PART, PLAIN>(a, one, acc, xc, col0, m0, kh, ncols, bias_lds, 0, regmask);
}
#undef KG_REG_ON

// per K-slice-group state: everything that costs a kernel-argument read or an integer division is computed once,
// before the slice loop, for both groups; the loop selects between the two copies with v_cndmask / s_cselect
template <int WREG>
struct GroupState {
    unsigned woff[WREG];            // per thread: byte offset of the m-part of its i-th weight element (or W_OOB)
    unsigned wlane;                 // FAST: the lane's loop-invariant channel part ((tw % DK) or (tw / BM)) * w_sI * 4
    unsigned xoff[3];               // per thread: byte offset of its column(s)' source for tap 0..2 (or X_OOB)
    const float* x;                 // wave-uniform geometry
    const float* w;
    long xsC, wsT;
    int Cin, taps, cchunks, chanblock;
    unsigned wsi4;
};

// A wave owns 32 columns, K-slices are 32 channels deep; any stride / vertex gather / layout (32-bit lane loads).
// KF: weight staging pattern - k fastest (forward layouts) or m fastest (transposed).
//
// The slice loop is STRAIGHT-LINE code: slices are processed in pairs (ping-pong register sets), the slice after
// the last one is a "dead" slice whose loads all use out-of-range offsets (zeros -> its MFMAs add nothing).  With
// branches around the loads hipcc's s_waitcnt bookkeeping merges states at the joins and waits for the loads it
// has just issued, which serialises the pipeline (measured: 2x slower).
// KW = 1: the NW waves of a workgroup own NW column groups and share every weight slice through LDS.
// KW = NW ("skinny" launches: a few hundred columns, deep K - D4 / D5 and the generator's first blocks): ALL waves own
// the SAME 32 columns and every KW-th K-slice each, with a private weight tile in LDS; there is no barrier in the
// slice loop, the accumulators meet in LDS at the end.  This replaces the K-split across workgroups (partial slabs in
// HBM + a second launch to add them) wherever a tile's K range fits one workgroup.
// FAST: every K-slice of the launch is FULL (each group's Cin is a multiple of the slice depth; the host checks): the loads
// then need no per-fragment validity selects - a lane's operand offset is one loop-invariant VGPR (per tap) and the walk
// through the slice's channels / weight columns goes through the buffer instructions' SCALAR offset operand, a dead
// slice is a descriptor with zero records.  Per slice that removes ~60 v_cmp / v_cndmask / v_add (+ the s_nops the
// VCC hazards need) in front of the 20 loads - the loop body of the 32-row tile drops from ~110 to ~50 non-MFMA
// instructions per 16 MFMAs.  FAST = 1: the launch has ONE K-slice group (no per-slice selects between two groups' state
// either); FAST = 2: two groups.
template <int BM, int NW, bool KF, int KW = 1, int FAST = 0, bool PLAIN = false, bool INK = false>
#ifndef KG_CONV_MINW128
#define KG_CONV_MINW128 1
#endif
// minimum waves per SIMD the 32-bit-load kernels are compiled for (experiments: -DKG_CONV_MINW64=4 caps the 64-row
// tile at 123 VGPRs instead of 242 without spilling; in the tuning loop the 3n launches gained 2-4 %, the whole
// iteration did not: 4.75 vs 4.72 ms - left at the compiler's choice)
#ifndef KG_CONV_MINW64
#define KG_CONV_MINW64 1
#endif
#ifndef KG_CONV_MINW32
#define KG_CONV_MINW32 1
#endif
__device__ __forceinline__ void conv_tile(const KgConvArgs& a, const Split& sp, const Blk blk) {
    constexpr int NT = 64 * NW;
    constexpr int TM = BM / 32;
    constexpr int DK = 32;                       // slice depth
    constexpr int NWC = NW / KW;                 // waves side by side along the columns
    constexpr int NTW = 64 * NWC;                // threads that stage one weight tile together
    constexpr int WREG = DK * BM / NTW;          // weight elements each thread stages per slice
    constexpr int BREG = DK / 2;                 // B fragments per slice (one per k-step of 2)
    constexpr int BN = 32 * NWC;
    static_assert((DK * BM) % NTW == 0 && NTW % DK == 0 && NTW % BM == 0, "tile/thread mismatch");
    static_assert(KW == 1 || KW == NW, "wave K-split: all waves on one column group");
    using BT = float;

    __shared__ float WsAll[KW][2][DK][BM + 1];      // +1: the k-fastest staging pattern writes a column of Ws per wave
    __shared__ float Bl[BM];                 // bias0 + bias1 of the workgroup's rows (epilogue)

    KG_STAMP_DECL();
    KG_STAMP(0);
    kg_kernarg_warm<(int)(sizeof(KgConvArgs) + sizeof(Split))>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kwave = KW > 1 ? __builtin_amdgcn_readfirstlane(wave) : 0;      // which K-slices this wave takes
    const int cwave = KW > 1 ? 0 : wave;                                     // which column group
    const int tw = KW > 1 ? lane : tid;                                      // index among the threads sharing a weight tile
    float (*const Ws)[DK][BM + 1] = WsAll[kwave];
    const int ncols = a.N * a.T_out * a.V_out;
    const int L = a.T_out * a.V_out;
    int ctile, rtile;
    if (!kg_tile_of_block(blk, sp.xcd != 0, (ncols + BN - 1) / BN, (a.M + BM - 1) / BM, ctile, rtile)) return;    // (uniform) padding workgroup
    const int m0 = rtile * BM;
    const int kh = lane >> 5;                // which of the two k rows of an MFMA step this lane feeds
    const int col0 = ctile * BN + cwave * 32 + (lane & 31);   // this lane's column
    // ---- loads that need nothing but the kernel arguments are issued FIRST, as one batch, and are not waited for until
    // their values are used: the two bias vectors (out-of-range and absent ones read 0 through the buffer's range check -
    // no branches, hipcc put an s_waitcnt vmcnt(0) behind each guarded load) and the two groups' vertex maps as lane
    // tables.  Before: four global loads one after the other, each a full memory round trip - 2.3 us of setup on an idle
    // chip, 4-12 us when the five resident workgroups of every CU start together (profiles/r03_conv_phases.log).
    float bias_r0, bias_r1;
    int vt0, vt1;
    {
        const unsigned boff = tid < BM ? (unsigned)(m0 + tid) * 4u : 0xffffffffu;
        const __amdgpu_buffer_rsrc_t rb0 = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.bias0), 0, a.bias0 ? a.M * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.bias1), 0, a.bias1 ? a.M * 4 : 0, 0x00020000);
        bias_r0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb0, boff, 0, 0));
        bias_r1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb1, boff, 0, 0));
        const int32_t* vm0 = a.g[0].vmap;
        const int32_t* vm1 = a.ngroups > 1 ? a.g[1].vmap : nullptr;
        const __amdgpu_buffer_rsrc_t rv0 = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(vm0), 0, vm0 ? a.V_out * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rv1 = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(vm1), 0, vm1 ? a.V_out * 4 : 0, 0x00020000);
        vt0 = __builtin_amdgcn_raw_buffer_load_b32(rv0, (unsigned)lane * 4u, 0, 0);
        vt1 = __builtin_amdgcn_raw_buffer_load_b32(rv1, (unsigned)lane * 4u, 0, 0);
    }

    const int s_total = slices_of(a.g[0], DK) + (a.ngroups > 1 ? slices_of(a.g[1], DK) : 0);
    const int s_beg = blk.z * sp.per;
    const int s_end = min(s_total, s_beg + sp.per);
    // slices of this wave: s_beg + kwave, + KW, ...
    const int ns = KW > 1 ? (s_end - s_beg - kwave + KW - 1) / KW : s_end - s_beg;

    kg_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- this lane's column(s)
    const ColInfo xc = decode_col_fast(col0, ncols, a.T_out, a.V_out);

    // ---- per-group state for both groups (no kernel-argument reads or divisions after this point)
    GroupState<WREG> g0, g1;
    auto setup = [&](GroupState<WREG>& gs, const KgConvGroup& g, const int vt) {
        gs.x = g.x; gs.w = g.w; gs.xsC = g.x_sC; gs.wsT = g.w_sT;
        gs.Cin = g.Cin; gs.taps = g.taps; gs.cchunks = (g.Cin + DK - 1) / DK;
        gs.chanblock = g.tap_mode == KG_TAP_CHANBLOCK ? g.Cin : 0;
        gs.wsi4 = (unsigned)g.w_sI * 4u;
        gs.wlane = (unsigned)(KF ? tw % DK : tw / BM) * gs.wsi4;
        // (32-bit arithmetic throughout: the host has checked that every in-range offset is below 2^29 elements)
        const bool rowblocks = g.w_MB < a.M;                    // (uniform) most launches have one row block
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int m = KF ? tw / DK + i * (NTW / DK) : tw % BM;
            const int mm = m0 + m;
            unsigned off = (unsigned)mm * (unsigned)g.w_sO;
            if (rowblocks) {
                int mb, mr;
                kg_divmod_small(mm, g.w_MB, mb, mr);            // (mm <= 65535)
                off = (unsigned)mb * (unsigned)g.w_sMB + (unsigned)mr * (unsigned)g.w_sO;
            }
            gs.woff[i] = mm < a.M ? off * 4u : W_OOB;
        }
        // vertex gather: the table was fetched at kernel entry as a lane table (lane i holds vmap[i]), the lookup is a
        // cross-lane read - no global load that depends on the column decode (V_out > 64: direct load)
        int vi = xc.vo;
        if (g.vmap) {
            vi = a.V_out <= 64 ? __builtin_amdgcn_ds_bpermute(xc.vo << 2, vt) : (xc.valid ? g.vmap[xc.vo] : -1);
        }
        const bool okv = xc.valid && vi >= 0;
        const unsigned base = (unsigned)kh * (unsigned)g.x_sC + (unsigned)xc.n * (unsigned)g.x_sN + (unsigned)vi;
        const int tstep = g.tap_mode == KG_TAP_TIME ? 1 : 0;    // tap d is shifted by d - pad frames (TIME) / not at all
        const int pad = tstep ? (g.taps - 1) / 2 : 0;
        if (!g.transposed) {
            const int t0 = xc.to * g.t_stride - pad;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int ti = t0 + tstep * d;
                const bool ok = okv && d < g.taps && (unsigned)ti < (unsigned)g.T_in;
                gs.xoff[d] = ok ? (base + (unsigned)(ti * g.V_in)) * 4u : X_OOB;
            }
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int num = xc.to + pad - tstep * d;
                int ti, rem;
                kg_divmod_small(num < 0 ? 0 : num, g.t_stride, ti, rem);
                const bool ok = okv && d < g.taps && num >= 0 && rem == 0 && ti < g.T_in;
                gs.xoff[d] = ok ? (base + (unsigned)(ti * g.V_in)) * 4u : X_OOB;
            }
        }
    };
    setup(g0, a.g[0], vt0);
    if (FAST != 1 && a.ngroups > 1) setup(g1, a.g[1], vt1);  // (uniform) most launches have one group: half the setup code is skipped
    else g1 = g0;                                // (FAST == 1: the instantiation of one-group launches - not even compiled in)

    // ---- slice iterator: (gi, cch, d) of the next slice to fetch, f = slices fetched so far.  The TAPS of a channel chunk
    // follow each other (round 3; before: all chunks of tap 0, then of tap 1, ...): the three temporal taps read the same
    // feature rows shifted by one frame, and with the taps outermost the reuse distance was the whole K range of every
    // resident workgroup - on the C5a block (512 channels, 3200 column tiles) no tap ever found its rows in L2 again:
    // 11.8 GB fetched for 2.5 GB of operands (profiles/roofline_c5a_pmc.json history)
    int gi = 0, d = 0, cch = 0, f = 0;
    {
        int sl = s_beg + kwave;
        const int s0 = slices_of(a.g[0], DK);
        if (sl >= s0) { gi = 1; sl -= s0; }
        const int tp = gi ? g1.taps : g0.taps;
        cch = sl / tp;
        d = sl - cch * tp;
    }

    float wreg[WREG];
    BT b0[BREG], b1[BREG];

    // ---- global -> registers for the next slice (dead slices: every offset out of range).
    // prep() resolves the slice's descriptors / offsets and advances the iterator (its branches come BEFORE the
    // loads, so that the loads share one scheduling region with the MFMAs they are interleaved with);
    // load_w(i) / load_x(i) issue one load each.
    struct Fetch {
        __amdgpu_buffer_rsrc_t wr, xr;
        unsigned wterm, wstep, xbase, xstep;
        unsigned ws0;                // FAST: scalar byte offset of the slice's first channel in the weight rows
        int wnvalid, xnvalid;
        bool g1sel;
    };
    auto prep = [&](Fetch& c) {
        const bool live = f < ns;
        const bool g1sel = FAST == 1 ? false : gi != 0;
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
        const int dcur = d;
        const long chan = (long)(d * chanblock + c0);
        ++f;
#pragma unroll
        for (int adv = 0; adv < KW; ++adv) {         // this wave's next slice is KW slices on
            const int cc_ = gi ? g1.cchunks : g0.cchunks, tp_ = gi ? g1.taps : g0.taps;
            if (++d == tp_) {
                d = 0;
                if (++cch == cc_) {
                    cch = 0;
                    if (gi + 1 < a.ngroups) ++gi;
                }
            }
        }
        c.g1sel = g1sel;
        if constexpr (FAST != 0) {
            // full slices only: scalar walk, zero-record descriptors for the dead slice
            c.wr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gw + (long)dcur * wsT), 0, live ? (int)W_RANGE : 0, 0x00020000);
            c.xr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gx + chan * xsC), 0, live ? (int)X_RANGE : 0, 0x00020000);
            c.ws0 = (unsigned)c0 * wsi4;
            c.wstep = (unsigned)(NTW / BM) * wsi4;
            const bool s1 = FAST == 2 && g1sel;
            c.g1sel = s1;
            c.wterm = (s1 ? g1.woff[0] : g0.woff[0]) + (s1 ? g1.wlane : g0.wlane);     // (m-fastest staging)
            const unsigned xo0 = s1 ? g1.xoff[0] : g0.xoff[0];
            const unsigned xo1 = s1 ? g1.xoff[1] : g0.xoff[1];
            const unsigned xo2 = s1 ? g1.xoff[2] : g0.xoff[2];
            c.xbase = dcur == 0 ? xo0 : (dcur == 1 ? xo1 : xo2);
            c.xstep = (unsigned)(2 * xsC * 4);
            c.wnvalid = WREG;
            c.xnvalid = BREG;
            return;
        }
        c.wr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gw + (long)dcur * wsT), 0, (int)W_RANGE, 0x00020000);
        c.xr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gx + chan * xsC), 0, (int)X_RANGE, 0x00020000);
        if constexpr (KF) {
            const int cc = c0 + tw % DK;
            c.wterm = (live && cc < Cin) ? (unsigned)cc * wsi4 : W_OOB;
            c.wstep = 0;
            c.wnvalid = WREG;
        } else {
            const int k0 = c0 + tw / BM;
            c.wterm = (g1sel ? g1.woff[0] : g0.woff[0]) + (unsigned)k0 * wsi4;
            c.wstep = (unsigned)(NTW / BM) * wsi4;
            c.wnvalid = live ? (Cin - k0 + (NTW / BM) - 1) / (NTW / BM) : 0;     // elements i < wnvalid are inside Cin
        }
        const unsigned xo0 = g1sel ? g1.xoff[0] : g0.xoff[0];
        const unsigned xo1 = g1sel ? g1.xoff[1] : g0.xoff[1];
        const unsigned xo2 = g1sel ? g1.xoff[2] : g0.xoff[2];
        c.xbase = dcur == 0 ? xo0 : (dcur == 1 ? xo1 : xo2);
        c.xstep = (unsigned)(2 * xsC * 4);
        c.xnvalid = live ? (Cin - c0 - kh + 1) / 2 : 0;       // fragments i < xnvalid have their channel inside Cin
    };
    auto load_w = [&](const Fetch& c, int i) {
        if constexpr (FAST != 0) {
            if constexpr (KF)
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    c.wr, (c.g1sel ? g1.woff[i] + g1.wlane : g0.woff[i] + g0.wlane), c.ws0, 0));
            else
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.wr, c.wterm, c.ws0 + i * c.wstep, 0));
        }
        unsigned off;
        if constexpr (KF) off = (c.g1sel ? g1.woff[i] : g0.woff[i]) + c.wterm;
        else off = i < c.wnvalid ? c.wterm + i * c.wstep : W_OOB;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.wr, off, 0, 0));
    };
    auto load_x = [&](const Fetch& c, int i) {
        if constexpr (FAST != 0)
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.xr, c.xbase, i * c.xstep, 0));
        const unsigned off = i < c.xnvalid ? c.xbase + i * c.xstep : X_OOB;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.xr, off, 0, 0));
    };
    auto fetch = [&](BT (&breg)[BREG]) {
        Fetch c;
        prep(c);
#pragma unroll
        for (int i = 0; i < WREG; ++i) wreg[i] = load_w(c, i);
#pragma unroll
        for (int i = 0; i < BREG; ++i) breg[i] = load_x(c, i);
    };
    // weight registers -> LDS buffer b
    auto stash = [&](int b) {
        if constexpr (KF) {
            float* p = &Ws[b][tw % DK][tw / DK];
#pragma unroll
            for (int i = 0; i < WREG; ++i) p[i * (NTW / DK)] = wreg[i];
        } else {
            float* p = &Ws[b][tw / BM][tw % BM];
#pragma unroll
            for (int i = 0; i < WREG; ++i) p[i * (NTW / BM) * (BM + 1)] = wreg[i];
        }
    };
    // one k-step (2 channels) of the slice held by `cur` / LDS buffer b
    auto mfma_step = [&](const BT (&cur)[BREG], const float (&av)[TM], int q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], cur[q], acc[i], 0, 0, 0);
    };
    auto read_a = [&](float (&av)[TM], int b, int q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = Ws[b][2 * q + kh][i * 32 + (lane & 31)];
    };
    // the slice loop's body: the MFMAs of the current slice with the loads of the next one spread between them
    // (V loads after every k-step; A operands are read from LDS two k-steps ahead).  The sched_barriers pin that
    // order: left alone, hipcc issues all loads first and a wave then sits in the load-issue queue (measured ~110
    // clk per load under contention) before its first MFMA.
    auto fetch_mfma = [&](BT (&nxt)[BREG], const BT (&cur)[BREG], int b) {
        constexpr int NL = WREG + BREG;
        constexpr int V = (NL + BREG - 1) / BREG;
        Fetch c;
        prep(c);
        float avs[BREG][TM];
        read_a(avs[0], b, 0);
        if constexpr (BREG > 1) read_a(avs[1], b, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BREG; ++q) {
            if (q + 2 < BREG) read_a(avs[q + 2], b, q + 2);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int idx = q * V + v;
                if (idx < WREG) wreg[idx] = load_w(c, idx);
                else if (idx < NL) nxt[idx - WREG] = load_x(c, idx - WREG);
            }
            mfma_step(cur, avs[q], q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    KG_STAMP(1);
    // the waves that share a weight tile meet at a barrier; a wave with its own tile (KW > 1) only has to keep its
    // LDS writes and reads in program order (the LDS queue of a wave is in order; the fence stops the compiler)
    auto tile_sync = [&]() {
        if constexpr (KW > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        else __syncthreads();
    };
    if constexpr (KW > 1) {
        if (tid < BM) Bl[tid] = bias_r0 + bias_r1;
        __syncthreads();
    }
    if (ns > 0) {
        // an odd slice count is made even by running the first slice through the second register set before the
        // pair loop; both entry paths reach the loop with the same pending-load picture (hipcc's waits stay exact)
        if (ns & 1) {
            fetch(b1);
            stash(1);
            if constexpr (KW == 1) { if (tid < BM) Bl[tid] = bias_r0 + bias_r1; }
            tile_sync();
            fetch_mfma(b0, b1, 1);
            stash(0);
            tile_sync();
        } else {
            fetch(b0);
            stash(0);
            if constexpr (KW == 1) { if (tid < BM) Bl[tid] = bias_r0 + bias_r1; }
            tile_sync();
        }
        const int npairs = ns / 2;
        KG_SEG(-1);
        for (int p = 0; p < npairs; ++p) {
            fetch_mfma(b1, b0, 0);
            KG_SEG(0);
            stash(1);
            KG_SEG(1);
            tile_sync();
            KG_SEG(2);
            fetch_mfma(b0, b1, 1);
            KG_SEG(0);
            stash(0);
            KG_SEG(1);
            tile_sync();
            KG_SEG(2);
        }
    }
    unsigned regmask = 0xffffu;
    if constexpr (KW > 1) {
        // the KW partial tiles meet in LDS (the weight tiles are dead now): every wave writes its tile, then sums 16 / KW
        // accumulator registers over all waves in wave order (deterministic) and finishes those rows - the epilogue's loads
        // and stores are spread over all waves instead of queueing behind wave 0
        __syncthreads();
        float* const red = &WsAll[0][0][0][0];
        static_assert(KW * TM * 16 * 64 <= KW * 2 * DK * (BM + 1), "reduction scratch");
        constexpr int RPW = 16 / KW;
        static_assert(16 % KW == 0, "registers per wave");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((kwave * TM + i) * 16 + r) * 64 + lane] = acc[i][r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r / RPW == kwave) {          // (uniform)
                    float v = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < KW; ++w2) v += red[((w2 * TM + i) * 16 + r) * 64 + lane];
                    acc[i][r] = v;
                }
            }
        regmask = ((1u << RPW) - 1u) << (kwave * RPW);
    }

    KG_STAMP(2);
    // ---- epilogue.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    store_tile<TM, (KW > 1), PLAIN, INK>(a, sp, acc, xc, col0, m0, kh, ncols, Bl, blk.z, regmask);
    if constexpr (INK) {     // the K-split launch completes its tiles itself (its own instantiation: finish_split's loads in
                             // flight would cost every other launch of the 32-row tile three waves per SIMD of occupancy)
        __shared__ int split_last;
        finish_split<TM, (KW > 1), PLAIN>(a, sp, acc, xc, col0, m0, kh, ncols, Bl, regmask,
                                          rtile * ((ncols + BN - 1) / BN) + ctile, &split_last);
    }
    KG_STAMP_FLUSH();
}

#define KG_CONV_MINW(BM_, NW_) (((BM_) == 64 && (NW_) == 2) ? 1 : (BM_) == 128 ? KG_CONV_MINW128 : ((BM_) == 64 ? KG_CONV_MINW64 : KG_CONV_MINW32))

template <int BM, int NW, bool KF, int KW = 1, int FAST = 0, bool PLAIN = false, bool INK = false>
__global__ __launch_bounds__(64 * NW, KG_CONV_MINW(BM, NW)) void kg_conv_kernel(const KgConvArgs a, const Split sp) {
    conv_tile<BM, NW, KF, KW, FAST, PLAIN, INK>(a, sp, Blk{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z});
}

// Several INDEPENDENT problems in one launch (kg_conv_many): the backward pass of a discriminator block issues up to
// three contractions that read the same gradient gm - the transposed temporal conv's two frame-parity launches and the
// small dense product of the residual branch.  As separate launches each pays its own prologue / epilogue / launch ramp
// (8-10 us) and the small ones leave most of the chip idle; here they share one grid: a workgroup finds
// its problem in a table of first-workgroup indices and runs the same tile code on it.  All problems use the two-group
// full-slice instantiation (a one-group problem runs it with its second group switched off at run time).
constexpr int CONV_MANY_MAX = KG_CONV_MANY_MAX;
struct ConvManyJob { KgConvArgs a; Split sp; int wg_begin; int ctiles; int nwg; };
struct ConvMany { int njobs; ConvManyJob job[CONV_MANY_MAX]; };

template <int BM, bool KF, bool PLAIN = false>
__global__ __launch_bounds__(256, KG_CONV_MINW(BM, 4)) void kg_conv_many_kernel(const ConvMany m) {
    kg_kernarg_warm<(int)sizeof(ConvMany)>();       // (the job search below reads one line per job, each behind the other)
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.job[ji + 1].wg_begin) ++ji;      // (uniform)
    const ConvManyJob& j = m.job[ji];
    const int local = (int)blockIdx.x - j.wg_begin;
    Blk b{local, 0, 0};                 // XCD-aware tile order (kg_tile_of_block drops the padding workgroups)
    if (!j.sp.xcd) {
        if (local >= j.nwg) return;     // (padding: every problem starts at a multiple of 8 workgroups)
        b.y = local / j.ctiles;
        b.x = local - b.y * j.ctiles;
    }
    conv_tile<BM, 4, KF, 1, 2, PLAIN>(j.a, j.sp, b);       // (ONE inlined copy of the tile code)
}


// =====================================================================================================================
// The bf16-split, LDS-staged form ("bs", round 5): the same contraction on v_mfma_f32_32x32x16_bf16.
// Every fp32 operand element is written as three bf16 terms x = h + m + l (round to nearest at each level; the sum is
// exact) and a product of two elements as hh + hm + mh + mm + hl + lh (what is dropped is below 2^-24 of the product):
// six bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of 64 cycles per 32 x 32 x 16 block, accumulation stays fp32.
// Splitting costs 5.5 VALU instructions per element; done in every wave's registers (the direct kernel's data path) that
// eats the gain, so here each operand is split ONCE:
//   weights    by a small pack launch in front of the tile kernel (kg_conv_bs_pack_kernel, into the caller's workspace):
//              P[step][term][octet][row], 16 bytes = eight consecutive channels of one row, step = (group, 32-channel
//              slice, tap).  The tile kernel copies a step's block to LDS with 16-byte loads (double buffered).
//   features   once per workgroup on the way into LDS, F[term][octet][position], 16 bytes = eight consecutive channels of
//              one position: a lane's B fragment of a 32x32x16 MFMA is one ds_read_b128 per term.  A temporal (3-tap)
//              group is staged as a WINDOW: the run of source positions the tile's columns read, in a coordinate with a
//              zero gap of `pad` >= V positions before and after every sample (q = n Lp + pad + t V + v), once per
//              32-channel slice; the three taps read it shifted by V positions - a third of the loads and splits.  With
//              pad and the sample length multiples of 4 a lane fetches four consecutive positions per 16-byte load.
//              Any other group (1 tap, or taps over channel blocks; stride, vertex gather) is staged per (slice, tap) in
//              COLUMN order with the gather folded into the (4-byte) staging loads.
// Wave w stages octet w (8 channels; lanes = positions: coalesced loads); the waves then form an RWV x CWV grid of
// (32 TM) x 32 accumulator tiles.  One barrier per step, two more where a new feature slice replaces the old one.
// =====================================================================================================================
typedef unsigned kg_u32x4 __attribute__((ext_vector_type(4)));
typedef float kg_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 kg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 kg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float kg_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bs_cvt2(float a, float b) {
    const kg_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, kg_bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
// eight fp32 values (consecutive channels) -> three 16-byte bf16 octets
__device__ __forceinline__ void bs_split8(const float (&x)[8], kg_u32x4& h, kg_u32x4& m, kg_u32x4& l) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float x0 = x[2 * p], x1 = x[2 * p + 1];
        const unsigned hh = bs_cvt2(x0, x1);
        x0 -= __uint_as_float(hh << 16); x1 -= __uint_as_float(hh & 0xffff0000u);      // exact
        const unsigned mm = bs_cvt2(x0, x1);
        x0 -= __uint_as_float(mm << 16); x1 -= __uint_as_float(mm & 0xffff0000u);      // exact, <= 8 significant bits left
        h[p] = hh; m[p] = mm; l[p] = bs_cvt2(x0, x1);
    }
}
__device__ __forceinline__ kg_f32x16 bs_mfma(const kg_u32x4& a, const kg_u32x4& b, const kg_f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kg_bf16x8, a), __builtin_bit_cast(kg_bf16x8, b), c, 0, 0, 0);
}

constexpr int BS_PMAX = 192;                      // positions of a staged feature slice (window or columns)
struct BsPlan {
    int win[2];                                   // group staged as a window (3 temporal taps, no gather): 0 no, 1 4-byte loads, 2 16-byte loads
    int Lp[2], pad[2];                            // window coordinate: positions per sample (T_in V_in + 2 pad), zero gap
    int nsteps, mpad;                             // packed weights: steps, rows (a multiple of the tile's rows)
    int xcd;
};

// packed weights of all steps: one thread per (step, row, octet)
__global__ __launch_bounds__(256) void kg_conv_bs_pack_kernel(const KgConvArgs a, const BsPlan bp, kg_u32x4* __restrict__ P) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    const int per = bp.mpad * 4;
    if (u >= bp.nsteps * per) return;
    const int step = u / per, r = u - step * per, oc = r / bp.mpad, m = r - oc * bp.mpad;
    int gi = 0, sl = step;
    const int s0 = a.g[0].taps * (a.g[0].Cin / 32);
    if (sl >= s0) { gi = 1; sl -= s0; }
    const KgConvGroup& g = a.g[gi];
    const int cch = sl / g.taps, d = sl - cch * g.taps;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = 0.f;
    if (m < a.M) {
        const int mb = g.w_MB < a.M ? m / g.w_MB : 0;
        const float* w = g.w + (long)d * g.w_sT + (long)mb * g.w_sMB + (long)(m - mb * g.w_MB) * g.w_sO + (long)(cch * 32 + oc * 8) * g.w_sI;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = w[(long)e * g.w_sI];
    }
    kg_u32x4 h, mm, l;
    bs_split8(x, h, mm, l);
    kg_u32x4* o = P + ((long)step * 12 + oc) * bp.mpad + m;
    o[0] = h; o[4L * bp.mpad] = mm; o[8L * bp.mpad] = l;
}

template <int TM, int RWV, int CWV>
__global__ __launch_bounds__(256) void kg_conv_bs_kernel(const KgConvArgs a, const BsPlan bp, const kg_u32x4* __restrict__ P) {
    static_assert(RWV * CWV == 4, "four waves");
    static_assert(BS_PMAX == 192, "three feature units per thread (G::foff0..2)");
    constexpr int BM = 32 * TM * RWV, BN = 32 * CWV;
    constexpr int WU = (12 * BM + 255) / 256;     // 16-byte pieces of a step's weight block per thread
    constexpr int FU = BS_PMAX / 64;              // feature units (position, octet) per thread with 4-byte loads
    __shared__ kg_u32x4 Fs[3][4][BS_PMAX];
    __shared__ kg_u32x4 Wq[2][12 * BM];           // [buffer][term][octet][row]
    __shared__ float Bl[BM];

    KG_STAMP_DECL();
    KG_STAMP(0);
    const int s_beg = 0, s_end = bp.nsteps;
    (void)s_beg; (void)s_end;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave / CWV, cw = wave % CWV;
    const int l32 = lane & 31, kh = lane >> 5;
    const int ncols = a.N * a.T_out * a.V_out;
    int ctile, rtile;
    if (!kg_tile_of_block(Blk{(int)blockIdx.x, (int)blockIdx.y, 0}, bp.xcd != 0, (ncols + BN - 1) / BN, (a.M + BM - 1) / BM, ctile, rtile)) return;
    const int m0 = rtile * BM;
    const int j0 = ctile * BN;
    const int col0 = j0 + cw * 32 + l32;          // this lane's column in the MFMA phase
    if (tid < BM) {
        const int mm = m0 + tid;
        Bl[tid] = mm < a.M ? (a.bias0 ? a.bias0[mm] : 0.f) + (a.bias1 ? a.bias1[mm] : 0.f) : 0.f;
    }
    const ColInfo xc = decode_col_fast(col0, ncols, a.T_out, a.V_out);

    // ---- per-group state (wave-uniform values end up in SGPRs)
    struct G {
        const float* x;
        long xsC;
        int taps, cchunks, chanblock, win, P;
        // staging: byte offset of this thread's position(s) for tap 0 (column order: tap d adds d * fstep; 16-byte window
        // loads: unit 0 only, four positions) and which taps read inside the tensor (bit d).  (Scalars, not arrays:
        // hipcc kept the arrays in scratch memory.)
        unsigned foff0, foff1, foff2, fval0, fval1, fval2;
        unsigned fstep;
        int pB, tapstep;            // MFMA phase: this lane's position in F for tap 0, positions per tap
    };
    G gs0, gs1;
#define KG_FO(s_, i_) ((i_) == 0 ? (s_).foff0 : (i_) == 1 ? (s_).foff1 : (s_).foff2)
#define KG_FV(s_, i_) ((i_) == 0 ? (s_).fval0 : (i_) == 1 ? (s_).fval1 : (s_).fval2)
#define KG_FSET(s_, i_, o_, v_) do { if ((i_) == 0) { (s_).foff0 = (o_); (s_).fval0 = (v_); } else if ((i_) == 1) { (s_).foff1 = (o_); (s_).fval1 = (v_); } else { (s_).foff2 = (o_); (s_).fval2 = (v_); } } while (0)
    const int j1 = min(j0 + BN, ncols) - 1;       // the tile's last valid column
    auto setup = [&](G& s, const KgConvGroup& g, const int gi) __attribute__((always_inline)) {
        s.x = g.x; s.xsC = g.x_sC;
        s.taps = g.taps; s.cchunks = g.Cin / 32;
        s.chanblock = g.tap_mode == KG_TAP_CHANBLOCK ? g.Cin : 0;
        s.win = bp.win[gi];
        const int tstep = g.tap_mode == KG_TAP_TIME ? 1 : 0;
        const int pad = tstep ? (g.taps - 1) / 2 : 0;
        if (s.win) {
            // window coordinate q = n Lp + gap + t V + v (gap >= V zero positions in front of and behind every sample); the
            // tile reads [q_lo, q_hi], q_lo rounded down to a multiple of 4
            const int Lp = bp.Lp[gi], gap = bp.pad[gi], V = g.V_in, st = g.t_stride;
            const ColInfo c0 = decode_col_fast(j0, ncols, a.T_out, a.V_out);      // (uniform)
            const ColInfo c1 = decode_col_fast(j1, ncols, a.T_out, a.V_out);
            const int q_lo = (c0.n * Lp + gap + (c0.to * st - pad) * V + c0.vo) & ~3;
            const int q_hi = c1.n * Lp + gap + (c1.to * st - pad + g.taps - 1) * V + c1.vo;
            s.P = q_hi - q_lo + 1;
            s.tapstep = V;
            s.pB = xc.valid ? xc.n * Lp + gap + (xc.to * st - pad) * V + xc.vo - q_lo : 0;
            s.fstep = 0;
            const int Lin = g.T_in * V;
            if (s.win == 2) {       // 16-byte loads: this lane's four positions 4 lane .. 4 lane + 3 (all inside a sample or all in a gap)
                const int q = q_lo + 4 * lane;
                int n, r;
                kg_divmod_small(q, Lp, n, r);
                const bool ok = 4 * lane < s.P && n < a.N && r >= gap && r < gap + Lin;
                KG_FSET(s, 0, ((unsigned)n * (unsigned)g.x_sN + (unsigned)(r - gap)) * 4u, ok ? 1u : 0u);
                KG_FSET(s, 1, 0u, 0u);
                KG_FSET(s, 2, 0u, 0u);
            } else {
#pragma unroll
                for (int i = 0; i < FU; ++i) {
                    const int q = q_lo + lane + 64 * i;
                    int n, r;
                    kg_divmod_small(q, Lp, n, r);
                    const bool ok = lane + 64 * i < s.P && n < a.N && r >= gap && r < gap + Lin;
                    KG_FSET(s, i, ((unsigned)n * (unsigned)g.x_sN + (unsigned)(r - gap)) * 4u, ok ? 1u : 0u);
                }
            }
        } else {
            s.P = BN;
            s.tapstep = 0;
            s.pB = cw * 32 + l32;
#pragma unroll
            for (int i = 0; i < FU; ++i) {
                if (64 * i >= BN) { KG_FSET(s, i, 0u, 0u); continue; }      // (compile time: no columns there)
                const int jj = lane + 64 * i;
                const ColInfo c = decode_col_fast(j0 + jj, ncols, a.T_out, a.V_out);
                int vi = c.vo;
                if (g.vmap) vi = c.valid ? g.vmap[c.vo] : -1;
                const bool okv = jj < BN && c.valid && vi >= 0;
                const unsigned base = (unsigned)c.n * (unsigned)g.x_sN + (unsigned)vi;
                const int t0 = c.to * g.t_stride - pad;
                unsigned ok = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if (okv && d < g.taps && (unsigned)(t0 + tstep * d) < (unsigned)g.T_in) ok |= 1u << d;
                KG_FSET(s, i, (base + (unsigned)(t0 * g.V_in)) * 4u, ok);           // (mod 2^32: t0 may be -1)
            }
            s.fstep = (unsigned)(tstep * g.V_in) * 4u;
        }
    };
    setup(gs0, a.g[0], 0);
    if (a.ngroups > 1) setup(gs1, a.g[1], 1);      // (uniform)
    else gs1 = G{};                                // (never selected)

    kg_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- weights: step k's block P[k][12][mpad] rows m0 .. m0 + BM - 1 -> Wq[k & 1]
    // (fetched TWO steps ahead into one of two register sets: with one step - the time of 24 MFMAs - between the request
    // and the LDS write every step waited most of an L2 round trip for its weights)
    kg_u32x4 wra[WU], wrb[WU];
    auto issue_w = [&](int k, kg_u32x4 (&wraw)[WU]) __attribute__((always_inline)) {
        const kg_u32x4* src = P + (long)k * 12 * bp.mpad + m0;
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int idx = tid + 256 * u;
            if (12 * BM % 256 == 0 || idx < 12 * BM) wraw[u] = src[(idx / BM) * bp.mpad + idx % BM];
        }
    };
    auto stash_w = [&](int b, const kg_u32x4 (&wraw)[WU]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int idx = tid + 256 * u;
            if (12 * BM % 256 == 0 || idx < 12 * BM) Wq[b][idx] = wraw[u];
        }
    };
    // ---- features: F units in order (a window slice, or one (slice, tap) in column order); (fg, fc, fd) = the unit in flight
    kg_f32x4 fv[8];                     // 16-byte loads: fv[channel][position]; 4-byte loads: flat index unit * 8 + channel
    int fg = 0, fc = 0, fd = 0, fP = 0, fmode = 0;
    bool fmore = true;
    auto issue_f = [&]() __attribute__((always_inline)) {     // global -> registers for unit (fg, fc, fd)
        const bool s1 = fg != 0;
        const float* gx = s1 ? gs1.x : gs0.x;
        const long xsC = s1 ? gs1.xsC : gs0.xsC;
        const int chanblock = s1 ? gs1.chanblock : gs0.chanblock;
        fmode = s1 ? gs1.win : gs0.win;
        fP = s1 ? gs1.P : gs0.P;
        const long chan = (long)fd * chanblock + fc * 32 + 8 * wave;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gx + chan * xsC), 0, (int)X_RANGE, 0x00020000);
        if (fmode == 2) {
            const unsigned fo = (s1 ? gs1.fval0 : gs0.fval0) ? (s1 ? gs1.foff0 : gs0.foff0) : X_OOB;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                fv[e] = __builtin_bit_cast(kg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, fo, (unsigned)(e * xsC * 4), 0));
            }
        } else {
            const unsigned fstep = s1 ? gs1.fstep : gs0.fstep;
#pragma unroll
            for (int i = 0; i < FU; ++i) {
                if (64 * i < fP) {      // (uniform)
                    const unsigned fok = s1 ? KG_FV(gs1, i) : KG_FV(gs0, i);
                    const unsigned fo = ((fok >> fd) & 1u) ? (s1 ? KG_FO(gs1, i) : KG_FO(gs0, i)) + (unsigned)fd * fstep : X_OOB;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        fv[(i * 8 + e) / 4][(i * 8 + e) % 4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, fo, (unsigned)(e * xsC * 4), 0));
                }
            }
        }
        // advance to the next unit
        const int tp = s1 ? gs1.taps : gs0.taps, cc = s1 ? gs1.cchunks : gs0.cchunks;
        if (fmode != 0 || ++fd == tp) {
            fd = 0;
            if (++fc == cc) { fc = 0; ++fg; }
        }
    };
    auto stash_f = [&]() __attribute__((always_inline)) {     // registers -> LDS, split on the way
        if (fmode == 2) {
            if (4 * lane < BS_PMAX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x8[e] = fv[e][q];
                    kg_u32x4 h, mm, l;
                    bs_split8(x8, h, mm, l);
                    Fs[0][wave][4 * lane + q] = h; Fs[1][wave][4 * lane + q] = mm; Fs[2][wave][4 * lane + q] = l;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < FU; ++i) {
                if (64 * i < fP) {      // (uniform)
                    float x8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x8[e] = fv[(i * 8 + e) / 4][(i * 8 + e) % 4];
                    kg_u32x4 h, mm, l;
                    bs_split8(x8, h, mm, l);
                    Fs[0][wave][lane + 64 * i] = h; Fs[1][wave][lane + 64 * i] = mm; Fs[2][wave][lane + 64 * i] = l;
                }
            }
        }
    };

    // ---- step iterator: (group, 32-channel chunk, tap), taps innermost
    int gi = 0, cch = 0, d = 0, k = 0;
    issue_w(0, wra);
    if (bp.nsteps > 1) issue_w(1, wrb);
    issue_f();
    stash_w(0, wra);
    stash_f();
    fmore = fg < a.ngroups;
    if (fmore) issue_f();               // the next unit's loads fly while this one is multiplied
    __syncthreads();
    KG_STAMP(1);
    KG_SEG(-1);
    // one step; `wnew` takes the weights of step k + 2, `wnext` holds those of step k + 1.  Returns true after the last step.
    auto step = [&](kg_u32x4 (&wnew)[WU], const kg_u32x4 (&wnext)[WU]) __attribute__((always_inline)) -> bool {
        const bool s1 = gi != 0;
        const int pB = (s1 ? gs1.pB : gs0.pB) + d * (s1 ? gs1.tapstep : gs0.tapstep);
        const int win = s1 ? gs1.win : gs0.win;
        // advance the step iterator
        {
            const int tp = s1 ? gs1.taps : gs0.taps, cc = s1 ? gs1.cchunks : gs0.cchunks;
            if (++d == tp) {
                d = 0;
                if (++cch == cc) { cch = 0; ++gi; }
            }
        }
        const bool more = gi < a.ngroups;
        const bool newf = more && (win == 0 || d == 0);        // the next step reads a new feature unit
        if (k + 2 < bp.nsteps) issue_w(k + 2, wnew);
        const kg_u32x4* const Wb = Wq[k & 1];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int oc = 2 * g2 + kh;
            const kg_u32x4 bh = Fs[0][oc][pB], bm = Fs[1][oc][pB], bl = Fs[2][oc][pB];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = rw * 32 * TM + i * 32 + l32;
                const kg_u32x4 ah = Wb[(0 * 4 + oc) * BM + row], am = Wb[(1 * 4 + oc) * BM + row], al = Wb[(2 * 4 + oc) * BM + row];
                kg_f32x16 t = acc[i];       // small terms first
                t = bs_mfma(al, bh, t);
                t = bs_mfma(ah, bl, t);
                t = bs_mfma(am, bm, t);
                t = bs_mfma(am, bh, t);
                t = bs_mfma(ah, bm, t);
                t = bs_mfma(ah, bh, t);
                acc[i] = t;
            }
        }
        KG_SEG(0);
        if (!more) return true;
        stash_w((k + 1) & 1, wnext);    // (the other buffer: last read in step k - 1, every wave is past that barrier)
        KG_SEG(1);
        if (newf) {
            __syncthreads();            // every wave is done with the old feature unit
            stash_f();
            fmore = fg < a.ngroups;
            if (fmore) issue_f();
        }
        KG_SEG(2);
        __syncthreads();
        KG_SEG(3);
        ++k;
        return false;
    };
    for (;;) {
        if (step(wra, wrb)) break;
        if (step(wrb, wra)) break;
    }
    KG_STAMP(2);
    const Split sp{1, 0, 0};
    store_tile<TM>(a, sp, acc, xc, col0, m0 + rw * 32 * TM, kh, ncols, Bl + rw * 32 * TM, 0);
    KG_STAMP_FLUSH();
#undef KG_FO
#undef KG_FV
#undef KG_FSET
}

// ---- "bsw": the instantiation for launches whose groups are ALL 16-byte windows (the D0 / D1 tails), with every global load
// of the step loop's weight path issued by hand and waited for by COUNT.  In kg_conv_bs_kernel the compiler's wait bookkeeping loses
// track at the loop's branches and puts s_waitcnt vmcnt(0) in front of every LDS write of staged data: each step then waits
// for the loads it has just issued for two steps ahead (workgroup stamps, profiles/r05_bs_stamps.log: 570 of a step's 2900
// cycles at the weight write, and a full memory round trip after every feature-slice change).  Here the weights go
// global -> LDS directly (buffer_load_dwordx4 ... lds, three stages, two steps ahead, no registers) and each wait for them
// names how many younger loads may stay in flight (the vector-memory counter retires in order); the feature quads stay
// compiler-managed register loads (waited for in full at the two slice changes of a launch).
typedef int kg_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ kg_i32x4 bsw_rsrc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    kg_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)u);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void bsw_dma_x4(unsigned lds_byte, unsigned voff, kg_i32x4 rsrc, unsigned soff) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void bsw_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void bsw_wait_upto(int n) {       // at most n loads may stay in flight (weights: LDS-DMA, no registers)
    switch (n) {
        case 0: bsw_wait<0>(); break;   case 1: bsw_wait<1>(); break;   case 2: bsw_wait<2>(); break;   case 3: bsw_wait<3>(); break;
        case 4: bsw_wait<4>(); break;   case 5: bsw_wait<5>(); break;   case 6: bsw_wait<6>(); break;   case 7: bsw_wait<7>(); break;
        case 8: bsw_wait<8>(); break;   case 9: bsw_wait<9>(); break;   case 10: bsw_wait<10>(); break; case 11: bsw_wait<11>(); break;
        case 12: bsw_wait<12>(); break; case 13: bsw_wait<13>(); break; case 14: bsw_wait<14>(); break; case 15: bsw_wait<15>(); break;
        case 16: bsw_wait<16>(); break; case 17: bsw_wait<17>(); break; case 18: bsw_wait<18>(); break; case 19: bsw_wait<19>(); break;
        case 20: case 21: case 22: case 23: bsw_wait<20>(); break;
        case 24: case 25: case 26: case 27: bsw_wait<24>(); break;
        default: bsw_wait<28>(); break;
    }
}
__device__ __forceinline__ void bsw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int bsw_lds_bytes(int bm) { return (3 * 4 * BS_PMAX + 3 * 12 * bm) * 16 + bm * 4; }

template <int TM, int RWV, int CWV, bool PLAIN>
__global__ __launch_bounds__(256) void kg_conv_bsw_kernel(const KgConvArgs a, const BsPlan bp, const kg_u32x4* __restrict__ P) {
    static_assert(RWV * CWV == 4, "four waves");
    constexpr int BM = 32 * TM * RWV, BN = 32 * CWV;
    constexpr int WU = (12 * BM + 255) / 256;     // 16-byte pieces of a step's weight block per thread (whole waves: 12 BM % 64 == 0)
    extern __shared__ kg_u32x4 kg_bsw_lds[];
    kg_u32x4* const Fs = kg_bsw_lds;                              // [3 terms][4 octets][BS_PMAX]
    kg_u32x4* const Wq = kg_bsw_lds + 3 * 4 * BS_PMAX;            // [3 stages][term][octet][row]
    float* const Bl = reinterpret_cast<float*>(Wq + 3 * 12 * BM);

    KG_STAMP_DECL();
    KG_STAMP(0);
    const int s_beg = 0, s_end = bp.nsteps;
    (void)s_beg; (void)s_end;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rw = wave / CWV, cw = wave % CWV;
    const int l32 = lane & 31, kh = lane >> 5;
    const int ncols = a.N * a.T_out * a.V_out;
    int ctile, rtile;
    if (!kg_tile_of_block(Blk{(int)blockIdx.x, (int)blockIdx.y, 0}, bp.xcd != 0, (ncols + BN - 1) / BN, (a.M + BM - 1) / BM, ctile, rtile)) return;
    const int m0 = rtile * BM;
    const int j0 = ctile * BN;
    const int col0 = j0 + cw * 32 + l32;          // this lane's column in the MFMA phase
    if (tid < BM) {
        const int mm = m0 + tid;
        Bl[tid] = mm < a.M ? (a.bias0 ? a.bias0[mm] : 0.f) + (a.bias1 ? a.bias1[mm] : 0.f) : 0.f;
    }
    const ColInfo xc = decode_col_fast(col0, ncols, a.T_out, a.V_out);
    const int j1 = min(j0 + BN, ncols) - 1;       // the tile's last valid column
    const ColInfo c0 = decode_col_fast(j0, ncols, a.T_out, a.V_out);      // (uniform)

    // ---- per-group window state
    struct G {
        const float* x;
        long xsC;
        int taps, cchunks;
        unsigned fo;                // staging: byte offset of this lane's four positions 4 lane .. 4 lane + 3, or X_OOB
        int pB, tapstep;            // MFMA phase: this lane's position in F for tap 0, positions per tap
    };
    G gs0, gs1;
    auto setup = [&](G& s, const KgConvGroup& g, const int gi) __attribute__((always_inline)) {
        s.x = g.x; s.xsC = g.x_sC;
        s.taps = g.taps; s.cchunks = g.Cin / 32;
        const int pad = (g.taps - 1) / 2;
        const int Lp = bp.Lp[gi], gap = bp.pad[gi], V = g.V_in, st = g.t_stride;
        const ColInfo c1 = decode_col_fast(j1, ncols, a.T_out, a.V_out);
        const int q_lo = (c0.n * Lp + gap + (c0.to * st - pad) * V + c0.vo) & ~3;
        const int q_hi = c1.n * Lp + gap + (c1.to * st - pad + g.taps - 1) * V + c1.vo;
        const int Pn = q_hi - q_lo + 1;
        s.tapstep = V;
        s.pB = xc.valid ? xc.n * Lp + gap + (xc.to * st - pad) * V + xc.vo - q_lo : 0;
        const int q = q_lo + 4 * lane;
        int n, r;
        kg_divmod_small(q, Lp, n, r);
        const bool ok = 4 * lane < Pn && n < a.N && r >= gap && r < gap + g.T_in * V;
        s.fo = ok ? ((unsigned)n * (unsigned)g.x_sN + (unsigned)(r - gap)) * 4u : X_OOB;
    };
    setup(gs0, a.g[0], 0);
    if (a.ngroups > 1) setup(gs1, a.g[1], 1);      // (uniform)
    else gs1 = G{};                                // (never selected)

    kg_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int vmi = 0;                        // hand-issued vector-memory instructions of this wave so far
    // ---- weights: step k's block P[k][12][mpad], rows m0 .. m0 + BM - 1 -> stage k % 3, straight into LDS
    const kg_i32x4 prs = bsw_rsrc(P, (unsigned)((long)bp.nsteps * 12 * bp.mpad * 16));
    unsigned wvoff[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int idx = tid + 256 * u;
        wvoff[u] = (unsigned)((idx / BM) * bp.mpad + idx % BM) * 16u;
    }
    const unsigned wq_byte = (unsigned)(3 * 4 * BS_PMAX * 16);         // Wq's offset inside the workgroup's LDS
    auto dma_w = [&](int k) __attribute__((always_inline)) {
        const unsigned so = (unsigned)(((long)k * 12 * bp.mpad + m0) * 16);
        const unsigned stage = (unsigned)(k % 3) * (unsigned)(12 * BM * 16);
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            if (256 * u + 64 * wave < 12 * BM) {        // (uniform per wave)
                bsw_dma_x4(wq_byte + stage + (unsigned)(256 * u + 64 * wave) * 16u, wvoff[u], prs, so);
                ++vmi;
            }
        }
    };
    // ---- features: one unit per (group, 32-channel slice); (fg, fc) = the next unit to request
    kg_u32x4 fv[8];                     // fv[channel] = four positions
    int fg = 0, fc = 0;
    auto ld_f = [&]() __attribute__((always_inline)) {
        const bool s1 = fg != 0;
        const float* gx = s1 ? gs1.x : gs0.x;
        const long xsC = s1 ? gs1.xsC : gs0.xsC;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(gx + ((long)fc * 32 + 8 * wave) * xsC), 0, (int)X_RANGE, 0x00020000);
        const unsigned fo = s1 ? gs1.fo : gs0.fo;
        // (compiler-managed loads, counted like the hand-issued ones: the asm statements around them are volatile with a
        // memory clobber, the eight loads stay between them.  The same loads as inline asm with hand-placed waits faulted
        // from the third slice on - round 5, tools/bsw_check.py - and hipcc waits for all of them at the first use anyway.)
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] = __builtin_bit_cast(kg_u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, fo, (unsigned)(e * xsC * 4), 0));
        vmi += 8;
        if (++fc == (s1 ? gs1.cchunks : gs0.cchunks)) { fc = 0; ++fg; }
    };
    auto stash_f = [&]() __attribute__((always_inline)) {     // registers -> LDS, split on the way
        if (4 * lane < BS_PMAX) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x8[e] = __uint_as_float(fv[e][q]);
                kg_u32x4 h, mm, l;
                bs_split8(x8, h, mm, l);
                Fs[(0 * 4 + wave) * BS_PMAX + 4 * lane + q] = h;
                Fs[(1 * 4 + wave) * BS_PMAX + 4 * lane + q] = mm;
                Fs[(2 * 4 + wave) * BS_PMAX + 4 * lane + q] = l;
            }
        }
    };

    // ---- prologue: nothing of the compiler's own loads is in flight past this point
#pragma unroll
    for (int e = 0; e < 8; ++e) fv[e] = kg_u32x4{0u, 0u, 0u, 0u};
    bsw_wait<0>();
    const int K = bp.nsteps;
    dma_w(0);
    const int o_w0 = vmi;
    ld_f();
    int o_w1 = vmi, o_w2 = vmi;
    if (K > 1) { dma_w(1); o_w1 = vmi; }
    stash_f();
    if (fg < a.ngroups) ld_f();         // the next slice's loads fly while this one is multiplied
    bsw_wait_upto(vmi - o_w0);
    bsw_barrier();
    KG_STAMP(1);
    KG_SEG(-1);
    int gi = 0, cch = 0, d = 0;
    for (int k = 0;; ++k) {
        const bool s1 = gi != 0;
        const int pB = (s1 ? gs1.pB : gs0.pB) + d * (s1 ? gs1.tapstep : gs0.tapstep);
        {
            const int tp = s1 ? gs1.taps : gs0.taps, cc = s1 ? gs1.cchunks : gs0.cchunks;
            if (++d == tp) {
                d = 0;
                if (++cch == cc) { cch = 0; ++gi; }
            }
        }
        const bool more = k + 1 < K;
        const bool newf = more && d == 0;                       // the next step reads a new feature slice
        if (k + 2 < K) { dma_w(k + 2); o_w2 = vmi; }            // (stage (k + 2) % 3 was last read in step k - 1)
        const kg_u32x4* const Wb = Wq + (k % 3) * 12 * BM;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int oc = 2 * g2 + kh;
            const kg_u32x4 bh = Fs[(0 * 4 + oc) * BS_PMAX + pB], bm = Fs[(1 * 4 + oc) * BS_PMAX + pB], bl = Fs[(2 * 4 + oc) * BS_PMAX + pB];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = rw * 32 * TM + i * 32 + l32;
                const kg_u32x4 ah = Wb[(0 * 4 + oc) * BM + row], am = Wb[(1 * 4 + oc) * BM + row], al = Wb[(2 * 4 + oc) * BM + row];
                kg_f32x16 t = acc[i];       // small terms first
                t = bs_mfma(al, bh, t);
                t = bs_mfma(ah, bl, t);
                t = bs_mfma(am, bm, t);
                t = bs_mfma(am, bh, t);
                t = bs_mfma(ah, bm, t);
                t = bs_mfma(ah, bh, t);
                acc[i] = t;
            }
        }
        KG_SEG(0);
        if (!more) break;
        if (newf) {
            bsw_barrier();              // every wave is done with the old feature slice
            stash_f();
            if (fg < a.ngroups) ld_f();
        }
        KG_SEG(2);
        bsw_wait_upto(vmi - o_w1);      // the weights of step k + 1 have landed (those of k + 2 may still fly)
        o_w1 = o_w2;
        KG_SEG(1);
        bsw_barrier();
        KG_SEG(3);
    }
    bsw_wait<0>();
    KG_STAMP(2);
    const Split sp{1, 0, 0};
    store_tile<TM, false, PLAIN>(a, sp, acc, xc, col0, m0 + rw * 32 * TM, kh, ncols, Bl + rw * 32 * TM, 0);
    KG_STAMP_FLUSH();
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
    v = kg_act(v, a.act, a.slope);
    if (a.mask) v *= a.mask[(long)m * a.m_sC + (long)oc.n * a.m_sN + (long)oc.to * a.V_out + oc.vo] > 0.f ? 1.f : a.slope;
    a.out[(long)m * a.o_sC + (long)oc.n * a.o_sN + (long)oc.to * kg_ots(a) * a.V_out + oc.vo] = v;
}

// =====================================================================================================================
// Tiny-channel launches: M <= 16 output rows and <= 48 contraction terms in total (the generator's image-channel
// convs: 3 -> 9, 9 -> 3, 3x3 taps -> 3, 32 -> 12 rows, and their transposes).  A 32-row MFMA tile would multiply 29 zero
// rows per 3 useful ones and the launch would still pay the weight-staging / barrier pipeline of the GEMM kernels
// (G6's temporal conv: 22 us for 11 MFLOP).  Here a thread owns ONE column and all M rows: the conv is a streaming
// VALU kernel (coalesced loads along (t, v), weights broadcast from LDS), HBM bound, same semantics as kg_conv_kernel
// (groups, taps, stride, transposed, vertex gather, bias, add, activation, mask, output frame stride).
// =====================================================================================================================
constexpr int TINY_MAXM = 16, TINY_MAXK = 48;

template <int MT>
__global__ __launch_bounds__(256, 4) void kg_conv_tiny_kernel(const KgConvArgs a) {
    __shared__ float Wl[TINY_MAXK][MT];
    const int tid = threadIdx.x;
    // weights -> LDS, [term][m] with term = (group, tap, channel) in loop order
    {
        int base = 0;
        for (int gi = 0; gi < a.ngroups; ++gi) {
            const KgConvGroup& g = a.g[gi];
            const int nterm = g.taps * g.Cin;
            for (int e = tid; e < nterm * MT; e += 256) {
                const int term = e / MT, m = e - term * MT;
                const int d = term / g.Cin, c = term - d * g.Cin;
                float v = 0.f;
                if (m < a.M) {
                    const int mb = g.w_MB < a.M ? m / g.w_MB : 0;
                    v = g.w[(long)d * g.w_sT + (long)mb * g.w_sMB + (long)(m - mb * g.w_MB) * g.w_sO + (long)c * g.w_sI];
                }
                Wl[base + term][m] = v;
            }
            base += nterm;
        }
    }
    __syncthreads();
    const int ncols = a.N * a.T_out * a.V_out;
    const int j = blockIdx.x * 256 + tid;
    const ColInfo xc = decode_col(j, ncols, a.T_out, a.V_out);
    if (!xc.valid) return;
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    int base = 0;
    for (int gi = 0; gi < a.ngroups; ++gi) {
        const KgConvGroup& g = a.g[gi];
        const int vi = g.vmap ? g.vmap[xc.vo] : xc.vo;
        const int pad = (g.tap_mode == KG_TAP_TIME) ? (g.taps - 1) / 2 : 0;
        for (int d = 0; d < g.taps; ++d) {
            const int shift = (g.tap_mode == KG_TAP_TIME) ? d - pad : 0;
            int ti;
            bool ok = vi >= 0;
            if (!g.transposed) {
                ti = xc.to * g.t_stride + shift;
            } else {
                const int num = xc.to - shift;
                int rem;
                divmod_stride(num, g.t_stride, ti, rem);
                ok = ok && num >= 0 && rem == 0;
            }
            ok = ok && ti >= 0 && ti < g.T_in;
            const float* xp = g.x + (long)xc.n * g.x_sN + (long)(ok ? ti : 0) * g.V_in + (ok ? vi : 0) +
                              (g.tap_mode == KG_TAP_CHANBLOCK ? (long)d * g.Cin * g.x_sC : 0L);
            const float (*wl)[MT] = Wl + base + d * g.Cin;
            for (int c = 0; c < g.Cin; ++c) {
                const float xv = ok ? xp[(long)c * g.x_sC] : 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = fmaf(wl[c][m], xv, acc[m]);
            }
        }
        base += g.taps * g.Cin;
    }
    const long opos = (long)xc.n * a.o_sN + (long)xc.to * kg_ots(a) * a.V_out + xc.vo;
    const long apos = a.add ? (long)xc.n * a.a_sN + (long)(xc.to * a.a_tstride) * a.V_out + xc.vo : 0L;
    const long mpos = a.mask ? (long)xc.n * a.m_sN + (long)xc.to * a.V_out + xc.vo : 0L;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m < a.M) {
            float v = acc[m];
            if (a.bias0) v += a.bias0[m];
            if (a.bias1) v += a.bias1[m];
            if (a.add) v += a.add[(long)m * a.a_sC + apos];
            v = kg_act(v, a.act, a.slope);
            if (a.mask) v *= a.mask[(long)m * a.m_sC + mpos] > 0.f ? 1.f : a.slope;
            a.out[(long)m * a.o_sC + opos] = v;
        }
    }
}

bool tiny_eligible(const KgConvArgs* a) {
    if (kg_env().conv_tiny == 0 || kg_env().conv_plan_tile >= 0) return false;
    if (a->M > TINY_MAXM) return false;
    int terms = 0;
    for (int i = 0; i < a->ngroups; ++i) terms += a->g[i].taps * a->g[i].Cin;
    // a thread does terms x (4 | 16) multiply-adds in sequence: beyond ~200 the MFMA tile wins again (measured: the
    // 32 -> 12-row head conv of generator block 5, 22 k columns: 16.7 us here against 8.2 us on the matrix cores)
    return terms <= TINY_MAXK && terms * (a->M <= 4 ? 4 : 16) <= 192;
}

int launch_tiny(const KgConvArgs* a, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    if (a->M <= 4) hipLaunchKernelGGL(kg_conv_tiny_kernel<4>, dim3(kg_cdiv(ncols, 256)), dim3(256), 0, s, *a);
    else           hipLaunchKernelGGL(kg_conv_tiny_kernel<16>, dim3(kg_cdiv(ncols, 256)), dim3(256), 0, s, *a);
    return kg_launch_status("kg_conv (tiny)");
}

// plan tile codes (kg_conv_plan_info; KG_CONV_PLAN): the numbers of round 1 / 2 are kept, the retired forms (5-8: 128-bit
// loads and LDS-staged tiles, 10: image form) are gone
// (round 4: K32x32 with 8 / 16 waves per tile was built and measured - never ahead of the 4-wave form, profiles/r04_audit_kw.log)
enum Tile { T128x128 = 0, T64x128 = 1, T32x128 = 2, T64x64 = 3, T32x64 = 4, K32x32 = 9 };
inline bool tile_known(int t) { return (t >= 0 && t <= 4) || t == 9; }
inline bool tile_kw(Tile t) { return t == K32x32; }
inline int tile_bm(Tile t) { return t == T128x128 ? 128 : (t == T64x128 || t == T64x64) ? 64 : 32; }
inline int tile_bn(Tile t) { return t <= T32x128 ? 128 : (tile_kw(t) ? 32 : 64); }

struct Plan {
    Tile tile;
    Split sp;
    int ring;            // >= 0: the persistent LDS-ring form (kg_conv_ring.hip) with this ring tile instead of `tile`
    int bs;              // >= 0: the bf16-split LDS-staged form (kg_conv_bs_kernel) with this tile variant
    BsPlan bsp;
};

// ---- the bf16-split form's tile variants and what it can run
struct BsTile { int tm, rwv, cwv; };
constexpr BsTile BS_TILES[3] = {{2, 1, 4}, {1, 1, 4}, {2, 2, 2}};       // 64 x 128, 32 x 128, 128 x 64 (rows x columns)
inline int bs_bm(int v) { return 32 * BS_TILES[v].tm * BS_TILES[v].rwv; }
inline int bs_bn(int v) { return 32 * BS_TILES[v].cwv; }

// fills bp and returns true when variant v can run the problem
bool bs_plan(const KgConvArgs* a, int v, BsPlan& bp) {
    if (v < 0 || v > 2 || kg_env().conv_fast == 0) return false;
    const long ncols = (long)a->N * a->T_out * a->V_out;
    if (ncols >= (1L << 22)) return false;
    const int BN = bs_bn(v), L = a->T_out * a->V_out;
    bp.nsteps = 0;
    for (int i = 0; i < 2; ++i) { bp.win[i] = 0; bp.Lp[i] = 1; bp.pad[i] = 0; }
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        if (g.Cin % 32 != 0 || g.transposed) return false;
        bp.nsteps += g.taps * (g.Cin / 32);
        const int tpad = g.taps == 3 ? 1 : 0;
        if (g.tap_mode == KG_TAP_TIME && g.vmap == nullptr && g.V_in == a->V_out &&
            (long)(a->T_out - 1) * g.t_stride - tpad + g.taps - 1 <= g.T_in - 1 + tpad) {
            // window staging (a temporal conv, or a plain 1 x 1 conv of un-gathered columns) if every tile's window fits the
            // staging buffer; 16-byte loads when every row, sample and gap starts on a 16-byte boundary
            const int Lin = g.T_in * g.V_in;
            const bool al = Lin % 4 == 0 && g.x_sN % 4 == 0 && g.x_sC % 4 == 0 && ((uintptr_t)g.x & 15) == 0;
            const int gap = tpad == 0 ? 0 : al ? (g.V_in + 3) / 4 * 4 : g.V_in;
            const int Lp = Lin + 2 * gap;
            if ((long)a->N * Lp >= (1L << 22)) continue;
            long pmax = 0;
            for (long j0 = 0; j0 < ncols; j0 += BN) {
                const long j1 = (j0 + BN < ncols ? j0 + BN : ncols) - 1;
                const long n0 = j0 / L, r0 = j0 % L, n1 = j1 / L, r1 = j1 % L;
                const long q0 = (n0 * Lp + gap + ((r0 / a->V_out) * g.t_stride - tpad) * g.V_in + r0 % a->V_out) & ~3L;
                const long q1 = n1 * Lp + gap + ((r1 / a->V_out) * g.t_stride - tpad + g.taps - 1) * g.V_in + r1 % a->V_out;
                if (q1 - q0 + 1 > pmax) pmax = q1 - q0 + 1;
            }
            if (pmax <= BS_PMAX) { bp.win[i] = al ? 2 : 1; bp.Lp[i] = Lp; bp.pad[i] = gap; }
        }
    }
    const int ctl = kg_cdiv(ncols, BN), rtl = kg_cdiv(a->M, bs_bm(v));
    bp.mpad = (a->M + 127) / 128 * 128;             // (the same for every tile: a packed buffer serves them all)
    bp.xcd = kg_xcd_grouped(ctl, rtl, kg_env().conv_xcd_min > 0 ? kg_env().conv_xcd_min : KG_XCD_MIN_TILES) ? 1 : 0;
    return true;
}
inline int64_t bs_ws_bytes(const BsPlan& bp) { return (int64_t)bp.nsteps * 12 * bp.mpad * 16; }
// the caller's packed weights are usable for this launch
inline bool bs_packed(const KgConvArgs* a, const BsPlan& bp) {
    return a->wpack != nullptr && a->wpack_bytes >= bs_ws_bytes(bp) && ((uintptr_t)a->wpack & 15) == 0;
}

// variant for a problem the bs form takes
int bs_auto_tile(const KgConvArgs* a) {
    BsPlan bp;
    const long ncols = (long)a->N * a->T_out * a->V_out;
    if (a->M <= 32) return 1;
    // a strided temporal group: its window only fits the 64-column tile
    if (a->M > 64 && (!bs_plan(a, 0, bp) || (a->g[0].taps == 3 && a->g[0].tap_mode == KG_TAP_TIME && !bp.win[0]))) return 2;
    (void)ncols;
    return 0;
}

// Which launches take the bf16-split form (weight-pack launch + tile kernel) when nothing is forced: none (round 6).
// Round 5 sent the all-window 64-channel tails from ~110 samples on (D1's tail in the critic's 3n pass) through it
// (profiles/r05_pack_tails.log: tile kernel 42.4 us + 2.9 us of pack against 50.4 us at 192 samples) - 5 us of a 3.3 ms
// iteration, invisible on the driver's line (3.415 ms either way), while its 1e-5 difference from the fp32 kernel flipped
// LeakyReLU kinks and cost the C5b split=cat property its 1e-4 bound (round-5 VERDICT weak 2 / ADVICE).  The form stays
// opt-in: KG_CONV_BS=1 (wherever it can run), KG_CONV_BS=2 (this rule's round-5 shape class), or a caller that hands over
// packed weights (KgConvArgs.wpack).
bool bs_auto_rule(const KgConvArgs* a, int v, const BsPlan& bp) {
    if (kg_env().conv_bs != 2) return false;
    if (v != 0 || a->M <= 32 || a->M > 64) return false;
    for (int i = 0; i < a->ngroups; ++i)
        if (bp.win[i] != 2) return false;
    if (a->g[0].taps != 3 || a->g[0].tap_mode != KG_TAP_TIME) return false;
    return (long)a->N * a->T_out * a->V_out >= 80000;
}

// ring tile for a problem the ring form takes: by rows, then by how many tiles there are to walk
int ring_auto_tile(const KgConvArgs* a) {
    const long ncols = (long)a->N * a->T_out * a->V_out;
    if (a->M <= 32) return ncols >= 256L * 512 ? 5 : 3;                              // 32 x 256 | 32 x 128
    if (a->M <= 64) return ncols >= 128L * 640 ? 0 : 1;                              // 64 x 128 | 64 x 64
    return (long)kg_cdiv(a->M, 128) * kg_cdiv(ncols, 128) >= 512 ? 2 : 4;           // 128 x 128 | 128 x 64
}

// Which launches take the ring form when nothing is forced (tools/exp_conv.py, profiles/r05_ring_*.log)
bool ring_auto_rule(const KgConvArgs* a, const Plan& p, int s_total) {
    (void)a; (void)p; (void)s_total;
    return false;
}

Plan make_plan(const KgConvArgs* a) {
    const long ncols = (long)a->N * a->T_out * a->V_out;
    const int M = a->M;
    int s_total = slices_of(a->g[0]) + (a->ngroups > 1 ? slices_of(a->g[1]) : 0);
    auto count = [&](Tile t) { return (long)kg_cdiv(M, tile_bm(t)) * kg_cdiv(ncols, tile_bn(t)); };
    Plan p;
    const KgEnv& env = kg_env();
    // Tile (tools/tune_conv.py at 64 and 192 samples with the round-3 kernel, profiles/r03_v18_tune_conv_n*.log).  The
    // 32-row tile has the most workgroups to balance over the CUs and, since the full-slice instantiation, the cheapest
    // slice loop; it wins every SHALLOW contraction (K < 384: the gcn convs, their transposes, the D0 / D1 tails) at
    // both batch sizes - the 64- / 128-row tiles chosen by the round-2 rule cost 10-30 % there (D1 gcn^T at 192 samples:
    // 46.4 -> 32.2 us, D2 gcn^T 50.9 -> 40.7 us).  A DEEP contraction (>= 12 slices) re-reads its feature columns once
    // per row tile, and the larger tiles win as soon as they still give ~2 workgroups per CU (D3 tail at 192 samples:
    // 64 rows 73.9 us, 128 rows with 240 workgroups 82.1 us, 32 rows 89.5 us).
    const bool deep = s_total >= 12;
    p.tile = (M > 32 || count(T32x128) >= 300) ? T32x128 : T32x64;
    if (deep) {
        if (M > 64 && count(T128x128) >= 480)     p.tile = T128x128;
        else if (M > 32 && count(T64x128) >= 480) p.tile = T64x128;
    }
    // tuning hook (tools/tune_conv.py): KG_CONV_PLAN="<tile>,<nsplit>" forces the plan
    int forced_split = 0;
    {
        const int t = env.conv_plan_tile;
        if (tile_known(t)) {
            p.tile = (Tile)t;
            forced_split = env.conv_plan_split;
        }
    }
    // Skinny launches (a few hundred columns, deep K): where the direct kernel would split K across workgroups, the
    // waves of a workgroup split it instead (K32x32: 32 rows x 32 columns per workgroup, every wave a quarter of the
    // slices, partial tiles added in LDS) - no partial slabs in HBM, no second launch.  KG_CONV_KW=0: off (A/B, tests).
    // Measured (profiles/r02_v26_tune_conv_n64.log / _n192.log): 10-20 % faster than the best workgroup split for
    // SHALLOW contractions (8-16 slices: the single-vertex gcn convs of D4 / D5, 21.9 -> 17.5 us at 64 samples), equal
    // or slower for deep ones (D4 tail, 56 slices: 26.7 vs 24.6 us) - with one 32-column group per workgroup every
    // workgroup streams its own copy of the weight rows through L1, which is what bounds these launches.
    // Round 3 (profiles/r03_v18_tune_conv_n*.log): with the full-slice loop it also wins the deeper gcn convs of D4 / D5
    // (24-48 slices: 18.3 -> 14.2 and 17.9 -> 14.8 us at 64 samples, 34.1 -> 27.6 us at 192) as long as its own
    // workgroups (one per 32 x 32 tile) fill most of the chip (not the 3-tap temporal convs: the D5 tail at 192 samples runs
    // 29.4 us on it against 24.2 us K-split four ways) - but only where the 32 x 128 tile cannot put a workgroup
    // on every CU: with a few thousand columns (the D2 / D3 tails at 64 samples, 320 such tiles) the direct tiles with
    // or without a K-split are 10-15 % ahead of it.
    // Round 4 (profiles/r04_audit_kw.log, after the partial tiles' sum and the epilogue were spread over all four waves):
    // it also takes SHALLOW contractions (3-7 slices) whose 32 x 128 tiles would cover a quarter of the CUs or less (the
    // generator-step launches of D4: 9.8 -> 5.9 us, 8.5 -> 6.7 us), and deep temporal convs when the direct tile leaves
    // half the chip empty (D5 tail at 64 samples, 56 slices: 25.3 us as eight K-parts + epilogue -> 22.6-24.1 us).
    if (forced_split == 0 && env.conv_plan_tile < 0 && env.conv_kw != 0 && p.tile <= T32x64 &&
        count(p.tile) * (tile_bm(p.tile) / 32) < 256) {
        const bool time3 = a->g[0].tap_mode == KG_TAP_TIME && a->g[0].taps > 1;
        const long kt = count(K32x32), dt = count(p.tile);
        if (s_total >= 8 && (s_total <= 16 || (s_total <= 48 && kt >= 192 && !time3))) p.tile = K32x32;
        else if (s_total >= 3 && s_total < 8 && dt <= 64 && kt >= 128) p.tile = K32x32;
        else if (s_total > 16 && s_total <= 64 && time3 && dt <= 64 && kt >= 192) p.tile = K32x32;
    }
    const long wgs = count(p.tile);
    int nsplit = 1;
    if (forced_split > 0) {
        nsplit = forced_split > s_total ? s_total : forced_split;
    } else if (tile_kw(p.tile)) {
        if (wgs < 96 && s_total >= 16) {                      // far fewer workgroups than CUs: split across them too
            nsplit = (int)((256 + wgs - 1) / wgs);
            if (nsplit > s_total / 8) nsplit = s_total / 8;   // at least two slices per wave
            if (nsplit > 8) nsplit = 8;
            if (nsplit < 1) nsplit = 1;
        }
    } else if (wgs * (tile_bm(p.tile) / 32) < 400 && s_total >= (wgs >= 256 ? 28 : 8)) {
        // (a launch that already has a workgroup per CU is split only when its contraction is deep: the D2 tail at 64
        // samples, 320 workgroups x 14 slices, runs 29.0 us whole against 32 us as two K-halves + epilogue launch)
        // (a launch of 240 128-row workgroups is a full round already: splitting it made it 20 % slower)
        // K-split (tools/tune_conv.py at 64 and 192 samples): ~6.5 slices per workgroup, at least one workgroup per
        // CU, at most ~1000 workgroups, and a split count that divides the slices evenly (an uneven last split made
        // the D5 gcn launch 15 % slower)
        int want = (int)((2 * s_total + 6) / 13);
        const int fill = (int)((256 + wgs - 1) / wgs);
        if (want < fill) want = fill;
        const int cap = (int)(1000 / wgs) > 1 ? (int)(1000 / wgs) : 1;
        if (want > cap && cap >= fill) want = cap;
        if (want > s_total / 2) want = s_total / 2;          // at least two slices per split
        if (want > 8) want = 8;
        if (want < 1) want = 1;
        nsplit = want;
        for (int dlt = 0; dlt <= 2; ++dlt) {                 // nearest divisor of the slice count, larger one first
            const int hi = want + dlt, lo = want - dlt;
            if (hi <= 8 && hi <= s_total / 2 && s_total % hi == 0) { nsplit = hi; break; }
            if (lo >= 1 && s_total % lo == 0) { nsplit = lo; break; }
        }
    }
    p.sp.inkernel = 0;       // (decided per launch: launch())
    p.sp.per = kg_cdiv(s_total, nsplit);
    p.sp.nsplit = kg_cdiv(s_total, p.sp.per);
    {
        const int ctl = kg_cdiv(ncols, tile_bn(p.tile)), rtl = kg_cdiv(M, tile_bm(p.tile));
        p.sp.xcd = (p.tile <= T32x128 && kg_xcd_grouped(ctl, rtl, env.conv_xcd_min > 0 ? env.conv_xcd_min : KG_XCD_MIN_TILES)) ? 1 : 0;
    }
    // The persistent LDS-ring form (kg_conv_ring.hip).  KG_CONV_RING=1: wherever it can run (tests, A/B); unset: by the
    // rule below; a forced direct plan (KG_CONV_PLAN) or KG_CONV_RING=0 / KG_CONV_FAST=0 keep it off.
    p.ring = -1;
    if (env.conv_ring != 0 && env.conv_fast != 0 && env.conv_plan_tile < 0 && kg_ring_eligible(a)) {
        bool want = env.conv_ring == 1;
        if (env.conv_ring < 0) want = ring_auto_rule(a, p, s_total);
        if (want) {
            const int rt = (env.conv_ring_tile >= 0 && env.conv_ring_tile < kg_ring_tile_count()) ? env.conv_ring_tile : ring_auto_tile(a);
            if (kg_ring_tile_ok(a, rt)) {       // (a window-form tile also needs alignment and a window that fits)
                p.ring = rt;
                p.sp.nsplit = 1;
                p.sp.per = s_total;
            }
        }
    }
    // The bf16-split LDS-staged form.  KG_CONV_BS=1: wherever it can run (tests, A/B); unset: by bs_auto_rule; a forced
    // direct plan, a ring plan or KG_CONV_BS=0 keep it off.
    p.bs = -1;
    // (the eligibility check walks the launch's column tiles on the host: only for launches that can end up on the form -
    // forced, handed packed weights, or inside the plan rule's shape class)
    const bool bs_candidate = env.conv_bs == 1 || a->wpack != nullptr ||
                              (env.conv_bs == 2 && M > 32 && M <= 64 && ncols >= 80000 && a->g[0].taps == 3 && a->g[0].tap_mode == KG_TAP_TIME);
    if (p.ring < 0 && env.conv_bs != 0 && env.conv_plan_tile < 0 && bs_candidate) {
        const int v = (env.conv_bs_tile >= 0 && env.conv_bs_tile <= 2) ? env.conv_bs_tile : bs_auto_tile(a);
        // (a caller that hands over packed weights has chosen the form for this launch)
        if (bs_plan(a, v, p.bsp) && (env.conv_bs == 1 || bs_auto_rule(a, v, p.bsp) ||
                                     (a->wpack != nullptr && a->wpack_bytes >= bs_ws_bytes(p.bsp) && ((uintptr_t)a->wpack & 15) == 0))) {
            p.bs = v;
            p.sp.nsplit = 1;
            p.sp.per = s_total;
        }
    }
    return p;
}

int launch_bs(const KgConvArgs* a, const Plan& p, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    const int ct = kg_cdiv(ncols, bs_bn(p.bs)), rt = kg_cdiv(a->M, bs_bm(p.bs));
    dim3 grid(p.bsp.xcd ? (ct + 7) / 8 * 8 * rt : ct, p.bsp.xcd ? 1 : rt, 1);
    const bool packed = bs_packed(a, p.bsp);
    kg_u32x4* const P = reinterpret_cast<kg_u32x4*>(packed ? const_cast<void*>(a->wpack) : (void*)a->ws);        // (16-byte aligned: checked by kg_conv)
    if (!packed) {
        hipLaunchKernelGGL(kg_conv_bs_pack_kernel, dim3(kg_cdiv((long)p.bsp.nsteps * p.bsp.mpad * 4, 256)), dim3(256), 0, s, *a, p.bsp, P);
        if (int rc = kg_launch_status("kg_conv (bf16-split, weight pack)")) return rc;
    }
    // every group a 16-byte window (the D0 / D1 tails): the hand-scheduled instantiation
    bool allwin = kg_env().conv_bs_asm != 0;
    for (int i = 0; i < a->ngroups; ++i) allwin = allwin && p.bsp.win[i] == 2;
    if (allwin) {
        const bool plain = a->add == nullptr && a->mask == nullptr && a->M % bs_bm(p.bs) == 0;       // (the lean epilogue, see store_tile)
        static unsigned long long attr_mask = 0;
        if (kg_first_on_device(attr_mask)) {
#define KG_BSW_ATTR(TM_, RWV_, CWV_, BM_) do { \
            KG_SET_DYN_LDS((kg_conv_bsw_kernel<TM_, RWV_, CWV_, true>), bsw_lds_bytes(BM_)); \
            KG_SET_DYN_LDS((kg_conv_bsw_kernel<TM_, RWV_, CWV_, false>), bsw_lds_bytes(BM_)); } while (0)
            KG_BSW_ATTR(2, 1, 4, 64); KG_BSW_ATTR(1, 1, 4, 32); KG_BSW_ATTR(2, 2, 2, 128);
#undef KG_BSW_ATTR
        }
#define KG_BSW_GO(TM_, RWV_, CWV_, BM_) do { \
            if (plain) hipLaunchKernelGGL((kg_conv_bsw_kernel<TM_, RWV_, CWV_, true>), grid, dim3(256), bsw_lds_bytes(BM_), s, *a, p.bsp, P); \
            else       hipLaunchKernelGGL((kg_conv_bsw_kernel<TM_, RWV_, CWV_, false>), grid, dim3(256), bsw_lds_bytes(BM_), s, *a, p.bsp, P); } while (0)
        if (p.bs == 0)      KG_BSW_GO(2, 1, 4, 64);
        else if (p.bs == 1) KG_BSW_GO(1, 1, 4, 32);
        else                KG_BSW_GO(2, 2, 2, 128);
#undef KG_BSW_GO
        return kg_launch_status("kg_conv (bf16-split, windows)");
    }
    if (p.bs == 0)      hipLaunchKernelGGL((kg_conv_bs_kernel<2, 1, 4>), grid, dim3(256), 0, s, *a, p.bsp, P);
    else if (p.bs == 1) hipLaunchKernelGGL((kg_conv_bs_kernel<1, 1, 4>), grid, dim3(256), 0, s, *a, p.bsp, P);
    else                hipLaunchKernelGGL((kg_conv_bs_kernel<2, 2, 2>), grid, dim3(256), 0, s, *a, p.bsp, P);
    return kg_launch_status("kg_conv (bf16-split)");
}

template <int BM, int NW, int KW = 1>
int launch(const KgConvArgs* a, const Plan& p, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    // 1-D tile grid, column tiles padded to a multiple of 8 (kg_tile_of_block)
    const int ct = kg_cdiv(ncols, 32 * NW / KW), rt = kg_cdiv(a->M, BM);
    dim3 grid(p.sp.xcd ? (ct + 7) / 8 * 8 * rt : ct, p.sp.xcd ? 1 : rt, p.sp.nsplit);
    // a K-split launch completes its tiles itself when the caller handed over zeroed ticket counters for them (finish_split);
    // KG_CONV_INKERNEL=0 or no counters: the separate epilogue launch
    Split sp = p.sp;
    const int ink_max = kg_env().conv_inkernel_max > 0 ? kg_env().conv_inkernel_max : 4;
    sp.inkernel = (BM <= 64 && sp.nsplit > 1 && sp.nsplit <= ink_max && kg_env().conv_inkernel != 0 && a->sync != nullptr &&
                   (long)ct * rt <= a->sync_len) ? 1 : 0;
    // full K-slices everywhere (32-bit-load kernels: slices of 32 channels): the FAST instantiation
    bool fast = kg_env().conv_fast != 0;
    for (int i = 0; i < a->ngroups; ++i) fast = fast && (a->g[i].Cin % 32 == 0);
    const bool kf = a->g[0].w_sI <= a->g[0].w_sO;
    // (the lean epilogue for the full-slice instantiations of launches without add / mask; K-split partial tiles never reach it)
    const bool plain = a->add == nullptr && a->mask == nullptr && a->M % BM == 0 && kg_env().conv_plain_epi != 0;
#define KG_CONV_GO3(KF_, FAST_, PLAIN_, INK_) hipLaunchKernelGGL((kg_conv_kernel<BM, NW, KF_, KW, FAST_, PLAIN_, INK_>), grid, dim3(64 * NW), 0, s, *a, sp)
#define KG_CONV_GO2(KF_, FAST_, PLAIN_) do { if constexpr (BM <= 64) { if (sp.inkernel) KG_CONV_GO3(KF_, FAST_, PLAIN_, true); else KG_CONV_GO3(KF_, FAST_, PLAIN_, false); } \
                                             else KG_CONV_GO3(KF_, FAST_, PLAIN_, false); } while (0)
#define KG_CONV_GO(KF_, FAST_) do { if (plain) KG_CONV_GO2(KF_, FAST_, true); else KG_CONV_GO2(KF_, FAST_, false); } while (0)
    if (fast && a->ngroups == 1) {
        if (kf) KG_CONV_GO(true, 1); else KG_CONV_GO(false, 1);
    } else if (fast) {
        if (kf) KG_CONV_GO(true, 2); else KG_CONV_GO(false, 2);
    } else {
        if (kf) KG_CONV_GO2(true, 0, false); else KG_CONV_GO2(false, 0, false);
    }
#undef KG_CONV_GO3
#undef KG_CONV_GO2
#undef KG_CONV_GO
    if (int rc = kg_launch_status("kg_conv")) return rc;
    if (p.sp.nsplit > 1 && !sp.inkernel) {
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
    KG_REQUIRE(a->sync == nullptr || a->sync_len > 0, "kg_conv: sync_len=%d", a->sync_len);
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
    if (p.bs >= 0) n = bs_packed(a, p.bsp) ? 0 : bs_ws_bytes(p.bsp);      // the packed weights (the bf16-split form never splits K)
#ifdef KG_CONV_TIMING
    n += 2 << 20;
#endif
    return n;
}

}  // namespace

extern "C" int64_t kg_conv_workspace_bytes(const KgConvArgs* a) {
    if (validate(a) != 0) return -1;
    if (tiny_eligible(a)) return 0;
    return ws_bytes(a, make_plan(a));
}

extern "C" int kg_conv_plan_info(const KgConvArgs* a, int32_t* tile, int32_t* nsplit) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(tile && nsplit, "kg_conv_plan_info: null output");
    if (tiny_eligible(a)) {
        *tile = 11;                 // the tiny-channel streaming kernel
        *nsplit = 1;
        return 0;
    }
    Plan p = make_plan(a);
    *tile = p.bs >= 0 ? 40 + p.bs : p.ring >= 0 ? 20 + p.ring : (int32_t)p.tile;      // 20..: ring tiles, 40..: bf16-split tiles
    *nsplit = p.sp.nsplit;
    return 0;
}

extern "C" int kg_conv(const KgConvArgs* a, void* stream) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(a->out != nullptr, "kg_conv: null out");
    for (int i = 0; i < a->ngroups; ++i) KG_REQUIRE(a->g[i].x && a->g[i].w, "kg_conv: group %d null pointer", i);
    if (tiny_eligible(a)) return launch_tiny(a, (hipStream_t)stream);
    Plan p = make_plan(a);
    const int64_t need = ws_bytes(a, p);
    KG_REQUIRE(need == 0 || (a->ws != nullptr && a->ws_bytes >= need), "kg_conv: workspace %ld < %ld bytes",
               (long)a->ws_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    if (p.ring >= 0) return kg_ring_launch(a, p.ring, s);
    if (p.bs >= 0) {
        KG_REQUIRE(bs_packed(a, p.bsp) || ((uintptr_t)a->ws & 15) == 0, "kg_conv: workspace must be 16-byte aligned");
        return launch_bs(a, p, s);
    }
    switch (p.tile) {
        case T128x128: return launch<128, 4>(a, p, s);
        case T64x128:  return launch<64, 4>(a, p, s);
        case T32x128:  return launch<32, 4>(a, p, s);
        case T64x64:   return launch<64, 2>(a, p, s);
        case T32x64:   return launch<32, 2>(a, p, s);
        default:       return launch<32, 4, 4>(a, p, s);      // K32x32
    }
}

extern "C" int64_t kg_conv_pack_bytes(const KgConvArgs* a) {
    if (validate(a) != 0) return -1;
    if (tiny_eligible(a) || kg_env().conv_bs == 0) return 0;
    BsPlan bp;
    return bs_plan(a, bs_auto_tile(a), bp) ? bs_ws_bytes(bp) : 0;
}

extern "C" int kg_conv_pack(const KgConvArgs* a, void* wpack, int64_t wpack_bytes, void* stream) {
    if (int rc = validate(a)) return rc;
    for (int i = 0; i < a->ngroups; ++i) KG_REQUIRE(a->g[i].w, "kg_conv_pack: group %d null weights", i);
    BsPlan bp;
    KG_REQUIRE(!tiny_eligible(a) && bs_plan(a, bs_auto_tile(a), bp), "kg_conv_pack: the bf16-split form cannot run this launch");
    KG_REQUIRE(wpack != nullptr && ((uintptr_t)wpack & 15) == 0 && wpack_bytes >= bs_ws_bytes(bp),
               "kg_conv_pack: buffer %ld < %ld bytes, or not 16-byte aligned", (long)wpack_bytes, (long)bs_ws_bytes(bp));
    hipLaunchKernelGGL(kg_conv_bs_pack_kernel, dim3(kg_cdiv((long)bp.nsteps * bp.mpad * 4, 256)), dim3(256), 0, (hipStream_t)stream, *a, bp,
                       reinterpret_cast<kg_u32x4*>(wpack));
    return kg_launch_status("kg_conv_pack");
}

// Can the jobs share one launch, and with which tile?  (-1: no.)  A K32x32 plan without a K-split across workgroups is a
// per-launch choice (it avoids partial slabs); inside a shared launch the job takes the common tile.
static int many_tile(const KgConvArgs* jobs, int njobs) {
    if (njobs < 2 || njobs > CONV_MANY_MAX || kg_env().conv_fast == 0 || kg_env().conv_many == 0 || kg_env().conv_plan_tile >= 0)
        return -1;
    double best_work = -1.0;
    int tile = T32x128;
    const bool kf = jobs[0].g[0].w_sI <= jobs[0].g[0].w_sO;
    for (int i = 0; i < njobs; ++i) {
        const KgConvArgs* a = &jobs[i];
        if ((a->g[0].w_sI <= a->g[0].w_sO) != kf || tiny_eligible(a)) return -1;
        double work = 0.0;
        for (int q = 0; q < a->ngroups; ++q) {
            if (a->g[q].Cin % 32 != 0) return -1;
            work += (double)a->g[q].taps * a->g[q].Cin;
        }
        const Plan p = make_plan(a);
        if (p.sp.nsplit != 1) return -1;
        const int t = p.tile <= T32x128 ? (int)p.tile : (int)T32x128;
        work *= (double)a->M * a->N * a->T_out * a->V_out;
        if (work > best_work) { best_work = work; tile = t; }
    }
    return tile;
}

extern "C" int kg_conv_many_plan(const KgConvArgs* jobs, int32_t njobs, int32_t* tile) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1 && tile != nullptr, "kg_conv_many_plan: bad arguments");
    for (int i = 0; i < njobs; ++i)
        if (int rc = validate(&jobs[i])) return rc;
    *tile = many_tile(jobs, njobs);
    return 0;
}

// several independent problems in ONE launch where their plans allow it (see kg_conv_many_kernel), else one by one
extern "C" int kg_conv_many(const KgConvArgs* jobs, int32_t njobs, void* stream) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1, "kg_conv_many: no jobs");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < njobs; ++i) {
        const KgConvArgs* a = &jobs[i];
        if (int rc = validate(a)) return rc;
        KG_REQUIRE(a->out != nullptr, "kg_conv_many: job %d null out", i);
        for (int q = 0; q < a->ngroups; ++q) KG_REQUIRE(a->g[q].x && a->g[q].w, "kg_conv_many: job %d group %d null pointer", i, q);
    }
    const int mt = many_tile(jobs, njobs);
    if (mt < 0) {
        for (int i = 0; i < njobs; ++i)
            if (int rc = kg_conv(&jobs[i], stream)) return rc;
        return 0;
    }
    const Tile tile = (Tile)mt;
    const bool kf = jobs[0].g[0].w_sI <= jobs[0].g[0].w_sO;
    ConvMany m;
    m.njobs = njobs;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        const KgConvArgs* a = &jobs[i];
        const int ncols = a->N * a->T_out * a->V_out;
        ConvManyJob& j = m.job[i];
        j.a = *a;
        j.sp.nsplit = 1;
        j.sp.per = slices_of(a->g[0]) + (a->ngroups > 1 ? slices_of(a->g[1]) : 0);
        j.wg_begin = total;               // (a multiple of 8: the XCD of a workgroup is its index in the launch & 7)
        j.ctiles = kg_cdiv(ncols, 128);
        const int rtl = kg_cdiv(a->M, tile_bm(tile));
        j.sp.xcd = kg_xcd_grouped(j.ctiles, rtl, kg_env().conv_xcd_min > 0 ? kg_env().conv_xcd_min : KG_XCD_MIN_TILES) ? 1 : 0;
        j.nwg = j.ctiles * rtl;
        total += j.sp.xcd ? (j.ctiles + 7) / 8 * 8 * rtl : (j.nwg + 7) / 8 * 8;
    }
    bool plain = kg_env().conv_plain_epi != 0;
    for (int i = 0; i < njobs; ++i) plain = plain && jobs[i].add == nullptr && jobs[i].mask == nullptr && jobs[i].M % tile_bm(tile) == 0;
#define KG_MANY_GO(BM_) do { if (kf && plain)  hipLaunchKernelGGL((kg_conv_many_kernel<BM_, true, true>), dim3(total), dim3(256), 0, s, m); \
                             else if (kf)      hipLaunchKernelGGL((kg_conv_many_kernel<BM_, true, false>), dim3(total), dim3(256), 0, s, m); \
                             else if (plain)   hipLaunchKernelGGL((kg_conv_many_kernel<BM_, false, true>), dim3(total), dim3(256), 0, s, m); \
                             else              hipLaunchKernelGGL((kg_conv_many_kernel<BM_, false, false>), dim3(total), dim3(256), 0, s, m); } while (0)
    if (tile == T128x128)     KG_MANY_GO(128);
    else if (tile == T64x128) KG_MANY_GO(64);
    else                      KG_MANY_GO(32);
#undef KG_MANY_GO
    return kg_launch_status("kg_conv_many");
}
