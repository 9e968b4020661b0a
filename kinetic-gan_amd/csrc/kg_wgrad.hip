// kg_wgrad: weight gradient of the tap GEMM,
//   dW(d, m, c) = sum_j G[m, j] * X[c (+ d*Cin), src(j, d)]
// a GEMM whose contraction runs over the batch's columns j = (n, t, v).  64x64 output tile per
// workgroup (4 waves x one 32x32 v_mfma_f32_32x32x2_f32 tile), both operand tiles staged in LDS
// column-contiguous ([row][64+1], conflict-free for the row-per-lane fragment reads), the column
// range split across workgroups into partial slabs that a second kernel sums in a fixed order
// (deterministic; no atomics).
// Reference op covered: the weight half of aten::convolution_backward for tgcn.py:61,
// discriminator.py:99-105,115-120 and generator.py:134-140,154-159.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "kg_common.h"

namespace {

constexpr int BM = 64, BN = 64, BJ = 64, NT = 256;
#ifndef KG_WGRAD_PJ
#define KG_WGRAD_PJ 64
#endif
constexpr int PJ = KG_WGRAD_PJ;       // columns per chunk of the per-tap kernel


// splits [sbeg[p], sbeg[p+1]) walk the columns of operand pair p in ranges of cps[p] columns
// lmul / lshr, vmul / vshr: floor(j / (T_out V_out)) and floor(r / V_out) as umulhi(j, mul) >> shr (j < 2^31; found on the
// host): the column decode of every chunk otherwise costs two ~30-instruction integer divisions per thread.
// full: every staged row of every tile is inside the tensors (M and Cin multiples of the tile) - the FULL instantiation
// of wgrad_tile walks the rows through the buffer loads' scalar offset without per-row validity selects.  It is NOT used
// by kg_wgrad_many_kernel: a second set of inlined tile variants made hipcc copy the whole job table (3.5 KB of kernel
// arguments) to scratch memory and raised the kernel to 171 VGPRs - the launch ran 2x slower (622 vs 296 us).
struct Plan { int tiles_m, tiles_n, splits; int sbeg[4]; int cps[3]; unsigned lmul, lshr, vmul, vshr; int full; };

inline void wg_magic(unsigned d, unsigned& mul, unsigned& shr) {
    mul = 0; shr = 0;
    if (d <= 1) return;
    unsigned lg = 0;
    while ((1u << lg) < d) ++lg;
    const unsigned p = 31 + lg;
    mul = (unsigned)(((1ull << p) + d - 1) / d);
    shr = p - 32;
}

inline int pair_N(const KgWgradArgs* a, int p) { return p == 0 ? a->N : a->extra[p - 1].N; }

// Tile variants of the per-tap kernel.  The 4 waves of a workgroup form a GM x GN x GK grid: a wave owns WM x WN MFMA
// tiles (32 x 32 weights each) of the workgroup's (32 GM WM) x (32 GN WN) tile and every GK-th part of a chunk's
// columns (GK > 1: the waves' accumulators are added through LDS at the end).
//   V_BIG   128 x 128, chunks of 32 columns: an operand fragment read from LDS feeds two MFMAs and a staged element
//           four (the 256 / 512-channel layers, where most of the work is)
//   V_6464  64 x 64, the general tile
//   V_6432 / V_3264 / V_3232: layers with <= 32 input channels and / or output rows (D0, D1's gcn and residual, the
//           generator's last blocks): the 64 x 64 tile would multiply zero rows (D0 gcn: 3 of 64 channels in use)
//           (V_3232 walks 128 columns per chunk: 16 MFMAs per wave between two barriers instead of 8)
// Fragment reads (round 4): ds_read_b128 on the small tiles, ds_read_b64 on V_BIG (wgrad_tile's RW; VGPRs 119 -> 125) -
// a quarter / half of the LDS read instructions and waits per MFMA; critic pass 375 -> 361 us with both.
enum { V_BIG = 0, V_6464, V_6432, V_3264, V_3232, V_COUNT };
#ifndef KG_WG_RW
#define KG_WG_RW 4
#endif
#ifndef KG_WG_RWBIG
#define KG_WG_RWBIG 2
#endif
#ifndef KG_WG_PJ3232
#define KG_WG_PJ3232 128
#endif
constexpr int PJ_3232 = KG_WG_PJ3232;      // columns per chunk of the 32 x 32 tile
constexpr int RW_S = KG_WG_RW, RW_B = KG_WG_RWBIG;     // columns per fragment read (wgrad_tile's RW): small tiles / V_BIG
struct Tile { int bm, bn, pj; float cost; int rw; };       // cost of one chunk relative to V_6464's (measured)
constexpr Tile TILES[V_COUNT] = {{128, 128, 32, 2.0f * 64 / PJ, RW_B}, {64, 64, PJ, 1.0f, RW_S}, {64, 32, PJ, 0.65f, RW_S},
                                 {32, 64, PJ, 0.65f, RW_S}, {32, 32, PJ_3232, 0.4f * PJ_3232 / PJ, RW_S}};
#ifdef KG_WGRAD_NO_BIG          // A/B builds (tools/gpu_ab.sh)
inline int tile_variant(const KgWgradArgs*) { return V_6464; }
#else
inline int tile_variant(const KgWgradArgs* a) {
    // the 128 x 128 tile only from 4096 columns (all operand pairs together) on: the 512-channel layers of D4 / D5 have
    // 768-1536 columns per pair - a handful of chunks per workgroup and 64 KB partial slabs each; on 64 x 64 tiles they
    // run 10-50 % faster one by one (D4 res 31.5 -> 16.5 us) and the iteration 1.3 % (round 4, with the 3072-workgroup
    // budget; KG_WGRAD_BIGCOLS / KG_WGRAD_BUDGET, profiles/r04_wgrad_tiles.log)
    long cols = 0;
    for (int q = 0; q <= a->nextra; ++q) cols += (long)pair_N(a, q) * a->T_out * a->V_out;
    const long bigcols = kg_env().wgrad_bigcols >= 0 ? kg_env().wgrad_bigcols : 4096;
    if (a->M >= 128 && a->Cin >= 128 && cols >= bigcols) return V_BIG;
    const bool m32 = a->M <= 32, c32 = a->Cin <= 32;
    return m32 && c32 ? V_3232 : (c32 ? V_6432 : (m32 ? V_3264 : V_6464));
}
#endif
// ONE staging buffer per operand (round 4): the next chunk sits in registers while the current one is multiplied, and
// goes to LDS between two barriers.  The second buffer saved one barrier per chunk but held a workgroup at 67 KB of LDS -
// two workgroups, two waves per SIMD; with 34 KB four workgroups share a CU and the other three fill the matrix pipe
// while one stages (critic pass 312 -> 282 us; the double-buffer / no-load / no-MFMA ablation builds of round 4 are kept as
// tools/probe/wgrad_ablation_switches.patch).
constexpr int LDS_BUFS = 1;
constexpr size_t tile_lds(int v) {
    return (size_t)LDS_BUFS * (TILES[v].bm + TILES[v].bn) * (TILES[v].pj + (TILES[v].rw == 1 ? 1 : TILES[v].rw)) * sizeof(float);
}
constexpr size_t TILE_LDS_MAX = tile_lds(V_BIG) > tile_lds(V_6464) ? tile_lds(V_BIG) : tile_lds(V_6464);
// the bf16-split tiles: three bf16 terms per element, rows of pj * 2 + 16 bytes; at least the GK-way reduction scratch
constexpr size_t tile_lds_bs(int v) {
    return (size_t)3 * (TILES[v].bm + TILES[v].bn) * (2 * TILES[v].pj + 16);
}
constexpr size_t TILE_LDS_BS_MAX = tile_lds_bs(V_BIG) > tile_lds_bs(V_6464) ? tile_lds_bs(V_BIG) : tile_lds_bs(V_6464);
static_assert(tile_lds_bs(V_3232) >= 3 * 16 * 64 * 4 && tile_lds_bs(V_6432) >= 2 * 16 * 64 * 4, "reduction scratch of the GK > 1 tiles");

// per_target > 0: chunks per split asked for by the caller (kg_wgrad_many balances all layers of a pass against each
// other); 0: a single layer, aim at ~768 workgroups
Plan make_plan(const KgWgradArgs* a, long per_target = 0, Tile t = TILES[V_6464]) {
    Plan p;
    p.tiles_m = kg_cdiv(a->M, t.bm);
    p.tiles_n = kg_cdiv(a->Cin, t.bn);
    const long tiles = (long)p.tiles_m * p.tiles_n * a->taps;
    const int npairs = 1 + a->nextra;
    const int PJ = t.pj;
    long chunks_all = 0;
    int chunks[3] = {0, 0, 0};
    for (int q = 0; q < npairs; ++q) {
        chunks[q] = kg_cdiv((long)pair_N(a, q) * a->T_out * a->V_out, PJ);
        chunks_all += chunks[q];
    }
    long s = per_target > 0 ? (chunks_all + per_target - 1) / per_target : (768 + tiles - 1) / tiles;
    if (s > chunks_all) s = chunks_all;
    if (s > 128) s = 128;
    if (s < 1) s = 1;
    const int per = kg_cdiv(chunks_all, s);                   // chunks per split, the same for every pair
    p.sbeg[0] = 0;
    for (int q = 0; q < 3; ++q) {
        p.cps[q] = per * PJ;
        p.sbeg[q + 1] = p.sbeg[q] + (q < npairs ? kg_cdiv(chunks[q], per) : 0);
    }
    p.splits = p.sbeg[3];
    wg_magic((unsigned)(a->T_out * a->V_out), p.lmul, p.lshr);
    wg_magic((unsigned)a->V_out, p.vmul, p.vshr);
    p.full = (a->M % t.bm == 0 && a->Cin % t.bn == 0) ? 1 : 0;
    return p;
}

// end of a tile: the waves that share an output tile add their accumulators through LDS, then the tile goes to its partial
// slab [split][tap][M][Cin] (or straight into dw: single-split layers of kg_wgrad_many)
// SLAB: no job of the launch writes its gradient directly (every layer has several splits: the critic pass) - the direct
// path (read-modify-write of dw) is not compiled in.
template <int GM, int GN, int GK, int WM, int WN, bool SLAB = false>
__device__ __forceinline__ void wgrad_finish(float* const lds, const KgWgradArgs& a, const Plan& p, kg_f32x16 (&acc)[WM][WN],
                                             const int m0, const int c0, const int d, const int split) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wk = wave / (GM * GN), wmn = wave % (GM * GN);
    const int wm = wmn / GN, wn = wmn % GN;
    if constexpr (GK > 1) {
        // the waves that share an output tile add their accumulators through LDS (free after the loop's last barrier)
        float* const red = lds;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        red[((((wk - 1) * GM * GN + wmn) * WM * WN + i * WN + k) * 16 + r) * 64 + lane] = acc[i][k][r];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int w2 = 0; w2 < GK - 1; ++w2)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int k = 0; k < WN; ++k)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[i][k][r] += red[(((w2 * GM * GN + wmn) * WM * WN + i * WN + k) * 16 + r) * 64 + lane];
    }
    // a single split (few columns: the generator's first blocks, 64-1280 columns) writes / adds straight into the
    // gradient: no partial slab, no reduction job (kg_wgrad_many leaves such layers out of the reduction launch)
    const bool direct = !SLAB && p.splits == 1 && a.defer_reduce == 2;
    float* slab = a.ws + ((long)split * a.taps + d) * (long)a.M * a.Cin;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int k = 0; k < WN; ++k) {
            const int c = c0 + (wn * WN + k) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < a.M && c < a.Cin) {
                    if (direct) {
                        float* o = a.dw + (long)d * a.w_sT + (long)m * a.w_sO + (long)c * a.w_sI;
                        *o = a.accumulate ? *o + acc[i][k][r] : acc[i][k][r];
                    } else {
                        slab[(long)m * a.Cin + c] = acc[i][k][r];
                    }
                }
            }
        }
}

template <int GM, int GN, int GK, int WM, int WN, int PJ, bool FULL = false, int RW = 1, bool SLAB = false>
__device__ __forceinline__ void wgrad_tile(float* const lds, const KgWgradArgs& a, const Plan& p, const int tile, const int d,
                                           const int split) {
    static_assert(GM * GN * GK == NT / 64, "wave grid");
    constexpr int BM = 32 * GM * WM, BN = 32 * GN * WN;
    // RW: columns a lane takes per fragment read (ds_read_b32 / b64 / b128).  A lane's RW consecutive columns feed RW
    // MFMAs in turn: MFMA e of a group contracts columns {c0 + e, c0 + RW + e} (lanes 0-31 / 32-63) - any pairing of the
    // chunk's columns is a valid contraction order as long as both operands use it.  Row pitch PJ + 1 (RW = 1) or PJ + RW:
    // 16-byte aligned rows, and the 16 lanes one ds_read_b128 pass serves fall on 64 different banks.
    constexpr int LDP = PJ + (RW == 1 ? 1 : RW);
    typedef float GsT[BM][LDP];
    typedef float XsT[BN][LDP];
    typedef float FragT __attribute__((ext_vector_type(RW)));
    GsT* const Gs = reinterpret_cast<GsT*>(lds);                       // [LDS_BUFS][BM][LDP]
    XsT* const Xs = reinterpret_cast<XsT*>(lds + LDS_BUFS * BM * LDP);   // [LDS_BUFS][BN][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (GM * GN), wmn = wave % (GM * GN);
    const int wm = wmn / GN, wn = wmn % GN;
    const int m0 = (tile / p.tiles_n) * BM;
    const int c0 = (tile % p.tiles_n) * BN;
    // operand pair of this split (uniform)
    const int pr = (split >= p.sbeg[1] ? 1 : 0) + (split >= p.sbeg[2] ? 1 : 0);
    const int pN = pr == 0 ? a.N : a.extra[pr - 1].N;
    const float* const pg = pr == 0 ? a.g : a.extra[pr - 1].g;
    const float* const px_ = pr == 0 ? a.x : a.extra[pr - 1].x;
    const long g_sN = pr == 0 ? a.g_sN : a.extra[pr - 1].g_sN, g_sC = pr == 0 ? a.g_sC : a.extra[pr - 1].g_sC;
    const long x_sN = pr == 0 ? a.x_sN : a.extra[pr - 1].x_sN, x_sC = pr == 0 ? a.x_sC : a.extra[pr - 1].x_sC;
    const int ncols = pN * a.T_out * a.V_out;
    const int L = a.T_out * a.V_out;
    const int cps = p.cps[pr];
    const int jbeg = (split - p.sbeg[pr]) * cps;
    const int jend = min(ncols, jbeg + cps);
    const int shift = (a.tap_mode == KG_TAP_TIME) ? d - (a.taps - 1) / 2 : 0;
    const int choff = (a.tap_mode == KG_TAP_CHANBLOCK) ? d * a.Cin : 0;

    const int cj = tid & (PJ - 1);   // this thread's column inside a chunk
    const int r0 = tid / PJ;         // first row it stages (rows r0, r0 + RSTEP, ...)
    constexpr int RPG = BM / (NT / PJ), RPX = BN / (NT / PJ);   // rows per thread of g / of x
    static_assert(BM % (NT / PJ) == 0 && BN % (NT / PJ) == 0, "staging pattern");

    kg_f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int k = 0; k < WN; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    float greg[RPG], xreg[RPX];
    // raw buffer loads: wave-uniform descriptor (tensor base + this tile's first row), 32-bit byte offsets,
    // out-of-range offset == reads as 0 (rows beyond M / Cin, padding frames, dropped vertices, ragged tail)
    constexpr unsigned RANGE = 0x80000000u, OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(pg + (long)m0 * g_sC), 0, (int)RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(px_ + (long)(choff + c0) * x_sC), 0, (int)RANGE, 0x00020000);
    constexpr int RSTEP = NT / PJ;
    const int g_nvalid = (a.M - m0 - r0 + RSTEP - 1) / RSTEP;       // staged rows i < nvalid are inside the tensor
    const int x_nvalid = (a.Cin - c0 - r0 + RSTEP - 1) / RSTEP;
    const unsigned g_step = (unsigned)(RSTEP * g_sC * 4), x_step = (unsigned)(RSTEP * x_sC * 4);
    // global -> registers, software pipelined against the MFMAs of the current chunk.  prep() resolves the chunk's
    // per-thread base offsets (one column per thread: decode + time shift / stride / vertex gather);
    // load_g(i) / load_x(i) issue one row each.  jc >= jend: every offset out of range (reads as 0).
    unsigned gb = OOB, xb = OOB;
    auto prep = [&](int jc) {
        const int j = jc + cj;
        gb = OOB; xb = OOB;
        if (j < jend) {
            const int n = L == 1 ? j : (int)(__umulhi((unsigned)j, p.lmul) >> p.lshr), r = j - n * L;
            const int to = a.V_out == 1 ? r : (int)(__umulhi((unsigned)r, p.vmul) >> p.vshr), vo = r - to * a.V_out;
            gb = (unsigned)(((long)r0 * g_sC + (long)n * g_sN + r) * 4);
            int vi = a.vmap ? a.vmap[vo] : vo;
            int ti = to * a.t_stride + shift;
            if (vi >= 0 && ti >= 0 && ti < a.T_in)
                xb = (unsigned)(((long)r0 * x_sC + (long)n * x_sN + (long)ti * a.V_in + vi) * 4);
        }
    };
    auto load_g = [&](int i) {
        if constexpr (FULL) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, gb, i * g_step, 0));
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, i < g_nvalid ? gb + i * g_step : OOB, 0, 0));
    };
    auto load_x = [&](int i) {
        if constexpr (FULL) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xb, i * x_step, 0));
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, i < x_nvalid ? xb + i * x_step : OOB, 0, 0));
    };
    auto stash = [&](int b) {
        float* pg = &Gs[b][r0][cj];
        float* px = &Xs[b][r0][cj];
#pragma unroll
        for (int i = 0; i < RPG; ++i) pg[i * RSTEP * LDP] = greg[i];
#pragma unroll
        for (int i = 0; i < RPX; ++i) px[i * RSTEP * LDP] = xreg[i];
    };

    if (jbeg < jend) {
        prep(jbeg);
#pragma unroll
        for (int i = 0; i < RPG; ++i) greg[i] = load_g(i);
#pragma unroll
        for (int i = 0; i < RPX; ++i) xreg[i] = load_x(i);
        stash(0);
        __syncthreads();
        int b = 0;
        for (int jc = jbeg; jc < jend; jc += PJ, b ^= (LDS_BUFS - 1)) {
            // one scheduling step per MFMA: the operands of step q+2 are read from LDS, one row of the NEXT chunk
            // are requested from memory, MFMA q issues.  (All 32 loads up front made a wave sit in the load-issue
            // queue before its first MFMA; see kg_conv.hip.)
            prep(jc + PJ);
            constexpr int KS = PJ / 2 / GK;                             // this wave's k-steps (2 columns each) per chunk
            static_assert(KS % RW == 0, "fragment read width");
            constexpr int KG = KS / RW;                                 // fragment reads per operand tile and chunk
            const float* ga = &Gs[b][wm * 32 * WM + (lane & 31)][RW * (lane >> 5) + 2 * KS * wk];
            const float* xa = &Xs[b][wn * 32 * WN + (lane & 31)][RW * (lane >> 5) + 2 * KS * wk];
            constexpr int LPS = (RPG + RPX + KS / 2 - 1) / (KS / 2);    // the next chunk's rows go out during the first KS/2 steps
            FragT av[KG][WM], bv[KG][WN];
            auto read_ab = [&](int q) {
#pragma unroll
                for (int i = 0; i < WM; ++i) av[q][i] = *reinterpret_cast<const FragT*>(ga + i * 32 * LDP + 2 * RW * q);
#pragma unroll
                for (int k = 0; k < WN; ++k) bv[q][k] = *reinterpret_cast<const FragT*>(xa + k * 32 * LDP + 2 * RW * q);
            };
            auto frag = [](const FragT& f, int e) -> float { return f[e]; };
            constexpr int RD = RW == 1 ? 2 : 1;                         // fragment reads in flight
#pragma unroll
            for (int q = 0; q < RD; ++q) read_ab(q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                if (q % RW == 0 && q / RW + RD < KG) read_ab(q / RW + RD);
#pragma unroll
                for (int l = 0; l < LPS; ++l) {
                    const int idx = q * LPS + l;
                    if (idx < RPG) greg[idx] = load_g(idx);
                    else if (idx < RPG + RPX) xreg[idx - RPG] = load_x(idx - RPG);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int k = 0; k < WN; ++k)
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag(av[q / RW][i], q % RW), frag(bv[q / RW][k], q % RW),
                                                                         acc[i][k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (LDS_BUFS == 1) __syncthreads();
            stash(b ^ (LDS_BUFS - 1));
            __syncthreads();
        }
    }

    wgrad_finish<GM, GN, GK, WM, WN, SLAB>(lds, a, p, acc, m0, c0, d, split);
}

// =====================================================================================================================
// The bf16-split form of the tile (round 5; opt-in with KG_WGRAD_SPLIT=1 - profiles/r05_wgrad_split.log: the 128 x 128-tile
// layers gain 9-20 %, the narrow tiles lose 3-15 %, the 16-layer critic pass as a whole 480 -> 491 us).
// Every fp32 operand element x is written as three bf16 terms x = h + m + l (round to nearest at each level: 24 mantissa
// bits, the sum is exact) and a product of two elements as hh + hm + mh + mm + hl + lh (the three dropped terms are below
// 2^-24 of the product - fp32 rounding level); v_mfma_f32_32x32x16_bf16 runs 16x the rate of v_mfma_f32_32x32x2_f32, so
// the six products of a 32 x 32 x 16 block cost 192 matrix-pipe cycles where fp32 costs 512, and they still accumulate in
// fp32.  What made the same trick useless in kg_conv (its operands go from memory straight into every
// wave's registers, each wave would split them again and again) is free here: both operand tiles are staged through LDS
// anyway, so an element is split ONCE per workgroup on its way into LDS (5.5 VALU instructions) however many waves and
// MFMAs read it.  The contraction index of this GEMM is the column j - contiguous in memory for both operands - so a
// lane's eight consecutive k of a 32x32x16 fragment are one 16-byte LDS read per term.
//   staging: a thread owns the column PAIR (cp, cp + PJ/2) of a chunk (two coalesced 4-byte loads per staged row - any
//   pairing of the chunk's columns is a valid contraction order as long as both operands use it) and writes one packed
//   bf16 pair per term: LDS rows [term][row][PJ bf16 + 16 bytes], pitch chosen so that a 16-lane ds_read_b128 pass hits 64
//   different banks (PJ = 32: 80 B, 64: 144 B, 128: 272 B).
// =====================================================================================================================
typedef unsigned kg_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 kg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 kg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float kg_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bs_cvt2(float a, float b) {
    const kg_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, kg_bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void bs_split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = bs_cvt2(x0, x1);
    x0 -= __uint_as_float(h << 16); x1 -= __uint_as_float(h & 0xffff0000u);          // exact
    m = bs_cvt2(x0, x1);
    x0 -= __uint_as_float(m << 16); x1 -= __uint_as_float(m & 0xffff0000u);          // exact, <= 8 significant bits left
    l = bs_cvt2(x0, x1);
}
__device__ __forceinline__ kg_f32x16 bs_mfma(const kg_u32x4& a, const kg_u32x4& b, const kg_f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(kg_bf16x8, a), __builtin_bit_cast(kg_bf16x8, b), c, 0, 0, 0);
}

constexpr int bs_pitch(int pj) { return 2 * pj + 16; }       // bytes per LDS row

template <int GM, int GN, int GK, int WM, int WN, int PJ>
__device__ __forceinline__ void wgrad_tile_bs(float* const lds, const KgWgradArgs& a, const Plan& p, const int tile, const int d,
                                              const int split) {
    static_assert(GM * GN * GK == NT / 64, "wave grid");
    constexpr int BM = 32 * GM * WM, BN = 32 * GN * WN;
    constexpr int PITCH = bs_pitch(PJ);
    constexpr int HP = PJ / 2;                 // column pairs per chunk
    constexpr int RSTEP = NT / HP;             // rows one staging pass covers
    constexpr int RPG = BM / RSTEP, RPX = BN / RSTEP;      // staging passes of g / of x (two loads each)
    static_assert(NT % HP == 0 && BM % RSTEP == 0 && BN % RSTEP == 0, "staging pattern");
    static_assert(PJ % (16 * GK) == 0, "k-groups per wave");
    char* const Gs = reinterpret_cast<char*>(lds);         // [3 terms][BM][PITCH]
    char* const Xs = Gs + 3 * BM * PITCH;                  // [3 terms][BN][PITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (GM * GN), wmn = wave % (GM * GN);
    const int wm = wmn / GN, wn = wmn % GN;
    const int m0 = (tile / p.tiles_n) * BM;
    const int c0 = (tile % p.tiles_n) * BN;
    // operand pair of this split (uniform)
    const int pr = (split >= p.sbeg[1] ? 1 : 0) + (split >= p.sbeg[2] ? 1 : 0);
    const int pN = pr == 0 ? a.N : a.extra[pr - 1].N;
    const float* const pg = pr == 0 ? a.g : a.extra[pr - 1].g;
    const float* const px_ = pr == 0 ? a.x : a.extra[pr - 1].x;
    const long g_sN = pr == 0 ? a.g_sN : a.extra[pr - 1].g_sN, g_sC = pr == 0 ? a.g_sC : a.extra[pr - 1].g_sC;
    const long x_sN = pr == 0 ? a.x_sN : a.extra[pr - 1].x_sN, x_sC = pr == 0 ? a.x_sC : a.extra[pr - 1].x_sC;
    const int ncols = pN * a.T_out * a.V_out;
    const int L = a.T_out * a.V_out;
    const int cps = p.cps[pr];
    const int jbeg = (split - p.sbeg[pr]) * cps;
    const int jend = min(ncols, jbeg + cps);
    const int shift = (a.tap_mode == KG_TAP_TIME) ? d - (a.taps - 1) / 2 : 0;
    const int choff = (a.tap_mode == KG_TAP_CHANBLOCK) ? d * a.Cin : 0;

    const int cp = tid & (HP - 1);   // this thread's column pair inside a chunk: columns cp and cp + HP
    const int r0 = tid / HP;         // first row it stages (rows r0, r0 + RSTEP, ...)

    kg_f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int k = 0; k < WN; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    float greg[RPG][2], xreg[RPX][2];
    constexpr unsigned RANGE = 0x80000000u, OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(pg + (long)m0 * g_sC), 0, (int)RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(px_ + (long)(choff + c0) * x_sC), 0, (int)RANGE, 0x00020000);
    const int g_nvalid = (a.M - m0 - r0 + RSTEP - 1) / RSTEP;       // staged rows i < nvalid are inside the tensor
    const int x_nvalid = (a.Cin - c0 - r0 + RSTEP - 1) / RSTEP;
    const unsigned g_step = (unsigned)(RSTEP * g_sC * 4), x_step = (unsigned)(RSTEP * x_sC * 4);
    unsigned gb[2] = {OOB, OOB}, xb[2] = {OOB, OOB};
    auto prep = [&](int jc) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = jc + cp + e * HP;
            gb[e] = OOB; xb[e] = OOB;
            if (j < jend) {
                const int n = L == 1 ? j : (int)(__umulhi((unsigned)j, p.lmul) >> p.lshr), r = j - n * L;
                const int to = a.V_out == 1 ? r : (int)(__umulhi((unsigned)r, p.vmul) >> p.vshr), vo = r - to * a.V_out;
                gb[e] = (unsigned)(((long)r0 * g_sC + (long)n * g_sN + r) * 4);
                const int vi = a.vmap ? a.vmap[vo] : vo;
                const int ti = to * a.t_stride + shift;
                if (vi >= 0 && ti >= 0 && ti < a.T_in)
                    xb[e] = (unsigned)(((long)r0 * x_sC + (long)n * x_sN + (long)ti * a.V_in + vi) * 4);
            }
        }
    };
    auto load_g = [&](int i, int e) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, i < g_nvalid ? gb[e] + i * g_step : OOB, 0, 0));
    };
    auto load_x = [&](int i, int e) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, i < x_nvalid ? xb[e] + i * x_step : OOB, 0, 0));
    };
    // registers -> LDS: split on the way
    auto stash = [&]() {
        char* const qg = Gs + r0 * PITCH + cp * 4;
        char* const qx = Xs + r0 * PITCH + cp * 4;
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            unsigned h, m, l;
            bs_split_pair(greg[i][0], greg[i][1], h, m, l);
            *reinterpret_cast<unsigned*>(qg + i * RSTEP * PITCH) = h;
            *reinterpret_cast<unsigned*>(qg + i * RSTEP * PITCH + BM * PITCH) = m;
            *reinterpret_cast<unsigned*>(qg + i * RSTEP * PITCH + 2 * BM * PITCH) = l;
        }
#pragma unroll
        for (int i = 0; i < RPX; ++i) {
            unsigned h, m, l;
            bs_split_pair(xreg[i][0], xreg[i][1], h, m, l);
            *reinterpret_cast<unsigned*>(qx + i * RSTEP * PITCH) = h;
            *reinterpret_cast<unsigned*>(qx + i * RSTEP * PITCH + BN * PITCH) = m;
            *reinterpret_cast<unsigned*>(qx + i * RSTEP * PITCH + 2 * BN * PITCH) = l;
        }
    };

    if (jbeg < jend) {
        prep(jbeg);
#pragma unroll
        for (int i = 0; i < RPG; ++i) { greg[i][0] = load_g(i, 0); greg[i][1] = load_g(i, 1); }
#pragma unroll
        for (int i = 0; i < RPX; ++i) { xreg[i][0] = load_x(i, 0); xreg[i][1] = load_x(i, 1); }
        stash();
        __syncthreads();
        constexpr int KGW = PJ / 16 / GK;                    // this wave's k-groups (16 columns each) per chunk
        constexpr int NLD = 2 * (RPG + RPX);                 // loads of the next chunk, spread over the k-groups
        constexpr int LPS = (NLD + KGW - 1) / KGW;
        const char* const ga = Gs + (wm * 32 * WM + (lane & 31)) * PITCH + (16 * KGW * wk + 8 * (lane >> 5)) * 2;
        const char* const xa = Xs + (wn * 32 * WN + (lane & 31)) * PITCH + (16 * KGW * wk + 8 * (lane >> 5)) * 2;
        for (int jc = jbeg; jc < jend; jc += PJ) {
            prep(jc + PJ);
            kg_u32x4 af[2][WM][3], bf[2][WN][3];
            auto read_ab = [&](int q) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
#pragma unroll
                    for (int i = 0; i < WM; ++i)
                        af[q & 1][i][t] = *reinterpret_cast<const kg_u32x4*>(ga + t * BM * PITCH + i * 32 * PITCH + q * 32);
#pragma unroll
                    for (int k = 0; k < WN; ++k)
                        bf[q & 1][k][t] = *reinterpret_cast<const kg_u32x4*>(xa + t * BN * PITCH + k * 32 * PITCH + q * 32);
                }
            };
            read_ab(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < KGW; ++q) {
                if (q + 1 < KGW) read_ab(q + 1);
#pragma unroll
                for (int l = 0; l < LPS; ++l) {
                    const int idx = q * LPS + l;
                    if (idx < 2 * RPG) greg[idx >> 1][idx & 1] = load_g(idx >> 1, idx & 1);
                    else if (idx < NLD) xreg[(idx - 2 * RPG) >> 1][idx & 1] = load_x((idx - 2 * RPG) >> 1, idx & 1);
                }
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int k = 0; k < WN; ++k) {
                        const kg_u32x4 ah = af[q & 1][i][0], am = af[q & 1][i][1], al = af[q & 1][i][2];
                        const kg_u32x4 bh = bf[q & 1][k][0], bm = bf[q & 1][k][1], bl = bf[q & 1][k][2];
                        kg_f32x16 t = acc[i][k];              // small terms first
                        t = bs_mfma(al, bh, t);
                        t = bs_mfma(ah, bl, t);
                        t = bs_mfma(am, bm, t);
                        t = bs_mfma(am, bh, t);
                        t = bs_mfma(ah, bm, t);
                        t = bs_mfma(ah, bh, t);
                        acc[i][k] = t;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            stash();
            __syncthreads();
        }
    }

    wgrad_finish<GM, GN, GK, WM, WN>(lds, a, p, acc, m0, c0, d, split);
}

__global__ __launch_bounds__(NT) void kg_wgrad_kernel(const KgWgradArgs a, const Plan p) {
    extern __shared__ float kg_wlds[];
    wgrad_tile<2, 2, 1, 1, 1, PJ, false, RW_S>(kg_wlds, a, p, blockIdx.x, blockIdx.y, blockIdx.z);
}
__global__ __launch_bounds__(NT) void kg_wgrad_bs_kernel(const KgWgradArgs a, const Plan p) {
    extern __shared__ float kg_wlds[];
    wgrad_tile_bs<2, 2, 1, 1, 1, PJ>(kg_wlds, a, p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The weight gradients of SEVERAL layers in one launch.  A backward pass of D produces 16 of them (three convs per
// block), 19 for G; launched one by one they range from 6 to 90 us and the small ones are pure launch latency,
// while each had to split its columns ~100-fold to put enough workgroups on the chip (slab traffic: 160 MB per
// critic step).  Here the layers share one grid, every layer split just finely enough that all workgroups of the
// launch carry about the same number of column chunks.
#ifndef KG_WG_MANY_MAX
#define KG_WG_MANY_MAX 20
#endif
constexpr int MANY_MAX = KG_WG_MANY_MAX;          // jobs per launch (352 B of kernel arguments each: a backward pass of D has 16, of G 19)
struct ManyJob { KgWgradArgs a; Plan p; int wg_begin; int variant; };
// wg_begin[]: the jobs' first workgroups side by side (two lines of kernel arguments: found through job[i].wg_begin the search
// of a workgroup read one line per job, each a scalar-cache miss behind the other on a CU's first wave)
struct ManyArgs { int njobs; int wg_begin[MANY_MAX];
    ManyJob job[MANY_MAX]; };


template <bool SLAB>
__global__ __launch_bounds__(NT) void kg_wgrad_many_kernel(const ManyArgs m) {
    extern __shared__ float kg_wlds[];
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.wg_begin[ji + 1]) ++ji;       // (uniform)
    kg_kernarg_warm_at<(int)sizeof(ManyJob), (int)sizeof(ManyArgs)>((unsigned)(offsetof(ManyArgs, job) + ji * sizeof(ManyJob)));
    const ManyJob& j = m.job[ji];
    int local = blockIdx.x - j.wg_begin;
    const int tiles = j.p.tiles_m * j.p.tiles_n;
    // XCD-aware order: workgroup i of a launch runs on XCD i % 8 (each with its own L2).  The Q = tiles x taps
    // workgroups that walk the SAME column range (one split) read the same g / x chunks: they sit 8 apart in the
    // launch order - on one XCD, dispatched together - so that one of them pulls a chunk into that L2 and the others
    // hit it, instead of eight L2s fetching it once each.
    const int Q = tiles * j.a.taps, S8 = j.p.splits & ~7;
    int q, split;
    if (local < S8 * Q) {
        const int grp = local / (8 * Q), rem = local - grp * 8 * Q;
        split = grp * 8 + (rem & 7);
        q = rem >> 3;
    } else {
        const int r = j.p.splits - S8, rem = local - S8 * Q;
        split = S8 + rem % r;
        q = rem / r;
    }
    const int tile = q % tiles;
    const int d = q / tiles;
    switch (j.variant) {                                            // (uniform)
        case V_BIG:  wgrad_tile<2, 2, 1, 2, 2, 32, false, RW_B, SLAB>(kg_wlds, j.a, j.p, tile, d, split); break;
        case V_6432: wgrad_tile<2, 1, 2, 1, 1, PJ, false, RW_S, SLAB>(kg_wlds, j.a, j.p, tile, d, split); break;
        case V_3264: wgrad_tile<1, 2, 2, 1, 1, PJ, false, RW_S, SLAB>(kg_wlds, j.a, j.p, tile, d, split); break;
        case V_3232: wgrad_tile<1, 1, 4, 1, 1, PJ_3232, false, RW_S, SLAB>(kg_wlds, j.a, j.p, tile, d, split); break;
        default:     wgrad_tile<2, 2, 1, 1, 1, PJ, false, RW_S, SLAB>(kg_wlds, j.a, j.p, tile, d, split); break;
    }
}

// workgroup -> (tile, tap, split) of a job in XCD-aware order (see kg_wgrad_many_kernel)
__device__ __forceinline__ void many_locate(const ManyJob& j, int local, int& tile, int& d, int& split) {
    const int tiles = j.p.tiles_m * j.p.tiles_n;
    const int Q = tiles * j.a.taps, S8 = j.p.splits & ~7;
    int q;
    if (local < S8 * Q) {
        const int grp = local / (8 * Q), rem = local - grp * 8 * Q;
        split = grp * 8 + (rem & 7);
        q = rem >> 3;
    } else {
        const int r = j.p.splits - S8, rem = local - S8 * Q;
        split = S8 + rem % r;
        q = rem / r;
    }
    tile = q % tiles;
    d = q / tiles;
}

// the same launch on the bf16-split tiles (a kernel of its own: a second set of inlined tile variants in
// kg_wgrad_many_kernel made hipcc copy the job table to scratch, see Plan)
__global__ __launch_bounds__(NT) void kg_wgrad_many_bs_kernel(const ManyArgs m) {
    extern __shared__ float kg_wlds[];
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < m.njobs && (int)blockIdx.x >= m.wg_begin[ji + 1]) ++ji;       // (uniform)
    kg_kernarg_warm_at<(int)sizeof(ManyJob), (int)sizeof(ManyArgs)>((unsigned)(offsetof(ManyArgs, job) + ji * sizeof(ManyJob)));
    const ManyJob& j = m.job[ji];
    int tile, d, split;
    many_locate(j, blockIdx.x - j.wg_begin, tile, d, split);
    switch (j.variant) {                                            // (uniform)
        case V_BIG:  wgrad_tile_bs<2, 2, 1, 2, 2, 32>(kg_wlds, j.a, j.p, tile, d, split); break;
        case V_6432: wgrad_tile_bs<2, 1, 2, 1, 1, PJ>(kg_wlds, j.a, j.p, tile, d, split); break;
        case V_3264: wgrad_tile_bs<1, 2, 2, 1, 1, PJ>(kg_wlds, j.a, j.p, tile, d, split); break;
        case V_3232: wgrad_tile_bs<1, 1, 4, 1, 1, PJ_3232>(kg_wlds, j.a, j.p, tile, d, split); break;
        default:     wgrad_tile_bs<2, 2, 1, 1, 1, PJ>(kg_wlds, j.a, j.p, tile, d, split); break;
    }
}

__global__ __launch_bounds__(256) void kg_wgrad_reduce_kernel(const KgWgradArgs a, int splits) {
    const long per = (long)a.taps * a.M * a.Cin;
    const long i = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = kg_slab_sum_256(a.ws, per, i, i < per, splits);
    if (i >= per || threadIdx.x >= 64) return;
    const int c = (int)(i % a.Cin);
    const long q = i / a.Cin;
    const int m = (int)(q % a.M);
    const int d = (int)(q / a.M);
    float* o = a.dw + (long)d * a.w_sT + (long)m * a.w_sO + (long)c * a.w_sI;
    *o = a.accumulate ? *o + s : s;
}

// first workgroup of every job in the 1-D grid (jobs differ in size by three orders of magnitude: a (job, block) grid
// sized for the largest job launched ~4x more workgroups than it used)
struct ReduceBegin { int beg[KG_WGRAD_REDUCE_MAX_JOBS + 1]; };

// 256 outputs per workgroup and 16-byte slab loads where a job's element count and workspace allow it (round 3: with 64
// outputs and 4-byte loads the two launches of an iteration moved their 112 MB of slabs at 1.9 TB/s); the order of the
// additions per output is the scalar form's (bit-identical)
__host__ __device__ inline bool reduce_vec(const KgWgradReduceJob& j) {
    return (((long)j.taps * j.M * j.Cin) & 3) == 0 && (((unsigned long long)j.ws) & 15ull) == 0;
}

__global__ __launch_bounds__(256) void kg_wgrad_reduce_many_kernel(const KgWgradReduceJobs js, const ReduceBegin rb) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < js.njobs && (int)blockIdx.x >= rb.beg[ji + 1]) ++ji;     // (uniform)
    const KgWgradReduceJob& j = js.job[ji];
    const long per = (long)j.taps * j.M * j.Cin;
    auto store = [&](long i, float s) {
        const int c = (int)(i % j.Cin);
        const long q = i / j.Cin;
        const int m = (int)(q % j.M);
        const int d = (int)(q / j.M);
        float* o = j.dw + (long)d * j.w_sT + (long)m * j.w_sO + (long)c * j.w_sI;
        *o = j.accumulate ? *o + s : s;
    };
    if (reduce_vec(j)) {                                                     // (uniform)
        __shared__ f4 red4[4][64];
        const int o = threadIdx.x & 63, sub = threadIdx.x >> 6;
        const long i4 = (long)(blockIdx.x - rb.beg[ji]) * 256 + 4 * o;
        f4 s = {0.f, 0.f, 0.f, 0.f};
        if (i4 < per) {
#pragma unroll 8
            for (int k = sub; k < j.splits; k += 4) s += *reinterpret_cast<const f4*>(j.ws + (long)k * per + i4);
        }
        red4[sub][o] = s;
        __syncthreads();
        if (sub != 0 || i4 >= per) return;
        const f4 t = (red4[0][o] + red4[1][o]) + (red4[2][o] + red4[3][o]);
#pragma unroll
        for (int e = 0; e < 4; ++e) store(i4 + e, t[e]);
        return;
    }
    const long i = (long)(blockIdx.x - rb.beg[ji]) * 64 + (threadIdx.x & 63);
    const float s = kg_slab_sum_256(j.ws, per, i, i < per, j.splits);
    if (i >= per || threadIdx.x >= 64) return;
    store(i, s);
}

// both tile kernels take their LDS as dynamic shared memory (the variants of one launch share the allocation)
bool wgrad_lds_attr() {
    (void)hipFuncSetAttribute((const void*)kg_wgrad_many_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS_MAX);
    (void)hipFuncSetAttribute((const void*)kg_wgrad_many_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS_MAX);
    (void)hipFuncSetAttribute((const void*)kg_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds(V_6464));
    (void)hipFuncSetAttribute((const void*)kg_wgrad_many_bs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE_LDS_BS_MAX);
    (void)hipFuncSetAttribute((const void*)kg_wgrad_bs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds_bs(V_6464));
    return true;
}

int validate(const KgWgradArgs* a) {
    KG_REQUIRE(a != nullptr, "kg_wgrad: null args");
    KG_REQUIRE(a->N > 0 && a->M > 0 && a->T_out > 0 && a->V_out > 0 && a->Cin > 0 && a->T_in > 0 && a->V_in > 0,
               "kg_wgrad: bad dims");
    KG_REQUIRE((long)a->N * a->T_out * a->V_out < (1L << 31), "kg_wgrad: too many columns");
    KG_REQUIRE(a->taps == 1 || a->taps == 3, "kg_wgrad: taps=%d", a->taps);
    KG_REQUIRE(a->tap_mode == KG_TAP_TIME || a->tap_mode == KG_TAP_CHANBLOCK, "kg_wgrad: tap_mode");
    KG_REQUIRE(a->t_stride >= 1, "kg_wgrad: t_stride");
    KG_REQUIRE(a->vmap != nullptr || a->V_in == a->V_out, "kg_wgrad: V_in != V_out without vmap");
    KG_REQUIRE(a->nextra >= 0 && a->nextra <= 2, "kg_wgrad: nextra=%d", a->nextra);
    // 32-bit byte offsets inside one tile of up to 128 rows (buffer-load addressing), for every operand pair
    for (int q = 0; q <= a->nextra; ++q) {
        const int n = q == 0 ? a->N : a->extra[q - 1].N;
        const long gN = q == 0 ? a->g_sN : a->extra[q - 1].g_sN, gC = q == 0 ? a->g_sC : a->extra[q - 1].g_sC;
        const long xN = q == 0 ? a->x_sN : a->extra[q - 1].x_sN, xC = q == 0 ? a->x_sC : a->extra[q - 1].x_sC;
        KG_REQUIRE(n > 0 && (long)n * a->T_out * a->V_out < (1L << 31), "kg_wgrad: pair %d N=%d", q, n);
        const long gspan = 128L * gC + (long)(n - 1) * gN + (long)a->T_out * a->V_out;
        const long xspan = 128L * xC + (long)(n - 1) * xN + (long)a->T_in * a->V_in;
        KG_REQUIRE(gC >= 0 && gN >= 0 && xC >= 0 && xN >= 0 && gspan < (1L << 29) && xspan < (1L << 29),
                   "kg_wgrad: pair %d tensors too large for 32-bit tile offsets (%ld / %ld elements)", q, gspan, xspan);
    }
    return 0;
}

}  // namespace

extern "C" int64_t kg_wgrad_workspace_bytes(const KgWgradArgs* a) {
    if (validate(a) != 0) return -1;
    const int splits = make_plan(a).splits;
    return (int64_t)splits * a->taps * a->M * a->Cin * (int64_t)sizeof(float);
}

extern "C" int kg_wgrad(const KgWgradArgs* a, void* stream) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(a->g && a->x && a->dw && a->ws, "kg_wgrad: null pointer");
    for (int q = 0; q < a->nextra; ++q) KG_REQUIRE(a->extra[q].g && a->extra[q].x, "kg_wgrad: pair %d null pointer", q + 1);
    Plan p = make_plan(a);
    const int64_t need = (int64_t)p.splits * a->taps * a->M * a->Cin * (int64_t)sizeof(float);
    KG_REQUIRE(a->ws_bytes >= need, "kg_wgrad: workspace %ld < %ld bytes", (long)a->ws_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p.tiles_m * p.tiles_n, a->taps, p.splits);
    static const bool lds_ok = wgrad_lds_attr();
    (void)lds_ok;
    if (kg_env().wgrad_split != 0) hipLaunchKernelGGL(kg_wgrad_bs_kernel, grid, dim3(NT), tile_lds_bs(V_6464), s, *a, p);
    else                           hipLaunchKernelGGL(kg_wgrad_kernel, grid, dim3(NT), tile_lds(V_6464), s, *a, p);
    if (int rc = kg_launch_status("kg_wgrad")) return rc;
    if (a->defer_reduce) return 0;
    const long per = (long)a->taps * a->M * a->Cin;
    hipLaunchKernelGGL(kg_wgrad_reduce_kernel, dim3(kg_cdiv(per, 64)), dim3(256), 0, s, *a, p.splits);
    return kg_launch_status("kg_wgrad_reduce");
}

namespace {

// common plan of a multi-layer launch: every layer takes the tile variant that fits it and is split into workgroups
// of about the same COST (chunks x the variant's cost per chunk), ~6144 of them per pass (1024 are resident at a time;
// 2048 / 3072 with the two-buffer tiles of rounds 2-4: 512 resident)
float many_cost_target(const KgWgradArgs* jobs, int njobs) {
    double work = 0;
    for (int i = 0; i < njobs; ++i) {
        const KgWgradArgs* a = &jobs[i];
        const Tile t = TILES[tile_variant(a)];
        const long tiles = (long)kg_cdiv(a->M, t.bm) * kg_cdiv(a->Cin, t.bn) * a->taps;
        long chunks = 0;
        for (int q = 0; q <= a->nextra; ++q) chunks += kg_cdiv((long)pair_N(a, q) * a->T_out * a->V_out, t.pj);
        work += (double)tiles * chunks * t.cost;
    }
    const int budget = kg_env().wgrad_budget > 0 ? kg_env().wgrad_budget : 6144;       // KG_WGRAD_BUDGET (tuning)
    return (float)(work / (double)budget);
}

Plan many_plan(const KgWgradArgs* a, float cost_target) {
    const Tile t = TILES[tile_variant(a)];
    long per = (long)(cost_target / t.cost + 0.5f);
    const long floor_ = 256 / t.pj;                          // at least 256 columns per workgroup
    return make_plan(a, per < floor_ ? floor_ : per, t);
}

}  // namespace

extern "C" int64_t kg_wgrad_many_workspace_bytes(const KgWgradArgs* jobs, int32_t njobs) {
    if (jobs == nullptr || njobs < 1) { kg_set_error("kg_wgrad_many: no jobs"); return -1; }
    for (int i = 0; i < njobs; ++i)
        if (validate(&jobs[i]) != 0) return -1;
    const float target = many_cost_target(jobs, njobs);
    int64_t total = 0;
    for (int i = 0; i < njobs; ++i)
        total += (int64_t)many_plan(&jobs[i], target).splits * jobs[i].taps * jobs[i].M * jobs[i].Cin * (int64_t)sizeof(float);
    return total;
}

extern "C" int kg_wgrad_many(const KgWgradArgs* jobs, int32_t njobs, float* ws, int64_t ws_bytes, void* stream) {
    KG_REQUIRE(jobs != nullptr && njobs >= 1, "kg_wgrad_many: no jobs");
    for (int i = 0; i < njobs; ++i) {
        if (int rc = validate(&jobs[i])) return rc;
        KG_REQUIRE(jobs[i].g && jobs[i].x && jobs[i].dw, "kg_wgrad_many: job %d null pointer", i);
        for (int k = 0; k < i; ++k)
            KG_REQUIRE(jobs[k].dw != jobs[i].dw, "kg_wgrad_many: jobs %d and %d write the same dw", k, i);
    }
    static const bool lds_ok = wgrad_lds_attr();
    (void)lds_ok;
    const float target = many_cost_target(jobs, njobs);
    hipStream_t s = (hipStream_t)stream;
    int64_t off = 0;
    KgWgradReduceJobs rj;
    rj.njobs = 0;
    ManyArgs m;
    m.njobs = 0;
    int wgs = 0;
    size_t lds = 0;
    const bool bs = kg_env().wgrad_split != 0;
    auto flush_compute = [&]() -> int {
        if (m.njobs == 0) return 0;
        bool slab = true;                  // no job of this launch writes dw itself
        for (int i = 0; i < m.njobs; ++i) slab = slab && m.job[i].a.defer_reduce != 2;
        if (bs)        hipLaunchKernelGGL(kg_wgrad_many_bs_kernel, dim3(wgs), dim3(NT), lds, s, m);
        else if (slab) hipLaunchKernelGGL(kg_wgrad_many_kernel<true>, dim3(wgs), dim3(NT), lds, s, m);
        else           hipLaunchKernelGGL(kg_wgrad_many_kernel<false>, dim3(wgs), dim3(NT), lds, s, m);
        m.njobs = 0;
        wgs = 0;
        lds = 0;
        return kg_launch_status("kg_wgrad_many");
    };
    auto flush_reduce = [&]() -> int {
        if (rj.njobs == 0) return flush_compute();
        if (int rc = flush_compute()) return rc;               // the slabs of these jobs must have been enqueued
        const int rc = kg_wgrad_reduce_many(&rj, stream);
        rj.njobs = 0;
        return rc;
    };
    // (launch order = the caller's order; sorting the layers by workgroup cost, costliest first, measured 367 -> 379 us)
    // Workgroups are dispatched in index order as slots free up, so only the LAST ones shape the tail of the launch:
    // the layers in the first KG_WG_EARLYF of the pass's cost are cut into workgroups of KG_WG_EARLYX times the common
    // cost - half the partial slabs (and reduction traffic) for the wide layers a backward pass of D emits first.
#ifndef KG_WG_EARLYX
#define KG_WG_EARLYX 2.0f
#endif
#ifndef KG_WG_EARLYF
#define KG_WG_EARLYF 0.5f
#endif
    const float total_cost = target * (float)(kg_env().wgrad_budget > 0 ? kg_env().wgrad_budget : 6144);
    float cost_before = 0.f;
    for (int i = 0; i < njobs; ++i) {
        ManyJob& j = m.job[m.njobs];
        j.a = jobs[i];
        j.variant = tile_variant(&jobs[i]);
        const float job_target = cost_before < KG_WG_EARLYF * total_cost ? target * KG_WG_EARLYX : target;
        {
            const Tile t = TILES[j.variant];
            long chunks = 0;
            for (int q = 0; q <= jobs[i].nextra; ++q) chunks += kg_cdiv((long)pair_N(&jobs[i], q) * jobs[i].T_out * jobs[i].V_out, t.pj);
            cost_before += (float)((long)kg_cdiv(jobs[i].M, t.bm) * kg_cdiv(jobs[i].Cin, t.bn) * jobs[i].taps * chunks) * t.cost;
        }
        j.p = many_plan(&jobs[i], job_target);
        lds = std::max(lds, bs ? tile_lds_bs(j.variant) : tile_lds(j.variant));
        const int64_t bytes = (int64_t)j.p.splits * j.a.taps * j.a.M * j.a.Cin * (int64_t)sizeof(float);
        KG_REQUIRE(ws != nullptr && off + bytes <= ws_bytes, "kg_wgrad_many: workspace %ld < %ld bytes", (long)ws_bytes,
                   (long)(off + bytes));
        j.a.ws = ws + off / (int64_t)sizeof(float);
        j.a.ws_bytes = bytes;
        off += bytes;
        j.wg_begin = wgs;
        m.wg_begin[m.njobs] = wgs;
        wgs += j.p.tiles_m * j.p.tiles_n * j.a.taps * j.p.splits;
        if (j.p.splits == 1) {
            j.a.defer_reduce = 2;                              // the tile kernel writes dw itself
        } else {
            j.a.defer_reduce = 1;
            KgWgradReduceJob& r = rj.job[rj.njobs++];
            r.ws = j.a.ws; r.dw = j.a.dw;
            r.w_sT = j.a.w_sT; r.w_sO = j.a.w_sO; r.w_sI = j.a.w_sI;
            r.taps = j.a.taps; r.M = j.a.M; r.Cin = j.a.Cin; r.splits = j.p.splits; r.accumulate = j.a.accumulate;
        }
        if (++m.njobs == MANY_MAX)
            if (int rc = flush_compute()) return rc;
        if (rj.njobs == KG_WGRAD_REDUCE_MAX_JOBS)
            if (int rc = flush_reduce()) return rc;
    }
    return flush_reduce();
}


extern "C" int kg_wgrad_reduce_many(const KgWgradReduceJobs* jobs, void* stream) {
    KG_REQUIRE(jobs != nullptr, "kg_wgrad_reduce_many: null jobs");
    KG_REQUIRE(jobs->njobs >= 1 && jobs->njobs <= KG_WGRAD_REDUCE_MAX_JOBS, "kg_wgrad_reduce_many: njobs=%d", jobs->njobs);
    ReduceBegin rb;
    rb.beg[0] = 0;
    for (int i = 0; i < jobs->njobs; ++i) {
        const KgWgradReduceJob& j = jobs->job[i];
        KG_REQUIRE(j.ws && j.dw && j.taps >= 1 && j.M >= 1 && j.Cin >= 1 && j.splits >= 1,
                   "kg_wgrad_reduce_many: job %d is malformed", i);
        const long per = (long)j.taps * j.M * j.Cin;
        rb.beg[i + 1] = rb.beg[i] + kg_cdiv(per, reduce_vec(j) ? 256 : 64);
        // jobs of one launch run in different workgroups and add into dw without atomics: two jobs must never
        // share a destination (the caller reduces further contributions to one weight in a later launch)
        for (int k = 0; k < i; ++k)
            KG_REQUIRE(jobs->job[k].dw != j.dw, "kg_wgrad_reduce_many: jobs %d and %d write the same dw", k, i);
    }
    hipLaunchKernelGGL(kg_wgrad_reduce_many_kernel, dim3(rb.beg[jobs->njobs]), dim3(256), 0, (hipStream_t)stream, *jobs,
                       rb);
    return kg_launch_status("kg_wgrad_reduce_many");
}
