// kg_conv, persistent LDS-ring form ("ring"): the same tap GEMM as kg_conv.hip
//
//   out[m, j] = act( sum_g sum_d sum_c W_g(d,m,c) * X_g[c, src_g(j,d)] + bias + add ) * lrelu'(mask)
//
// for launches whose K-slices are all full (every group's Cin a multiple of 32) and that have enough (row tile, column
// tile) pairs to keep one or a few PERSISTENT workgroups per CU busy for several tiles.  Reference ops covered:
// discriminator.py:99-105,115-120,130-136 and tgcn.py:61 (the same launches as kg_conv_kernel's full-slice instantiation).
//
// Why a second kernel shape.  kg_conv_kernel gives every workgroup ONE output tile: fetch -> MFMA loop ->
// store, and all workgroups of a dispatch round run those phases at the same time - the chip first asks for every tile's
// first slices at once, then all matrix pipes run, then all tiles are written at once; 40-60 % of a launch is spent outside
// the MFMA loops (profiles/r04_conv_cu_timeline.log).  Here a workgroup WALKS a list of tiles and the three phases of
// neighbouring tiles overlap inside it:
//   * both operands of a K-slice - 32 channels x BN columns of features, BM rows x 32 channels of weights - arrive through
//     LDS-DMA (buffer_load_dword ... lds: no VGPR staging, out-of-range lanes write zeros = the conv's zero padding,
//     dropped vertices, ragged columns and rows) into a ring of NSTAGE slots, issued NSTAGE-1 slices AHEAD of the MFMAs
//     and across tile boundaries, counted s_waitcnt vmcnt(N), ONE s_barrier per slice;
//   * the per-lane global address of a DMA carries the tap's frame shift / frame stride / vertex gather (features) or the
//     XOR swizzle that makes the k-contiguous weight rows conflict-free to read back (weights): the LDS image itself is
//     lane-linear, as the DMA requires;
//   * the feature slab is shared by all row waves of the workgroup (kg_conv_kernel fetches it once per 32 output rows);
//   * a tile's stores are issued as soon as its last slice is done and are NOT waited for: the counted waits of the next
//     two slices leave them in flight (the vector-memory counter retires in order), so they drain under the next tile's
//     MFMAs;
//   * the MFMA loop of a slice is the same 16 k-steps whatever the slice's group / tap / geometry is - all addressing
//     generality lives on the DMA side.
// Tiles are dealt to the workgroups round-robin in an XCD-aware order (the row tiles of a column tile on one XCD).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "kg_common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr unsigned X_OOB = 0x80000000u;     // feature / output descriptors: 2 GiB, valid offsets below
constexpr unsigned W_OOB = 0x40000000u;     // weight descriptor: 1 GiB
constexpr int RING_MAXM = 512;              // rows whose bias sum is kept in LDS

__device__ __forceinline__ void dma_dword(unsigned lds_byte, unsigned voff, v4i rsrc_, unsigned soff_) {
    // (the descriptor and the scalar offset must be SGPRs for the assembler: under register pressure hipcc keeps uniform
    // values in VGPRs - v_readfirstlane of an SGPR value folds away, of a VGPR-held one it costs one instruction)
    v4i rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc_[0]);
    rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc_[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc_[2]);
    rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc_[3]);
    const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most n vector-memory operations of this wave may still be in flight (rounded DOWN to a multiple of 4: waiting for
// more than necessary is always safe; the counter has 6 bits)
__device__ __forceinline__ void wait_vm_upto(int n) {
    switch (n >> 2) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<4>(); break;
        case 2: wait_vm<8>(); break;
        case 3: wait_vm<12>(); break;
        case 4: wait_vm<16>(); break;
        case 5: wait_vm<20>(); break;
        case 6: wait_vm<24>(); break;
        case 7: wait_vm<28>(); break;
        case 8: wait_vm<32>(); break;
        case 9: wait_vm<36>(); break;
        case 10: wait_vm<40>(); break;
        case 11: wait_vm<44>(); break;
        case 12: wait_vm<48>(); break;
        case 13: wait_vm<52>(); break;
        case 14: wait_vm<56>(); break;
        default: wait_vm<60>(); break;
    }
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ v4i make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long u = (unsigned long long)p;
    v4i r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)u);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000;
    return r;
}

__device__ __forceinline__ void divmod_small(int x, int d, int& q, int& r) {      // exact below 2^22
    q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
    r = x - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}

struct Col { int n, to, vo; bool valid; };
__device__ __forceinline__ Col decode_col(int j, int ncols, int T_out, int V_out) {
    Col c;
    c.valid = j < ncols;
    const int jj = c.valid ? j : 0;
    int rr;
    divmod_small(jj, T_out * V_out, c.n, rr);
    divmod_small(rr, V_out, c.to, c.vo);
    return c;
}

struct RingPlan {
    int grid;            // persistent workgroups (a multiple of 8)
    int ctiles, rtiles;  // column / row tiles
    int slices;          // K-slices per tile
};

// workgroup b's i-th tile: t = b + i * grid; XCD-aware order (kg_conv.hip, kg_tile_of_block): xcd = t & 7 holds column
// tiles ct = 8 * cgrp + xcd, the row tiles of one column tile follow each other on that XCD.  A workgroup's valid tiles
// are a prefix of its list (grid is a multiple of 8: its xcd never changes, ct grows with i).
__device__ __forceinline__ bool tile_of(int t, const RingPlan& pl, int& ct, int& rt) {
    const int xcd = t & 7, slot = t >> 3;
    const int cgrp = slot / pl.rtiles;
    rt = slot - cgrp * pl.rtiles;
    ct = cgrp * 8 + xcd;
    return ct < pl.ctiles;
}

// RW x CW waves; a wave owns TMW x TNW MFMA tiles of 32 x 32: BM = 32 RW TMW rows, BN = 32 CW TNW columns per tile.
// KF: weights k-contiguous in memory (forward layouts: LDS image [m][32 k], XOR-swizzled) or m-contiguous (transposed:
// image [k][BM]).  NSTAGE ring slots, NSTAGE-1 slices of look-ahead.
template <int RW, int CW, int TMW, int TNW, bool KF, int NSTAGE, int MINW>
__global__ __launch_bounds__(64 * RW * CW, MINW) void kg_conv_ring_kernel(const KgConvArgs a, const RingPlan pl) {
    constexpr int NW = RW * CW;
    constexpr int BM = 32 * RW * TMW, BN = 32 * CW * TNW;
    static_assert(BN % 64 == 0, "a DMA moves 64 columns");
    constexpr int NCS = BN / 64;                 // column slots per lane on the DMA side
    constexpr int XD = 32 * NCS / NW;            // feature DMAs per wave and slice
    constexpr int WD = BM / 2 / NW;              // weight DMAs per wave and slice
    static_assert((32 * NCS) % NW == 0 && (BM / 2) % NW == 0 && XD >= 1 && WD >= 1, "DMA split");
    constexpr int DPS = XD + WD;
    constexpr int PER = (DPS + 15) / 16;         // DMAs issued behind one k-step
    constexpr int LA = NSTAGE - 1;
    constexpr int STAGE_F = 32 * BN + 32 * BM;   // floats per ring slot
    constexpr int NACC = TMW * TNW;
    constexpr int NPF = 3;                       // LDS operand reads this many k-steps ahead of their MFMAs
    extern __shared__ float kg_ring_lds[];       // [NSTAGE][STAGE_F] | bias[RING_MAXM] | vmap[2][64]
    float* const Bl = kg_ring_lds + NSTAGE * STAGE_F;
    int* const Vm = (int*)(Bl + RING_MAXM);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, l32 = lane & 31;
    const int rw = wave % RW, cw = wave / RW;
    const int ncols = a.N * a.T_out * a.V_out;
    const int G = pl.grid, b = blockIdx.x;
    const int S = pl.slices;

    // ---- tiles of this workgroup
    int my_tiles = 0;
    {
        const int per = (pl.ctiles + 7) / 8 * 8 * pl.rtiles;      // padded tile count
        for (int t = b; t < per; t += G) {
            int ct, rt;
            if (!tile_of(t, pl, ct, rt)) break;
            ++my_tiles;
        }
    }
    if (my_tiles == 0) return;
    const int total = my_tiles * S;

    // ---- once per launch: bias sums and vertex maps -> LDS
    for (int m = tid; m < a.M; m += 64 * NW) Bl[m] = (a.bias0 ? a.bias0[m] : 0.f) + (a.bias1 ? a.bias1[m] : 0.f);
    if (tid < 128) {
        const int gi = tid >> 6, v = tid & 63;
        const int32_t* vm = gi < a.ngroups ? a.g[gi].vmap : nullptr;
        Vm[tid] = (vm && v < a.V_out) ? vm[v] : v;
    }
    wg_barrier();       // the DMA side reads the vertex maps in its very first tile setup, before the slice loop's barriers

    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(void*)kg_ring_lds);

    // =============================== DMA side ===============================
    unsigned xoff[2][3][NCS];       // byte offset of this lane's column(s) for group g, tap d (or X_OOB)
    unsigned woff[2][WD];           // byte offset of this lane's weight element(s) inside a (tap, slice) block (or W_OOB)
    int d_it = 0;                   // tile counter of the DMA side
    int d_gi = 0, d_cch = 0, d_d = 0;
    bool d_fresh = true;            // the next slice is the first of a tile: lane offsets are due

    auto dma_tile_setup = [&](int it) {
        int ct, rt;
        tile_of(b + it * G, pl, ct, rt);
        const int m0 = rt * BM;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            if (gq < a.ngroups) {       // (uniform)
                const KgConvGroup& g = a.g[gq];
                const int tstep = g.tap_mode == KG_TAP_TIME ? 1 : 0;
                const int pad = tstep ? (g.taps - 1) / 2 : 0;
#pragma unroll
                for (int p = 0; p < NCS; ++p) {
                    const Col xc = decode_col(ct * BN + p * 64 + lane, ncols, a.T_out, a.V_out);
                    const int vi = Vm[gq * 64 + (xc.vo & 63)];
                    const bool okv = xc.valid && vi >= 0;
                    const unsigned base = (unsigned)xc.n * (unsigned)g.x_sN + (unsigned)vi;
                    if (!g.transposed) {
                        const int t0 = xc.to * g.t_stride - pad;
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const int ti = t0 + tstep * d;
                            const bool ok = okv && d < g.taps && (unsigned)ti < (unsigned)g.T_in;
                            xoff[gq][d][p] = ok ? (base + (unsigned)(ti * g.V_in)) * 4u : X_OOB;
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const int num = xc.to + pad - tstep * d;
                            int ti, rem;
                            divmod_small(num < 0 ? 0 : num, g.t_stride, ti, rem);
                            const bool ok = okv && d < g.taps && num >= 0 && rem == 0 && ti < g.T_in;
                            xoff[gq][d][p] = ok ? (base + (unsigned)(ti * g.V_in)) * 4u : X_OOB;
                        }
                    }
                }
                const bool rowblocks = g.w_MB < a.M;        // (uniform)
#pragma unroll
                for (int j = 0; j < WD; ++j) {
                    const int f = (wave * WD + j) * 64 + lane;
                    int m, k;
                    if constexpr (KF) {
                        m = f >> 5;
                        k = (f & 31) ^ (m & 31);
                    } else {
                        k = f / BM;
                        m = f % BM;
                    }
                    const int mm = m0 + m;
                    unsigned off = (unsigned)mm * (unsigned)g.w_sO;
                    if (rowblocks) {
                        int mb, mr;
                        divmod_small(mm, g.w_MB, mb, mr);
                        off = (unsigned)mb * (unsigned)g.w_sMB + (unsigned)mr * (unsigned)g.w_sO;
                    }
                    woff[gq][j] = mm < a.M ? (off + (unsigned)k * (unsigned)g.w_sI) * 4u : W_OOB;
                }
            }
        }
    };

    // per-group uniform state, read from the kernel arguments ONCE.  One buffer descriptor per operand and group for the
    // whole launch: the walk through taps / channel chunks goes through the DMA's SCALAR offset (not range-checked for
    // raw buffers; the host has checked that every operand spans less than 4 GiB).
    struct GU { unsigned xs4, wsT4, wsi4; int cch, taps, chanblock; };
    GU gu[2];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
        const KgConvGroup& g = a.g[gq < a.ngroups ? gq : 0];
        gu[gq].xs4 = (unsigned)g.x_sC * 4u; gu[gq].wsT4 = (unsigned)g.w_sT * 4u; gu[gq].wsi4 = (unsigned)g.w_sI * 4u;
        gu[gq].cch = g.Cin / 32; gu[gq].taps = g.taps; gu[gq].chanblock = g.tap_mode == KG_TAP_CHANBLOCK ? g.Cin : 0;
    }
    const v4i xr0 = make_rsrc(a.g[0].x, X_OOB), wr0 = make_rsrc(a.g[0].w, W_OOB);
    const v4i xr1 = make_rsrc(a.g[a.ngroups > 1 ? 1 : 0].x, X_OOB), wr1 = make_rsrc(a.g[a.ngroups > 1 ? 1 : 0].w, W_OOB);
    auto sel_rsrc = [](bool g1, const v4i& r0, const v4i& r1) {
        v4i r;
        r[0] = g1 ? r1[0] : r0[0];
        r[1] = g1 ? r1[1] : r0[1];
        r[2] = r0[2];
        r[3] = r0[3];
        return r;
    };
    const int ngroups = a.ngroups;

    struct Prep {
        bool g1;                // which group's descriptors
        unsigned xs0;           // scalar byte offset of the slice's first channel row
        unsigned xs4;           // byte stride between channels
        unsigned ws0;           // scalar byte offset of the slice's (tap, first channel) in the weight rows
        unsigned slot;          // LDS byte address of the ring slot
        unsigned xcur[NCS];
        unsigned wcur[WD];
    };
    // resolve slice number `gs` of this workgroup's stream and advance the iterator.  Slices beyond the last one are
    // "dead": every lane offset out of range (the DMAs write zeros into a slot nobody reads any more), so that every
    // iteration issues the same number of vector-memory operations and the counted waits stay constants.
    auto prep = [&](Prep& c, int gs) {
        c.slot = lds0 + (unsigned)(gs % NSTAGE) * (unsigned)(STAGE_F * 4);
        if (gs >= total) {          // (uniform)
            c.g1 = false; c.xs0 = 0; c.xs4 = 0; c.ws0 = 0;
#pragma unroll
            for (int p = 0; p < NCS; ++p) c.xcur[p] = X_OOB;
#pragma unroll
            for (int j = 0; j < WD; ++j) c.wcur[j] = W_OOB;
            return;
        }
        if (d_fresh) {
            dma_tile_setup(d_it);
            d_fresh = false;
        }
        const bool g1 = d_gi != 0;
        const unsigned xs4 = g1 ? gu[1].xs4 : gu[0].xs4;
        const unsigned wsT4 = g1 ? gu[1].wsT4 : gu[0].wsT4;
        const unsigned wsi4 = g1 ? gu[1].wsi4 : gu[0].wsi4;
        const int cchn = g1 ? gu[1].cch : gu[0].cch;
        const int tapsn = g1 ? gu[1].taps : gu[0].taps;
        const int chanblock = g1 ? gu[1].chanblock : gu[0].chanblock;
        const int c0 = d_cch * 32;
        c.g1 = g1;
        c.xs4 = xs4;
        c.xs0 = (unsigned)(d_d * chanblock + c0) * xs4;
        c.ws0 = (unsigned)d_d * wsT4 + (unsigned)c0 * wsi4;
#pragma unroll
        for (int p = 0; p < NCS; ++p) {
            const unsigned o0 = g1 ? xoff[1][0][p] : xoff[0][0][p];
            const unsigned o1 = g1 ? xoff[1][1][p] : xoff[0][1][p];
            const unsigned o2 = g1 ? xoff[1][2][p] : xoff[0][2][p];
            c.xcur[p] = d_d == 0 ? o0 : (d_d == 1 ? o1 : o2);
        }
#pragma unroll
        for (int j = 0; j < WD; ++j) c.wcur[j] = g1 ? woff[1][j] : woff[0][j];
        // advance: taps of a channel chunk follow each other, then the next chunk, then the next group, then the next tile
        if (++d_d == tapsn) {
            d_d = 0;
            if (++d_cch == cchn) {
                d_cch = 0;
                if (++d_gi == ngroups) {
                    d_gi = 0;
                    ++d_it;
                    d_fresh = true;
                }
            }
        }
    };
    // DMA number e (compile-time) of a prepared slice: the feature pieces first, then the weight pieces
    auto dma_one = [&](const Prep& c, int e) {
        if (e < XD) {
            const int row = wave * (32 / NW) + e / NCS, piece = e % NCS;     // (piece: compile-time index)
            dma_dword(c.slot + (unsigned)(row * BN + piece * 64) * 4u, c.xcur[piece], sel_rsrc(c.g1, xr0, xr1), c.xs0 + (unsigned)row * c.xs4);
        } else {
            const int j = e - XD;
            const int I = wave * WD + j;
            dma_dword(c.slot + (unsigned)(32 * BN + I * 64) * 4u, c.wcur[j], sel_rsrc(c.g1, wr0, wr1), c.ws0);
        }
    };

    // =============================== MFMA side ===============================
    kg_f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int mrow = rw * 32 * TMW + l32;              // this lane's A row inside the tile (first MFMA row block)
    const int ccol = cw * 32 * TNW + l32;              // this lane's B column inside the tile (first column block)
    const int aswz = kh ^ l32;

    // epilogue of the tile (ct, rt): bias, add, activation, mask; every lane issues ALL its stores (absent rows / columns
    // through out-of-range offsets) so that the number of vector-memory operations per wave is a constant
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.out), 0, (int)X_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.add), 0, a.add ? (int)X_OOB : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.mask), 0, a.mask ? (int)X_OOB : 0, 0x00020000);
    const int ots = a.o_tstride > 1 ? a.o_tstride : 1;
    const bool has_add = a.add != nullptr, has_mask = a.mask != nullptr;      // (uniform)
    auto epilogue = [&](int it) {
        int ct, rt;
        tile_of(b + it * G, pl, ct, rt);
        const int mbase = rt * BM + rw * 32 * TMW + 4 * kh;       // row of register r = 0 in row block 0
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) {
            const Col xc = decode_col(ct * BN + cw * 32 * TNW + tn * 32 + l32, ncols, a.T_out, a.V_out);
            const unsigned ocol = xc.valid ? ((unsigned)xc.n * (unsigned)a.o_sN + (unsigned)(xc.to * ots * a.V_out + xc.vo)) * 4u : X_OOB;
            const unsigned acol = ((unsigned)xc.n * (unsigned)a.a_sN + (unsigned)(xc.to * a.a_tstride * a.V_out + xc.vo)) * 4u;
            const unsigned mcol = ((unsigned)xc.n * (unsigned)a.m_sN + (unsigned)(xc.to * a.V_out + xc.vo)) * 4u;
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) {
                const kg_f32x16& av = acc[tm * TNW + tn];
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                    v[r] = av[r] + Bl[row < a.M ? row : 0];
                }
                if (has_add) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = mbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                        const unsigned off = (xc.valid && row < a.M) ? acol + (unsigned)row * (unsigned)a.a_sC * 4u : X_OOB;
                        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_add, off, 0, 0));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += rv[r];
                }
                float mv[16];
                if (has_mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = mbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                        const unsigned off = (xc.valid && row < a.M) ? mcol + (unsigned)row * (unsigned)a.m_sC * 4u : X_OOB;
                        mv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_msk, off, 0, 0));
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                    float o = kg_act(v[r], a.act, a.slope);
                    if (has_mask) o *= mv[r] > 0.f ? 1.f : a.slope;
                    const unsigned off = row < a.M ? ocol + (unsigned)row * (unsigned)a.o_sC * 4u : X_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), r_out,
                                                          ocol == X_OOB ? X_OOB : off, 0, 0);
                }
            }
        }
    };
    // vector-memory operations an epilogue issues per wave (constant: see above)
    const int NST = 16 * NACC * (1 + (has_add ? 1 : 0) + (has_mask ? 1 : 0));

    // ---- prologue: the first LA slices at once, the next one prepared
    Prep cn;
    {
#pragma unroll 1
        for (int gs = 0; gs < LA; ++gs) {
            prep(cn, gs);
#pragma unroll
            for (int e = 0; e < DPS; ++e) dma_one(cn, e);
        }
        prep(cn, LA);
    }

    int s_in_tile = 0, c_it = 0;
    unsigned ephist = 0;            // bit i: an epilogue ran at the end of iteration g - 1 - i
    constexpr int PQ = (DPS + PER - 1) / PER <= 12 ? 12 : 14;     // k-step behind which the NEXT iteration's slice is prepared
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
        // slice g has landed when at most the DMAs of the LA - 1 younger slices (and the stores / loads of epilogues issued
        // after its DMAs: those of the last LA iterations) are outstanding
        wait_vm_upto(DPS * (LA - 1) + NST * __builtin_popcount(ephist & ((1u << LA) - 1u)));
        wg_barrier();
        const Prep c = cn;
        const float* xs = kg_ring_lds + (g % NSTAGE) * STAGE_F;
        const float* ws = xs + 32 * BN;
        const float* bp = xs + kh * BN + ccol;
        auto lda = [&](int q, int tm) -> float {
            if constexpr (KF) return ws[(mrow + tm * 32) * 32 + ((2 * q) ^ aswz)];
            else return ws[(2 * q + kh) * BM + mrow + tm * 32];
        };
        auto ldb = [&](int q, int tn) -> float { return bp[2 * q * BN + tn * 32]; };
        float av[NPF][TMW], bv[NPF][TNW];
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) av[q][tm] = lda(q, tm);
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) bv[q][tn] = ldb(q, tn);
        }
        auto kstep = [&](int q) {
            float a_[TMW], b_[TNW];
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) a_[tm] = av[q % NPF][tm];
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) b_[tn] = bv[q % NPF][tn];
            if (q + NPF < 16) {
#pragma unroll
                for (int tm = 0; tm < TMW; ++tm) av[q % NPF][tm] = lda(q + NPF, tm);
#pragma unroll
                for (int tn = 0; tn < TNW; ++tn) bv[q % NPF][tn] = ldb(q + NPF, tn);
            }
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNW; ++tn)
                    acc[tm * TNW + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[tm], b_[tn], acc[tm * TNW + tn], 0, 0, 0);
            // the DMAs of slice g + LA ride behind the first k-steps
#pragma unroll
            for (int e = q * PER; e < (q + 1) * PER; ++e)
                if (e < DPS) dma_one(c, e);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int q = 0; q <= PQ; ++q) kstep(q);
        // the address work of slice g + LA + 1 behind a late k-step (under the MFMAs in flight)
        prep(cn, g + 1 + LA);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = PQ + 1; q < 16; ++q) kstep(q);
        ephist <<= 1;
        if (++s_in_tile == S) {
            s_in_tile = 0;
            epilogue(c_it);
            ++c_it;
            ephist |= 1u;
#pragma unroll
            for (int i = 0; i < NACC; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
    }
}

// =====================================================================================================================
// Ring v2 ("window" form).  Measured on ring v1 (profiles/r05_ring_v1_ablations.log) and on an experiment that made the
// MFMAs of kg_conv_kernel 2.7x cheaper without making it faster (profiles/r05_split_bf16_experiment.log): what these
// contractions run into is the CU's vector-memory INSTRUCTION rate - about one 64-lane instruction per 11 cycles, whatever
// its width - and a 4-byte-per-lane load or DMA moves only 256 bytes.  kg_conv_kernel's 32-row tile issues 1.5 such
// instructions per MFMA and 4 SIMDs x 1.5 / 64 cycles is exactly that rate.  So here every operand moves as 16 bytes per lane:
//   * features: the columns of a tile and ALL taps of a K-slice read ONE contiguous window of the source rows (a temporal
//     tap is a shift by V positions, a frame stride makes the window twice as long).  The window of 32 channels arrives
//     as buffer_load_dwordx4 ... lds (1 KiB per instruction, 16-byte aligned start); the tap shift, the frame stride and
//     the conv's zero padding are per-lane LDS read addresses (an invalid (column, tap) reads the zero chunk at the end of
//     the row).  Groups that gather vertices (the residual branch of a down-sampling block) keep v1's per-lane 4-byte DMA.
//   * weights: the (row tile x 32 channels x taps) block of a K-slice as 16-byte DMA in the layout the weight tensor has
//     in memory - [m][c][tap] (temporal conv), [m][c] (1x1 convs), [c][m] / [c][m][tap] (their transposes) - read back
//     as 16-byte fragments (k-contiguous layouts, XOR-swizzled on the DMA's source side) or 4-byte ones (m-contiguous).
//     The k -> MFMA-step assignment follows the layout (any order is right as long as both operands use the same one).
// A K-slice = 32 channels x all taps: 48 MFMAs per accumulator and barrier for the temporal convs.
// =====================================================================================================================
enum { WL_TAPROW = 0, WL_ROW = 1, WL_COL = 2, WL_COLTAP = 3 };

struct RingWGroup {
    int xmode;          // 0: window (16-byte DMA), 1: gather (4-byte DMA per column; one tap)
    int wl;             // weight layout (WL_*)
    int taps;           // MFMA taps of a slice: 3 for temporal convs, 1 otherwise (channel-block taps are more slices)
    int nslice;         // slices of the group
    int cpb;            // slices per channel block (KG_TAP_CHANBLOCK) or nslice
    int rowlen;         // floats per source channel row that may be read (N * T_in * V_in)
    int need4;          // 16-byte pieces of a window row that can hold operands (the rest of the row is not fetched)
};
struct RingWPlan {
    int grid, ctiles, rtiles, slices;
    int stagger;        // s_sleep units (64 cycles) the workgroups of the grid's second half wait before their first tile
    RingWGroup g[2];
};

__device__ __forceinline__ void dma_x4(unsigned lds_byte, unsigned voff, v4i rsrc_, unsigned soff_) {
    v4i rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc_[0]);
    rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc_[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc_[2]);
    rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc_[3]);
    const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory", "m0");
}

typedef float kg_f32x4 __attribute__((ext_vector_type(4)));

template <int RW, int CW, int TMW, int TNW, int PWS, int NSTAGE, int MINW>
__global__ __launch_bounds__(64 * RW * CW, MINW) void kg_conv_ringw_kernel(const KgConvArgs a, const RingWPlan pl) {
    constexpr int NW = RW * CW;
    constexpr int BM = 32 * RW * TMW, BN = 32 * CW * TNW;
    static_assert(BN % 64 == 0 && PWS % 4 == 0 && PWS >= BN, "window");
    constexpr int NCS = BN / 64;
    constexpr int PW4 = PWS / 4;                                 // 16-byte chunks per window row (the last one stays zero)
    constexpr int XW = (32 * PW4 + 64 * NW - 1) / (64 * NW);     // window DMAs per wave and slice
    constexpr int XG = 32 * NCS / NW;                            // gather DMAs per wave and slice
    constexpr int W3 = (BM * 24 + 64 * NW - 1) / (64 * NW);      // weight DMAs per wave and slice, three taps
    constexpr int W1 = (BM * 8 + 64 * NW - 1) / (64 * NW);       // one tap
    static_assert((32 * NCS) % NW == 0, "gather split");
    constexpr int XMAX = XW > XG ? XW : XG;
    constexpr int LA = NSTAGE - 1;
    constexpr int XS_F = (NW * XW * 256 > 32 * PWS) ? NW * XW * 256 : 32 * PWS;       // floats of the feature slab (whole DMAs)
    constexpr int WS_F = (NW * W3 * 256 > 96 * BM) ? NW * W3 * 256 : 96 * BM;         // weight slab (three taps; whole DMAs)
    constexpr int STAGE_F = XS_F + WS_F;
    constexpr int NACC = TMW * TNW;
    extern __shared__ float kg_ring_lds[];                       // [NSTAGE][STAGE_F] | bias[RING_MAXM] | vmap[2][64]
    float* const Bl = kg_ring_lds + NSTAGE * STAGE_F;
    int* const Vm = (int*)(Bl + RING_MAXM);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, l32 = lane & 31;
    const int rw = wave % RW, cw = wave / RW;
    const int ncols = a.N * a.T_out * a.V_out;
    const int G = pl.grid, b = blockIdx.x;
    const int S = pl.slices;

    int my_tiles = 0;
    {
        const int per = (pl.ctiles + 7) / 8 * 8 * pl.rtiles;
        RingPlan tp; tp.ctiles = pl.ctiles; tp.rtiles = pl.rtiles;
        for (int t = b; t < per; t += G) {
            int ct, rt;
            if (!tile_of(t, tp, ct, rt)) break;
            ++my_tiles;
        }
    }
    if (my_tiles == 0) return;
    const int total = my_tiles * S;
    RingPlan tp;
    tp.ctiles = pl.ctiles; tp.rtiles = pl.rtiles; tp.grid = G; tp.slices = S;
    if (pl.stagger > 0 && b >= G / 2) {         // (uniform) the CU's second workgroup runs half a tile behind the first
        for (int i = 0; i < pl.stagger; i += 16) __builtin_amdgcn_s_sleep(16);
    }

    for (int m = tid; m < a.M; m += 64 * NW) Bl[m] = (a.bias0 ? a.bias0[m] : 0.f) + (a.bias1 ? a.bias1[m] : 0.f);
    if (tid < 128) {
        const int gi = tid >> 6, v = tid & 63;
        const int32_t* vm = gi < a.ngroups ? a.g[gi].vmap : nullptr;
        Vm[tid] = (vm && v < a.V_out) ? vm[v] : v;
    }
    wg_barrier();       // the DMA side reads the vertex maps in its very first tile setup, before the slice loop's barriers
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(void*)kg_ring_lds);

    // ---- per-group uniform state
    struct GU { unsigned xs4, wsT4, wsi4; int xmode, wl, taps, nslice, cpb, rowlen, need4, chanblock; int xsN, Vin, Tin, ts, tr, pad; };
    GU gu[2];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
        const int gs = gq < a.ngroups ? gq : 0;
        const KgConvGroup& g = a.g[gs];
        const RingWGroup& pg = pl.g[gs];
        gu[gq].xs4 = (unsigned)g.x_sC * 4u; gu[gq].wsT4 = (unsigned)g.w_sT * 4u; gu[gq].wsi4 = (unsigned)g.w_sI * 4u;
        gu[gq].xmode = pg.xmode; gu[gq].wl = pg.wl; gu[gq].taps = pg.taps; gu[gq].nslice = pg.nslice; gu[gq].cpb = pg.cpb;
        gu[gq].rowlen = pg.rowlen; gu[gq].need4 = pg.need4; gu[gq].chanblock = g.tap_mode == KG_TAP_CHANBLOCK ? g.Cin : 0;
        gu[gq].xsN = (int)g.x_sN; gu[gq].Vin = g.V_in; gu[gq].Tin = g.T_in; gu[gq].ts = g.t_stride; gu[gq].tr = g.transposed;
        gu[gq].pad = (g.tap_mode == KG_TAP_TIME) ? (g.taps - 1) / 2 : 0;
    }
    const v4i xr0 = make_rsrc(a.g[0].x, X_OOB), wr0 = make_rsrc(a.g[0].w, W_OOB);
    const v4i xr1 = make_rsrc(a.g[a.ngroups > 1 ? 1 : 0].x, X_OOB), wr1 = make_rsrc(a.g[a.ngroups > 1 ? 1 : 0].w, W_OOB);
    auto sel_rsrc = [](bool g1, const v4i& r0, const v4i& r1) {
        v4i r;
        r[0] = g1 ? r1[0] : r0[0];
        r[1] = g1 ? r1[1] : r0[1];
        r[2] = r0[2];
        r[3] = r0[3];
        return r;
    };
    const int ngroups = a.ngroups;

    // first source position (floats from the channel row's start, multiple of 4, may be negative) of the window the column
    // tile ct reads in group gq: the position of its first column under the earliest tap
    auto window_lo = [&](int ct, int gq) __attribute__((always_inline)) -> int {
        const GU& u = gu[gq];
        const Col c0 = decode_col(ct * BN, ncols, a.T_out, a.V_out);
        const int lo = c0.n * u.xsN + (c0.to * u.ts - u.pad) * u.Vin + c0.vo;
        return lo & ~3;
    };

    // =============================== DMA side ===============================
    unsigned xg_off[NCS];           // gather group: byte offset of this lane's column(s) (or X_OOB)
    unsigned wv_off[2][W3];         // weights: byte offset of this lane's 16-byte piece(s) inside a slice's block (or W_OOB)
    int d_win[2];                   // window start of the DMA side's tile per group (uniform)
    int d_it = 0, d_gi = 0, d_sl = 0;
    bool d_fresh = true;

    // the weight offsets of row tile 0; where every row tile is full and the rows are not blocked, row tile rt's are these
    // plus rt * BM rows (w_fast: uniform)
    unsigned wv0[2][W3];
    unsigned wrow4[2];              // bytes one output row advances a weight offset
    bool w_fast = true;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
        const KgConvGroup& g = a.g[gq < a.ngroups ? gq : 0];
        const int wl = pl.g[gq < a.ngroups ? gq : 0].wl;
        wrow4[gq] = wl == WL_COL ? 4u : (wl == WL_COLTAP ? 12u : (unsigned)g.w_sO * 4u);
        if (gq < a.ngroups && (g.w_MB < a.M || a.M % BM != 0)) w_fast = false;
    }
    auto dma_tile_setup = [&](int it, bool first) __attribute__((always_inline)) {
        int ct, rt;
        tile_of(b + it * G, tp, ct, rt);
        const int m0 = (first && w_fast) ? 0 : rt * BM;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            if (gq < ngroups) {     // (uniform)
                const KgConvGroup& g = a.g[gq];
                const GU& u = gu[gq];
                d_win[gq] = 0;
                if (u.xmode == 0) {
                    d_win[gq] = __builtin_amdgcn_readfirstlane(window_lo(ct, gq));
                } else {
#pragma unroll
                    for (int p = 0; p < NCS; ++p) {
                        const Col xc = decode_col(ct * BN + p * 64 + lane, ncols, a.T_out, a.V_out);
                        const int vi = Vm[gq * 64 + (xc.vo & 63)];
                        const int ti = xc.to * u.ts;
                        const bool ok = xc.valid && vi >= 0 && ti < u.Tin;
                        xg_off[p] = ok ? ((unsigned)xc.n * (unsigned)u.xsN + (unsigned)(ti * u.Vin + vi)) * 4u : X_OOB;
                    }
                }
                if (w_fast && !first) {                      // (uniform) the row tile only shifts the offsets
#pragma unroll
                    for (int j = 0; j < W3; ++j) wv_off[gq][j] = wv0[gq][j] == W_OOB ? W_OOB : wv0[gq][j] + (unsigned)(rt * BM) * wrow4[gq];
                    continue;
                }
                const bool rowblocks = g.w_MB < a.M;        // (uniform)
                auto woffm = [&](int mm) -> unsigned {
                    unsigned off = (unsigned)mm * (unsigned)g.w_sO;
                    if (rowblocks) {
                        int mb, mr;
                        divmod_small(mm, g.w_MB, mb, mr);
                        off = (unsigned)mb * (unsigned)g.w_sMB + (unsigned)mr * (unsigned)g.w_sO;
                    }
                    return off;
                };
                const int wper = (u.wl == WL_TAPROW || u.wl == WL_COLTAP) ? W3 : W1;       // DMAs per wave: as dma_one numbers them
#pragma unroll
                for (int j = 0; j < W3; ++j) {
                    const int f = (wave * wper + j) * 64 + lane;    // 16-byte piece number inside the slice's weight block
                    unsigned off = W_OOB;
                    if (u.wl == WL_TAPROW) {                        // [BM][96]: 24 pieces per row, swizzled in blocks of 8
                        const int m = f / 24, pc = f - m * 24;
                        const int i = (pc & ~7) | ((pc & 7) ^ ((m >> 1) & 7));
                        if (m < BM && m0 + m < a.M) off = (woffm(m0 + m) + 4u * (unsigned)i) * 4u;
                    } else if (u.wl == WL_ROW) {                    // [BM][32]: 8 pieces per row
                        const int m = f >> 3, pc = f & 7;
                        const int i = pc ^ ((m >> 1) & 7);
                        if (j < wper && m < BM && m0 + m < a.M) off = (woffm(m0 + m) + 4u * (unsigned)i) * 4u;
                    } else if (u.wl == WL_COL) {                    // [32][BM]: BM / 4 pieces per channel
                        const int k = f / (BM / 4), mc = f - k * (BM / 4);
                        if (j < wper && k < 32 && m0 + 4 * mc < a.M) off = ((unsigned)k * (unsigned)g.w_sI + woffm(m0 + 4 * mc)) * 4u;
                    } else {                                        // WL_COLTAP [32][3 BM]: 3 BM / 4 pieces per channel
                        const int k = f / (3 * BM / 4), pc = f - k * (3 * BM / 4);
                        if (k < 32 && 3 * m0 + 4 * pc < 3 * a.M) off = ((unsigned)k * (unsigned)g.w_sI + 3u * (unsigned)m0 + 4u * (unsigned)pc) * 4u;
                    }
                    wv_off[gq][j] = off;
                    if (first) wv0[gq][j] = off;
                }
            }
        }
        if (first && w_fast) {          // (the offsets above were row tile 0's)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int j = 0; j < W3; ++j)
                    if (gq < ngroups) wv_off[gq][j] = wv0[gq][j] == W_OOB ? W_OOB : wv0[gq][j] + (unsigned)(rt * BM) * wrow4[gq];
        }
    };

    struct Prep {
        bool g1, gather;
        int nx, nw;             // DMAs of this slice per wave: features, weights
        unsigned xs0, xs4, ws0, slot;
        int win, rowlen, need4;
        unsigned wcur[W3];
    };
    auto prep = [&](Prep& c, int gs) __attribute__((always_inline)) {
        c.slot = lds0 + (unsigned)(gs % NSTAGE) * (unsigned)(STAGE_F * 4);
        if (gs >= total) {          // (uniform) dead slice: same DMA counts as a window / three-tap one, nothing fetched
            c.g1 = false; c.gather = false; c.nx = XW; c.nw = W3; c.xs0 = 0; c.xs4 = 0; c.ws0 = 0; c.win = 0; c.rowlen = 0; c.need4 = 0;
#pragma unroll
            for (int j = 0; j < W3; ++j) c.wcur[j] = W_OOB;
            return;
        }
        if (d_fresh) {
            dma_tile_setup(d_it, d_it == 0);
            d_fresh = false;
        }
        const bool g1 = d_gi != 0;
        const unsigned xs4 = g1 ? gu[1].xs4 : gu[0].xs4;
        const unsigned wsT4 = g1 ? gu[1].wsT4 : gu[0].wsT4;
        const unsigned wsi4 = g1 ? gu[1].wsi4 : gu[0].wsi4;
        const int cpb = g1 ? gu[1].cpb : gu[0].cpb;
        const int nsl = g1 ? gu[1].nslice : gu[0].nslice;
        const int chanblock = g1 ? gu[1].chanblock : gu[0].chanblock;
        const int taps = g1 ? gu[1].taps : gu[0].taps;
        // slice d_sl of the group = (channel block blk, 32-channel chunk cc inside it)
        const int blk = d_sl / cpb, cc = d_sl - blk * cpb;
        c.g1 = g1;
        c.gather = (g1 ? gu[1].xmode : gu[0].xmode) != 0;
        c.nx = c.gather ? XG : XW;
        c.nw = taps == 3 ? W3 : W1;
        c.xs4 = xs4;
        c.xs0 = (unsigned)(blk * chanblock + cc * 32) * xs4;
        c.ws0 = (unsigned)blk * wsT4 + (unsigned)(cc * 32) * wsi4;
        c.win = g1 ? d_win[1] : d_win[0];
        c.rowlen = g1 ? gu[1].rowlen : gu[0].rowlen;
        c.need4 = g1 ? gu[1].need4 : gu[0].need4;
#pragma unroll
        for (int j = 0; j < W3; ++j) c.wcur[j] = g1 ? wv_off[1][j] : wv_off[0][j];
        if (++d_sl == nsl) {
            d_sl = 0;
            if (++d_gi == ngroups) {
                d_gi = 0;
                ++d_it;
                d_fresh = true;
            }
        }
    };
    // DMA number e of a prepared slice (e compile-time: every register array index below is a constant): the weight pieces
    // e < W3 (issued while e < nw), then the feature pieces e - W3 (while < nx)
    auto dma_one = [&](const Prep& c, int e) __attribute__((always_inline)) {
        if (e < W3) {
            if (e < c.nw)                       // (uniform)
                dma_x4(c.slot + (unsigned)(XS_F * 4) + (unsigned)((wave * c.nw + e) * 1024), c.wcur[e], sel_rsrc(c.g1, wr0, wr1), c.ws0);
            return;
        }
        const int ex = e - W3;
        if (ex >= c.nx) return;                 // (uniform)
        if (c.gather) {
            // (ex < XG) 4 bytes per lane: row = wave * (32 / NW) + ex / NCS, column slot ex % NCS
            if (ex < XG) {
                const int row = wave * (32 / NW) + ex / NCS, piece = ex % NCS;
                dma_dword(c.slot + (unsigned)(row * PWS + piece * 64) * 4u, xg_off[piece], sel_rsrc(c.g1, xr0, xr1), c.xs0 + (unsigned)row * c.xs4);
            }
        } else if (ex < XW) {
            const int f = (wave * XW + ex) * 64 + lane;         // 16-byte piece of the [32][PW4] window
            const int row = f / PW4, cp = f - row * PW4;
            const int pos = c.win + 4 * cp;
            const bool ok = row < 32 && cp < c.need4 && pos >= 0 && pos < c.rowlen;
            const unsigned vo = ok ? (unsigned)row * c.xs4 + (unsigned)pos * 4u : X_OOB;
            dma_x4(c.slot + (unsigned)((wave * XW + ex) * 1024), vo, sel_rsrc(c.g1, xr0, xr1), c.xs0);
        }
    };
    constexpr int DMAX = XMAX + W3;             // most DMAs a slice issues per wave
    constexpr int PER1 = (DMAX + 15) / 16;      // DMAs behind one step of a 16-step slice

    // =============================== MFMA side ===============================
    kg_f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int mrow = rw * 32 * TMW + l32;
    const int ccol = cw * 32 * TNW + l32;
    // LDS byte offsets (inside the weight slab) of this lane's 16-byte fragments, first row block:
    //   TAPROW: fragment t = 0..11 holds k' = 48 kh + 4 t .. + 3 (k' = 3 c + tap);  ROW: fragment t = 0..3 holds c = 16 kh + 4 t ..
    unsigned a3off[12], a1off[4];
    {
        const int sw = (mrow >> 1) & 7;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const int i = 12 * kh + t;
            a3off[t] = (unsigned)((mrow * 24 + ((i & ~7) | ((i & 7) ^ sw))) * 16);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) a1off[t] = (unsigned)((mrow * 8 + ((4 * kh + t) ^ sw)) * 16);
    }
    unsigned ocol[TNW];             // byte offset of this lane's output column(s) (incl. its 4 kh rows), or X_OOB
    unsigned bb[2][TNW][3];         // LDS byte offset (inside the feature slab, incl. the 16 kh channel rows) per group, column block, tap

    const int ots_ = a.o_tstride > 1 ? a.o_tstride : 1;
    const unsigned okh = (unsigned)(4 * kh) * (unsigned)a.o_sC * 4u;
    auto mfma_tile_setup = [&](int it) __attribute__((always_inline)) {
        int ct, rt;
        tile_of(b + it * G, tp, ct, rt);
        Col xcs[TNW];
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) {
            xcs[tn] = decode_col(ct * BN + cw * 32 * TNW + tn * 32 + l32, ncols, a.T_out, a.V_out);
            ocol[tn] = xcs[tn].valid ? ((unsigned)xcs[tn].n * (unsigned)a.o_sN + (unsigned)(xcs[tn].to * ots_ * a.V_out + xcs[tn].vo)) * 4u + okh : X_OOB;
        }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            if (gq < ngroups) {
                const GU& u = gu[gq];
                const unsigned khrow = (unsigned)(16 * kh * PWS * 4);
                if (u.xmode != 0) {
#pragma unroll
                    for (int tn = 0; tn < TNW; ++tn)
#pragma unroll
                        for (int d = 0; d < 3; ++d) bb[gq][tn][d] = khrow + (unsigned)((ccol + tn * 32) * 4);
                } else {
                    const int win = window_lo(ct, gq);
                    const int tstep = u.pad ? 1 : 0;
#pragma unroll
                    for (int tn = 0; tn < TNW; ++tn) {
                        const Col xc = xcs[tn];
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const int ti = u.tr ? xc.to + u.pad - tstep * d : xc.to * u.ts - u.pad + tstep * d;
                            const bool ok = xc.valid && d < u.taps && ti >= 0 && ti < u.Tin;
                            const int pos = xc.n * u.xsN + ti * u.Vin + xc.vo - win;
                            bb[gq][tn][d] = khrow + (unsigned)((ok ? pos : PWS - 1) * 4);
                        }
                    }
                }
            }
        }
    };

    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.out), 0, (int)X_OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_add = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.add), 0, a.add ? (int)X_OOB : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.mask), 0, a.mask ? (int)X_OOB : 0, 0x00020000);
    const bool has_add = a.add != nullptr, has_mask = a.mask != nullptr;
    const bool mfull = a.M % 32 == 0;                   // (uniform) no ragged row block
    const unsigned osc4 = (unsigned)a.o_sC * 4u;
    // epilogue of tile `it` (whose columns mfma_tile_setup decoded): the row of a register goes through the store's SCALAR
    // offset, the column (and the lane's 4 kh rows) through ocol; every lane issues all its stores (see v1)
    auto epilogue = [&](int it) __attribute__((always_inline)) {
        int ct, rt;
        tile_of(b + it * G, tp, ct, rt);
        const int rowb = rt * BM + rw * 32 * TMW;       // (uniform) first row of this wave's row blocks
        const float* const blp = Bl + rowb + 4 * kh;
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) {
            Col xc;
            unsigned acol = 0, mcol = 0;
            if (has_add || has_mask) {      // (uniform, rare: identity residual / linearised blocks)
                xc = decode_col(ct * BN + cw * 32 * TNW + tn * 32 + l32, ncols, a.T_out, a.V_out);
                acol = ((unsigned)xc.n * (unsigned)a.a_sN + (unsigned)(xc.to * a.a_tstride * a.V_out + xc.vo)) * 4u + (unsigned)(4 * kh) * (unsigned)a.a_sC * 4u;
                mcol = ((unsigned)xc.n * (unsigned)a.m_sN + (unsigned)(xc.to * a.V_out + xc.vo)) * 4u + (unsigned)(4 * kh) * (unsigned)a.m_sC * 4u;
            }
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) {
                const kg_f32x16& av = acc[tm * TNW + tn];
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                    v[r] = av[r] + (((mfull && rowb + tm * 32 < a.M) || rowb + rr + 4 * kh < a.M) ? blp[rr] : 0.f);
                }
                if (has_add) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                        const bool ok = ocol[tn] != X_OOB && ((mfull && rowb + tm * 32 < a.M) || rowb + rr + 4 * kh < a.M);
                        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_add, ok ? acol : X_OOB, (unsigned)(rowb + rr) * (unsigned)a.a_sC * 4u, 0));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += rv[r];
                }
                float mv[16];
                if (has_mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                        const bool ok = ocol[tn] != X_OOB && ((mfull && rowb + tm * 32 < a.M) || rowb + rr + 4 * kh < a.M);
                        mv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_msk, ok ? mcol : X_OOB, (unsigned)(rowb + rr) * (unsigned)a.m_sC * 4u, 0));
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                    float o = kg_act(v[r], a.act, a.slope);
                    if (has_mask) o *= mv[r] > 0.f ? 1.f : a.slope;
                    const unsigned vo = ((mfull && rowb + tm * 32 < a.M) || rowb + rr + 4 * kh < a.M) ? ocol[tn] : X_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), r_out, vo, (unsigned)(rowb + rr) * osc4, 0);
                }
            }
        }
    };
    const int NST = 16 * NACC * (1 + (has_add ? 1 : 0) + (has_mask ? 1 : 0));

    // ---- the MFMA loop of one slice, by weight layout.  `c` = the prepared slice whose DMAs ride behind the first steps.
    // B operand of step (channel cl inside the lane half's 16, tap d): xs[(16 kh + cl) * PWS + position(column, d)]
    Prep cn;                // the slice prepared for the NEXT iteration's DMAs
    int gnow = 0;           // iteration counter of the slice loop
    auto slice_taps3 = [&](const char* xs, const char* ws, const unsigned (&bt)[TNW][3], const Prep& c, auto colmajor) {
        constexpr bool COLM = decltype(colmajor)::value;            // WL_COLTAP instead of WL_TAPROW
        // step s = 0..47: k' = 48 kh + s = 3 cl + d with cl = s / 3, d = s % 3
        kg_f32x4 af[2][TMW];
        auto lda4 = [&](int t, kg_f32x4 (&o)[TMW]) {
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) o[tm] = *(const kg_f32x4*)(ws + a3off[t] + tm * (32 * 24 * 16));
        };
        auto lda1 = [&](int s, int tm) -> float {                   // [32][3 BM]: row 16 kh + cl, element 3 m + d
            return *(const float*)(ws + (((16 * kh + s / 3) * 3 * BM + 3 * (mrow + tm * 32) + s % 3) * 4));
        };
        auto ldb = [&](int s, int tn) -> float { return *(const float*)(xs + bt[tn][s % 3] + (s / 3) * (PWS * 4)); };
        constexpr int NPF = 3;
        float bv[NPF][TNW], a1[NPF][TMW];
        if constexpr (!COLM) lda4(0, af[0]);
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) bv[s][tn] = ldb(s, tn);
            if constexpr (COLM) {
#pragma unroll
                for (int tm = 0; tm < TMW; ++tm) a1[s][tm] = lda1(s, tm);
            }
        }
        auto step = [&](int s) __attribute__((always_inline)) {
            float a_[TMW], b_[TNW];
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) b_[tn] = bv[s % NPF][tn];
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) {
                if constexpr (COLM) a_[tm] = a1[s % NPF][tm];
                else a_[tm] = af[(s >> 2) & 1][tm][s & 3];
            }
            if (s + NPF < 48) {
#pragma unroll
                for (int tn = 0; tn < TNW; ++tn) bv[s % NPF][tn] = ldb(s + NPF, tn);
                if constexpr (COLM) {
#pragma unroll
                    for (int tm = 0; tm < TMW; ++tm) a1[s % NPF][tm] = lda1(s + NPF, tm);
                }
            }
            if constexpr (!COLM) {
                if ((s & 3) == 0 && s + 4 < 48) lda4((s >> 2) + 1, af[((s >> 2) + 1) & 1]);
            }
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNW; ++tn)
                    acc[tm * TNW + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[tm], b_[tn], acc[tm * TNW + tn], 0, 0, 0);
            if (s < DMAX) dma_one(c, s);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int s = 0; s < 40; ++s) step(s);
        prep(cn, gnow + 1 + LA);                // the next slice's address work, under the MFMAs in flight
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 40; s < 48; ++s) step(s);
    };
    auto slice_tap1 = [&](const char* xs, const char* ws, const unsigned (&bt)[TNW][3], const Prep& c, auto colmajor) {
        constexpr bool COLM = decltype(colmajor)::value;            // WL_COL instead of WL_ROW
        // step q = 0..15: channel 16 kh + q
        kg_f32x4 af[2][TMW];
        auto lda4 = [&](int t, kg_f32x4 (&o)[TMW]) {
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) o[tm] = *(const kg_f32x4*)(ws + a1off[t] + tm * (32 * 8 * 16));
        };
        auto lda1 = [&](int q, int tm) -> float { return *(const float*)(ws + (((16 * kh + q) * BM + mrow + tm * 32) * 4)); };
        auto ldb = [&](int q, int tn) -> float { return *(const float*)(xs + bt[tn][0] + q * (PWS * 4)); };
        constexpr int NPF = 3;
        float bv[NPF][TNW], a1[NPF][TMW];
        if constexpr (!COLM) lda4(0, af[0]);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) bv[q][tn] = ldb(q, tn);
            if constexpr (COLM) {
#pragma unroll
                for (int tm = 0; tm < TMW; ++tm) a1[q][tm] = lda1(q, tm);
            }
        }
        auto step = [&](int q) __attribute__((always_inline)) {
            float a_[TMW], b_[TNW];
#pragma unroll
            for (int tn = 0; tn < TNW; ++tn) b_[tn] = bv[q % NPF][tn];
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) {
                if constexpr (COLM) a_[tm] = a1[q % NPF][tm];
                else a_[tm] = af[(q >> 2) & 1][tm][q & 3];
            }
            if (q + NPF < 16) {
#pragma unroll
                for (int tn = 0; tn < TNW; ++tn) bv[q % NPF][tn] = ldb(q + NPF, tn);
                if constexpr (COLM) {
#pragma unroll
                    for (int tm = 0; tm < TMW; ++tm) a1[q % NPF][tm] = lda1(q + NPF, tm);
                }
            }
            if constexpr (!COLM) {
                if ((q & 3) == 0 && q + 4 < 16) lda4((q >> 2) + 1, af[((q >> 2) + 1) & 1]);
            }
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int tn = 0; tn < TNW; ++tn)
                    acc[tm * TNW + tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[tm], b_[tn], acc[tm * TNW + tn], 0, 0, 0);
#pragma unroll
            for (int e = q * PER1; e < (q + 1) * PER1; ++e)
                if (e < DMAX) dma_one(c, e);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int q = 0; q < 12; ++q) step(q);
        prep(cn, gnow + 1 + LA);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 12; q < 16; ++q) step(q);
    };

    // ---- prologue
    int pend[LA > 1 ? LA - 1 : 1];          // vector-memory operations of the LA - 1 youngest issued slices (oldest first)
    {
#pragma unroll 1
        for (int gs = 0; gs < LA; ++gs) {
            prep(cn, gs);
#pragma unroll
            for (int e = 0; e < DMAX; ++e) dma_one(cn, e);
            if (LA > 1 && gs >= 1) pend[gs - 1] = cn.nx + cn.nw;
        }
        prep(cn, LA);
    }

    int c_it = 0, c_gi = 0, c_sl = 0;
    unsigned ephist = 0;
    mfma_tile_setup(0);
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
        {
            int n = NST * __builtin_popcount(ephist & ((1u << LA) - 1u));
            if (LA > 1) {
#pragma unroll
                for (int i = 0; i < LA - 1; ++i) n += pend[i];
            }
            wait_vm_upto(n);
        }
        wg_barrier();
        const Prep c = cn;
        if (LA > 1) {
#pragma unroll
            for (int i = 0; i + 1 < LA - 1; ++i) pend[i] = pend[i + 1];
            pend[LA - 2 >= 0 ? LA - 2 : 0] = c.nx + c.nw;
        }
        const char* xs = (const char*)(kg_ring_lds + (g % NSTAGE) * STAGE_F);
        const char* ws = xs + XS_F * 4;
        const bool g1 = c_gi != 0;
        const int wl = g1 ? gu[1].wl : gu[0].wl;
        // (uniform) the slice loop of this group's weight layout; the next slice's address work follows it
        gnow = g;
        unsigned bt[TNW][3];
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn)
#pragma unroll
            for (int d = 0; d < 3; ++d) bt[tn][d] = g1 ? bb[1][tn][d] : bb[0][tn][d];
        if (wl == WL_TAPROW)      slice_taps3(xs, ws, bt, c, std::false_type{});
        else if (wl == WL_COLTAP) slice_taps3(xs, ws, bt, c, std::true_type{});
        else if (wl == WL_ROW)    slice_tap1(xs, ws, bt, c, std::false_type{});
        else                      slice_tap1(xs, ws, bt, c, std::true_type{});
        ephist <<= 1;
        const int nsl = g1 ? gu[1].nslice : gu[0].nslice;
        if (++c_sl == nsl) {
            c_sl = 0;
            if (++c_gi == ngroups) {
                c_gi = 0;
                epilogue(c_it);
                ++c_it;
                ephist |= 1u;
#pragma unroll
                for (int i = 0; i < NACC; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                if (c_it < my_tiles) mfma_tile_setup(c_it);
            }
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------

// ring tile codes (kg_conv_plan_info reports 20 + code): 0..5 ring v1 (4-byte DMA, any full-slice problem), 6.. ring v2 (window form)
enum RingTile { R64x128 = 0, R64x64 = 1, R128x128 = 2, R32x128 = 3, R128x64 = 4, R32x256 = 5,
                W64x128 = 6, W128x128 = 7, W128x64 = 8, W32x256 = 9, W64x64 = 10, RING_TILES = 11 };
struct RingTileInfo { int bm, bn, waves, nstage, wgpc, pws; };
constexpr RingTileInfo kRingTiles[RING_TILES] = {
    {64, 128, 8, 4, 1, 0},     // 24 KB per slot
    {64, 64, 4, 3, 3, 0},      // 16 KB per slot, three workgroups per CU
    {128, 128, 8, 4, 1, 0},    // 32 KB per slot
    {32, 128, 4, 3, 2, 0},     // 20 KB per slot
    {128, 64, 4, 3, 2, 0},     // 24 KB per slot
    {32, 256, 8, 4, 1, 0},     // 36 KB per slot
    {64, 128, 8, 3, 1, 192},   // window form: 48 KB per slot
    {128, 128, 8, 2, 1, 192},  // 72 KB per slot
    {128, 64, 8, 2, 1, 192},   // 72 KB per slot (the window of a stride-2 conv is twice its columns)
    {32, 256, 8, 2, 1, 320},   // 52 KB per slot
    {64, 64, 4, 2, 2, 96},     // 36 KB per slot, TWO workgroups of four waves per CU: one's tile change runs under the other's MFMAs
};

size_t ring_lds_bytes(const RingTileInfo& t) {
    size_t slot = (size_t)(32 * t.bn + 32 * t.bm) * 4;
    if (t.pws) {        // window form: both slabs hold whole 1-KiB DMAs of all waves (kg_conv_ringw_kernel: XS_F, WS_F)
        const size_t xw = (32 * (t.pws / 4) + 64 * t.waves - 1) / (64 * t.waves), w3 = (t.bm * 24 + 64 * t.waves - 1) / (64 * t.waves);
        const size_t xs = std::max<size_t>(t.waves * xw * 256, 32 * t.pws), ws = std::max<size_t>(t.waves * w3 * 256, 96 * t.bm);
        slot = (xs + ws) * 4;
    }
    return (size_t)t.nstage * slot + RING_MAXM * 4 + 128 * 4;
}

void ring_grid(const KgConvArgs* a, const RingTileInfo& ti, int& ctiles, int& rtiles, int& grid) {
    const int ncols = a->N * a->T_out * a->V_out;
    ctiles = kg_cdiv(ncols, ti.bn);
    rtiles = kg_cdiv(a->M, ti.bm);
    const long padded = (long)((ctiles + 7) / 8 * 8) * rtiles;
    long g = 256L * ti.wgpc;
    if (padded < g) g = (padded + 7) / 8 * 8;
    grid = (int)g;
}

template <int RW, int CW, int TMW, int TNW, int NSTAGE, int MINW>
int launch_ring(const KgConvArgs* a, const RingTileInfo& ti, hipStream_t s) {
    RingPlan pl;
    ring_grid(a, ti, pl.ctiles, pl.rtiles, pl.grid);
    pl.slices = 0;
    for (int i = 0; i < a->ngroups; ++i) pl.slices += a->g[i].taps * (a->g[i].Cin / 32);
    const size_t lds = ring_lds_bytes(ti);
    const bool kf = a->g[0].w_sI <= a->g[0].w_sO;
    if (kf) {
        auto kern = kg_conv_ring_kernel<RW, CW, TMW, TNW, true, NSTAGE, MINW>;
        static bool attr = false;       // idempotent; a race only repeats the call
        if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(64 * RW * CW), lds, s, *a, pl);
    } else {
        auto kern = kg_conv_ring_kernel<RW, CW, TMW, TNW, false, NSTAGE, MINW>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(64 * RW * CW), lds, s, *a, pl);
    }
    return kg_launch_status("kg_conv (ring)");
}

// ring v2: can this problem run on tile `ti`, and with which per-group modes?
bool ringw_plan(const KgConvArgs* a, const RingTileInfo& ti, RingWPlan& pl) {
    ring_grid(a, ti, pl.ctiles, pl.rtiles, pl.grid);
    pl.slices = 0;
    int gathers = 0;
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        RingWGroup& pg = pl.g[i];
        const bool chanblock = g.tap_mode == KG_TAP_CHANBLOCK;
        if (g.Cin % 32 != 0) return false;
        if (g.transposed && g.t_stride != 1) return false;
        pg.taps = (!chanblock && g.taps == 3) ? 3 : 1;
        pg.nslice = (chanblock ? g.taps : 1) * (g.Cin / 32);
        pg.cpb = g.Cin / 32;
        pg.rowlen = a->N * g.T_in * g.V_in;
        pl.slices += pg.nslice;
        // ---- features
        const int s = g.t_stride;
        const bool dense = g.vmap == nullptr && g.V_in == a->V_out && g.x_sN == (int64_t)g.T_in * g.V_in &&
                           g.T_in == (g.transposed ? a->T_out : a->T_out * s) &&
                           ((uintptr_t)g.x % 16 == 0) && g.x_sC % 4 == 0 && pg.rowlen % 4 == 0;
        const int cross = (ti.bn - 1 + a->V_out - 1) / a->V_out;
        const int span = 3 + (ti.bn - 1) + cross * (s - 1) * g.V_in + (pg.taps - 1) * g.V_in + 1;
        pg.need4 = (span + 3) / 4;
        if (dense && span <= ti.pws - 4) {
            pg.xmode = 0;
        } else if (pg.taps == 1 && !g.transposed && gathers == 0) {
            pg.xmode = 1;
            ++gathers;
        } else {
            return false;
        }
        // ---- weights (16-byte pieces: alignment of every piece)
        if ((uintptr_t)g.w % 16 != 0) return false;
        const bool rowblocks = g.w_MB < a->M;
        if (pg.taps == 3) {
            if (rowblocks || g.w_sT != 1) return false;
            if (g.w_sI == 3 && g.w_sO % 4 == 0) pg.wl = WL_TAPROW;
            else if (g.w_sO == 3 && g.w_sI % 4 == 0 && a->M % 4 == 0) pg.wl = WL_COLTAP;
            else return false;
        } else {
            if (chanblock && g.w_sT % 4 != 0) return false;
            if (rowblocks && (g.w_sMB % 4 != 0 || g.w_MB % 4 != 0)) return false;
            if (g.w_sI == 1 && g.w_sO % 4 == 0) pg.wl = WL_ROW;
            else if (g.w_sO == 1 && g.w_sI % 4 == 0 && a->M % 4 == 0) pg.wl = WL_COL;
            else return false;
        }
    }
    if (a->ngroups == 1) pl.g[1] = pl.g[0];
    return true;
}

template <int RW, int CW, int TMW, int TNW, int PWS, int NSTAGE, int MINW>
int launch_ringw(const KgConvArgs* a, const RingTileInfo& ti, hipStream_t s) {
    RingWPlan pl;
    if (!ringw_plan(a, ti, pl)) { kg_set_error("kg_conv (ring window form): problem not eligible"); return -1; }
    pl.stagger = ti.wgpc > 1 ? kg_env().conv_ring_stagger : 0;
    const size_t lds = ring_lds_bytes(ti);
    auto kern = kg_conv_ringw_kernel<RW, CW, TMW, TNW, PWS, NSTAGE, MINW>;
    static bool attr = false;       // idempotent; a race only repeats the call
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(64 * RW * CW), lds, s, *a, pl);
    return kg_launch_status("kg_conv (ring, window form)");
}

}  // namespace

// Can the ring form run this problem at all?  Full K-slices, both groups' weights in one orientation (validated by kg_conv),
// rows and offsets within the LDS / 32-bit budgets of the kernel.
bool kg_ring_eligible(const KgConvArgs* a) {
    if (a->M > RING_MAXM || a->V_out > 64) return false;
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        if (g.Cin % 32 != 0) return false;
        // the channel / tap walk goes through the DMA's scalar byte offset: the whole operand within 2 GiB (beyond that
        // the C5a gcn contraction, 1536 rows of 1.6 MB, read wrong rows: tools/time_c5a_ring.py)
        if ((long)(g.tap_mode == KG_TAP_CHANBLOCK ? g.taps : 1) * g.Cin * g.x_sC >= (1L << 29)) return false;
    }
    const long ospan = (long)(a->M - 1) * a->o_sC + (long)(a->N - 1) * a->o_sN + (long)a->T_out * (a->o_tstride > 1 ? a->o_tstride : 1) * a->V_out;
    if (a->o_sC < 0 || a->o_sN < 0 || ospan >= (1L << 29)) return false;
    if (a->add) {
        const long aspan = (long)(a->M - 1) * a->a_sC + (long)(a->N - 1) * a->a_sN + (long)a->T_out * (a->a_tstride > 1 ? a->a_tstride : 1) * a->V_out;
        if (a->a_sC < 0 || a->a_sN < 0 || aspan >= (1L << 29)) return false;
    }
    if (a->mask) {
        const long mspan = (long)(a->M - 1) * a->m_sC + (long)(a->N - 1) * a->m_sN + (long)a->T_out * a->V_out;
        if (a->m_sC < 0 || a->m_sN < 0 || mspan >= (1L << 29)) return false;
    }
    return true;
}

// a window-form tile (6..) additionally needs the problem's geometry to fit its window and 16-byte alignment
bool kg_ring_tile_ok(const KgConvArgs* a, int tile) {
    if (tile < 0 || tile >= RING_TILES || !kg_ring_eligible(a)) return false;
    if (kRingTiles[tile].pws == 0) return true;
    RingWPlan pl;
    return ringw_plan(a, kRingTiles[tile], pl);
}

int kg_ring_tile_count() { return RING_TILES; }

void kg_ring_tile_dims(int tile, int* bm, int* bn, int* wgpc) {
    *bm = kRingTiles[tile].bm; *bn = kRingTiles[tile].bn; *wgpc = kRingTiles[tile].wgpc;
}

int kg_ring_launch(const KgConvArgs* a, int tile, hipStream_t s) {
    switch (tile) {
        case R64x128:  return launch_ring<2, 4, 1, 1, 4, 2>(a, kRingTiles[tile], s);
        case R64x64:   return launch_ring<2, 2, 1, 1, 3, 3>(a, kRingTiles[tile], s);
        case R128x128: return launch_ring<4, 2, 1, 2, 4, 2>(a, kRingTiles[tile], s);
        case R32x128:  return launch_ring<1, 4, 1, 1, 3, 2>(a, kRingTiles[tile], s);
        case R128x64:  return launch_ring<2, 2, 2, 1, 3, 2>(a, kRingTiles[tile], s);
        case R32x256:  return launch_ring<1, 8, 1, 1, 4, 2>(a, kRingTiles[tile], s);
        case W64x128:  return launch_ringw<2, 4, 1, 1, 192, 3, 2>(a, kRingTiles[tile], s);
        case W128x128: return launch_ringw<4, 2, 1, 2, 192, 2, 2>(a, kRingTiles[tile], s);
        case W128x64:  return launch_ringw<4, 2, 1, 1, 192, 2, 2>(a, kRingTiles[tile], s);
        case W32x256:  return launch_ringw<1, 8, 1, 1, 320, 2, 2>(a, kRingTiles[tile], s);
        case W64x64:   return launch_ringw<2, 2, 1, 1, 96, 2, 2>(a, kRingTiles[tile], s);
        default: kg_set_error("kg_conv (ring): unknown tile %d", tile); return -1;
    }
}
