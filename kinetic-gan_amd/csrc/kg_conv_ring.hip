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
#ifndef KG_RING_NODMA
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
#endif
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most n vector-memory operations of this wave may still be in flight (rounded DOWN to a multiple of 4: waiting for
// more than necessary is always safe; the counter has 6 bits)
__device__ __forceinline__ void wait_vm_upto(int n) {
#ifdef KG_RING_NOWAIT
    return;
#endif
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
#ifndef KG_RING_NOBARRIER
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
__device__ __forceinline__ void wg_barrier() {}
#endif

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
#ifdef KG_RING_NOSTORE
                    if (o == 123.456f)
#endif
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

// ---- host side -----------------------------------------------------------------------------------------------------------

// ring tile codes (kg_conv_plan_info reports 20 + code)
enum RingTile { R64x128 = 0, R64x64 = 1, R128x128 = 2, R32x128 = 3, R128x64 = 4, R32x256 = 5, RING_TILES = 6 };
struct RingTileInfo { int bm, bn, waves, nstage, wgpc; };
constexpr RingTileInfo kRingTiles[RING_TILES] = {
    {64, 128, 8, 4, 1},     // 24 KB per slot
    {64, 64, 4, 3, 3},      // 16 KB per slot, three workgroups per CU
    {128, 128, 8, 4, 1},    // 32 KB per slot
    {32, 128, 4, 3, 2},     // 20 KB per slot
    {128, 64, 4, 3, 2},     // 24 KB per slot
    {32, 256, 8, 4, 1},     // 36 KB per slot
};

size_t ring_lds_bytes(const RingTileInfo& t) {
    return (size_t)t.nstage * (32 * t.bn + 32 * t.bm) * 4 + RING_MAXM * 4 + 128 * 4;
}

template <int RW, int CW, int TMW, int TNW, int NSTAGE, int MINW>
int launch_ring(const KgConvArgs* a, const RingTileInfo& ti, hipStream_t s) {
    const int ncols = a->N * a->T_out * a->V_out;
    RingPlan pl;
    pl.ctiles = kg_cdiv(ncols, ti.bn);
    pl.rtiles = kg_cdiv(a->M, ti.bm);
    pl.slices = 0;
    for (int i = 0; i < a->ngroups; ++i) pl.slices += a->g[i].taps * (a->g[i].Cin / 32);
    const long padded = (long)((pl.ctiles + 7) / 8 * 8) * pl.rtiles;
    long grid = 256L * ti.wgpc;
    if (padded < grid) grid = (padded + 7) / 8 * 8;
    pl.grid = (int)grid;
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

}  // namespace

// Can the ring form run this problem at all?  Full K-slices, both groups' weights in one orientation (validated by kg_conv),
// rows and offsets within the LDS / 32-bit budgets of the kernel.
bool kg_ring_eligible(const KgConvArgs* a) {
    if (a->M > RING_MAXM || a->V_out > 64) return false;
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        if (g.Cin % 32 != 0) return false;
        // the channel / tap walk goes through 32-bit scalar byte offsets
        if ((long)(g.tap_mode == KG_TAP_CHANBLOCK ? g.taps : 1) * g.Cin * g.x_sC >= (1L << 30)) return false;
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
        default: kg_set_error("kg_conv (ring): unknown tile %d", tile); return -1;
    }
}
