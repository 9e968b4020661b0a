// kg_conv, image form: the tap GEMM of the discriminator's wide-and-thin layers (D0 / D1: <= 64 output rows, <= 224
// contraction depth, 45 k - 135 k columns) with
//   * the FEATURE operand streamed global -> LDS by the DMA form of the buffer loads (buffer_load_dwordx4 ... lds:
//     1 KB per wave instruction, no VGPR round trip) into a double-buffered IMAGE of the tile: every channel row of
//     the launch x (tile columns + one frame of halo on both sides); the three temporal taps read the same image at
//     columns shifted by -V / 0 / +V, so a feature element crosses the vector memory path once instead of three times;
//   * ALL weights of a wave's 32 output rows resident in registers as MFMA A operands (K/2 VGPRs, loaded once per
//     workgroup through an LDS transposition);
//   * persistent workgroups (one per CU) that walk the column tiles: per tile one barrier, one DMA batch for the next
//     tile, K/2 MFMAs per wave fed by ds_read_b32, epilogue.
// The direct kernel (kg_conv.hip) loads every B fragment with a 4-byte lane load per MFMA; tools/probe/ showed that this
// path - not issue slots, LDS or barriers - bounds those launches (24.4 / 57.4 us at 64 / 192 samples on the D1 tail,
// 0.33 / 0.43 of the fp32 MFMA peak); this form runs them in 20 / 47 us.
// Scope (everything else stays with kg_conv.hip): one or two K-slice groups of TAP_TIME taps with frame stride 1, no
// vertex map, T_in = T_out, V_in = V_out <= 16, sample-contiguous planes, exactly the channel / tap combinations
// instantiated below, dense weight tensors.  Forward and transposed (time-flipped) taps, biases, residual add,
// activation and derivative mask are supported as in kg_conv.
#include <stdio.h>

#include "kg_common.h"

#ifndef KG_IMG_UN
#define KG_IMG_UN 8
#endif

namespace {

constexpr int NT = 256;
constexpr int HALO = 16;            // image columns in front of / behind the tile (>= V, multiple of 4)

// one wave instruction: every lane fetches 16 bytes at its byte offset, the wave's 1 KB lands at `ldsp` in lane order
// (a plain device function: inside the kernel template the address-space cast made hipcc drop the kernels' host stubs
// without a diagnostic)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* ldsp, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)ldsp, 16, off, 0, 0, 0);
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

typedef KgImgWeights ImgWeights;
typedef KgImgArgs ImgArgs;

// RT row tiles x CG column groups of 32 = 4 waves; group 0: C0 channels x TAPS0 temporal taps, group 1: C1 channels x 1
template <int RT, int C0, int TAPS0, int C1>
__global__ __launch_bounds__(NT, 2) void kg_conv_img_kernel(const KgConvArgs a, const ImgArgs ia) {
    constexpr int CG = 4 / RT;
    constexpr int TC = 32 * CG;                 // columns per tile
    constexpr int IW = TC + 2 * HALO;           // image width (floats)
    constexpr int SEG = IW / 4;                 // 16-byte segments per image row
    constexpr int ROWS = C0 + C1;
    constexpr int IMG = ROWS * IW;
    constexpr int K = TAPS0 * C0 + C1, KS = K / 2;
    constexpr int BMW = 32 * RT;                // rows per workgroup
    constexpr int WP = BMW + 1;
    static_assert(K % 2 == 0 && (ROWS * SEG) % 64 == 0, "image shape");
    extern __shared__ float lds[];
    float* const Im = lds;                      // [2][ROWS][IW]; first used as [K][WP] weight scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
    const int ri = wave / CG, cg = wave % CG;
    const int ncols = a.N * a.T_out * a.V_out;
    const int V = a.V_out, L = a.T_out * a.V_out;
    const int m0 = blockIdx.y * BMW;

    // ---- weights -> LDS [k][m] (coalesced reads of the dense box, decoded to (tap, row, channel)) -> registers
    {
        // e -> (i0, i1, i2) with two multiply-high divisions (magic numbers from the host, exact for e < 2^16 * n) and
        // branch-free role selection: no runtime-indexed private array (that went to scratch memory: 30 us of preload)
        auto stage = [&](const KgConvGroup& g, const ImgWeights& w, int taps, int cin, int kbase) {
            const int total = taps * a.M * cin;
            constexpr int UN = KG_IMG_UN;            // loads in flight per thread (one by one the loop ran at one memory
            for (int e0 = tid; e0 < total; e0 += NT * UN) {     // latency per element: 26 us of preload)
                float v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int e = e0 + u * NT;
                    v[u] = e < total ? g.w[e] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int e = e0 + u * NT;
                    const int rest = (int)__umulhi((unsigned)e, w.magic0), i0 = e - rest * w.n0;
                    const int i2 = (int)__umulhi((unsigned)rest, w.magic1), i1 = rest - i2 * w.n1;
                    const int d = w.r0 == 0 ? i0 : (w.r1 == 0 ? i1 : i2);
                    const int m = w.r0 == 1 ? i0 : (w.r1 == 1 ? i1 : i2);
                    const int c = w.r0 == 2 ? i0 : (w.r1 == 2 ? i1 : i2);
                    if (e < total && m >= m0 && m < m0 + BMW) Im[(kbase + d * cin + c) * WP + (m - m0)] = v[u];
                }
            }
        };
        // rows beyond M (ragged last row tile) must read as zero
        for (int e = tid; e < K * WP; e += NT) Im[e] = 0.f;
        __syncthreads();
        stage(a.g[0], ia.w[0], TAPS0, C0, 0);
        if constexpr (C1 > 0) stage(a.g[1], ia.w[1], 1, C1, TAPS0 * C0);
        __syncthreads();
    }
    float wr[KS];                               // A operands: W[32 ri + l31][2 q + kh]
#pragma unroll
    for (int q = 0; q < KS; ++q) wr[q] = Im[(2 * q + kh) * WP + 32 * ri + l31];
    const int mrow = m0 + 32 * ri;              // first row of this wave
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mrow + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float b = 0.f;
        if (m < a.M) {
            if (a.bias0) b += a.bias0[m];
            if (a.bias1) b += a.bias1[m];
        }
        bias[r] = b;
    }
    __syncthreads();                            // the scratch becomes the image buffers

    // ---- DMA plan of one image: flat 16-byte segments s = 64 j + lane, j = wave, wave + 4, ...
    const long ext0 = (long)(C0 - 1) * a.g[0].x_sC + ncols;
    const __amdgpu_buffer_rsrc_t d0 = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(a.g[0].x), 0, (int)(ext0 * 4 > 0x7fffffffL ? 0x7fffffffL : ext0 * 4), 0x00020000);
    constexpr int N0 = C0 * SEG / 64, N1 = C1 * SEG / 64;
    constexpr int J0 = (N0 + 3) / 4, J1 = (N1 + 3) / 4;
    unsigned off0[J0], off1[J1 > 0 ? J1 : 1];
#pragma unroll
    for (int i = 0; i < J0; ++i) {
        const int j = wave + 4 * i, s = 64 * j + lane, row = s / SEG, sg = s - row * SEG;
        off0[i] = j < N0 ? (unsigned)(((long)row * a.g[0].x_sC + 4 * sg - HALO) * 4) : 0x80000000u;
    }
    __amdgpu_buffer_rsrc_t d1 = d0;
    if constexpr (C1 > 0) {
        const long ext1 = (long)(C1 - 1) * a.g[1].x_sC + ncols;
        d1 = __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(a.g[1].x), 0,
                                               (int)(ext1 * 4 > 0x7fffffffL ? 0x7fffffffL : ext1 * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < J1; ++i) {
            const int j = wave + 4 * i, s = 64 * j + lane, row = s / SEG, sg = s - row * SEG;
            off1[i] = j < N1 ? (unsigned)(((long)row * a.g[1].x_sC + 4 * sg - HALO) * 4) : 0x80000000u;
        }
    }
    // a segment in front of column 0 has a "negative" offset: as an unsigned byte offset it is out of range -> zeros
    auto dma = [&](int tile, int b) {
        float* const im = Im + b * IMG;
        const bool live = tile < ia.ntiles;
        const unsigned tb = (unsigned)(tile * TC * 4);
#pragma unroll
        for (int i = 0; i < J0; ++i) {
            const int j = wave + 4 * i;
            if (j < N0)
                dma16(d0, im + 256 * j, (live && off0[i] != 0x80000000u) ? off0[i] + tb : 0x80000000u);
        }
        if constexpr (C1 > 0) {
#pragma unroll
            for (int i = 0; i < J1; ++i) {
                const int j = wave + 4 * i;
                if (j < N1)
                    dma16(d1, im + C0 * IW + 256 * j, (live && off1[i] != 0x80000000u) ? off1[i] + tb : 0x80000000u);
            }
        }
    };

    const bool has_add = a.add != nullptr, has_mask = a.mask != nullptr;
    const int act = a.act;
    const float slope = a.slope;
    auto plane_desc = [&](const float* p, long sC) {
        const long ext = p ? ((long)(a.M - 1) * sC + ncols) * 4 : 0;
        return __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(p), 0, (int)(ext > 0x7fffffffL ? 0x7fffffffL : ext), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t outd = plane_desc(a.out, a.o_sC), addd = plane_desc(a.add, a.a_sC),
                                 maskd = plane_desc(a.mask, a.m_sC);
    // per accumulator row: byte offsets of the row in out / add / mask (out of range for rows >= M), once per launch
    unsigned ro[16], ra[16], rm_[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mrow + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const bool ok = m < a.M;
        ro[r] = ok ? (unsigned)((long)m * a.o_sC * 4) : 0x80000000u;
        ra[r] = ok ? (unsigned)((long)m * a.a_sC * 4) : 0x80000000u;
        rm_[r] = ok ? (unsigned)((long)m * a.m_sC * 4) : 0x80000000u;
    }
    int tile = blockIdx.x, b = 0;
    dma(tile, 0);
    const bool flip = a.g[0].transposed != 0;   // transposed taps: tap d reads frame t - (d - 1)
    for (; tile < ia.ntiles; tile += gridDim.x, b ^= 1) {
        wait_dma();
        __syncthreads();                        // image b is complete; nobody reads image b ^ 1 any more
        dma(tile + gridDim.x, b ^ 1);
        const int col = tile * TC + 32 * cg + l31;
        // temporal zero padding: the outer taps of this lane's column exist only inside its sample's frame range
        const int t = (col % L) / V;
        const bool lo_ok = t >= 1, hi_ok = t + 1 < a.T_out;              // frame t - 1 / t + 1 exists
        const float* const im = Im + b * IMG + HALO + 32 * cg + l31 + kh * IW;
        kg_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        constexpr int CH = 8;
        // tap d of group 0 reads the image shv[d] columns away; forward: (d - 1) V, transposed: the mirror image
        const int sh0 = flip ? V : -V;
        const bool ok0 = flip ? hi_ok : lo_ok, ok2 = flip ? lo_ok : hi_ok;
        auto bread = [&](int q) -> float {
            if (q < TAPS0 * (C0 / 2)) {
                const int d = q / (C0 / 2), c2 = q % (C0 / 2);
                if constexpr (TAPS0 == 3) {
                    if (d == 1) return im[2 * c2 * IW];
                    const float v = im[2 * c2 * IW + (d == 0 ? sh0 : -sh0)];
                    return (d == 0 ? ok0 : ok2) ? v : 0.f;
                } else {
                    return im[2 * c2 * IW];
                }
            }
            return im[(C0 + 2 * (q - TAPS0 * (C0 / 2))) * IW];
        };
        float bq[2][CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) bq[0][i] = bread(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < (KS + CH - 1) / CH; ++c) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if ((c + 1) * CH + i < KS) bq[(c + 1) & 1][i] = bread((c + 1) * CH + i);
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (c * CH + i < KS) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[c * CH + i], bq[c & 1][i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue without per-element branches: residual / mask operands come through buffer loads whose offset is
        // out of range for rows >= M and columns >= ncols (zeros), stores likewise are dropped by the hardware
        float av[16], mv[16];
        // (a column >= ncols makes every offset of the lane out of range: 0x80000000 + anything < 2^31 stays >= 2^31)
        const unsigned cb = col < ncols ? (unsigned)col * 4u : 0x80000000u;
#pragma unroll
        for (int r = 0; r < 16; ++r) { av[r] = 0.f; mv[r] = 1.f; }
        if (has_add) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                av[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(addd, (ra[r] | cb) >= 0x80000000u ? 0x80000000u : ra[r] + cb, 0, 0));
        }
        if (has_mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(maskd, (rm_[r] | cb) >= 0x80000000u ? 0x80000000u : rm_[r] + cb, 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + bias[r] + av[r];
            if (act == KG_ACT_LRELU) v = v > 0.f ? v : v * slope;
            else if (act == KG_ACT_TANH) v = tanhf(v);
            if (has_mask) v *= mv[r] > 0.f ? 1.f : slope;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), outd,
                                                  (ro[r] | cb) >= 0x80000000u ? 0x80000000u : ro[r] + cb, 0, 0);
        }
    }
}

// dense (taps, M, Cin) box?  strides (w_sT, w_sO, w_sI) must be a permutation of (1, n_a, n_a n_b)
bool dense_box(const KgConvGroup& g, int M, ImgWeights* w) {
    if (g.w_MB < M) return false;
    long st[3] = {g.taps > 1 ? g.w_sT : -1, g.w_sO, g.w_sI};
    const int ext[3] = {g.taps, M, g.Cin};
    int order[3] = {0, 1, 2};
    if (g.taps == 1) st[0] = (long)M * g.Cin;           // a single tap: place it as the slowest dimension
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (st[order[j]] < st[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    if (st[order[0]] != 1 || st[order[1]] != ext[order[0]] || st[order[2]] != (long)ext[order[0]] * ext[order[1]]) return false;
    w->n0 = ext[order[0]];
    w->n1 = ext[order[1]];
    // floor(x / n) == umulhi(x, magic) for x < 2^16 with magic = floor(2^32 / n) + 1 (n < 2^16): the boxes here have
    // at most 64 * 224 elements
    w->magic0 = (unsigned)((1ull << 32) / (unsigned)w->n0 + 1);
    w->magic1 = (unsigned)((1ull << 32) / (unsigned)w->n1 + 1);
    if ((long)ext[0] * ext[1] * ext[2] >= 65536) return false;
    w->r0 = order[0]; w->r1 = order[1]; w->r2 = order[2];
    return true;
}

template <int RT, int C0, int TAPS0, int C1>
int launch_img(const KgConvArgs* a, const ImgArgs& ia, hipStream_t s) {
    constexpr int TC = 32 * (4 / RT), IW = TC + 2 * HALO;
    constexpr size_t img = (size_t)2 * (C0 + C1) * IW * sizeof(float);
    constexpr size_t wsc = (size_t)(TAPS0 * C0 + C1) * (32 * RT + 1) * sizeof(float);
    constexpr size_t smem = img > wsc ? img : wsc;
    auto kern = kg_conv_img_kernel<RT, C0, TAPS0, C1>;
    static bool attr_done = false;              // idempotent; a race only repeats the call
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    const int rt = kg_cdiv(a->M, 32 * RT);
    int wgs = 256 / rt;                         // one persistent workgroup per CU
    if (wgs > ia.ntiles) wgs = ia.ntiles;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(kern, dim3(wgs, rt), dim3(NT), smem, s, *a, ia);
    return kg_launch_status("kg_conv (image)");
}

}  // namespace

// 0: not eligible; else the variant id (kg_conv_img_launch dispatches on it)
int kg_conv_img_variant(const KgConvArgs* a, KgImgArgs* ia) {
    if (a->ngroups < 1 || a->ngroups > 2 || a->o_tstride > 1 || a->V_out > HALO) return 0;
    const long L = (long)a->T_out * a->V_out, ncols = (long)a->N * L;
    if (ncols >= (1L << 28) || ncols < 32768) return 0;                 // the weight preload + first image cost ~6 us per launch
    if (a->N > 1 && a->o_sN != L) return 0;
    if ((long)a->M * a->o_sC >= (1L << 29) || (a->add && (long)a->M * a->a_sC >= (1L << 29)) ||
        (a->mask && (long)a->M * a->m_sC >= (1L << 29)))
        return 0;                                                        // 32-bit byte offsets in the epilogue
    if (a->add && ((a->N > 1 && a->a_sN != L) || a->a_tstride != 1)) return 0;
    if (a->mask && a->N > 1 && a->m_sN != L) return 0;
    for (int i = 0; i < a->ngroups; ++i) {
        const KgConvGroup& g = a->g[i];
        if (g.vmap || g.t_stride != 1 || g.tap_mode != KG_TAP_TIME || g.T_in != a->T_out || g.V_in != a->V_out) return 0;
        if (a->N > 1 && g.x_sN != L) return 0;
        if (g.x_sC % 4 != 0 || ((uintptr_t)g.x & 15) != 0) return 0;   // 16-byte DMA segments
        if ((long)g.Cin * g.x_sC >= (1L << 28)) return 0;
        if (!dense_box(g, a->M, &ia->w[i])) return 0;
    }
    if (a->ngroups == 2 && (a->g[1].taps != 1 || a->g[1].transposed)) return 0;
    const int c0 = a->g[0].Cin, t0 = a->g[0].taps, c1 = a->ngroups == 2 ? a->g[1].Cin : 0;
    int v = 0;
    if (a->M <= 64 && a->M > 32) {
        if (c0 == 64 && t0 == 3 && c1 == 32) v = 1;        // D1 tail
        else if (c0 == 64 && t0 == 3 && c1 == 0) v = 2;    // D1 temporal conv, transposed or alone
    } else if (a->M <= 32) {
        if (c0 == 32 && t0 == 3 && c1 == 0) v = 3;         // D0 tail / its transposed temporal conv
        else if (c0 == 64 && t0 == 1 && c1 == 0) v = 4;    // D1 residual conv, transposed
    }
    if (v == 0) return 0;
    const int tc = (v <= 2) ? 64 : 128;
    ia->ntiles = kg_cdiv(ncols, tc);
    return v;
}

int kg_conv_img_launch(const KgConvArgs* a, int variant, const KgImgArgs& ia, hipStream_t s) {
    switch (variant) {
        case 1: return launch_img<2, 64, 3, 32>(a, ia, s);
        case 2: return launch_img<2, 64, 3, 0>(a, ia, s);
        case 3: return launch_img<1, 32, 3, 0>(a, ia, s);
        default: return launch_img<1, 64, 1, 0>(a, ia, s);
    }
}
