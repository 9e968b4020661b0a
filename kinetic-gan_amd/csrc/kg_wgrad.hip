// kg_wgrad: weight gradient of the tap GEMM,
//   dW(d, m, c) = sum_j G[m, j] * X[c (+ d*Cin), src(j, d)]
// a GEMM whose contraction runs over the batch's columns j = (n, t, v).  64x64 output tile per
// workgroup (4 waves x one 32x32 v_mfma_f32_32x32x2_f32 tile), both operand tiles staged in LDS
// column-contiguous ([row][64+1], conflict-free for the row-per-lane fragment reads), the column
// range split across workgroups into partial slabs that a second kernel sums in a fixed order
// (deterministic; no atomics).
// Reference op covered: the weight half of aten::convolution_backward for tgcn.py:61,
// discriminator.py:99-105,115-120 and generator.py:134-140,154-159.
#include "kg_common.h"

namespace {

constexpr int BM = 64, BN = 64, BJ = 64, NT = 256;

struct Plan { int tiles_m, tiles_n, splits, cols_per_split; };

Plan make_plan(const KgWgradArgs* a) {
    Plan p;
    const long ncols = (long)a->N * a->T_out * a->V_out;
    p.tiles_m = kg_cdiv(a->M, BM);
    p.tiles_n = kg_cdiv(a->Cin, BN);
    const long tiles = (long)p.tiles_m * p.tiles_n * a->taps;
    const int chunks = kg_cdiv(ncols, BJ);
    long s = (768 + tiles - 1) / tiles;
    if (s > chunks) s = chunks;
    if (s > 128) s = 128;
    if (s < 1) s = 1;
    int cps = kg_cdiv(chunks, s) * BJ;
    p.cols_per_split = cps;
    p.splits = kg_cdiv(ncols, cps);
    return p;
}

__global__ __launch_bounds__(NT) void kg_wgrad_kernel(const KgWgradArgs a, const Plan p) {
    __shared__ float Gs[2][BM][BJ + 1];
    __shared__ float Xs[2][BN][BJ + 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int m0 = (tile / p.tiles_n) * BM;
    const int c0 = (tile % p.tiles_n) * BN;
    const int d = blockIdx.y;
    const int split = blockIdx.z;
    const int ncols = a.N * a.T_out * a.V_out;
    const int L = a.T_out * a.V_out;
    const int jbeg = split * p.cols_per_split;
    const int jend = min(ncols, jbeg + p.cols_per_split);
    const int shift = (a.tap_mode == KG_TAP_TIME) ? d - (a.taps - 1) / 2 : 0;
    const int choff = (a.tap_mode == KG_TAP_CHANBLOCK) ? d * a.Cin : 0;

    const int cj = tid & (BJ - 1);   // this thread's column inside a chunk
    const int r0 = tid / BJ;         // first row it stages (rows r0, r0+4, ...)
    constexpr int RPT = BM / (NT / BJ);   // rows per thread per operand (16)

    kg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float greg[RPT], xreg[RPT];
    // raw buffer loads: wave-uniform descriptor (tensor base + this tile's first row), 32-bit byte offsets,
    // out-of-range offset == reads as 0 (rows beyond M / Cin, padding frames, dropped vertices, ragged tail)
    constexpr unsigned RANGE = 0x80000000u, OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(a.g + (long)m0 * a.g_sC), 0, (int)RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        kg_uniform_ptr(a.x + (long)(choff + c0) * a.x_sC), 0, (int)RANGE, 0x00020000);
    constexpr int RSTEP = NT / BJ;
    const int g_nvalid = (a.M - m0 - r0 + RSTEP - 1) / RSTEP;       // staged rows i < nvalid are inside the tensor
    const int x_nvalid = (a.Cin - c0 - r0 + RSTEP - 1) / RSTEP;
    const unsigned g_step = (unsigned)(RSTEP * a.g_sC * 4), x_step = (unsigned)(RSTEP * a.x_sC * 4);
    auto fetch = [&](int jc) {       // global -> registers (software pipeline: overlaps the MFMAs below)
        const int j = jc + cj;
        unsigned gb = OOB, xb = OOB;
        if (j < jend) {
            int n = j / L, r = j - n * L;
            int to = r / a.V_out, vo = r - to * a.V_out;
            gb = (unsigned)(((long)r0 * a.g_sC + (long)n * a.g_sN + r) * 4);
            int vi = a.vmap ? a.vmap[vo] : vo;
            int ti = to * a.t_stride + shift;
            if (vi >= 0 && ti >= 0 && ti < a.T_in)
                xb = (unsigned)(((long)r0 * a.x_sC + (long)n * a.x_sN + (long)ti * a.V_in + vi) * 4);
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            greg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                gr, i < g_nvalid ? gb + i * g_step : OOB, 0, 0));
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            xreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                xr, i < x_nvalid ? xb + i * x_step : OOB, 0, 0));
    };
    auto stash = [&](int b) {
        float* pg = &Gs[b][r0][cj];
        float* px = &Xs[b][r0][cj];
#pragma unroll
        for (int i = 0; i < RPT; ++i) pg[i * RSTEP * (BJ + 1)] = greg[i];
#pragma unroll
        for (int i = 0; i < RPT; ++i) px[i * RSTEP * (BJ + 1)] = xreg[i];
    };

    if (jbeg < jend) {
        fetch(jbeg);
        stash(0);
        __syncthreads();
        int b = 0;
        for (int jc = jbeg; jc < jend; jc += BJ, b ^= 1) {
            const bool more = jc + BJ < jend;
            if (more) fetch(jc + BJ);
            __builtin_amdgcn_sched_barrier(0);   // loads -> MFMAs -> (wait + LDS writes), see kg_conv.hip
#pragma unroll
            for (int kk = 0; kk < BJ; kk += 2) {
                const int col = kk + (lane >> 5);
                float av = Gs[b][wm * 32 + (lane & 31)][col];
                float bv = Xs[b][wn * 32 + (lane & 31)][col];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) stash(b ^ 1);
            __syncthreads();
        }
    }

    // partial slab [split][tap][M][Cin]
    float* slab = a.ws + ((long)split * a.taps + d) * (long)a.M * a.Cin;
    const int c = c0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && c < a.Cin) slab[(long)m * a.Cin + c] = acc[r];
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
    a.dw[(long)d * a.w_sT + (long)m * a.w_sO + (long)c * a.w_sI] = s;
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
    // 32-bit byte offsets inside one 64-row tile (buffer-load addressing)
    const long gspan = 64L * a->g_sC + (long)(a->N - 1) * a->g_sN + (long)a->T_out * a->V_out;
    const long xspan = 64L * a->x_sC + (long)(a->N - 1) * a->x_sN + (long)a->T_in * a->V_in;
    KG_REQUIRE(a->g_sC >= 0 && a->g_sN >= 0 && a->x_sC >= 0 && a->x_sN >= 0 && gspan < (1L << 29) && xspan < (1L << 29),
               "kg_wgrad: tensors too large for 32-bit tile offsets (%ld / %ld elements)", gspan, xspan);
    return 0;
}

}  // namespace

extern "C" int64_t kg_wgrad_workspace_bytes(const KgWgradArgs* a) {
    if (validate(a) != 0) return -1;
    Plan p = make_plan(a);
    return (int64_t)p.splits * a->taps * a->M * a->Cin * (int64_t)sizeof(float);
}

extern "C" int kg_wgrad(const KgWgradArgs* a, void* stream) {
    if (int rc = validate(a)) return rc;
    KG_REQUIRE(a->g && a->x && a->dw && a->ws, "kg_wgrad: null pointer");
    Plan p = make_plan(a);
    const int64_t need = (int64_t)p.splits * a->taps * a->M * a->Cin * (int64_t)sizeof(float);
    KG_REQUIRE(a->ws_bytes >= need, "kg_wgrad: workspace %ld < %ld bytes", (long)a->ws_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p.tiles_m * p.tiles_n, a->taps, p.splits);
    hipLaunchKernelGGL(kg_wgrad_kernel, grid, dim3(NT), 0, s, *a, p);
    if (int rc = kg_launch_status("kg_wgrad")) return rc;
    const long per = (long)a->taps * a->M * a->Cin;
    hipLaunchKernelGGL(kg_wgrad_reduce_kernel, dim3(kg_cdiv(per, 64)), dim3(256), 0, s, *a, p.splits);
    return kg_launch_status("kg_wgrad_reduce");
}
