// kg_map: the generator's label embedding + mapping network (reference generator.py:22-37 Mapping_Net, :80-85 label_emb +
// cat + mlp) as kernels of this library - nn.Linear(D, D) + LeakyReLU(0.2), four (mlp4) or eight (mlp8) times, on the n
// latents of a step.  Stock PyTorch-ROCm spent 57 launches and 0.25 ms per iteration here (hipBLASLt GEMM + LeakyReLU per
// layer forward; LeakyReLU backward, two GEMMs, a column sum and two adds per layer backward; embedding, cat, scatter)
// for 0.1 % of the iteration's FLOPs: every one of those launches is pure latency.
//
//   kg_linear_fwd : y[n, o] = act( sum_i xin[n, i] W[o, i] + b[o] ),  xin = [emb[labels[n], 0:J) | x[n, 0:Din-J)]
//                   (the embedding lookup and the cat of generator.py:80-82 are folded into the first layer's operand
//                    load; J = 0 for the other layers) - one launch per layer
//   kg_linear_bwd : g' = g * act'(y)  (LeakyReLU derivative on the layer's OUTPUT y)
//                   gx[n, i] = sum_o g'[n, o] W[o, i]        (only the first `gx_cols` columns; the first layer needs the
//                                                              J embedding columns only: z carries no gradient)
//                   dW[o, i] (+)= sum_n g'[n, o] xin[n, i],   db[o] (+)= sum_n g'[n, o]
//                   - ONE launch per layer: the workgroups of the two products share the grid
//   kg_embed_bwd  : demb[l, j] (+)= sum_{n : labels[n] = l} gx[n, j]   (samples visited in index order: deterministic)
//
// All contractions run on v_mfma_f32_32x32x2_f32 (exact fp32).  Operands go straight from global memory / L2 into MFMA
// registers: an operand whose contraction index runs along its rows (W and xin in the forward pass, g in gx) is read with
// one 4- / 2- / 1-float load per lane covering that many consecutive k (the k -> MFMA-step assignment is free as long as
// both operands agree), an operand whose rows ARE the contraction index is read one coalesced row per k.  K is split over
// the sixteen waves of a workgroup (interleaved chunks; 1024 threads, 64 KB of static LDS), the partial tiles are summed through LDS in wave order
// (deterministic); no partial slabs in HBM, no second launch.
#include "kg_common.h"

namespace {

constexpr int NTM = 1024;           // 16 waves
constexpr int NWV = 16;
constexpr int NJB = 6;              // chunks (forward) / sample pairs (dW) whose loads a wave has in flight at once
constexpr int NJA = 5;              // the same for gx (three operands per chunk; 16 waves x 128 registers)
constexpr unsigned OOB = 0x80000000u;
typedef float kg_f32x4 __attribute__((ext_vector_type(4)));
typedef float kg_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, long bytes) {
    // (ranges are clamped to 2 GiB - 1: every tensor of this path is a few MB)
    return __builtin_amdgcn_make_buffer_rsrc(kg_uniform_ptr(p), 0, p ? (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes) : 0, 0x00020000);
}
__device__ __forceinline__ float load1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// VW consecutive floats at byte offset `off` (out of range: zeros)
template <int VW>
__device__ __forceinline__ void load_vec(__amdgpu_buffer_rsrc_t r, unsigned off, float (&v)[VW]) {
    if constexpr (VW == 4) {
        const kg_f32x4 t = __builtin_bit_cast(kg_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else if constexpr (VW == 2) {
        const kg_f32x2 t = __builtin_bit_cast(kg_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
        v[0] = t[0]; v[1] = t[1];
    } else {
        v[0] = load1(r, off);
    }
}
// elements of a vector that ran past the end of its row (k + t >= len) are zero
template <int VW>
__device__ __forceinline__ void mask_tail(float (&v)[VW], int k, int len) {
    if (k + VW > len) {
#pragma unroll
        for (int t = 0; t < VW; ++t)
            if (k + t >= len) v[t] = 0.f;
    }
}

// These launches are latency, not throughput: 36-126 workgroups, a few microseconds each, operands cold in L2.  Every
// wave therefore issues ALL loads of its share of the contraction before the first MFMA (one memory round trip), the
// contraction is split over the 16 waves of a workgroup, and the partial tiles meet in LDS where wave w sums accumulator
// register w of all waves (fixed order: deterministic) and stores that register's two rows.
__device__ __forceinline__ float reduce_rows(const kg_f32x16& acc, float* red, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) v += red[(w * 16 + wave) * 64 + lane];
    return v;
}
__device__ __forceinline__ int row_of_reg(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// the layer input xin[n, k .. k+VW): embedding columns (k < J) from emb[labels[n]], the rest from x; k and J are multiples
// of VW (host-checked): a vector never straddles the seam
template <int VW>
__device__ __forceinline__ void load_xin(const KgLinearArgs& a, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t re, int n, long lab, int k,
                                         bool ok, float (&v)[VW]) {
    const bool in_emb = k < a.J;
    const unsigned off_x = (unsigned)((long)n * a.x_ld + (k - a.J)) * 4u;
    const unsigned off_e = (unsigned)(lab * a.J + k) * 4u;
    load_vec<VW>(in_emb ? re : rx, (ok && k < a.Din) ? (in_emb ? off_e : off_x) : OOB, v);
    mask_tail<VW>(v, k, a.Din);
}

// ---- forward: D[n, o] tile 32 x 32; A = xin rows (m = n), B = W rows (column = o), both read along k -------------------
template <int VW>
__global__ __launch_bounds__(NTM) void kg_linear_fwd_kernel(const KgLinearArgs a) {
    __shared__ float red[NWV * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(a.x, (long)a.N * a.x_ld * 4), re = rsrc_of(a.emb, (long)a.L * a.J * 4);
    const __amdgpu_buffer_rsrc_t rw = rsrc_of(a.w, (long)a.Dout * a.Din * 4), rb = rsrc_of(a.bias, (long)a.Dout * 4);
    const int n = n0 + l32, o = o0 + l32;
    const bool n_ok = n < a.N, o_ok = o < a.Dout;
    const float bias = load1(rb, o_ok ? (unsigned)o * 4u : OOB);          // (absent bias: zero-length buffer, reads 0)
    long lab = 0;
    if (a.J > 0 && n_ok) {
        lab = a.labels[n];
        if (lab < 0 || lab >= a.L) lab = -1;        // out of range: the sample's embedding part reads as NaN below
    }
    kg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int CH = 2 * VW;                      // k per chunk: half-wave h takes [kc + VW h, kc + VW h + VW)
    const int nch = (a.Din + CH - 1) / CH;
    for (int c0 = wave; c0 < nch; c0 += NWV * NJB) {
        float av[NJB][VW], bv[NJB][VW];
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int c = c0 + NWV * j;
            const int k = c * CH + VW * h;
            const bool live = c < nch;
            load_xin<VW>(a, rx, re, n, lab < 0 ? 0 : lab, k, live && n_ok, av[j]);
            load_vec<VW>(rw, (live && o_ok && k < a.Din) ? (unsigned)((long)o * a.Din + k) * 4u : OOB, bv[j]);
            mask_tail<VW>(bv[j], k, a.Din);
            if (lab < 0 && k < a.J) {
#pragma unroll
                for (int t = 0; t < VW; ++t) av[j][t] = __builtin_nanf("");
            }
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j)
#pragma unroll
            for (int t = 0; t < VW; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][t], bv[j][t], acc, 0, 0, 0);
    }
    const float v = reduce_rows(acc, red, wave, lane);
    // D layout: column (lane & 31) = o, row of register r = n - n0: a store covers 32 consecutive o
    const int nn = n0 + row_of_reg(wave, h);
    if (o_ok && nn < a.N) a.y[(long)nn * a.y_ld + o] = kg_act(v + bias, a.act, a.slope);
}

// ---- backward: job A (gx) and job B (dW, db) share one grid ------------------------------------------------------------
template <int VW>
__global__ __launch_bounds__(NTM) void kg_linear_bwd_kernel(const KgLinearArgs a, int nwg_a, int itiles_a, int igroups_b) {
    __shared__ float red[NWV * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const __amdgpu_buffer_rsrc_t rg = rsrc_of(a.g, (long)a.N * a.g_ld * 4), ry = rsrc_of(a.y, (long)a.N * a.y_ld * 4);
    kg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float slope = a.slope;
    const bool lrelu = a.act == KG_ACT_LRELU;
    auto dact = [&](float gv, float yv) { return (lrelu && !(yv > 0.f)) ? gv * slope : gv; };
    if ((int)blockIdx.x < nwg_a) {
        // gx[n, i] = sum_o g'[n, o] W[o, i]: A = g' rows (m = n, read along o), B = W[o, :] (column = i, one row per o)
        const int it = blockIdx.x % itiles_a, nt = blockIdx.x / itiles_a;
        const int n0 = nt * 32, i0 = it * 32;
        const __amdgpu_buffer_rsrc_t rw = rsrc_of(a.w, (long)a.Dout * a.Din * 4);
        const int n = n0 + l32, i = i0 + l32;
        const bool n_ok = n < a.N, i_ok = i < a.gx_cols;
        constexpr int CH = 2 * VW;
        const int nch = (a.Dout + CH - 1) / CH;
        for (int c0 = wave; c0 < nch; c0 += NWV * NJA) {
            float gv[NJA][VW], yv[NJA][VW], bv[NJA][VW];
#pragma unroll
            for (int j = 0; j < NJA; ++j) {
                const int c = c0 + NWV * j;
                const int k = c * CH + VW * h;
                const bool live = c < nch && k < a.Dout;
                load_vec<VW>(rg, (live && n_ok) ? (unsigned)((long)n * a.g_ld + k) * 4u : OOB, gv[j]);
                load_vec<VW>(ry, (live && n_ok) ? (unsigned)((long)n * a.y_ld + k) * 4u : OOB, yv[j]);
#pragma unroll
                for (int t = 0; t < VW; ++t)
                    bv[j][t] = load1(rw, (live && i_ok && k + t < a.Dout) ? (unsigned)((long)(k + t) * a.Din + i) * 4u : OOB);
            }
#pragma unroll
            for (int j = 0; j < NJA; ++j) {
                const int k = (c0 + NWV * j) * CH + VW * h;
#pragma unroll
                for (int t = 0; t < VW; ++t) {
                    const float av = k + t < a.Dout ? dact(gv[j][t], yv[j][t]) : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j][t], acc, 0, 0, 0);
                }
            }
        }
        const float v = reduce_rows(acc, red, wave, lane);
        const int nn = n0 + row_of_reg(wave, h);
        if (i_ok && nn < a.N) a.gx[(long)nn * a.gx_ld + i] = v;
        return;
    }
    // dW[o, i] (+)= sum_n g'[n, o] xin[n, i]: A = g'[n, :] (m = o, one row per n), B = xin[n, :] (column = i).  Four i tiles
    // per workgroup (wave & 3), the samples split four ways (wave >> 2): pair s = (samples 2s, 2s + 1) goes to group s & 3
    const int wg = (int)blockIdx.x - nwg_a;
    const int ot = wg / igroups_b, ig = wg - ot * igroups_b;
    const int itl = wave & 3, kg = wave >> 2;
    const int o0 = ot * 32, i0 = (ig * 4 + itl) * 32;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(a.x, (long)a.N * a.x_ld * 4), re = rsrc_of(a.emb, (long)a.L * a.J * 4);
    const int o = o0 + l32, i = i0 + l32;
    const bool o_ok = o < a.Dout, i_ok = i < a.Din;
    const bool from_emb = i < a.J;
    float bsum = 0.f;
    const int npairs = (a.N + 1) / 2;
    for (int s0 = kg; s0 < npairs; s0 += 4 * NJB) {
        float gv[NJB], yv[NJB], xv[NJB];
        long lab[NJB];
        if (from_emb) {
#pragma unroll
            for (int j = 0; j < NJB; ++j) {
                const int n = 2 * (s0 + 4 * j) + h;
                lab[j] = n < a.N ? (long)a.labels[n] : -1;
            }
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int n = 2 * (s0 + 4 * j) + h;
            const bool n_ok = n < a.N;
            gv[j] = load1(rg, (n_ok && o_ok) ? (unsigned)((long)n * a.g_ld + o) * 4u : OOB);
            yv[j] = load1(ry, (n_ok && o_ok) ? (unsigned)((long)n * a.y_ld + o) * 4u : OOB);
            if (from_emb) xv[j] = load1(re, (n_ok && lab[j] >= 0 && lab[j] < a.L) ? (unsigned)(lab[j] * a.J + i) * 4u : OOB);
            else          xv[j] = load1(rx, (n_ok && i_ok) ? (unsigned)((long)n * a.x_ld + (i - a.J)) * 4u : OOB);
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const float av = dact(gv[j], yv[j]);            // (absent samples loaded zeros)
            bsum += av;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xv[j], acc, 0, 0, 0);
        }
    }
    // the four sample groups of an i tile meet in LDS; group kg finishes registers 4 kg .. 4 kg + 3 (rows o) of its tile
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (a.dw && i_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * kg + q;
            float v = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) v += red[((k2 * 4 + itl) * 16 + r) * 64 + lane];
            const int oo = o0 + row_of_reg(r, h);
            if (oo < a.Dout) {
                float* p = a.dw + (long)oo * a.Din + i;
                *p = (a.accumulate ? *p : 0.f) + v;
            }
        }
    }
    if (a.db == nullptr || ig != 0) return;           // (uniform)
    // db[o]: the half-waves and the four sample groups of i tile 0 hold partial sums
    __syncthreads();
    bsum += __shfl_xor(bsum, 32, 64);
    if (itl == 0 && h == 0) red[kg * 32 + l32] = bsum;
    __syncthreads();
    if (wave == 0 && h == 0 && o_ok) {
        const float t = (red[l32] + red[32 + l32]) + (red[64 + l32] + red[96 + l32]);
        a.db[o] = (a.accumulate ? a.db[o] : 0.f) + t;
    }
}

// demb[l, j] (+)= sum over the samples of class l, in index order.  The labels go through LDS and the gradient rows are
// read unconditionally (selected afterwards): every load of a thread is independent of the others - a compare-then-load
// loop was one memory round trip per sample.
__global__ __launch_bounds__(256) void kg_embed_bwd_kernel(const KgLinearArgs a) {
    __shared__ int lab_s[256];
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool ok = e < a.L * a.J;
    const int l = ok ? e / a.J : 0, j = ok ? e - l * a.J : 0;
    float s = 0.f;
    for (int n0 = 0; n0 < a.N; n0 += 256) {
        const int nb = a.N - n0 < 256 ? a.N - n0 : 256;
        __syncthreads();
        if ((int)threadIdx.x < nb) lab_s[threadIdx.x] = (int)a.labels[n0 + threadIdx.x];
        __syncthreads();
        for (int q0 = 0; q0 < nb; q0 += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = (q0 + q < nb) ? a.gx[(long)(n0 + q0 + q) * a.gx_ld + j] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += (q0 + q < nb && lab_s[q0 + q] == l) ? v[q] : 0.f;
        }
    }
    if (ok) a.demb[e] = (a.accumulate ? a.demb[e] : 0.f) + s;
}

int validate(const KgLinearArgs* a, const char* what) {
    KG_REQUIRE(a != nullptr, "%s: null args", what);
    KG_REQUIRE(a->N > 0 && a->Din > 0 && a->Dout > 0, "%s: bad dims N=%d Din=%d Dout=%d", what, a->N, a->Din, a->Dout);
    KG_REQUIRE(a->J >= 0 && a->J <= a->Din, "%s: J=%d outside [0, Din=%d]", what, a->J, a->Din);
    KG_REQUIRE(a->J == 0 || (a->emb != nullptr && a->labels != nullptr && a->L > 0), "%s: embedding columns without table / labels", what);
    KG_REQUIRE(a->J == a->Din || a->x != nullptr, "%s: null x", what);
    KG_REQUIRE(a->x_ld >= a->Din - a->J, "%s: x_ld=%ld < %d", what, (long)a->x_ld, a->Din - a->J);
    KG_REQUIRE(a->act == KG_ACT_NONE || a->act == KG_ACT_LRELU, "%s: act=%d (none or LeakyReLU)", what, a->act);
    KG_REQUIRE((long)a->N * (a->x_ld > a->Dout ? a->x_ld : a->Dout) < (1L << 28) && (long)a->Dout * a->Din < (1L << 28),
               "%s: operand too large for 32-bit offsets", what);
    // every row-major operand the kernels address with 32-bit byte offsets (round-4 ADVICE: g, y, gx and the table too)
    KG_REQUIRE(a->y_ld >= 0 && a->g_ld >= 0 && a->gx_ld >= 0 && (long)a->N * a->y_ld < (1L << 28) && (long)a->N * a->g_ld < (1L << 28) &&
               (long)a->N * a->gx_ld < (1L << 28) && (long)a->L * a->J < (1L << 28),
               "%s: y / g / gx / embedding table too large for 32-bit offsets", what);
    return 0;
}

// widest vector (4, 2 or 1 floats) that divides every given row length: rows then start on a vector boundary
int vec_width(long ld_a, long ld_b, long ld_c) {
    auto ok = [&](int v) { return ld_a % v == 0 && ld_b % v == 0 && ld_c % v == 0; };
    if (ok(4)) return 4;
    if (ok(2)) return 2;
    return 1;
}

}  // namespace

extern "C" int kg_linear_fwd(const KgLinearArgs* a, void* stream) {
    if (int rc = validate(a, "kg_linear_fwd")) return rc;
    KG_REQUIRE(a->w != nullptr && a->y != nullptr, "kg_linear_fwd: null w / y");
    KG_REQUIRE(a->y_ld >= a->Dout, "kg_linear_fwd: y_ld=%ld < Dout=%d", (long)a->y_ld, a->Dout);
    // rows of x (ld x_ld, shifted by J), of emb (ld J) and of W (ld Din) must start on a vector boundary
    const bool al16 = ((uintptr_t)a->x % 16 == 0) && ((uintptr_t)a->w % 16 == 0) && (a->J == 0 || (uintptr_t)a->emb % 16 == 0);
    const int vw = al16 ? vec_width(a->x_ld, a->Din, a->J) : 1;      // (J = 0 divides)
    dim3 grid(kg_cdiv(a->Dout, 32), kg_cdiv(a->N, 32));
    hipStream_t s = (hipStream_t)stream;
    if (vw == 4)      hipLaunchKernelGGL(kg_linear_fwd_kernel<4>, grid, dim3(NTM), 0, s, *a);
    else if (vw == 2) hipLaunchKernelGGL(kg_linear_fwd_kernel<2>, grid, dim3(NTM), 0, s, *a);
    else              hipLaunchKernelGGL(kg_linear_fwd_kernel<1>, grid, dim3(NTM), 0, s, *a);
    return kg_launch_status("kg_linear_fwd");
}

extern "C" int kg_linear_bwd(const KgLinearArgs* a, void* stream) {
    if (int rc = validate(a, "kg_linear_bwd")) return rc;
    KG_REQUIRE(a->g != nullptr && a->y != nullptr && a->w != nullptr, "kg_linear_bwd: null g / y / w");
    KG_REQUIRE(a->g_ld >= a->Dout && a->y_ld >= a->Dout, "kg_linear_bwd: g_ld / y_ld < Dout");
    KG_REQUIRE(a->gx_cols >= 0 && a->gx_cols <= a->Din, "kg_linear_bwd: gx_cols=%d", a->gx_cols);
    KG_REQUIRE(a->gx_cols == 0 || (a->gx != nullptr && a->gx_ld >= a->gx_cols), "kg_linear_bwd: gx / gx_ld");
    KG_REQUIRE(a->dw != nullptr || a->db != nullptr || a->gx_cols > 0, "kg_linear_bwd: nothing to compute");
    const bool al16 = ((uintptr_t)a->g % 16 == 0) && ((uintptr_t)a->y % 16 == 0);
    const int vw = al16 ? vec_width(a->g_ld, a->y_ld, 0) : 1;
    const int itiles_a = kg_cdiv(a->gx_cols, 32);
    const int nwg_a = itiles_a * kg_cdiv(a->N, 32);
    const int igroups_b = kg_cdiv(kg_cdiv(a->Din, 32), 4);
    const int nwg_b = (a->dw || a->db) ? kg_cdiv(a->Dout, 32) * igroups_b : 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nwg_a + nwg_b);
    if (vw == 4)      hipLaunchKernelGGL(kg_linear_bwd_kernel<4>, grid, dim3(NTM), 0, s, *a, nwg_a, itiles_a > 0 ? itiles_a : 1, igroups_b);
    else if (vw == 2) hipLaunchKernelGGL(kg_linear_bwd_kernel<2>, grid, dim3(NTM), 0, s, *a, nwg_a, itiles_a > 0 ? itiles_a : 1, igroups_b);
    else              hipLaunchKernelGGL(kg_linear_bwd_kernel<1>, grid, dim3(NTM), 0, s, *a, nwg_a, itiles_a > 0 ? itiles_a : 1, igroups_b);
    return kg_launch_status("kg_linear_bwd");
}

extern "C" int kg_embed_bwd(const KgLinearArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr, "kg_embed_bwd: null args");
    KG_REQUIRE(a->N > 0 && a->L > 0 && a->J > 0, "kg_embed_bwd: bad dims N=%d L=%d J=%d", a->N, a->L, a->J);
    KG_REQUIRE(a->gx != nullptr && a->labels != nullptr && a->demb != nullptr && a->gx_ld >= a->J, "kg_embed_bwd: null pointer / gx_ld");
    hipLaunchKernelGGL(kg_embed_bwd_kernel, dim3(kg_cdiv((long)a->L * a->J, 256)), dim3(256), 0, (hipStream_t)stream, *a);
    return kg_launch_status("kg_embed_bwd");
}
