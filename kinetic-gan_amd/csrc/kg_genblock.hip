// Fused generator block (round 6, ABI v9): generator.st_gcn.forward (generator.py:168-182) and its backward as ONE launch
// each for the blocks whose per-sample working set fits LDS (the generator's last four: <= 128 input channels).
//
// Why: the staged form (kg_conv on the input grid -> kg_gen_expand -> kg_conv (tcn) -> kg_bn_fwd_many -> kg_affine_act;
// backward kg_gen_tail_stats -> _apply -> kg_conv^T -> kg_gen_fold -> kg_conv^T) is 5-7 launches per block and direction for
// a few MFLOP - every launch is a chain of dependent memory round trips (arguments -> operands -> result) of 5-12 us while
// the arithmetic is under a microsecond.  A grid barrier does not help: on this chip it costs more than a kernel boundary
// inside a hipGraph (profiles/r06_grid_barrier_probe.log: 4.0 us per barrier of 256 workgroups against 1.6-2.0 us per
// graph node).  What helps is to remove the round trips: everything between two BatchNorm statistics is per-sample
// arithmetic, so ONE workgroup carries ONE sample through the whole block with every intermediate in LDS; the only
// cross-sample step - the BatchNorm statistics - is finished by the last workgroup to arrive (ticket counter, partials
// through agent-scope stores / loads, merged in sample order: deterministic), and the block's normalise + noise +
// activation rides at the front of the NEXT block's launch.
//
// Arithmetic: the two channel contractions of a block run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: weights
// straight from L2 into the A operand - a lane reads 8 consecutive k of its row with two 16-byte loads - features from
// LDS) when they have >= 17 rows, as a column-per-thread VALU loop with LDS-broadcast weights below that (the 3-channel
// blocks); up-sampling / aggregation (U A_k, <= 33 terms per output) and its adjoint as small matrix-core products with
// functor-addressed LDS operands (gb_mm).  All fp32, FMA chains as in the staged kernels (other summation order only inside
// the BatchNorm merge: per-sample two-pass partials instead of 4096-element ones).
//
// Geometry: the kernels are templates over a geometry policy.  GbGeo<Cin, C, Kp, Tc, Vc, V, rep, residual, BatchNorm> makes
// every extent a compile-time constant (divisions by constants, the matrix-core / VALU choice and the residual kind resolved
// at compile time) - the six instantiations of GB_GEOMETRIES are the fusable blocks of the NTU and Human3.6M generators and
// run 1.5-2.5x faster than the run-time form GbRt, which serves every other shape (KG_GB_RT=1 forces it).
#include "kg_common.h"

namespace {

#ifndef KG_GB_UB
#define KG_GB_UB 4
#endif
#ifndef KG_GB_RD
#define KG_GB_RD 4
#endif
constexpr int NT = 512;           // 8 waves: two per SIMD hide each other's LDS / L2 latency
constexpr int NW = NT / 64;
constexpr int UB = KG_GB_UB;           // elements a thread has in flight in the streaming loops (all loads before the first use)
constexpr int GB_MAX_LDS = 150 * 1024;
constexpr int GB_VALU_MAXM = 32, GB_VALU_MAXMK = 2048;
typedef float gb_f4 __attribute__((ext_vector_type(4)));

// ---- phase stamps (debug builds only: -DKG_GB_STAMP; tools/time_genblock.py) ---------------------------------------------
#ifdef KG_GB_STAMP
__device__ long long kg_gb_stamps[2][16];
#define GB_STAMP(dir_, i_) do { if (blockIdx.x == 0 && threadIdx.x == 0) kg_gb_stamps[dir_][i_] = wall_clock64(); } while (0)
#else
#define GB_STAMP(dir_, i_) do { } while (0)
#endif

// ---- LDS layout (float offsets; host-computed) ------------------------------------------------------------------------
struct GbLayout {
    int x, yc, uo, z, r, bs, us, brs, wl, total;     // forward
    int du, dr, gz, gyc, gid, gx;                                   // backward (x, yc, uo, z, r unused there)
    int Nc, Nf, ZP, Mg, Mh;
    int mfma0, mfma1;                                               // which path the two contractions take
    FastDiv dNc, dNf, dV, dVc, dTc, dTcV, dTcVc;
};

__device__ __forceinline__ float gb_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// k -> operand offset maps (arithmetic: a table in LDS put a dependent LDS read in front of every operand read)
struct KLin { int s; __device__ __forceinline__ int operator()(int k) const { return k * s; } };                   // k * s
struct KTap {                      // k = 3 c + d  ->  c * sc + (flip ? 2 - d : d) * sd
    int sc, sd, flip;
    __device__ __forceinline__ int operator()(int k) const {
        const int c = (k * 0xAAAB) >> 17, d = k - 3 * c;         // k / 3 for k < 2^15
        return c * sc + (flip ? 2 - d : d) * sd;
    }
};
struct KSplit { int M0, s; __device__ __forceinline__ int operator()(int k) const { return (k < M0 ? k : k - M0) * s; } };

// Which path a contraction of M rows, K terms and N columns takes.  Matrix cores when the rows fill most of a 32-row tile, or
// when there are enough columns to give every wave a tile (N >= 64: padding 3-12 rows to 32 is then cheaper than the
// column-per-thread VALU loop, which keeps only N of the 512 threads busy).  The transposed-operand form (AT: 4-byte operand
// loads) zero-fills a ragged K, the row-major form reads 16 bytes at a time and needs K % 16 == 0.
__host__ __device__ constexpr bool gb_use_mfma(int M, int K, int N, bool at) {
    return (at || K % 16 == 0) && (M >= 17 || N >= 64);
}

// A(m, k) = (second ? p1 : p0)[m' * sm + kofs]
struct GbA {
    const float* p0; const float* p1;
    int M0;          // split point: rows (split_m) or k (else); a value >= M / K means "no split"
    int sm;          // element stride between rows m
    int split_m;
};
__device__ __forceinline__ const float* gb_aptr(const GbA& A, int m, int k, int kofs) {
    if (A.split_m) return (m < A.M0 ? A.p0 + (long)m * A.sm : A.p1 + (long)(m - A.M0) * A.sm) + kofs;
    return (k < A.M0 ? A.p0 : A.p1) + (long)m * A.sm + kofs;
}

// out[m * op + j] = bias[m] + sum_k A(m, k) * Bsrc[kb[k] + j]   for m < M, j < N;  K a multiple of 16.
// AT = false: A rows are contiguous in k (ka[k] == k): a lane loads 8 consecutive k with two 16-byte loads.
// AT = true : consecutive lanes (rows m) are adjacent in memory, one 4-byte load per k.
// The k -> MFMA-step assignment is free as long as both operands agree: in a chunk of 16 k, step s multiplies
// k = 8 * (lane >> 5) + s of the A lane (row lane & 31) with the same k of the B lane (column lane & 31).
template <bool AT, typename KA, typename KB>
__device__ __forceinline__ void gb_gemm_mfma(const GbA& A, const KA ka, int M, int K, const float* Bsrc, const KB kb, int N,
                                             float* out, int op, const float* bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i32 = lane & 31, kh = lane >> 5;
    const int RT = (M + 31) >> 5, CT = (N + 31) >> 5;
    const int nch = (K + 15) >> 4;             // (a ragged last chunk only in the AT form: zero-filled)
    for (int tile = wave; tile < RT * CT; tile += NW) {
        const int ct = tile / RT, rt = tile - ct * RT;
        const int m0 = rt * 32, j0 = ct * 32;
        const int mi = min(m0 + i32, M - 1), jc = min(j0 + i32, N - 1);
        kg_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float bz[16];                       // (requested before the K loop: no memory round trip in the epilogue)
#pragma unroll
        for (int r = 0; r < 16; ++r) bz[r] = bias ? bias[min(m0 + (r & 3) + 8 * (r >> 2) + 4 * kh, M - 1)] : 0.f;
        // The A operands of up to RD chunks are in flight: the weights come from L2 (~2 us under this kernel's low
        // occupancy), a chunk's eight MFMAs take 0.2 us - with K <= 16 RD the whole row is requested before the first MFMA
        constexpr int RD = KG_GB_RD;
        float ring[RD][8];
        auto load_a = [&](float (&d)[8], int ch) {
            const int kc = ch << 4;
            if constexpr (!AT) {
                const float* rp = gb_aptr(A, mi, 0, 0) + kc + 8 * kh;
                const gb_f4 lo = *reinterpret_cast<const gb_f4*>(rp), hi = *reinterpret_cast<const gb_f4*>(rp + 4);
                d[0] = lo[0]; d[1] = lo[1]; d[2] = lo[2]; d[3] = lo[3];
                d[4] = hi[0]; d[5] = hi[1]; d[6] = hi[2]; d[7] = hi[3];
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int k = kc + 8 * kh + s, kk = k < K ? k : 0;
                    const float v_ = *gb_aptr(A, mi, kk, ka(kk));
                    d[s] = k < K ? v_ : 0.f;
                }
            }
        };
        auto mul = [&](const float (&av)[8], int ch) {
            const int kc = ch << 4;
            float bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) { const int k = kc + 8 * kh + s; bv[s] = Bsrc[kb(k < K ? k : K - 1) + jc]; }
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < RD; ++i)
            if (i < nch) load_a(ring[i], i);
        for (int ch = 0; ch < nch; ch += RD) {
#pragma unroll
            for (int i = 0; i < RD; ++i) {
                if (ch + i < nch) {                 // (uniform)
                    mul(ring[i], ch + i);
                    if (ch + i + RD < nch) load_a(ring[i], ch + i + RD);
                }
            }
        }
        // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        const int col = j0 + i32;
        if (col < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M) out[row * op + col] = acc[r] + bz[r];
            }
        }
    }
}

// weights -> LDS as Wl[k][MV] (zero rows beyond M), any A layout
template <int MV, typename KA>
__device__ __forceinline__ void gb_stage_w(const GbA& A, const KA ka, int M, int K, float* Wl) {
    const int tot = K * MV;
    for (int e0 = threadIdx.x; e0 < tot; e0 += NT * UB) {
        float v[UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int e = e0 + i * NT;
            const int k = (e < tot ? e : 0) / MV, m = (e < tot ? e : 0) - k * MV;
            v[i] = (e < tot && m < M) ? *gb_aptr(A, m, k, ka(k)) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < UB; ++i)
            if (e0 + i * NT < tot) Wl[e0 + i * NT] = v[i];
    }
}
// the same contraction with a thread per column and the weights broadcast from LDS (M <= MV rows)
template <int MV, typename KB>
__device__ __forceinline__ void gb_gemm_valu(const float* Wl, int M, int K, const float* Bsrc, const KB kb, int N, float* out, int op,
                                             const float* bias) {
    float bz[MV];
#pragma unroll
    for (int m = 0; m < MV; ++m) bz[m] = (bias && m < M) ? bias[m] : 0.f;
    for (int j = threadIdx.x; j < N; j += NT) {
        float acc[MV];
#pragma unroll
        for (int m = 0; m < MV; ++m) acc[m] = 0.f;
#pragma unroll 3
        for (int k = 0; k < K; ++k) {
            const float b = Bsrc[kb(k) + j];
#pragma unroll
            for (int m4 = 0; m4 < MV; m4 += 4) {
                const gb_f4 w4 = *reinterpret_cast<const gb_f4*>(Wl + k * MV + m4);       // (broadcast, 16-byte LDS read)
                acc[m4 + 0] = fmaf(w4[0], b, acc[m4 + 0]); acc[m4 + 1] = fmaf(w4[1], b, acc[m4 + 1]);
                acc[m4 + 2] = fmaf(w4[2], b, acc[m4 + 2]); acc[m4 + 3] = fmaf(w4[3], b, acc[m4 + 3]);
            }
        }
#pragma unroll
        for (int m = 0; m < MV; ++m)
            if (m < M) out[m * op + j] = acc[m] + bz[m];
    }
}

// one contraction on whichever path the host chose (syncs before and after are the caller's)
template <bool AT, typename KA, typename KB>
__device__ __forceinline__ void gb_gemm(bool mfma, const GbA& A, const KA ka, int M, int K, const float* Bsrc, const KB kb, int N,
                                        float* out, int op, const float* bias, float* Wl) {
    if (mfma) {
        gb_gemm_mfma<AT>(A, ka, M, K, Bsrc, kb, N, out, op, bias);
        return;
    }
    if (M <= 4) {
        gb_stage_w<4>(A, ka, M, K, Wl);
        __syncthreads();
        gb_gemm_valu<4>(Wl, M, K, Bsrc, kb, N, out, op, bias);
    } else if (M <= 16) {
        gb_stage_w<16>(A, ka, M, K, Wl);
        __syncthreads();
        gb_gemm_valu<16>(Wl, M, K, Bsrc, kb, N, out, op, bias);
    } else {
        gb_stage_w<32>(A, ka, M, K, Wl);
        __syncthreads();
        gb_gemm_valu<32>(Wl, M, K, Bsrc, kb, N, out, op, bias);
    }
}

// out(m, j) = sum_k fa(m, k) * fb(k, j) on the matrix cores with every operand fetched through a functor (LDS-resident
// operands with arbitrary index maps: the up-sampling / aggregation products U A_k and their adjoints, which as VALU loops
// were bound by the LDS instruction rate - 66-88 ds_reads per output).  Step s of a 16-k chunk multiplies k = 2 s + (lane >> 5).
template <typename FA, typename FB, typename ST>
__device__ __forceinline__ void gb_mm(int M, int K, int N, FA fa, FB fb, ST st) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i32 = lane & 31, kh = lane >> 5;
    const int RT = (M + 31) >> 5, CT = (N + 31) >> 5;
    for (int tile = wave; tile < RT * CT; tile += NW) {
        const int ct = tile / RT, rt = tile - ct * RT;
        const int m = rt * 32 + i32, j = ct * 32 + i32;
        kg_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {
            float av[8], bv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int k = k0 + 2 * s + kh;
                av[s] = (m < M && k < K) ? fa(m, k) : 0.f;
                bv[s] = (j < N && k < K) ? fb(k, j) : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
        }
        if (j < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M) st(row, j, acc[r]);
            }
        }
    }
}

// sum over groups of P consecutive lanes (P a power of two <= 64), result in every lane of the group
__device__ __forceinline__ float gb_seg_sum(float v, int P) {
    for (int off = P >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// threads per row for `rows` independent row reductions on NT threads: the largest power of two <= min(64, NT / rows)
__device__ __forceinline__ int gb_row_threads(int rows) {
    int P = 64;
    while (P > 1 && P * rows > NT) P >>= 1;
    return P;
}

// rows x cols block of LDS (row pitch `pitch`, first element at src) -> sample n of a plane tensor
template <int D>      // D > 0: the column count as a compile-time divisor
__device__ __forceinline__ void gb_store_plane(const KgPlane& t, int n, const float* src, int pitch, int rows, int cols, const FastDiv& dc) {
    float* const base = t.p + (long)n * t.sN;
    const unsigned tot = (unsigned)(rows * cols);
    for (unsigned e0 = threadIdx.x; e0 < tot; e0 += NT * UB) {
        float v[UB];
        unsigned cc[UB], jj[UB];
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const unsigned e = e0 + i * NT;
            if constexpr (D > 0) { cc[i] = (e < tot ? e : 0u) / (unsigned)D; jj[i] = (e < tot ? e : 0u) - cc[i] * (unsigned)D; }
            else dc.divmod(e < tot ? e : 0u, cc[i], jj[i]);
            v[i] = src[cc[i] * pitch + jj[i]];
        }
#pragma unroll
        for (int i = 0; i < UB; ++i)
            if (e0 + i * NT < tot) base[(long)cc[i] * t.sC + jj[i]] = v[i];
    }
}

// sum over the 64 lanes, result in every lane
// ---- compile-time block geometries -------------------------------------------------------------------------------------
// The kernels below are templates on a geometry policy G: GbRt (CT = false) reads every dimension from the arguments - any
// block that fits; GbGeo<...> (CT = true) fixes them at compile time - no per-element divisions by run-time divisors, inner
// loops of known length, the residual / BatchNorm / partition branches resolved.  The launcher picks the instantiation that
// matches the block (the NTU and Human3.6M generators' last blocks) and falls back to GbRt.
struct GbRt {
    static constexpr bool CT = false;
    static constexpr int Cin = 1, C = 1, Kp = 1, Tc = 1, Vc = 1, T = 1, V = 1, rep = 1, RES = 0, BNT = 0, M0 = 0, M1 = 0;
};
template <int CIN_, int C_, int KP_, int TC_, int VC_, int V_, int REP_, int RES_, int BNT_>
struct GbGeo {
    static constexpr bool CT = true;
    static constexpr int Cin = CIN_, C = C_, Kp = KP_, Tc = TC_, Vc = VC_, T = TC_ * REP_, V = V_, rep = REP_, RES = RES_, BNT = BNT_;
};
// q = x / D, r = x % D: a constant divisor D when the geometry is compile-time, the host's magic number otherwise
#define GB_DIVMOD(D_, fd_, x_, q_, r_) do { if constexpr (G::CT) { q_ = (unsigned)(x_) / (unsigned)(D_); r_ = (unsigned)(x_) - q_ * (unsigned)(D_); } \
                                            else (fd_).divmod((unsigned)(x_), q_, r_); } while (0)

// ======================================================================================================================
// forward
// ======================================================================================================================
template <typename G>
__global__ __launch_bounds__(NT) void kg_genblock_fwd_kernel(const KgGenBlockArgs a, const GbLayout L) {
    extern __shared__ float lds[];
    constexpr int UBE = G::CT ? 8 : UB;      // elements in flight per thread in the streaming stages (compile-time geometries: the code per element is short)
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x;
    const int C = G::CT ? G::C : a.C, Cin = G::CT ? G::Cin : a.Cin, V = G::CT ? G::V : a.V, Vc = G::CT ? G::Vc : a.Vc;
    const int Tc = G::CT ? G::Tc : a.Tc, rep = G::CT ? G::rep : a.rep, Kp = G::CT ? G::Kp : a.Kp, T_ = G::CT ? G::T : a.T;
    const int res_kind = G::CT ? G::RES : a.res_kind;
    const int Nc = Tc * Vc, Nf = T_ * V, ZP = (T_ + 2) * V, Mg = Kp * C, Mh = Mg + (res_kind == 2 ? C : 0);
    const bool mfma0 = G::CT ? gb_use_mfma(Mh, Cin, Nc, false) : (L.mfma0 != 0);
    const bool mfma1 = G::CT ? gb_use_mfma(C, 3 * C, Nf, false) : (L.mfma1 != 0);
    float* const X = lds + L.x;
    float* const YC = lds + L.yc;
    float* const UO = lds + L.uo;
    float* const Z = lds + L.z;
    float* const R = lds + L.r;
    float* const Bs = lds + L.bs;
    float* const Us = lds + L.us;
    float* const Wl = lds + L.wl;
    float* const Brs = lds + L.brs;      // residual conv bias (C)
    const int h = a.N / a.groups;
    const int grp = n / h;
    GB_STAMP(0, 0);

    // ---- stage 0: the block's input (finished, or the previous block's pending tail applied here), adjacency, tables.
    // Every streaming loop of this kernel issues the loads of UBE elements per thread before the first use: one by one a
    // loop iteration is a full memory round trip (the first version ran 50-90 us per launch that way)
    {
        const unsigned tot = (unsigned)(Cin * Nc);
        if (a.x.p) {
            const float* xp = a.x.p + (long)n * a.x.sN;
            for (unsigned e0 = tid; e0 < tot; e0 += NT * UBE) {
                float v[UBE];
#pragma unroll
                for (int i = 0; i < UBE; ++i) {
                    const unsigned e = e0 + i * NT;
                    unsigned c, j;
                    GB_DIVMOD(G::Tc * G::Vc, L.dNc, e < tot ? e : 0u, c, j);
                    v[i] = e < tot ? xp[(long)c * a.x.sC + j] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < UBE; ++i)
                    if (e0 + i * NT < tot) X[e0 + i * NT] = v[i];
            }
        } else {
            const float* up = a.pu.p + (long)n * a.pu.sN;
            const float* rp = a.pr.p ? a.pr.p + (long)n * a.pr.sN : nullptr;
            const float* ct = a.pcoef_t ? a.pcoef_t + (long)grp * 4 * Cin : nullptr;
            const float* cr = a.pcoef_r ? a.pcoef_r + (long)grp * 4 * Cin : nullptr;
            const float* nz = (a.pnoise && a.pnw) ? a.pnoise + (long)n * Nc : nullptr;
            float* xo = a.xout.p + (long)n * a.xout.sN;
            for (unsigned e0 = tid; e0 < tot; e0 += NT * UBE) {
                float uv[UBE], rv[UBE], s0[UBE], b0[UBE], s1[UBE], b1[UBE], nv_[UBE], wv[UBE];
                unsigned cc[UBE], jj[UBE];
#pragma unroll
                for (int i = 0; i < UBE; ++i) {
                    const unsigned e = e0 + i * NT;
                    GB_DIVMOD(G::Tc * G::Vc, L.dNc, e < tot ? e : 0u, cc[i], jj[i]);
                    const unsigned c = cc[i], j = jj[i];
                    uv[i] = up[(long)c * a.pu.sC + j];
                    rv[i] = rp ? rp[(long)c * a.pr.sC + j] : 0.f;
                    s0[i] = ct ? ct[c] : 1.f;
                    b0[i] = ct ? ct[Cin + c] : 0.f;
                    s1[i] = cr ? cr[c] : 1.f;
                    b1[i] = cr ? cr[Cin + c] : 0.f;
                    nv_[i] = nz ? nz[j] : 0.f;
                    wv[i] = nz ? a.pnw[c] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < UBE; ++i) {
                    if (e0 + i * NT >= tot) continue;
                    float v = ct ? fmaf(uv[i], s0[i], b0[i]) : uv[i];
                    if (rp) v += cr ? fmaf(rv[i], s1[i], b1[i]) : rv[i];
                    if (nz) v = fmaf(wv[i], nv_[i], v);
                    v = kg_act(v, a.pact, a.slope);
                    X[e0 + i * NT] = v;
                    xo[(long)cc[i] * a.xout.sC + jj[i]] = v;
                }
            }
        }
    }
    for (int i = tid; i < Kp * Vc * V; i += NT) Bs[i] = a.b[i];
    for (int i = tid; i < Vc * V; i += NT) Us[i] = a.u ? a.u[i] : ((i / V) == (i % V) ? 1.f : 0.f);
    if (res_kind == 2 && a.br)
        for (int i = tid; i < C; i += NT) Brs[i] = a.br[i];
    for (int i = tid; i < C * 2 * V; i += NT) {          // zero halo frames of z (frame 0 and frame T + 1)
        const int c = i / (2 * V), q = i - c * 2 * V;
        Z[c * ZP + (q < V ? q : ZP - 2 * V + q)] = 0.f;
    }
    __syncthreads();
    GB_STAMP(0, 1);

    // ---- stage 1: yc = [W_gcn[:Mg]; W_res] x on the input grid
    {
        GbA A{a.wg, a.wr, Mg, Cin, 1};
        gb_gemm<false>(mfma0, A, KLin{1}, Mh, Cin, X, KLin{Nc}, Nc, YC, Nc, nullptr, Wl);
    }
    __syncthreads();
    GB_STAMP(0, 2);
    gb_store_plane<(G::CT ? G::Tc * G::Vc : 0)>(a.yc, n, YC, Nc, Mh, Nc, L.dNc);

    // ---- stage 2: z = sum_k yc_k (U A_k), r = yc_res U + b_res | x U, frames repeated - two small products on the matrix
    //      cores: rows m = (c, tc), contraction k = (partition, coarse vertex), columns = output vertices
    {
                gb_mm(C * Tc, Kp * Vc, V,
              [&](int m, int k) {
                  unsigned c, tc, kk, vc;
                  GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                  GB_DIVMOD(G::Vc, L.dVc, (unsigned)k, kk, vc);
                  return YC[(kk * C + c) * Nc + tc * Vc + vc];
              },
              [&](int k, int j) { return Bs[k * V + j]; },
              [&](int m, int j, float v) {
                  unsigned c, tc;
                  GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                  float* zp = Z + c * ZP + V + (tc * rep) * V + j;
                  for (int q = 0; q < rep; ++q) zp[q * V] = v;
              });
        if (res_kind != 0) {
            const float* src = res_kind == 2 ? YC + Mg * Nc : X;          // [C][Nc]
            const bool bias = res_kind == 2 && a.br != nullptr;
            gb_mm(C * Tc, Vc, V,
                  [&](int m, int k) {
                      unsigned c, tc;
                      GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                      return src[c * Nc + tc * Vc + k];
                  },
                  [&](int k, int j) { return Us[k * V + j]; },
                  [&](int m, int j, float v) {
                      unsigned c, tc;
                      GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                      if (bias) v += Brs[c];
                      float* rp = R + c * Nf + (tc * rep) * V + j;
                      for (int q = 0; q < rep; ++q) rp[q * V] = v;
                  });
        }
    }
    __syncthreads();
    GB_STAMP(0, 3);

    // ---- stage 3: u = W_tcn (*) z + b   (k = (c', tap): the weight row is contiguous in k; tap d reads frame t + d of the
    //      zero-padded z)
    {
        GbA A{a.wt, a.wt, 1 << 30, 3 * C, 1};
        gb_gemm<false>(mfma1, A, KLin{1}, C, 3 * C, Z, KTap{ZP, V, 0}, Nf, UO, Nf, a.bt, Wl);
    }
    __syncthreads();
    GB_STAMP(0, 4);

    // ---- stage 4: the BatchNorm partials of this sample FIRST (the ticket below waits for the wave's outstanding stores:
    //      with the tape stores in front of it every workgroup paid their drain before it could arrive), then the tape; a block
    //      without any BatchNorm writes the tape and its finished output
#define GB_STORE_TAPE()                                                                                          \
    do {                                                                                                        \
        gb_store_plane<(G::CT ? G::T * G::V : 0)>(a.z, n, Z + V, ZP, C, Nf, L.dNf);                              \
        gb_store_plane<(G::CT ? G::T * G::V : 0)>(a.uo, n, UO, Nf, C, Nf, L.dNf);                                \
        if (res_kind != 0 && a.r.p) gb_store_plane<(G::CT ? G::T * G::V : 0)>(a.r, n, R, Nf, C, Nf, L.dNf);      \
    } while (0)
    const bool bn_t = G::CT ? G::BNT != 0 : a.bn_t != 0, bn_r = res_kind == 2;
    GB_STAMP(0, 5);
    if (!bn_t && !bn_r) {
        GB_STORE_TAPE();
        if (a.out.p) {
            float* const ob = a.out.p + (long)n * a.out.sN;
            const float* nz = (a.noise && a.nw) ? a.noise + (long)n * Nf : nullptr;
            const unsigned tot = (unsigned)(C * Nf);
            for (unsigned e0 = tid; e0 < tot; e0 += NT * UBE) {
                float nv_[UBE], wv[UBE];
                unsigned cc[UBE], jj[UBE];
#pragma unroll
                for (int i = 0; i < UBE; ++i) {
                    const unsigned e = e0 + i * NT;
                    GB_DIVMOD(G::T * G::V, L.dNf, e < tot ? e : 0u, cc[i], jj[i]);
                    nv_[i] = nz ? nz[jj[i]] : 0.f;
                    wv[i] = nz ? a.nw[cc[i]] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < UBE; ++i) {
                    const unsigned e = e0 + i * NT;
                    if (e >= tot) continue;
                    float v = UO[e];
                    if (res_kind != 0) v += R[e];
                    if (nz) v = fmaf(wv[i], nv_[i], v);
                    ob[(long)cc[i] * a.out.sC + jj[i]] = kg_act(v, a.act, a.slope);
                }
            }
        }
        GB_STAMP(0, 6);
        return;
    }
    // partials: ws[((branch * N + n) * C + c) * 2] = (mean, centred sum of squares) over this sample's T * V elements; a row
    // (branch, channel) is reduced by P consecutive lanes (a whole wave per row left most lanes idle: 20-176 elements)
    {
        const int nbr = (bn_t ? 1 : 0) + (bn_r ? 1 : 0), rows = nbr * C;
        const int P = gb_row_threads(rows);
        const int row = tid / P, slot = tid - row * P;
        const bool live = row < rows;
        const int bi = live ? row / C : 0, c = live ? row - bi * C : 0;
        const int br = (bn_t && bi == 0) ? 0 : 1;
        const float* src = (br == 0 ? UO : R) + c * Nf;
        float sm = 0.f;
#pragma unroll 4
        for (int j = slot; j < Nf; j += P) sm += src[j];
        const float mean = gb_seg_sum(sm, P) / (float)Nf;
        float q = 0.f;
#pragma unroll 4
        for (int j = slot; j < Nf; j += P) { const float dd = src[j] - mean; q = fmaf(dd, dd, q); }
        q = gb_seg_sum(q, P);
        if (live && slot == 0) {
            float* part = a.ws + ((long)(br * a.N + n) * C + c) * 2;
            __hip_atomic_store(part + 0, mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    GB_STAMP(0, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int tk = __hip_atomic_fetch_add(a.counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (tk == (int)gridDim.x - 1);
    }
    __syncthreads();
    GB_STAMP(0, 7);
    if (!last) { GB_STORE_TAPE(); return; }
    // The last workgroup to arrive merges the samples: thread (branch, batch, channel) takes the samples of its batch 32 at
    // a time - 64 loads in flight, consecutive channels in consecutive lanes - forms the chunk's pooled (mean, M2) from the
    // equal-sized samples (mean = average of the means, M2 = sum [M2_s + L (mean_s - mean)^2]) and merges chunks in order
    // (Chan et al.): deterministic.  (mean, var) go to LDS, then thread (branch, channel) writes the coefficients and
    // applies the running-statistics updates batch by batch.
    // (the layer parameters of thread (branch, channel) are requested now and used behind the merge)
    float pf_gam = 1.f, pf_bet = 0.f, pf_rm = 0.f, pf_rv = 0.f;
    if (tid < 2 * C) {
        const int br = tid / C, c = tid - br * C;
        if (br == 0 ? bn_t : bn_r) {
            const KgGenBnLayer& bl = br == 0 ? a.bt_ : a.br_;
            pf_gam = bl.gamma ? bl.gamma[c] : 1.f; pf_bet = bl.beta ? bl.beta[c] : 0.f;
            pf_rm = bl.running_mean ? bl.running_mean[c] : 0.f; pf_rv = bl.running_var ? bl.running_var[c] : 0.f;
        }
    }
    float* const MV_ = Wl;                  // [2][groups][C][2]  (the weight staging area is dead; the tape tensors are still needed)
    {
        const int ntask = 2 * a.groups * C;
        constexpr int SB = 32;
        for (int t = tid; t < ntask; t += NT) {
            const int br = t / (a.groups * C), rem = t - br * a.groups * C, g = rem / C, c = rem - g * C;
            if (!(br == 0 ? bn_t : bn_r)) continue;
            const float* part = a.ws + (((long)br * a.N + (long)g * h) * C + c) * 2;
            float n_tot = 0.f, mean = 0.f, M2 = 0.f;
            for (int s0 = 0; s0 < h; s0 += SB) {
                float mk[SB], qk[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int sidx = s0 + i < h ? s0 + i : h - 1;
                    mk[i] = __hip_atomic_load(part + (long)sidx * C * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    qk[i] = __hip_atomic_load(part + (long)sidx * C * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int cnt = h - s0 < SB ? h - s0 : SB;
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < SB; ++i) sm += i < cnt ? mk[i] : 0.f;
                const float cm = sm / (float)cnt;
                float cq = 0.f;
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const float dd = mk[i] - cm;
                    cq += i < cnt ? fmaf((float)Nf * dd, dd, qk[i]) : 0.f;
                }
                const float nb = (float)cnt * (float)Nf, n_new = n_tot + nb, delta = cm - mean;
                mean += delta * (nb / n_new);
                M2 += cq + delta * delta * (n_tot * nb / n_new);
                n_tot = n_new;
            }
            MV_[((br * a.groups + g) * C + c) * 2 + 0] = mean;
            MV_[((br * a.groups + g) * C + c) * 2 + 1] = M2 / n_tot;
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += NT) {
        const int br = i / C, c = i - br * C;
        if (!(br == 0 ? bn_t : bn_r)) continue;
        const KgGenBnLayer& bl = br == 0 ? a.bt_ : a.br_;
        const float gam = i == tid ? pf_gam : (bl.gamma ? bl.gamma[c] : 1.f), bet = i == tid ? pf_bet : (bl.beta ? bl.beta[c] : 0.f);
        float rm = i == tid ? pf_rm : (bl.running_mean ? bl.running_mean[c] : 0.f), rv = i == tid ? pf_rv : (bl.running_var ? bl.running_var[c] : 0.f);
        const float n_tot = (float)h * (float)Nf;
        for (int g = 0; g < a.groups; ++g) {
            const float mean = MV_[((br * a.groups + g) * C + c) * 2 + 0], var = MV_[((br * a.groups + g) * C + c) * 2 + 1];
            const float rstd = 1.f / sqrtf(var + bl.eps);
            const float scale = gam * rstd;
            const float shift = bet - mean * scale;
            float* coef = bl.coef + (long)g * 4 * C;
            coef[0 * C + c] = scale;
            coef[1 * C + c] = shift;
            coef[2 * C + c] = mean;
            coef[3 * C + c] = rstd;
            const float unb = var * (n_tot / (n_tot > 1.f ? n_tot - 1.f : 1.f));
            rm = (1.f - bl.momentum) * rm + bl.momentum * mean;
            rv = (1.f - bl.momentum) * rv + bl.momentum * unb;
        }
        if (bl.running_mean) { bl.running_mean[c] = rm; bl.running_var[c] = rv; }
        if (bl.num_batches_tracked && c == 0) *bl.num_batches_tracked += a.groups;
    }
    if (tid == 0) __hip_atomic_store(a.counters, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    GB_STORE_TAPE();
}

// ======================================================================================================================
// backward
// ======================================================================================================================
template <typename G>
__global__ __launch_bounds__(NT) void kg_genblock_bwd_kernel(const KgGenBlockBwdArgs a, const GbLayout L) {
    extern __shared__ float lds[];
    constexpr int UBE = G::CT ? 8 : UB;      // elements in flight per thread in the streaming stages (compile-time geometries: the code per element is short)
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x;
    const int C = G::CT ? G::C : a.C, Cin = G::CT ? G::Cin : a.Cin, V = G::CT ? G::V : a.V, Vc = G::CT ? G::Vc : a.Vc;
    const int Tc = G::CT ? G::Tc : a.Tc, rep = G::CT ? G::rep : a.rep, Kp = G::CT ? G::Kp : a.Kp, T_ = G::CT ? G::T : a.T;
    const int res_kind = G::CT ? G::RES : a.res_kind;
    const int Nc = Tc * Vc, Nf = T_ * V, ZP = (T_ + 2) * V, Mg = Kp * C, Mh = Mg + (res_kind == 2 ? C : 0);
    const bool mfma0 = G::CT ? gb_use_mfma(Cin, Mh, Nc, true) : (L.mfma0 != 0);
    const bool mfma1 = G::CT ? gb_use_mfma(C, 3 * C, Nf, true) : (L.mfma1 != 0);
    float* const DU = lds + L.du;        // [C][ZP], zero halo frames
    float* const DR = lds + L.dr;        // [C][Nf]
    float* const GZ = lds + L.gz;        // [C][Nf]
    float* const GYC = lds + L.gyc;      // [Mh][Nc]
    float* const GID = lds + L.gid;      // [C][Nc]  (identity residual)
    float* const GX = lds + L.gx;        // [Cin][Nc]
    float* const Bs = lds + L.bs;
    float* const Us = lds + L.us;
    float* const Wl = lds + L.wl;
    const bool bn_t = G::CT ? G::BNT != 0 : a.bn_t != 0, bn_r = res_kind == 2, has_r = res_kind != 0;
    GB_STAMP(1, 0);

    // ---- stage 0: du / dr from the tail coefficients (all loads of UBE elements before the first use), adjacency, tables
    {
        const float* gp_ = a.g.p + (long)n * a.g.sN;
        const float* op_ = a.out.p + (long)n * a.out.sN;
        const float* up_ = bn_t ? a.uo.p + (long)n * a.uo.sN : nullptr;
        const float* rp_ = bn_r ? a.r.p + (long)n * a.r.sN : nullptr;
        float* dub = a.du.p + (long)n * a.du.sN;
        float* drb = a.dr.p ? a.dr.p + (long)n * a.dr.sN : nullptr;
        const unsigned tot = (unsigned)(C * Nf);
        for (unsigned e0 = tid; e0 < tot; e0 += NT * UBE) {
            float gv[UBE], ov[UBE], uv[UBE], rv[UBE], k0[UBE], k1[UBE], k2[UBE], k3[UBE], k4[UBE], k5[UBE];
            unsigned cc[UBE], jj[UBE];
#pragma unroll
            for (int i = 0; i < UBE; ++i) {
                const unsigned e = e0 + i * NT;
                GB_DIVMOD(G::T * G::V, L.dNf, e < tot ? e : 0u, cc[i], jj[i]);
                const unsigned c = cc[i], j = jj[i];
                gv[i] = gp_[(long)c * a.g.sC + j];
                ov[i] = op_[(long)c * a.out.sC + j];
                uv[i] = bn_t ? up_[(long)c * a.uo.sC + j] : 0.f;
                rv[i] = bn_r ? rp_[(long)c * a.r.sC + j] : 0.f;
                k0[i] = bn_t ? a.coef[0 * C + c] : 1.f; k1[i] = bn_t ? a.coef[1 * C + c] : 0.f; k2[i] = bn_t ? a.coef[2 * C + c] : 0.f;
                k3[i] = bn_r ? a.coef[3 * C + c] : 1.f; k4[i] = bn_r ? a.coef[4 * C + c] : 0.f; k5[i] = bn_r ? a.coef[5 * C + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < UBE; ++i) {
                const unsigned e = e0 + i * NT;
                if (e >= tot) continue;
                const float gp = gv[i] * kg_dact_from_out(ov[i], a.act, a.slope);
                const float du = bn_t ? fmaf(k0[i], gp, fmaf(k1[i], uv[i], k2[i])) : gp;
                const float dr = bn_r ? fmaf(k3[i], gp, fmaf(k4[i], rv[i], k5[i])) : gp;
                DU[cc[i] * ZP + V + jj[i]] = du;
                dub[(long)cc[i] * a.du.sC + jj[i]] = du;
                if (has_r) {
                    DR[e] = dr;
                    if (drb) drb[(long)cc[i] * a.dr.sC + jj[i]] = dr;
                }
            }
        }
    }
    for (int i = tid; i < Kp * Vc * V; i += NT) Bs[i] = a.b[i];
    for (int i = tid; i < Vc * V; i += NT) Us[i] = a.u ? a.u[i] : ((i / V) == (i % V) ? 1.f : 0.f);
    for (int i = tid; i < C * 2 * V; i += NT) {
        const int c = i / (2 * V), q = i - c * 2 * V;
        DU[c * ZP + (q < V ? q : ZP - 2 * V + q)] = 0.f;
    }
    __syncthreads();
    GB_STAMP(1, 1);

    // ---- stage 1: gz = W_tcn^T (*) du:  gz[c'] = sum_(c, d) wt[c][c'][d] du[c][t + 1 - d], k = 3 c + d,
    //      A(m = c', k) = wt[c * 3C + 3 c' + d]
    {
        GbA A{a.wt, a.wt, 1 << 30, 3, 0};
        gb_gemm<true>(mfma1, A, KTap{3 * C, 1, 0}, C, 3 * C, DU, KTap{ZP, V, 1}, Nf, GZ, Nf, nullptr, Wl);
    }
    __syncthreads();
    GB_STAMP(1, 2);

    // ---- stage 2: fold back to the input grid: gyc_k = fold(gz (U A_k)^T), residual rows fold(dr U^T) - on the matrix
    //      cores: rows m = (c, tc), contraction k = (repeated frame, vertex), columns (partition, coarse vertex);
    //      zf = gz summed over the repeated frames (the adjacency gradient's operand)
    {
        const int len = rep * V;
        gb_mm(C * Tc, len, Kp * Vc,
              [&](int m, int k) {
                  unsigned c, tc;
                  GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                  return GZ[c * Nf + tc * len + k];
              },
              [&](int k, int j) {
                  unsigned q, w;
                  GB_DIVMOD(G::V, L.dV, (unsigned)k, q, w);
                  return Bs[j * V + w];
              },
              [&](int m, int j, float v) {
                  unsigned c, tc, kk, vc;
                  GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                  GB_DIVMOD(G::Vc, L.dVc, (unsigned)j, kk, vc);
                  GYC[(kk * C + c) * Nc + tc * Vc + vc] = v;
              });
        if (has_r) {
            float* const dst = bn_r ? GYC + Mg * Nc : GID;                  // [C][Nc]
            gb_mm(C * Tc, len, Vc,
                  [&](int m, int k) {
                      unsigned c, tc;
                      GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                      return DR[c * Nf + tc * len + k];
                  },
                  [&](int k, int j) {
                      unsigned q, w;
                      GB_DIVMOD(G::V, L.dV, (unsigned)k, q, w);
                      return Us[j * V + w];
                  },
                  [&](int m, int j, float v) {
                      unsigned c, tc;
                      GB_DIVMOD(G::Tc, L.dTc, (unsigned)m, c, tc);
                      dst[c * Nc + tc * Vc + j] = v;
                  });
        }
        float* const zfb = a.zf.p + (long)n * a.zf.sN;
        const unsigned zitems = (unsigned)(C * Tc * V);
        for (unsigned e = tid; e < zitems; e += NT) {
            unsigned c, rem, tc, w;
            GB_DIVMOD(G::Tc * G::V, L.dTcV, e, c, rem);
            GB_DIVMOD(G::V, L.dV, rem, tc, w);
            const float* gp = GZ + c * Nf + (tc * rep) * V + w;
            float s = 0.f;
            for (int q = 0; q < rep; ++q) s += gp[q * V];
            zfb[(long)c * a.zf.sC + tc * V + w] = s;
        }
    }
    __syncthreads();
    GB_STAMP(1, 3);
    gb_store_plane<(G::CT ? G::Tc * G::Vc : 0)>(a.gyc, n, GYC, Nc, Mh, Nc, L.dNc);

    // ---- stage 3: gx = [W_gcn; W_res]^T gyc (+ identity branch): k = m, rows of W_gcn then rows of W_res
    {
        GbA A{a.wg, a.wr, Mg, 1, 0};
        gb_gemm<true>(mfma0, A, KSplit{Mg, Cin}, Cin, Mh, GYC, KLin{Nc}, Nc, GX, Nc, nullptr, Wl);
    }
    __syncthreads();
    if (res_kind == 1) {
        for (int e = tid; e < Cin * Nc; e += NT) GX[e] += GID[e];
        __syncthreads();
    }
    GB_STAMP(1, 4);
#define GB_STORE_GX() gb_store_plane<(G::CT ? G::Tc * G::Vc : 0)>(a.gx, n, GX, Nc, Cin, Nc, L.dNc)
    GB_STAMP(1, 5);
    if (!a.px.p) { GB_STORE_GX(); return; }

    // ---- stage 4: tail statistics of the PREVIOUS block over this sample: gp = gx * pact'(x).  A channel is reduced by P
    //      consecutive lanes, every load of a thread in flight before the first use
    {
        const bool pbn_t = a.pu.p != nullptr, pbn_r = a.pr.p != nullptr && a.pmean_r != nullptr;
        const float* xb = a.px.p + (long)n * a.px.sN;
        const float* ub = pbn_t ? a.pu.p + (long)n * a.pu.sN : nullptr;
        const float* rb = pbn_r ? a.pr.p + (long)n * a.pr.sN : nullptr;
        const float* nz = a.pnoise ? a.pnoise + (long)n * Nc : nullptr;
        float* part = a.ws + ((long)n * Cin) * 4;
        {
            const int P = gb_row_threads(Cin);
            const int row = tid / P, slot = tid - row * P;
            const bool live = row < Cin;
            const int c = live ? row : 0;
            const float mt = pbn_t ? a.pmean_t[c] : 0.f, mr = pbn_r ? a.pmean_r[c] : 0.f;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int j0 = slot; j0 < Nc; j0 += UBE * P) {
                float xv[UBE], uv[UBE], rv[UBE], zv[UBE];
#pragma unroll
                for (int i = 0; i < UBE; ++i) {
                    const int j = j0 + i * P < Nc ? j0 + i * P : 0;
                    xv[i] = xb[(long)c * a.px.sC + j];
                    uv[i] = pbn_t ? ub[(long)c * a.pu.sC + j] : 0.f;
                    rv[i] = pbn_r ? rb[(long)c * a.pr.sC + j] : 0.f;
                    zv[i] = nz ? nz[j] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < UBE; ++i) {
                    const int j = j0 + i * P;
                    if (j >= Nc) continue;
                    const float gp = GX[c * Nc + j] * kg_dact_from_out(xv[i], a.pact, a.slope);
                    s0 += gp;
                    s1 = fmaf(gp, uv[i] - mt, s1);
                    s2 = fmaf(gp, rv[i] - mr, s2);
                    s3 = fmaf(gp, zv[i], s3);
                }
            }
            s0 = gb_seg_sum(s0, P); s1 = gb_seg_sum(s1, P); s2 = gb_seg_sum(s2, P); s3 = gb_seg_sum(s3, P);
            if (live && slot == 0) {
                __hip_atomic_store(part + c * 4 + 0, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + c * 4 + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + c * 4 + 2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(part + c * 4 + 3, s3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        GB_STAMP(1, 6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int tk = __hip_atomic_fetch_add(a.counters, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (tk == (int)gridDim.x - 1);
        }
        __syncthreads();
        GB_STAMP(1, 7);
        if (!last) { GB_STORE_GX(); return; }
        // the last workgroup to arrive: P consecutive lanes per channel, each the samples slot, slot + P, ... (all loads in
        // flight), then a fixed-shape sum over the P lanes: deterministic; lane 0 of the group writes the coefficients and
        // adds the parameter gradients
        const float inv_n = 1.f / ((float)a.N * (float)Nc);
        {
            const int P = gb_row_threads(Cin);
            const int row = tid / P, slot = tid - row * P;
            const bool live = row < Cin;
            const int c = live ? row : 0;
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            constexpr int SB = 8;
            for (int s0 = slot; s0 < a.N; s0 += SB * P) {
                float v[SB][4];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int sidx = s0 + i * P < a.N ? s0 + i * P : 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[i][q] = __hip_atomic_load(a.ws + ((long)sidx * Cin + c) * 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int i = 0; i < SB; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] += s0 + i * P < a.N ? v[i][q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = gb_seg_sum(t[q], P);
            if (live && slot == 0) {
                float at = 1.f, bt = 0.f, ct = 0.f, ar = 1.f, br = 0.f, cr = 0.f;
                if (pbn_t) {
                    const float rstd = a.prstd_t[c], q = t[1] * rstd, mt = a.pmean_t[c];
                    at = (a.pgamma_t ? a.pgamma_t[c] : 1.f) * rstd;
                    bt = -at * rstd * q * inv_n;
                    ct = -at * t[0] * inv_n - bt * mt;
                    if (a.dgamma_t) a.dgamma_t[c] += q;
                    if (a.dbeta_t) a.dbeta_t[c] += t[0];
                }
                if (pbn_r) {
                    const float rstd = a.prstd_r[c], q = t[2] * rstd, mr = a.pmean_r[c];
                    ar = (a.pgamma_r ? a.pgamma_r[c] : 1.f) * rstd;
                    br = -ar * rstd * q * inv_n;
                    cr = -ar * t[0] * inv_n - br * mr;
                    if (a.dgamma_r) a.dgamma_r[c] += q;
                    if (a.dbeta_r) a.dbeta_r[c] += t[0];
                }
                if (a.pnoise && a.dnw) a.dnw[c] += t[3];
                a.pcoef[0 * Cin + c] = at; a.pcoef[1 * Cin + c] = bt; a.pcoef[2 * Cin + c] = ct;
                a.pcoef[3 * Cin + c] = ar; a.pcoef[4 * Cin + c] = br; a.pcoef[5 * Cin + c] = cr;
            }
        }
        if (tid == 0) __hip_atomic_store(a.counters, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        GB_STORE_GX();
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
struct Dims { int N, Cin, C, K, Kp, Tc, Vc, T, V, rep, res_kind; };

bool path_ok(int M, int K, int N, bool at, bool* mfma) {
    if (gb_use_mfma(M, K, N, at)) { *mfma = true; return true; }
    *mfma = false;
    return M <= GB_VALU_MAXM && (long)K * (M <= 4 ? 4 : M <= 16 ? 16 : 32) <= GB_VALU_MAXMK;
}

int check_dims(const Dims& d, const char* who) {
    KG_REQUIRE(d.N > 0 && d.Cin > 0 && d.C > 0 && d.Tc > 0 && d.Vc > 0 && d.V > 0 && d.rep >= 1 && d.T == d.Tc * d.rep, "%s: bad dims", who);
    KG_REQUIRE(d.K >= 1 && d.K <= 3 && d.Kp >= 1 && d.Kp <= d.K, "%s: K=%d Kp=%d", who, d.K, d.Kp);
    KG_REQUIRE(d.V <= 32 && d.Vc <= 32, "%s: V=%d / Vc=%d exceed 32 vertices", who, d.V, d.Vc);
    KG_REQUIRE(d.res_kind >= 0 && d.res_kind <= 2 && (d.res_kind != 1 || d.Cin == d.C), "%s: res_kind=%d", who, d.res_kind);
    return 0;
}

// fills the layout; returns the dynamic LDS bytes or -1 when the block does not fit this form
long make_layout(const Dims& d, bool bwd, GbLayout& L) {
    L = GbLayout{};
    L.Nc = d.Tc * d.Vc; L.Nf = d.T * d.V; L.ZP = (d.T + 2) * d.V;
    L.Mg = d.Kp * d.C; L.Mh = L.Mg + (d.res_kind == 2 ? d.C : 0);
    bool m0, m1;
    if (!bwd) {
        if (!path_ok(L.Mh, d.Cin, L.Nc, false, &m0) || !path_ok(d.C, 3 * d.C, L.Nf, false, &m1)) return -1;
        if (m0 && d.Cin % 4 != 0) return -1;
    } else {
        if (!path_ok(d.Cin, L.Mh, L.Nc, true, &m0) || !path_ok(d.C, 3 * d.C, L.Nf, true, &m1)) return -1;
    }
    L.mfma0 = m0; L.mfma1 = m1;
    int off = 0;
    auto take = [&](int nfl) { const int o = off; off += (nfl + 3) & ~3; return o; };
    if (!bwd) {
        const int a0 = d.Cin * L.Nc + L.Mh * L.Nc, a1 = d.C * L.Nf;
        const int ra = take(a0 > a1 ? a0 : a1);
        L.x = ra; L.yc = ra + d.Cin * L.Nc; L.uo = ra;
        L.z = take(d.C * L.ZP);
        L.r = take(d.res_kind != 0 ? d.C * L.Nf : 0);
    } else {
        L.du = take(d.C * L.ZP);
        L.dr = take(d.res_kind != 0 ? d.C * L.Nf : 0);
        L.gz = take(d.C * L.Nf);
        L.gyc = take(L.Mh * L.Nc);
        L.gid = take(d.res_kind == 1 ? d.C * L.Nc : 0);
        L.gx = take(d.Cin * L.Nc);
    }
    L.bs = take(d.Kp * d.Vc * d.V);
    L.us = take(d.Vc * d.V);
    L.brs = take(d.C);
    L.wl = take(GB_VALU_MAXMK);
    L.total = off;
    L.dNc = FastDiv::make((unsigned)L.Nc); L.dNf = FastDiv::make((unsigned)L.Nf);
    L.dV = FastDiv::make((unsigned)d.V); L.dVc = FastDiv::make((unsigned)d.Vc); L.dTc = FastDiv::make((unsigned)d.Tc);
    L.dTcV = FastDiv::make((unsigned)(d.Tc * d.V)); L.dTcVc = FastDiv::make((unsigned)(d.Tc * d.Vc));
    const long bytes = (long)off * 4;
    return bytes <= GB_MAX_LDS ? bytes : -1;
}

Dims dims_of(const KgGenBlockArgs* a) { return Dims{a->N, a->Cin, a->C, a->K, a->Kp, a->Tc, a->Vc, a->T, a->V, a->rep, a->res_kind}; }
Dims dims_of(const KgGenBlockBwdArgs* a) { return Dims{a->N, a->Cin, a->C, a->K, a->Kp, a->Tc, a->Vc, a->T, a->V, a->rep, a->res_kind}; }

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// the compile-time geometries: Cin, C, Kp, Tc, Vc, V, rep, residual kind, BatchNorm behind the temporal conv -
// the last four generator blocks of the NTU configurations (t_size 64) and the last two of Human3.6M (t_size 32)
#define GB_GEOMETRIES(X) \
    X(128, 64, 3, 4, 5, 5, 2, 2, 1); X(64, 32, 3, 8, 5, 11, 2, 2, 0); X(32, 3, 3, 16, 11, 11, 2, 2, 1); X(3, 3, 3, 32, 11, 25, 2, 1, 0); \
    X(32, 2, 3, 8, 7, 7, 2, 2, 1); X(2, 2, 3, 16, 7, 16, 2, 1, 0)
template <typename G>
bool geo_matches(const Dims& d, int bn_t) {
    return d.Cin == G::Cin && d.C == G::C && d.Kp == G::Kp && d.Tc == G::Tc && d.Vc == G::Vc && d.V == G::V && d.rep == G::rep &&
           d.res_kind == G::RES && (bn_t != 0) == (G::BNT != 0);
}

}  // namespace

extern "C" int64_t kg_genblock_lds_bytes(const KgGenBlockArgs* a) {
    KG_REQUIRE(a != nullptr, "kg_genblock_lds_bytes: null args");
    if (check_dims(dims_of(a), "kg_genblock_lds_bytes")) return -2;
    GbLayout L;
    const long b = make_layout(dims_of(a), false, L);
    if (b < 0) return -1;
    // 16-byte row loads of the matrix-core contractions
    if (L.mfma0 && !(aligned16(a->wg) && (a->res_kind != 2 || aligned16(a->wr)))) return -1;
    if (L.mfma1 && !aligned16(a->wt)) return -1;
    return b;
}

extern "C" int64_t kg_genblock_workspace_bytes(const KgGenBlockArgs* a) {
    KG_REQUIRE(a != nullptr, "kg_genblock_workspace_bytes: null args");
    if (check_dims(dims_of(a), "kg_genblock_workspace_bytes")) return -1;
    return (int64_t)2 * a->N * a->C * 2 * (int64_t)sizeof(float);
}

extern "C" int kg_genblock_fwd(const KgGenBlockArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr, "kg_genblock_fwd: null args");
    if (int rc = check_dims(dims_of(a), "kg_genblock_fwd")) return rc;
    KG_REQUIRE(a->groups >= 1 && a->N % a->groups == 0, "kg_genblock_fwd: N=%d is not a multiple of groups=%d", a->N, a->groups);
    KG_REQUIRE(4L * a->groups * a->C <= GB_VALU_MAXMK, "kg_genblock_fwd: groups=%d x C=%d statistics do not fit the merge area", a->groups, a->C);
    const int64_t lds = kg_genblock_lds_bytes(a);
    KG_REQUIRE(lds >= 0, "kg_genblock_fwd: the block does not fit the fused form (kg_genblock_lds_bytes)");
    KG_REQUIRE(a->wg && a->wt && a->b, "kg_genblock_fwd: null weight / adjacency pointer");
    KG_REQUIRE(a->res_kind != 2 || a->wr, "kg_genblock_fwd: conv residual without its weight");
    KG_REQUIRE(a->u != nullptr || a->Vc == a->V, "kg_genblock_fwd: Vc=%d != V=%d without an up-sampling matrix", a->Vc, a->V);
    KG_REQUIRE(a->x.p != nullptr || (a->pu.p != nullptr && a->xout.p != nullptr), "kg_genblock_fwd: neither a finished input nor a pending tail (pu, xout)");
    KG_REQUIRE(a->yc.p && a->z.p && a->uo.p && (a->res_kind == 0 || a->r.p), "kg_genblock_fwd: null tape tensor");
    const bool bn_t = a->bn_t != 0, bn_r = a->res_kind == 2;
    if (bn_t) KG_REQUIRE(a->bt_.coef, "kg_genblock_fwd: BatchNorm (tcn) without a coefficient buffer");
    if (bn_r) KG_REQUIRE(a->br_.coef, "kg_genblock_fwd: BatchNorm (residual) without a coefficient buffer");
    if (bn_t || bn_r) {
        KG_REQUIRE(a->ws && a->ws_bytes >= kg_genblock_workspace_bytes(a), "kg_genblock_fwd: workspace too small");
        KG_REQUIRE(a->counters && a->counters_len >= 1, "kg_genblock_fwd: a zeroed ticket counter is needed");
        KG_REQUIRE(2 * a->C <= 16 * NT, "kg_genblock_fwd: too many channels");
    }
    GbLayout L;
    make_layout(dims_of(a), false, L);
    static unsigned long long attr_mask = 0;
    const bool first = kg_first_on_device(attr_mask);
    const Dims d = dims_of(a);
#define GB_ATTR(...) do { using G_ = GbGeo<__VA_ARGS__>; KG_SET_DYN_LDS(kg_genblock_fwd_kernel<G_>, GB_MAX_LDS); } while (0)
    if (first) {                                    // (per device: every instantiation, before any of them launches)
        GB_GEOMETRIES(GB_ATTR);
        KG_SET_DYN_LDS(kg_genblock_fwd_kernel<GbRt>, GB_MAX_LDS);
    }
#undef GB_ATTR
#define GB_TRY(...) do { using G_ = GbGeo<__VA_ARGS__>; \
        if (geo_matches<G_>(d, a->bn_t) && !kg_env().gb_rt) { \
            hipLaunchKernelGGL(kg_genblock_fwd_kernel<G_>, dim3(a->N), dim3(NT), (size_t)lds, (hipStream_t)stream, *a, L); \
            return kg_launch_status("kg_genblock_fwd"); } } while (0)
    GB_GEOMETRIES(GB_TRY);
#undef GB_TRY
    hipLaunchKernelGGL(kg_genblock_fwd_kernel<GbRt>, dim3(a->N), dim3(NT), (size_t)lds, (hipStream_t)stream, *a, L);
    return kg_launch_status("kg_genblock_fwd");
}

extern "C" int64_t kg_genblock_bwd_lds_bytes(const KgGenBlockBwdArgs* a) {
    KG_REQUIRE(a != nullptr, "kg_genblock_bwd_lds_bytes: null args");
    if (check_dims(dims_of(a), "kg_genblock_bwd_lds_bytes")) return -2;
    GbLayout L;
    return make_layout(dims_of(a), true, L);
}

extern "C" int64_t kg_genblock_bwd_workspace_bytes(const KgGenBlockBwdArgs* a) {
    KG_REQUIRE(a != nullptr, "kg_genblock_bwd_workspace_bytes: null args");
    if (check_dims(dims_of(a), "kg_genblock_bwd_workspace_bytes")) return -1;
    return (int64_t)a->N * a->Cin * 4 * (int64_t)sizeof(float);
}

extern "C" int kg_genblock_bwd(const KgGenBlockBwdArgs* a, void* stream) {
    KG_REQUIRE(a != nullptr, "kg_genblock_bwd: null args");
    if (int rc = check_dims(dims_of(a), "kg_genblock_bwd")) return rc;
    const int64_t lds = kg_genblock_bwd_lds_bytes(a);
    KG_REQUIRE(lds >= 0, "kg_genblock_bwd: the block does not fit the fused form (kg_genblock_bwd_lds_bytes)");
    KG_REQUIRE(a->wg && a->wt && a->b && a->coef, "kg_genblock_bwd: null weight / adjacency / coefficient pointer");
    KG_REQUIRE(a->res_kind != 2 || a->wr, "kg_genblock_bwd: conv residual without its weight");
    KG_REQUIRE(a->u != nullptr || a->Vc == a->V, "kg_genblock_bwd: Vc=%d != V=%d without an up-sampling matrix", a->Vc, a->V);
    KG_REQUIRE(a->g.p && a->out.p, "kg_genblock_bwd: null g / out");
    KG_REQUIRE(!a->bn_t || a->uo.p, "kg_genblock_bwd: BatchNorm (tcn) without the taped u");
    KG_REQUIRE(a->res_kind != 2 || a->r.p, "kg_genblock_bwd: BatchNorm (residual) without the taped r");
    KG_REQUIRE(a->du.p && a->gyc.p && a->zf.p && a->gx.p, "kg_genblock_bwd: null output tensor");
    KG_REQUIRE(a->res_kind != 2 || a->dr.p, "kg_genblock_bwd: conv residual without dr");
    if (a->px.p) {
        KG_REQUIRE(a->Tc * a->Vc <= 512, "kg_genblock_bwd: previous block's statistics need Tc * Vc <= 512 (got %d)", a->Tc * a->Vc);
        KG_REQUIRE(a->pcoef, "kg_genblock_bwd: previous block's statistics without pcoef");
        KG_REQUIRE(a->pu.p == nullptr || (a->pmean_t && a->prstd_t), "kg_genblock_bwd: previous BatchNorm (tcn) needs its statistics");
        KG_REQUIRE(a->pmean_r == nullptr || (a->pr.p && a->prstd_r), "kg_genblock_bwd: previous BatchNorm (residual) needs r and its statistics");
        KG_REQUIRE(a->ws && a->ws_bytes >= kg_genblock_bwd_workspace_bytes(a), "kg_genblock_bwd: workspace too small");
        KG_REQUIRE(a->counters && a->counters_len >= 1, "kg_genblock_bwd: a zeroed ticket counter is needed");
    }
    GbLayout L;
    make_layout(dims_of(a), true, L);
    static unsigned long long attr_mask = 0;
    const bool first = kg_first_on_device(attr_mask);
    const Dims d = dims_of(a);
#define GB_ATTR(...) do { using G_ = GbGeo<__VA_ARGS__>; KG_SET_DYN_LDS(kg_genblock_bwd_kernel<G_>, GB_MAX_LDS); } while (0)
    if (first) {                                    // (per device: every instantiation, before any of them launches)
        GB_GEOMETRIES(GB_ATTR);
        KG_SET_DYN_LDS(kg_genblock_bwd_kernel<GbRt>, GB_MAX_LDS);
    }
#undef GB_ATTR
#define GB_TRY(...) do { using G_ = GbGeo<__VA_ARGS__>; \
        if (geo_matches<G_>(d, a->bn_t) && !kg_env().gb_rt) { \
            hipLaunchKernelGGL(kg_genblock_bwd_kernel<G_>, dim3(a->N), dim3(NT), (size_t)lds, (hipStream_t)stream, *a, L); \
            return kg_launch_status("kg_genblock_bwd"); } } while (0)
    GB_GEOMETRIES(GB_TRY);
#undef GB_TRY
    hipLaunchKernelGGL(kg_genblock_bwd_kernel<GbRt>, dim3(a->N), dim3(NT), (size_t)lds, (hipStream_t)stream, *a, L);
    return kg_launch_status("kg_genblock_bwd");
}

#ifdef KG_GB_STAMP
// debug builds only (tools/time_genblock.py): the phase stamps of workgroup 0 of the last forward (dir 0) / backward (dir 1) launch
extern "C" int kg_gb_read_stamps(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kg_gb_stamps), sizeof(long long) * 32, 0, hipMemcpyDeviceToHost);
}
#endif

