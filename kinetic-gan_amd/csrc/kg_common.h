// Shared helpers for the gfx950 kernels of libkgan_hip.so (see include/kgan_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kgan_hip.h"

void kg_set_error(const char* fmt, ...);

#define KG_REQUIRE(cond, ...)           \
    do {                                \
        if (!(cond)) {                  \
            kg_set_error(__VA_ARGS__);  \
            return -1;                  \
        }                               \
    } while (0)

static inline int kg_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        kg_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

static inline int kg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Test / tuning switches (environment variables), read ONCE when the library is loaded - no getenv on
// the launch path.  kg_reload_env() (tests only: they flip switches inside one process) reads them again.
struct KgEnv {
    int conv_plan_tile;   // KG_CONV_PLAN="<tile>,<nsplit>": tile or -1
    int conv_plan_split;
    int conv_kw;            // KG_CONV_KW: 0 = never split K across the waves of a workgroup (K32x32 tile), default on
    int conv_fast;          // KG_CONV_FAST: 0 = never the full-slice (scalar-offset) instantiation of kg_conv_kernel (A/B, tests)
    int conv_many;          // KG_CONV_MANY: 0 = kg_conv_many runs its problems one launch each (A/B, tests)
    int conv_xcd_min;       // KG_CONV_XCD_MIN: tiles from which a launch uses the XCD-aware tile order (tuning; 0 = built-in)
    int conv_tiny;          // KG_CONV_TINY: 0 = never the tiny-channel streaming kernel (A/B, tests of the MFMA tiles)
    int agg_stream;       // KG_AGG_STREAM: -1 unset, 0, 1
    int agg_mfma;         // KG_AGG_MFMA: -1 unset, 0, 1
    int agg_mfma_sub;     // KG_AGG_MFMA_SUB or 0
    int agg_mfma_grid;    // KG_AGG_MFMA_GRID or 0
    int agg_outer_mfma;   // KG_AGG_OUTER_MFMA: -1 unset, 0, 1
    int agg_outer_budget; // KG_AGG_OUTER_BUDGET: workgroups of one kg_agg_outer_many launch, dealt to its jobs by work (0: built-in)
    int wgrad_bigcols;    // KG_WGRAD_BIGCOLS: columns from which kg_wgrad(_many) takes its 128 x 128 tile (tuning; -1 = built-in 4096)
    int wgrad_budget;     // KG_WGRAD_BUDGET: workgroups of equal cost a kg_wgrad_many pass is cut into (0 = 6144)
    int wgrad_split;      // KG_WGRAD_SPLIT: 1 = the bf16-split tiles of kg_wgrad.hip (opt-in: measured no faster overall), default the fp32 tiles
    int aggconv_plan;     // KG_AGGCONV_PLAN "<BM><KS>" or 0
    int conv_ring;        // KG_CONV_RING: -1 unset (the plan decides), 0 = never the persistent LDS-ring form, 1 = wherever it can run
    int conv_ring_stagger; // KG_CONV_RING_STAGGER: s_sleep units the second workgroup of a CU starts late (window tiles with two workgroups per CU)
    int conv_ring_tile;   // KG_CONV_RING_TILE: force the ring tile (kg_conv_ring.hip: 0..5), -1 = automatic
    int conv_inkernel;    // KG_CONV_INKERNEL: 0 = K-split launches of kg_conv always finish in the separate epilogue launch (A/B, tests)
    int conv_inkernel_max;  // KG_CONV_INKERNEL_MAX: the largest K-split count completed in-kernel (default 4)
    int conv_plain_epi;   // KG_CONV_PLAIN_EPI: 0 = never the add- / mask-free epilogue instantiations of kg_conv (A/B, tests)
    int conv_bs_asm;      // KG_CONV_BS_ASM: 0 = never the hand-scheduled all-window instantiation of the bf16-split form (A/B, tests)
    int conv_bs;          // KG_CONV_BS: -1 unset / 0 = never the bf16-split LDS-staged form (a caller's wpack still selects it when unset), 1 = wherever it can run, 2 = the round-5 plan rule
    int gb_rt;            // KG_GB_RT=1: kg_genblock always on the run-time-geometry instantiation (tests, A/B)
    int conv_bs_tile;     // KG_CONV_BS_TILE: force its tile variant (0: 64 x 128, 1: 32 x 128, 2: 128 x 64), -1 = automatic
};
const KgEnv& kg_env();

// the persistent LDS-ring form of kg_conv (round 5: parity-green, 1.5-2x slower than the direct kernel at every training
// shape, never chosen by the plan).  It lives in tools/probe/kg_conv_ring.hip and is only compiled in by
// `build.py --with-ring` (-DKG_WITH_RING); the default library answers "not eligible".
#ifdef KG_WITH_RING
bool kg_ring_eligible(const KgConvArgs* a);
bool kg_ring_tile_ok(const KgConvArgs* a, int tile);
int kg_ring_tile_count();
void kg_ring_tile_dims(int tile, int* bm, int* bn, int* wgpc);
int kg_ring_launch(const KgConvArgs* a, int tile, hipStream_t s);
#else
inline bool kg_ring_eligible(const KgConvArgs*) { return false; }
inline bool kg_ring_tile_ok(const KgConvArgs*, int) { return false; }
inline int kg_ring_tile_count() { return 0; }
inline void kg_ring_tile_dims(int, int* bm, int* bn, int* wgpc) { *bm = *bn = *wgpc = 0; }
inline int kg_ring_launch(const KgConvArgs*, int, hipStream_t) { return -1; }
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: a launcher raises it once per device ordinal
// (bit d of a static mask; a race only repeats the calls) and reports a failing call instead of an opaque launch error
inline bool kg_first_on_device(unsigned long long& mask) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
    if ((mask >> d) & 1ull) return false;
    mask |= 1ull << d;
    return true;
}
#define KG_SET_DYN_LDS(kern_, bytes_) do { \
    hipError_t e_ = hipFuncSetAttribute((const void*)(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes_)); \
    if (e_ != hipSuccess) { kg_set_error("hipFuncSetAttribute(%s, %d bytes of LDS): %s", #kern_, (int)(bytes_), hipGetErrorString(e_)); return (int)e_; } } while (0)

// x / d and x % d for a launch-constant divisor without the ~30-instruction generic 32-bit division (x < 2^31):
// q = umulhi(x, mul) >> shr  (the round-up magic number of Granlund-Montgomery, found on the host)
struct FastDiv {
    unsigned d, mul, shr;
    __host__ static FastDiv make(unsigned d) {
        FastDiv f;
        f.d = d;
        if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
        unsigned lg = 0;
        while ((1u << lg) < d) ++lg;
        const unsigned p = 31 + lg;
        f.mul = (unsigned)(((1ull << p) + d - 1) / d);
        f.shr = p - 32;
        return f;
    }
    __device__ __forceinline__ unsigned div(unsigned x) const { return d <= 1 ? x : __umulhi(x, mul) >> shr; }
    __device__ __forceinline__ void divmod(unsigned x, unsigned& q, unsigned& r) const { q = div(x); r = x - q * d; }
};

typedef float kg_f32x16 __attribute__((ext_vector_type(16)));

// activation and its derivative expressed on the activation OUTPUT
__device__ __forceinline__ float kg_act(float v, int act, float slope) {
    if (act == KG_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == KG_ACT_TANH) return tanhf(v);
    return v;
}
__device__ __forceinline__ float kg_dact_from_out(float o, int act, float slope) {
    if (act == KG_ACT_LRELU) return o > 0.f ? 1.f : slope;
    if (act == KG_ACT_TANH) return 1.f - o * o;
    return 1.f;
}

// Touch every 64-byte line of the kernel-argument segment at kernel entry - one batch of scalar loads and one wait, in ONE
// assembly block.  The compiler fetches a by-value argument struct piece by piece as the code needs it, each first touch of
// a line a scalar-cache miss of its own in front of a dependent wait: a 400-byte struct is seven of them in a row on every
// CU's first wave.  After the block every line is in the scalar cache and the kernel's own loads hit.  (The loaded values
// are dropped; the wait is inside the block because the compiler treats an asm output as available at once and could hand
// the register to something else while the load is still in flight.)
// Measured (whole iteration, same box): kg_conv (KgConvArgs + Split = 420 bytes, 72 launches) 3.174 -> 3.14 ms; on the
// kernels with arguments of <= 4 lines (aggregation, generator blocks, mapping network) it costs 0.1-0.2 us per launch
// instead - only kg_conv uses it.  (experiment switch: -DKG_KARG_WARM=0)
#ifndef KG_KARG_WARM
#define KG_KARG_WARM 1
#endif
// (the base goes through readfirstlane: below -O2 the compiler handed a VGPR pair to the "s" operand of a base that was
// derived from a loop-found job index)
__device__ __forceinline__ unsigned long long kg_uniform_u64(unsigned long long u) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int LINES, int I>
struct KgKargTouchLoad {
    static __device__ __forceinline__ void go(unsigned (&t)[LINES], unsigned long long kp) {
        asm volatile("s_load_dword %0, %1, %2" : "=&s"(t[I]) : "s"(kp), "n"(I * 64));
        KgKargTouchLoad<LINES, I + 1>::go(t, kp);
    }
};
template <int LINES>
struct KgKargTouchLoad<LINES, LINES> {
    static __device__ __forceinline__ void go(unsigned (&)[LINES], unsigned long long) {}
};
template <int LINES, int I>
struct KgKargTouchUse {
    static __device__ __forceinline__ void go(const unsigned (&t)[LINES]) {
        asm volatile("" ::"s"(t[I]));
        KgKargTouchUse<LINES, I + 1>::go(t);
    }
};
template <int LINES>
struct KgKargTouchUse<LINES, LINES> {
    static __device__ __forceinline__ void go(const unsigned (&)[LINES]) {}
};
// the first BYTES of the segment (never a byte beyond them)
template <int BYTES>
__device__ __forceinline__ void kg_kernarg_warm() {
#if KG_KARG_WARM
    constexpr int LINES = (BYTES + 63) / 64;
    static_assert(BYTES % 4 == 0 && LINES <= 40, "kg_kernarg_warm: at most 40 lines (one SGPR each)");
    unsigned t[LINES];
    // (consecutive volatile asm statements keep their order: loads, wait, then the registers are released)
    KgKargTouchLoad<LINES, 0>::go(t, kg_uniform_u64((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr()));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KgKargTouchUse<LINES, 0>::go(t);
#endif
}
// BYTES at the (uniform) offset byte_off of an argument block of TOTAL bytes - a job of a table that a workgroup has
// located.  The window of touched dwords is moved back where it would reach beyond the block.
template <int BYTES, int TOTAL>
__device__ __forceinline__ void kg_kernarg_warm_at(unsigned byte_off) {
#if KG_KARG_WARM
    constexpr int LINES = (BYTES + 63) / 64 + 1;        // (the job need not start on a line)
    static_assert(TOTAL % 4 == 0 && TOTAL >= 64 * LINES && LINES <= 40, "kg_kernarg_warm_at: window");
    unsigned start = byte_off & ~63u;
    if (start + 64u * LINES > (unsigned)TOTAL) start = (unsigned)TOTAL - 64u * LINES;
    unsigned t[LINES];
    KgKargTouchLoad<LINES, 0>::go(t, kg_uniform_u64((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + start));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KgKargTouchUse<LINES, 0>::go(t);
#endif
}

// make a pointer provably wave-uniform for the compiler (else every buffer op gets a waterfall loop)
__device__ __forceinline__ void* kg_uniform_ptr(const void* p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (void*)(((unsigned long long)hi << 32) | lo);
}

// Deterministic sum over S partial slabs: ws[k*per + i], k < S.  Block = 256 threads = 64 outputs x 4
// slab lanes (one wave per lane: every load is one coalesced 256-byte row); returns the total in the
// threads of wave 0 (others return 0 and must not store).
__device__ __forceinline__ float kg_slab_sum_256(const float* ws, long per, long i, bool valid, int S) {
    __shared__ float kg_red[4][64];
    const int o = threadIdx.x & 63, sub = threadIdx.x >> 6;
    float s = 0.f;
    if (valid) {
#pragma unroll 8
        for (int k = sub; k < S; k += 4) s += ws[(long)k * per + i];
    }
    kg_red[sub][o] = s;
    __syncthreads();
    if (sub != 0) return 0.f;
    return (kg_red[0][o] + kg_red[1][o]) + (kg_red[2][o] + kg_red[3][o]);
}
