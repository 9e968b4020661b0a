/*
 * kgan_hip.h - C ABI of libkgan_hip.so: the MI355X (gfx950) st_gcn hot path of Kinetic-GAN.
 *
 * The reference has no FFI layer: its hot path is reached through Python
 * nn.Modules that call stock ATen ops.  Each entry point below names the reference op(s) it
 * replaces (file:line relative to the reference repo).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every tensor is fp32 and lives in device memory owned by the caller (borrowed from torch);
 *    the library never allocates, frees or synchronises; scratch is passed in by the caller,
 *    sized by the matching *_workspace_bytes() query;
 *  - a "plane tensor" is a logical (N, C, T, V) array whose (t, v) plane is contiguous:
 *    element (n,c,t,v) sits at  p + n*sN + c*sC + t*V + v   (strides in elements).  Both the
 *    reference's NCHW layout (sN = C*T*V, sC = T*V) and the channel-major layout this library
 *    prefers between blocks (sN = T*V, sC = N*T*V) are plane tensors;
 *  - `stream` is a hipStream_t passed as void* (the caller passes
 *    torch.cuda.current_stream().cuda_stream); kernels are only enqueued on it;
 *  - return value: 0 = ok, negative = invalid argument / unsupported shape (see
 *    kg_last_error()), positive = hipError_t from the launch;
 *  - re-entrant and thread-safe: no global mutable state except the thread-local error string.
 */
#ifndef KGAN_HIP_H
#define KGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KG_ABI_VERSION 9

enum { KG_ACT_NONE = 0, KG_ACT_LRELU = 1, KG_ACT_TANH = 2 };
enum { KG_TAP_TIME = 0,   /* tap d reads the input at time  t*stride + d - (taps-1)/2            */
       KG_TAP_CHANBLOCK = 1 /* tap d reads input channels [d*Cin, (d+1)*Cin) at the same time  */ };

/* ---- library info -------------------------------------------------------------------------- */
int         kg_abi_version(void);
const char* kg_arch(void);              /* "gfx950" */
const char* kg_last_error(void);        /* thread-local, valid until the next failing call    */
/* The KG_* test / tuning switches (environment variables) are read once when the library is loaded;
 * a test that flips one inside a running process calls this to have them read again.  Not for production use.  */
void        kg_reload_env(void);

/* ---- channel contraction ("tap GEMM") on the fp32 matrix cores ---------------------------------
 * One launch computes, for every output column j = (n, t, v):
 *
 *   out[m, j] = act( sum_g sum_d sum_c  W_g(d, m, c) * X_g[c (+ d*Cin_g if CHANBLOCK), src_g(j, d)]
 *                    + bias0[m] + bias1[m] + add[m, (n, t*add_tstride, v)] )  *  lrelu'(mask[m, j])
 *
 * (the last factor only when `mask` is given: slope where mask <= 0, else 1 - the LeakyReLU derivative expressed
 *  on the activation OUTPUT `mask`, which folds "g * act'(out)" of the consumer into the producing launch)
 *
 * forward  : src = (n, t*stride + shift_d, vmap ? vmap[v] : v)        (zero outside [0,T_in))
 * transposed: src = (n, (t - shift_d)/stride, vmap[v])  if divisible, in range and vmap[v] >= 0
 *             (the adjoint of `forward` w.r.t. its input; vmap is then the inverse vertex map)
 * W_g(d, m, c) = w + d*w_sT + (m / w_MB)*w_sMB + (m % w_MB)*w_sO + c*w_sI
 *
 * Replaces: the 1x1 conv of ConvTemporalGraphical (tgcn.py:48-55,61), the (3x1) temporal conv
 * (discriminator.py:99-105, generator.py:134-140), the 1x1 residual conv (discriminator.py:115-120,
 * generator.py:154-159), "+ res" (discriminator.py:130, generator.py:176), the vertex gather
 * `tensor[:,:,:,keep]` (discriminator.py:139-142), the nearest T/2 resize
 * (discriminator.py:134: only even frames are computed), LeakyReLU / tanh
 * (discriminator.py:136, generator.py:182), and - in transposed mode - their backward-data passes.
 */
typedef struct KgConvGroup {
    const float* x;  int64_t x_sN, x_sC;
    int32_t Cin, T_in, V_in;
    int32_t x_lead;                 /* floats in front of x that belong to the same allocation (>= 0); the
                                       128-bit load path needs >= 32 (it reads up to one frame before a row) */
    const int32_t* vmap;            /* device ptr, V_out entries, or NULL                         */
    const float* w;  int64_t w_sT, w_sO, w_sI, w_sMB;  int32_t w_MB;
    int32_t taps, tap_mode, t_stride, transposed;
} KgConvGroup;

typedef struct KgConvArgs {
    int32_t N, M, T_out, V_out;
    float* out;  int64_t o_sN, o_sC;
    int32_t ngroups;
    KgConvGroup g[2];
    const float* bias0;  const float* bias1;
    const float* add;  int64_t a_sN, a_sC;  int32_t a_tstride;
    int32_t act;  float slope;
    float* ws;  int64_t ws_bytes;    /* scratch for K-split partial sums (kg_conv_workspace_bytes)  */
    const float* mask;  int64_t m_sN, m_sC;   /* optional (N, M, T_out, V_out) plane tensor, see above      */
    int32_t* sync;  int32_t sync_len;         /* optional: sync_len zeroed ticket counters.  A K-split launch with at most
                                                 sync_len output tiles then completes them itself - the last workgroup of
                                                 a tile to arrive sums the partial slabs in split order and runs the
                                                 epilogue - instead of a second launch; the counters are
                                                 zero again when the launch ends.  NULL: the separate epilogue launch.
                                                 Launches that share the counters must not overlap.                   */
    int32_t o_tstride;                        /* 0 / 1: output frames follow each other; s > 1: output frame `to` is
                                                 written at frame to * s of `out` (a transposed stride-2 temporal conv
                                                 runs as two launches, one per output-frame parity, each with only
                                                 the taps that reach it)                                            */
    const void* wpack;  int64_t wpack_bytes;  /* optional (ABI v8): the groups' weights as kg_conv_pack wrote them.  A launch
                                                 the bf16-split form can run (kg_conv_pack_bytes > 0) then runs its tile
                                                 kernel on them - no weight-pack launch, no workspace; any other launch
                                                 ignores the field.  The caller re-packs when the weights change.    */
} KgConvArgs;

int64_t kg_conv_workspace_bytes(const KgConvArgs* a);   /* 0 when the launch needs no scratch (K-split partial slabs; the
                                                           packed weights of the opt-in bf16-split form, KG_CONV_BS=1:
                                                           then `ws` must be 16-byte aligned)                          */
/* which kernel configuration kg_conv would pick (tests / tuning): tile 0..4 = direct kernel with
 * BMxBN = 128x128, 64x128, 32x128, 64x64, 32x64; 9 = 32x32 with the waves splitting K; 11 = tiny-channel streaming
 * kernel; 20.. = tile of the persistent LDS-ring form (KG_CONV_RING); 40..42 = tile of the bf16-split form (KG_CONV_BS) */
int     kg_conv_plan_info(const KgConvArgs* a, int32_t* tile, int32_t* nsplit);
/* several independent problems (own operands, geometry and epilogue each) in ONE launch where the launcher's plans allow it
 * - full K-slices, no K-split, the same weight orientation - and one launch each otherwise; results are those of
 * kg_conv(job i) for every i.  The jobs must not write what another job of the call reads.                          */
#define KG_CONV_MANY_MAX 4
int     kg_conv_many(const KgConvArgs* jobs, int32_t njobs, void* stream);
/* tests / tuning: *tile = the plan tile (0 / 1 / 2) of the shared launch, or -1 when kg_conv_many would launch one by one */
int     kg_conv_many_plan(const KgConvArgs* jobs, int32_t njobs, int32_t* tile);
int     kg_conv(const KgConvArgs* a, void* stream);
/* The weights of a launch in the bf16-split form's layout: every fp32 weight as three bf16 terms,
 * [step = (group, 32-channel slice, tap)][term][8-channel octet][row padded to 128] x 16 bytes.  The layout depends on the
 * groups' weights, Cin, taps and M only - a buffer packed once serves launches of any batch size until the weights change.
 * kg_conv_pack_bytes: bytes to allocate (16-byte aligned), 0 when the bf16-split form cannot run the launch, -1 on bad
 * arguments.  Replaces nothing in the reference (a layout transform of discriminator.py:99-105,115-120's weights).       */
int64_t kg_conv_pack_bytes(const KgConvArgs* a);
int     kg_conv_pack(const KgConvArgs* a, void* wpack, int64_t wpack_bytes, void* stream);

/* ---- weight gradient of the tap GEMM -----------------------------------------------------------
 *   dW(d, m, c) = sum_j  G[m, j] * X[c (+ d*Cin if CHANBLOCK), src(j, d)]      (src as `forward`)
 * written to  dw + d*w_sT + m*w_sO + c*w_sI.  Split over column ranges into `splits` partial
 * slabs in `ws` (deterministic two-pass reduction, no atomics).
 * Replaces aten::convolution_backward's weight part for the three convs above.                 */
/* an additional (g, x) operand pair of the same layer geometry whose product is added into the same dW: the same
 * weight receives several gradient contributions in one WGAN-GP backward pass (the real+fake batch, the penalty's
 * forward graph and its double-backward graph); one launch over the concatenated column ranges replaces three. */
typedef struct KgWgradPair {
    int32_t N;
    const float* g;  int64_t g_sN, g_sC;
    const float* x;  int64_t x_sN, x_sC;
} KgWgradPair;

typedef struct KgWgradArgs {
    int32_t N, M, T_out, V_out;
    const float* g;  int64_t g_sN, g_sC;
    const float* x;  int64_t x_sN, x_sC;
    int32_t Cin, T_in, V_in;
    const int32_t* vmap;
    int32_t taps, tap_mode, t_stride;
    float* dw;  int64_t w_sT, w_sO, w_sI;
    float* ws;  int64_t ws_bytes;
    int32_t accumulate;             /* 0: dw = result; 1: dw += result (gradient accumulation in place, e.g. into
                                       the flat gradient bucket the all-reduce and the optimizer work on)        */
    int32_t nextra;                 /* 0..2 additional operand pairs                                               */
    KgWgradPair extra[2];
    int32_t defer_reduce;           /* 1: only write the partial slabs; the caller reduces them later with
                                       kg_wgrad_reduce_many (ws must stay alive until then)                        */
} KgWgradArgs;

int64_t kg_wgrad_workspace_bytes(const KgWgradArgs* a);   /* = splits * taps * M * Cin * 4                        */
int     kg_wgrad(const KgWgradArgs* a, void* stream);

/* The slab reductions of several deferred kg_wgrad launches in ONE launch (a backward pass produces the weight
 * gradients of all layers back to back; 17-19 reductions of a few microseconds each become one).              */
typedef struct KgWgradReduceJob {
    const float* ws;  float* dw;
    int64_t w_sT, w_sO, w_sI;
    int32_t taps, M, Cin, splits, accumulate;
} KgWgradReduceJob;
#define KG_WGRAD_REDUCE_MAX_JOBS 24
typedef struct KgWgradReduceJobs {
    int32_t njobs;
    KgWgradReduceJob job[KG_WGRAD_REDUCE_MAX_JOBS];
} KgWgradReduceJobs;
int     kg_wgrad_reduce_many(const KgWgradReduceJobs* jobs, void* stream);

/* The weight gradients of SEVERAL layers (jobs[0..njobs), each a complete KgWgradArgs with its operand pairs, dw,
 * strides and accumulate flag; jobs[i].ws / ws_bytes / defer_reduce are ignored) in shared launches: the column
 * ranges of all layers are split against ONE workgroup budget and the slab reductions follow in the same call.
 * No two jobs may write the same dw.  Replaces one kg_wgrad + one reduction per layer of a backward pass
 * (aten::convolution_backward's weight halves of all of discriminator.py:99-120 / generator.py:134-159).        */
int64_t kg_wgrad_many_workspace_bytes(const KgWgradArgs* jobs, int32_t njobs);
int     kg_wgrad_many(const KgWgradArgs* jobs, int32_t njobs, float* ws, int64_t ws_bytes, void* stream);

/* ---- spatial graph aggregation -------------------------------------------------------------------
 * A is (K, V, W) row-major fp32 in device memory (the effective adjacency A[lvl]*importance,
 * optionally restricted to kept columns, or the up-sampling matrix with K = 1); it is staged in
 * LDS once per workgroup.
 *
 *  expand : out[k*C + c, (n, t', w)] = sum_v x[c, (n, t'/rep, v)] * A[k, v, w]       T' = T*rep
 *  reduce : out[c, (n, t, w)] = sum_{q<fold} sum_k sum_v y[k*C + c, (n, t*fold+q, v)] * A[k, v, w]
 *  outer  : dA[k, v, w] = sum_{c, n, t'} x[c, (n, t'/rep, v)] * y[k*C + c, (n, t', w)]
 *
 * Replaces torch.einsum('nkctv,kvw->nctw') (tgcn.py:66) in both operand orders, its two
 * gradients, upsample_s + the nearest T up-sampling of the generator (generator.py:172,185-200;
 * K = 1, A = U) and their adjoints.                                                             */
typedef struct KgAggArgs {
    int32_t N, C, K, V, W;          /* C = channels of the un-expanded side                       */
    int32_t T;                      /* frames of the (t,V) side: x for expand/outer, out for reduce */
    int32_t rep;                    /* expand/outer: rep; reduce: fold                             */
    const float* a;                 /* (K, V, W)                                                   */
    const float* x;  int64_t x_sN, x_sC;   /* expand/outer: x (C ch, V) ; reduce: y (K*C ch, V)    */
    const float* y;  int64_t y_sN, y_sC;   /* outer only: y (K*C ch, W)                            */
    float* out;  int64_t o_sN, o_sC;       /* expand: (K*C ch, W); reduce: (C ch, W); outer: dA    */
    float* ws;  int64_t ws_bytes;          /* outer only                                           */
    int32_t a_transposed;                  /* expand / reduce: `a` is stored (K, W, V): A[k][v][w] = a[(k*W + w)*V + v]
                                              (the adjoint passes use A^T without materialising it)              */
    int32_t defer_sum;                     /* outer: 1 = only the partial slabs are written to ws; the caller finishes
                                              several launches at once with kg_agg_outer_sum_many                 */
    /* reduce only (fold = 1), optional epilogue - the backward pass of a discriminator block (ABI v5):
     *   out[n,c,t,w] = ( aggregate + [t % r_tstride == 0 and r_inv[w] >= 0] res[n,c,t/r_tstride,r_inv[w]] ) * lrelu'(mask[n,c,t,w])
     * res: (N, C, r_T, r_V) plane tensor, the residual branch's input gradient at the frames / vertices the branch reads
     * (discriminator.py:115-120,134,139-142); r_inv (W): vertex of res that w reads or -1, NULL = identity;
     * mask: (N, C, T, W) activation output of the previous block (slope where <= 0).  Either may be NULL.          */
    const float* res;  int64_t r_sN, r_sC;  int32_t r_T, r_V, r_tstride;  const int32_t* r_inv;
    const float* mask;  int64_t m_sN, m_sC;  float slope;
} KgAggArgs;

int     kg_agg_expand(const KgAggArgs* a, void* stream);
int     kg_agg_reduce(const KgAggArgs* a, void* stream);
int64_t kg_agg_outer_workspace_bytes(const KgAggArgs* a);
int     kg_agg_outer(const KgAggArgs* a, void* stream);
/* the slab reductions of several deferred kg_agg_outer launches (one per block of a backward pass) in ONE launch */
int     kg_agg_outer_slabs(const KgAggArgs* a);          /* partial slabs kg_agg_outer writes for these arguments */
typedef struct KgOuterSumJob { const float* ws; float* out; int32_t nout, slabs; } KgOuterSumJob;
#define KG_OUTER_SUM_MAX_JOBS 16
typedef struct KgOuterSumJobs { int32_t njobs; KgOuterSumJob job[KG_OUTER_SUM_MAX_JOBS]; } KgOuterSumJobs;
int     kg_agg_outer_sum_many(const KgOuterSumJobs* jobs, void* stream);
/* several kg_agg_outer problems (every job with its own ws / ws_bytes / out; defer_sum ignored) in one launch for the
 * matrix-core form plus one launch for all slab sums: the adjacency gradients of a whole backward pass            */
int     kg_agg_outer_many(const KgAggArgs* jobs, int32_t njobs, void* stream);

/* ---- fused aggregation + gcn contraction of a discriminator block ("disc block forward", first half) -----------
 *   out[m, (n,t,w)] = sum_k sum_c W(k,m,c) * ( sum_v x[c, (n,t,v)] * A[k,v,w] )  + add[m, (n, t*a_tstride, w)]
 * = ConvTemporalGraphical (tgcn.py:58-68) in aggregate-first order restricted to the vertices the block keeps
 * (discriminator.py:125-142): sum_k (W_k x) A_k == sum_k W_k (x A_k).  The K*Cin aggregated planes are formed in
 * LDS / registers as the B operand of the MFMA and never written to HBM - unless `xa` is given, then they are also
 * stored (the gcn weight gradient needs them).  The adjacency's fixed sparsity pattern comes as a neighbour table:
 * nbr[(k*W + w)*4 + p] = p-th source vertex v with A[k,v,w] != 0, or -1; pcount[k] = most entries of any column of
 * A_k (supported: <= 1 / 4 / 1, what graph_ntu.py / graph_h36m.py produce at every level); the VALUES are read
 * from `a` (the live A[lvl] * edge_importance, (K,V,W), or stored (K,W,V) with a_transposed).
 * kg_aggconv_supported: 1 if this launch geometry can take the fused kernel (else use kg_agg_expand + kg_conv).  */
typedef struct KgAggConvArgs {
    int32_t N, Cin, M, T, V, W, K;
    const float* x;  int64_t x_sN, x_sC;           /* (N, Cin, T, V) plane tensor                              */
    const float* a;  int32_t a_transposed;
    const int32_t* nbr;  int32_t pcount[3];
    const float* w;  int64_t w_sT, w_sO, w_sI;     /* W(k, m, c) = w + k*w_sT + m*w_sO + c*w_sI                */
    float* out;  int64_t o_sN, o_sC;               /* (N, M, T, W)                                             */
    const float* add;  int64_t a_sN, a_sC;  int32_t a_tstride;   /* optional; a_tstride 0 = one frame for all t */
    float* xa;  int64_t xa_sN, xa_sC;              /* optional (N, K*Cin, T, W)                                */
} KgAggConvArgs;

int kg_aggconv_supported(const KgAggConvArgs* a);
int kg_aggconv(const KgAggConvArgs* a, void* stream);

/* ---- generator st_gcn block on the coarse grid ("fused G-block", generator.py:168-182) -------------------------------
 * The block's 1x1 convs commute with upsample_s (x U, generator.py:185-200) and the nearest frame repeat
 * (generator.py:172), so ONE kg_conv launch computes y = [W_gcn; W_res] x on the block's INPUT grid (N, *, Tc, Vc) and
 *
 *  kg_gen_expand:  z[c,(n,t',w)]  = sum_k sum_vc y[k*C + c,(n,t'/rep,vc)] * B_k[vc,w],        B_k = U A_k   (Vc x V)
 *                  r[cr,(n,t',w)] = sum_vc rs[cr,(n,t'/rep,vc)] * U[vc,w] + rbias[cr]          T' = Tc*rep
 *                  = upsample_s + F.interpolate + tgcn.py:66's einsum of the gcn branch, and upsample_s + interpolate
 *                  (+ bias) of the residual branch (rs = the residual conv's rows of y, or the block input itself for an
 *                  identity residual); the up-sampled input and the fine-grid conv output never exist
 *  kg_gen_fold:    the adjoint:  y_out[k*C + c,(n,tc,vc)] = sum_{q<rep} sum_w z[c,(n,tc*rep+q,w)] * B_k[vc,w]
 *                                rs_out[cr,(n,tc,vc)]     = sum_{q<rep} sum_w r[cr,(n,tc*rep+q,w)] * U[vc,w]
 *                                zf[c,(n,tc,w)]           = sum_{q<rep} z[c,(n,tc*rep+q,w)]          (optional)
 *                  (z, r hold the gradients w.r.t. z and r here; zf feeds the adjacency gradient:
 *                   d B_k[vc,w] = sum zf[c,(.,w)] * y[k*C + c,(.,vc)] = kg_agg_outer(zf, y) transposed)
 *  kg_gen_adj_finish: out[k,v,w] (+)= a[k,v,w] * sum_vc u[vc,v] * dbt[k,w,vc]     for k < Kd, 0 beyond
 *                  = d edge_importance from the (Kd, V, Vc) outer products of all blocks in ONE launch
 *                  (generator.py:92-93: A[lvl] * importance; d A_k = U^T d B_k).
 * a: (K, V, V); u: (Vc, V) or NULL (no spatial up-sampling: Vc == V).  Either branch may be absent (z / r NULL).
 * Tensors are limited to 2^31 elements per launch (32-bit item indices).                                             */
typedef struct KgGenArgs {
    int32_t N, C, K, Cr, Tc, Vc, V, rep;
    const float* a;  const float* u;
    const float* b;                                          /* optional: the product U A_k (K, Vc, V) precomputed by
                                                                kg_gen_adj_prepare; NULL: formed from a and u in LDS   */
    const float* y;  float* y_out;  int64_t y_sN, y_sC;      /* (N, K*C, Tc, Vc): expand reads y, fold writes y_out */
    float* z;  int64_t z_sN, z_sC;                           /* (N, C, Tc*rep, V): expand writes, fold reads          */
    float* zf;  int64_t zf_sN, zf_sC;                        /* fold only, optional: (N, C, Tc, V)                     */
    const float* rs;  float* rs_out;  int64_t rs_sN, rs_sC;  /* (N, Cr, Tc, Vc)                                        */
    const float* rbias;                                      /* expand: (Cr) or NULL                                   */
    float* r;  int64_t r_sN, r_sC;                           /* (N, Cr, Tc*rep, V)                                     */
} KgGenArgs;

int kg_gen_expand(const KgGenArgs* a, void* stream);
int kg_gen_fold(const KgGenArgs* a, void* stream);

typedef struct KgGenAdjJob {
    const float* dbt;               /* (Kd, V, Vc) */
    const float* u;                 /* (Vc, V) or NULL */
    const float* a;                 /* (K, V, V) fixed adjacency A[lvl] or NULL (= 1) */
    float* out;                     /* (K, V, V) */
    int32_t K, Kd, V, Vc, accumulate;
} KgGenAdjJob;
#define KG_GEN_ADJ_MAX_JOBS 8
int kg_gen_adj_finish(const KgGenAdjJob* jobs, int32_t njobs, void* stream);

/* kg_gen_adj_prepare: aeff[k,v,w] = a[k,v,w] * imp[k,v,w] (generator.py:92-93: A[lvl] * importance; imp NULL = 1) and
 * b[k,vc,w] = sum_v u[vc,v] * aeff[k,v,w] (u NULL: b = aeff) for all blocks of a forward pass in ONE launch.       */
typedef struct KgGenPrepJob {
    const float* a;  const float* imp;  const float* u;
    float* aeff;  float* b;
    int32_t K, V, Vc;
} KgGenPrepJob;
int kg_gen_adj_prepare(const KgGenPrepJob* jobs, int32_t njobs, void* stream);

/* Backward of a generator block's tail  out = act( BN_t(u) + BN_r(r) + w_noise * noise )  (generator.py:142,160,176,179-182):
 *   kg_gen_tail_stats : with gp = g * act'(out) formed on the fly, the per-channel sums  sum gp, sum gp (u - mean_t),
 *                       sum gp (r - mean_r), sum gp * noise  ->  coef (6, C) = [a_t, b_t, c_t, a_r, b_r, c_r] with
 *                       d loss / d u = a_t gp + b_t u + c_t (BatchNorm2d backward in training mode; a branch without
 *                       BatchNorm: (1, 0, 0)), and d gamma / d beta of both layers and d w_noise ADDED into the given
 *                       (C) buffers (NULL: skipped) - e.g. their slices of the flat gradient bucket
 *   kg_gen_tail_apply : du (and dr) from g, out, u, r and coef in one pass
 * u == NULL: no BatchNorm on the tcn branch; mean_r == NULL: none on the residual branch (r may still be given: an
 * identity residual, dr = gp).  `counters`: >= C zeroed int32, left zeroed.                                          */
typedef struct KgGenTailArgs {
    int32_t N, C, T, V;  int32_t act;  float slope;
    const float* g;  int64_t g_sN, g_sC;
    const float* out;  int64_t o_sN, o_sC;
    const float* u;  int64_t u_sN, u_sC;  const float* mean_t;  const float* rstd_t;  const float* gamma_t;
    const float* r;  int64_t r_sN, r_sC;  const float* mean_r;  const float* rstd_r;  const float* gamma_r;
    const float* noise;                     /* (N, 1, T, V) contiguous or NULL                                 */
    float* coef;                            /* (6, C)                                                          */
    float* dgamma_t;  float* dbeta_t;  float* dgamma_r;  float* dbeta_r;  float* dnw;
    float* ws;  int64_t ws_bytes;  int32_t* counters;  int32_t counters_len;
    float* du;  int64_t du_sN, du_sC;
    float* dr;  int64_t dr_sN, dr_sC;       /* NULL: no residual branch                                        */
} KgGenTailArgs;
int64_t kg_gen_tail_workspace_bytes(const KgGenTailArgs* a);
int     kg_gen_tail_stats(const KgGenTailArgs* a, void* stream);
int     kg_gen_tail_apply(const KgGenTailArgs* a, void* stream);

/* ---- fused generator block (ABI v9): generator.st_gcn.forward (generator.py:168-182) and its backward as ONE launch each --
 * The launch-per-stage form above (kg_conv on the input grid -> kg_gen_expand -> kg_conv (tcn) -> kg_bn_fwd_many ->
 * kg_affine_act) is 5-7 launches per block for a few MFLOP: pure latency.  For the blocks whose per-sample working set
 * fits LDS (the generator's last four: <= 128 input channels) ONE workgroup carries ONE sample through the whole block:
 *
 *   fwd:  x    = finished input, or the PREVIOUS block's pending tail  pact(pu*s_t + b_t + pr*s_r + b_r + pnw*pnoise)
 *                (its BatchNorm coefficients became known at the previous launch's end; x is written to `xout`)
 *         yc   = [W_gcn[:Kp*C]; W_res] x                         on the input grid (Tc, Vc)          -> `yc`
 *         z, r = sum_k yc_k (U A_k) / yc_res U + b_res (or x U: identity residual), frames repeated  -> `z`, `r`
 *         u    = W_tcn (*) z + b_tcn                             3 temporal taps, zero padding       -> `u`
 *         per-sample (mean, centred sum of squares) of u and r per channel -> the last workgroup to arrive merges them in
 *         sample order (Chan et al.) into coef_t / coef_r (groups, 4, C) = [scale, shift, mean, rstd] and updates the
 *         running statistics batch by batch, exactly as kg_bn_fwd_many; a block without any BatchNorm finishes itself:
 *         out = act(u + r + nw * noise)                                                               -> `out`
 *   bwd:  du, dr from g, out, u, r and the tail coefficients `coef` (6, C) (kg_gen_tail_stats, or the previous fused
 *         backward launch);  gz = W_tcn^T (*) du;  gyc = fold(gz (U A_k)^T), fold(dr U^T);  zf = gz summed over repeated
 *         frames;  gx = [W_gcn; W_res]^T gyc (+ identity branch);  then the tail statistics of the block BEFORE this one
 *         (sums of gp = gx * pact'(x) against pu, pr, pnoise -> `pcoef` (6, Cin), parameter gradients ADDED into the given
 *         buffers), merged by the last workgroup to arrive.
 * Every tensor the deferred parameter-gradient launches read (x, yc, z, u, r / du, dr, gyc, zf) is written exactly as
 * the staged form writes it.  kg_genblock_lds_bytes: dynamic LDS the launch needs, or -1 when the block does not fit
 * (then the staged entry points apply).  `counters`: >= 1 zeroed int32, left zeroed.                                */
typedef struct KgPlane { float* p;  int64_t sN, sC; } KgPlane;             /* (N, C, T, V) plane tensor, NULL p = absent */
typedef struct KgGenBnLayer {                                             /* one training-mode BatchNorm2d of the block */
    const float* gamma;  const float* beta;  float* running_mean;  float* running_var;  int64_t* num_batches_tracked;
    float momentum, eps;
    float* coef;                                                          /* out: (groups, 4, C)                        */
} KgGenBnLayer;
typedef struct KgGenBlockArgs {
    int32_t N, groups;                      /* N samples = `groups` batches stacked along N (BatchNorm statistics per batch) */
    int32_t Cin, C, K, Kp;                  /* channels in / out; partitions of the gcn weight / of them that act (1 at V = 1) */
    int32_t Tc, Vc, T, V, rep;              /* input grid, output grid, T = Tc * rep                                     */
    int32_t res_kind;                       /* 0 none, 1 identity (Cin == C), 2 conv + BatchNorm                          */
    int32_t bn_t;                           /* BatchNorm behind the temporal conv                                        */
    int32_t act;  float slope;
    KgPlane x;                              /* finished input (N, Cin, Tc, Vc), or absent: the pending tail below       */
    KgPlane pu, pr;  const float* pcoef_t;  const float* pcoef_r;  const float* pnoise;  const float* pnw;  int32_t pact;
    KgPlane xout;
    const float* wg;  const float* wr;  const float* br;  const float* wt;  const float* bt;
    const float* b;                         /* (Kp, Vc, V) = U (A * importance), kg_gen_adj_prepare                     */
    const float* u;                         /* (Vc, V) up-sampling matrix or NULL (Vc == V)                             */
    KgPlane yc, z, r, uo;                   /* tape: (N, Kp*C [+ C], Tc, Vc), (N, C, T, V) x 3; r absent for res_kind 0  */
    KgGenBnLayer bt_, br_;                  /* used when bn_t / res_kind == 2                                           */
    const float* noise;  const float* nw;  KgPlane out;       /* self-finishing blocks (no BatchNorm at all)            */
    float* ws;  int64_t ws_bytes;  int32_t* counters;  int32_t counters_len;
} KgGenBlockArgs;
typedef struct KgGenBlockBwdArgs {
    int32_t N;                              /* samples of the differentiated batch                                      */
    int32_t Cin, C, K, Kp, Tc, Vc, T, V, rep, res_kind, bn_t, act;  float slope;
    KgPlane g, out, uo, r;                  /* d loss / d out, and the block's taped out / u / r                        */
    const float* coef;                      /* (6, C) tail coefficients of THIS block                                   */
    const float* wg;  const float* wr;  const float* wt;  const float* b;  const float* u;
    KgPlane du, dr, gyc, zf, gx;            /* out.  dr absent: no residual branch, or du == dr (no BatchNorm at all);
                                               gyc (N, Kp*C [+ C], Tc, Vc); zf (N, C, Tc, V); gx (N, Cin, Tc, Vc)       */
    /* tail statistics of the PREVIOUS block (absent px: skipped)                                                        */
    KgPlane px, pu, pr;  const float* pnoise;  int32_t pact;
    const float* pmean_t;  const float* prstd_t;  const float* pgamma_t;
    const float* pmean_r;  const float* prstd_r;  const float* pgamma_r;
    float* pcoef;  float* dgamma_t;  float* dbeta_t;  float* dgamma_r;  float* dbeta_r;  float* dnw;
    float* ws;  int64_t ws_bytes;  int32_t* counters;  int32_t counters_len;
} KgGenBlockBwdArgs;
int64_t kg_genblock_lds_bytes(const KgGenBlockArgs* a);
int64_t kg_genblock_workspace_bytes(const KgGenBlockArgs* a);
int     kg_genblock_fwd(const KgGenBlockArgs* a, void* stream);
int64_t kg_genblock_bwd_lds_bytes(const KgGenBlockBwdArgs* a);
int64_t kg_genblock_bwd_workspace_bytes(const KgGenBlockBwdArgs* a);
int     kg_genblock_bwd(const KgGenBlockBwdArgs* a, void* stream);

/* ---- per-channel reductions over (n, t, v) ---------------------------------------------------------
 *   out[0*C + c] = sum x ;  out[1*C + c] = sum x*(y - shift[c])   (y == NULL: sum (x - shift[c])^2)
 * shift (C floats, may be NULL = 0) makes the second moment a centred one: BatchNorm2d batch
 * statistics are taken in two passes (mean, then sum (x-mean)^2) to avoid the E[x^2]-mean^2
 * cancellation.  Used for conv bias gradients and BatchNorm2d (generator.py:142,160).            */
typedef struct KgRowsumArgs {
    int32_t N, C, T, V;
    const float* x;  int64_t x_sN, x_sC;
    const float* y;  int64_t y_sN, y_sC;
    const float* shift;
    int32_t want_second;            /* 0: out (1, C) = sum x; 1: out (2, C) = [sum x, sum x*(y-shift)] ((x-shift)^2 without y);
                                       2: out (1, C) = the second sum alone (NoiseInjection.weight gradient)    */
    float* out;                     /* (2, C) or (1, C)                                            */
    float* ws;  int64_t ws_bytes;
    int32_t accumulate;             /* 0: out = sums; 1: out += sums                               */
    float* out2;                    /* optional second destination of the same sums (two biases that share one
                                       gradient: tcn + residual conv of a D block), same accumulate rule        */
} KgRowsumArgs;

int64_t kg_rowsum_workspace_bytes(const KgRowsumArgs* a);
int     kg_rowsum(const KgRowsumArgs* a, void* stream);
/* several reductions (jobs[i].ws / ws_bytes are ignored; no two jobs may share a destination) in one launch + one
 * finishing launch: the bias gradients of all convs of a backward pass                                          */
int64_t kg_rowsum_many_workspace_bytes(const KgRowsumArgs* jobs, int32_t njobs);
int     kg_rowsum_many(const KgRowsumArgs* jobs, int32_t njobs, float* ws, int64_t ws_bytes, void* stream);

/* ---- pointwise epilogues ---------------------------------------------------------------------------
 * kg_act_bwd : out = g * act'(ref)  where ref is the activation OUTPUT
 *              (LeakyReLU: ref > 0 ? 1 : slope; tanh: 1 - ref^2)      discriminator.py:136, generator.py:182
 * kg_affine_act: out = act( x*sx[c] + bx[c] + r*sr[c] + br[c] + nw[c]*noise[n,t,v] )
 *              the generator block tail: BatchNorm2d normalise/affine of the tcn branch and of the
 *              residual branch, "+ res", NoiseInjection and the activation in one pass
 *              (generator.py:142,160,176,179-182).  r / noise / any scale vector may be NULL.     */
typedef struct KgEltArgs {
    int32_t N, C, T, V;
    const float* x;  int64_t x_sN, x_sC;
    const float* r;  int64_t r_sN, r_sC;
    const float* noise;                 /* (N, 1, T, V) contiguous                                 */
    const float* sx; const float* bx; const float* sr; const float* br; const float* nw;
    float* out;  int64_t o_sN, o_sC;
    int32_t act;  float slope;
    int32_t groups;                     /* kg_affine_act: > 1 = that many batches stacked along N, each with its own  */
    int64_t coef_gs;                    /* sx/bx/sr/br: batch q reads them at + q * coef_gs floats (nw is shared)      */
} KgEltArgs;

int kg_act_bwd(const KgEltArgs* a, void* stream);      /* x = g, r = ref                           */
int kg_affine_act(const KgEltArgs* a, void* stream);

/* ---- BatchNorm2d statistics + coefficients in one launch (generator.py:142,160: tcn.1 / residual.1) -------
 * One workgroup per channel, two passes over the channel (mean, then sum (x-mean)^2: no E[x^2]-mean^2
 * cancellation).  kg_bn_fwd writes coef (4, C) = [scale, shift, mean, rstd] with
 *     rstd = 1/sqrt(var_biased + eps), scale = gamma*rstd, shift = beta - mean*scale
 * so that BN(x) = x*scale + shift is applied by kg_affine_act, and in training mode updates
 * running_mean / running_var (unbiased variance, momentum) and increments num_batches_tracked exactly as
 * torch.nn.BatchNorm2d does.  training = 0: no reduction, mean/var are the running statistics.
 * kg_bn_bwd: from g (= dL/d(BN output)) and the BN input x it writes coef (5, C) = [a, b, c, dgamma, dbeta]:
 *     dgamma = sum g*(x-mean)*rstd, dbeta = sum g,
 *     training: dL/dx = a*g + b*x + c with a = gamma*rstd, b = -a*rstd*dgamma/n, c = -a*dbeta/n - b*mean
 *     eval    : dL/dx = a*g            with a = gamma*rstd (b = c = 0)
 * (applied by kg_affine_act: out = g*a + c + x*b).  Replaces ~17 / ~11 per-channel-vector launches.   */
typedef struct KgBnArgs {
    int32_t N, C, T, V;
    const float* x;  int64_t x_sN, x_sC;
    const float* g;  int64_t g_sN, g_sC;       /* bwd only                                             */
    const float* gamma; const float* beta;     /* (C) or NULL (= 1 / 0)                                */
    float* running_mean; float* running_var;   /* (C) or NULL; updated by fwd in training mode         */
    int64_t* num_batches_tracked;              /* or NULL; += 1 by fwd in training mode                */
    const float* mean; const float* rstd;      /* bwd only: the statistics kg_bn_fwd used              */
    float momentum, eps;                       /* fwd: momentum < 0 = torch's momentum=None (cumulative moving
                                                  average: factor 1 / num_batches_tracked incl. this batch, formed
                                                  on the device from the live counter)                            */
    int32_t training;
    float* coef;                               /* fwd (4, C); bwd (5, C)                               */
} KgBnArgs;

int kg_bn_fwd(const KgBnArgs* a, void* stream);
int kg_bn_bwd(const KgBnArgs* a, void* stream);

/* Training-mode statistics of up to 4 BatchNorm layers in ONE launch, each over `groups` independent batches stacked
 * along N (a.N = samples PER batch; batch q = samples [q N, (q+1) N) of a.x): a.coef is (groups, 4, C), the running
 * statistics take the batches' updates in order, num_batches_tracked += groups - the same results as `groups`
 * kg_bn_fwd calls per layer (generator.py:142,160 on the two syntheses of a WGAN-GP iteration, kinetic-gan.py:143,167).
 * `counters`: >= sum of C zeroed int32, left zeroed.                                                                 */
typedef struct KgBnJob {
    KgBnArgs a;
    int32_t groups;
} KgBnJob;

/* the backward sums of up to 4 layers (kg_bn_bwd arguments each, coef (5, C)) in one launch; chunked partial sums,
 * the last workgroup of a channel finishes (`counters` as above)                                                    */
int64_t kg_bn_bwd_many_workspace_bytes(const KgBnArgs* jobs, int32_t njobs);
int     kg_bn_bwd_many(const KgBnArgs* jobs, int32_t njobs, float* ws, int64_t ws_bytes, int32_t* counters,
                       int32_t counters_len, void* stream);
int64_t kg_bn_fwd_many_workspace_bytes(const KgBnJob* jobs, int32_t njobs);
int     kg_bn_fwd_many(const KgBnJob* jobs, int32_t njobs, float* ws, int64_t ws_bytes, int32_t* counters,
                       int32_t counters_len, void* stream);

/* ---- WGAN-GP gradient penalty (kinetic-gan.py:112-113) --------------------------------------------------------
 *   kg_gp_fwd: nrm[n] = |g_n|_2 over (c, t, v);  gp[0] = mean_n (nrm[n] - 1)^2
 *   kg_gp_bwd: out = g * (2/N) * (1 - 1/nrm[n]) * gout[0]      (0 where nrm[n] == 0; gout: upstream gradient, device)
 * Replaces gradients.view(N,-1).norm(2, dim=1), (norm - 1)**2, .mean() and their autograd backward (~25 launches). */
typedef struct KgGpArgs {
    int32_t N, C, T, V;
    const float* g;  int64_t g_sN, g_sC;
    float* nrm;                         /* (N)  written by fwd, read by bwd                            */
    float* gp;                          /* (1)  fwd                                                    */
    const float* gout;                  /* (1)  bwd                                                    */
    float* out;  int64_t o_sN, o_sC;    /* bwd: (N, C, T, V) plane tensor                              */
} KgGpArgs;

int kg_gp_fwd(const KgGpArgs* a, void* stream);
int kg_gp_bwd(const KgGpArgs* a, void* stream);

/* ---- flat-buffer Adam (kinetic-gan.py:77-78: Adam(lr, betas=(b1,b2)), eps 1e-8, no weight decay) --
 * p, g, m, v are flat fp32 buffers of n elements; *step (device memory, so that a captured
 * hipGraph replays with the live value) is the 1-based step count.
 * grad_scale multiplies g first (1/world_size after the RCCL sum).                               */
int kg_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                 float lr, float b1, float b2, float eps, const int32_t* step, float grad_scale,
                 void* stream);
/* The same step with the loop's zero_grad() folded in (ABI v9; kinetic-gan.py:137,157): with zero_grad != 0 every gradient
 * element is set to 0 once it has been consumed - the next backward pass accumulates into a clean bucket without a fill
 * launch.  (*step stays the caller's 1-based count: advancing it inside the launch needed a ticket per workgroup, ~900
 * same-address atomics = 15 us, against 2 us for the `step += 1` launch.)                                              */
int kg_adam_step_fused(float* p, float* g, float* m, float* v, int64_t n,
                       float lr, float b1, float b2, float eps, const int32_t* step, float grad_scale,
                       int32_t zero_grad, void* stream);

/* ---- container-level fusions around the discriminator's blocks ---------------------------------------
 * kg_head_fwd   : v[n] = b + sum_c w[c] * mean_{t,v} h[n,c,t,v]           global average pool + Linear(latent, 1)
 *                                                                         (discriminator.py:68-72)
 * kg_head_bwd   : g[n,c,t,v] = gv[n] * w[c] / (T V) * (masked ? lrelu'(h[n,c,t,v]) : 1)
 *                 the backward pass's top gradient for d loss / d v = gv, with the last block's LeakyReLU derivative
 *                 (expressed on its output h, discriminator.py:136) already applied when `masked`
 * kg_head_wgrad : dw[c] (+)= sum_n gv[n] * mean_{t,v} h[n,c,t,v],  db (+)= sum_n gv[n]
 *                 (h = the last block's output for the Linear's first-order gradient, or the double backward's
 *                  cotangent of the top gradient: the WGAN-GP penalty differentiates gv * w / (T V) w.r.t. w)        */
typedef struct KgHeadArgs {
    int32_t N, C, T, V;
    const float* h;  int64_t h_sN, h_sC;
    const float* w;  const float* b;        /* (C), (1) or NULL                                                */
    float* v;                               /* fwd: (N)                                                        */
    const float* gv;                        /* bwd / wgrad: (N)                                                */
    float* g;  int64_t g_sN, g_sC;          /* bwd: (N, C, T, V) plane tensor                                  */
    float slope;  int32_t masked;
    float* dw;  float* db;  int32_t accumulate;   /* wgrad: (C), (1) or NULL                                   */
} KgHeadArgs;
int kg_head_fwd(const KgHeadArgs* a, void* stream);
int kg_head_bwd(const KgHeadArgs* a, void* stream);
int kg_head_wgrad(const KgHeadArgs* a, void* stream);

/* The label channels of discriminator block 0 (discriminator.py:57-60: the class embedding, broadcast over (t, v) and
 * concatenated in FRONT of x) are constant over (t, v): through the gcn (tgcn.py:61-66) they contribute a per-sample
 * bias   zl[n,c,w] = sum_k S[k,w] * sum_j Wc(k,c,j) E[label_n, j],   S[k,w] = sum_v ak[k,v,w]
 * with Wc(k,c,j) = w + k*w_sK + c*w_sC + j the first J input columns of the gcn weight and ak (K, V, W) the block's
 * masked kept-column adjacency - the (N, J, T, V) label planes are never built.
 * kg_label_bias_bwd (first order; gz (N, C, T, W) = gradient of the gcn output): demb[l,j] (+)= ..., dw (same addressing
 * as w) (+)= ..., dak[k,v,w] (+)= dS[k,w] for every v; NULL outputs are skipped.  Both directions work per CLASS first
 * (the bias depends on a sample only through its class) and keep those records in `ws`
 * (kg_label_bias_workspace_bytes); deterministic (a class's samples are visited in index order).                     */
typedef struct KgLabelBiasArgs {
    int32_t N, L, J, K, C, V, W, T;
    const int64_t* labels;                  /* (N) class of every sample                                       */
    const float* emb;                       /* (L, J) label_emb.weight                                         */
    const float* w;  int64_t w_sK, w_sC;
    const float* ak;                        /* (K, V, W)                                                       */
    float* zl;                              /* fwd: (N, C, W) contiguous                                       */
    const float* gz;  int64_t gz_sN, gz_sC; /* bwd: (N, C, T, W) plane tensor                                  */
    float* demb;  float* dw;  int32_t accumulate;
    float* dak;  int32_t dak_accumulate;
    float* ws;  int64_t ws_bytes;
} KgLabelBiasArgs;
int     kg_label_bias_fwd(const KgLabelBiasArgs* a, void* stream);
int64_t kg_label_bias_workspace_bytes(const KgLabelBiasArgs* a);
int     kg_label_bias_bwd(const KgLabelBiasArgs* a, void* stream);

/* kg_mix3: out (3N, C, T, V) = [real | fake | alpha[n] real + (1 - alpha[n]) fake] - the three batches the critic step
 * runs D on (kinetic-gan.py:97-99,146-148) as one tensor.                                                            */
typedef struct KgMixArgs {
    int32_t N, C, T, V;
    const float* real;  int64_t r_sN, r_sC;
    const float* fake;  int64_t f_sN, f_sC;
    const float* alpha;                     /* (N)                                                             */
    float* out;  int64_t o_sN, o_sC;        /* (3N, C, T, V)                                                   */
} KgMixArgs;
int kg_mix3(const KgMixArgs* a, void* stream);

/* kg_masked_adj_fwd: ak[i] = a[s] * imp[s], s = sel ? sel[i] : i, i < n   (discriminator.py:63-64 / generator.py:92-93:
 * A[lvl] * edge_importance of ALL blocks, flattened and concatenated, restricted to the kept columns through `sel`)
 * kg_masked_adj_bwd: dimp[s] (+)= g[i] * a[s]   (sel injective: every element has one writer)                        */
typedef struct KgMaskedAdjArgs {
    int32_t n;
    const float* a;  const float* imp;  const int64_t* sel;
    float* ak;
    const float* g;  float* dimp;  int32_t accumulate;
} KgMaskedAdjArgs;
int kg_masked_adj_fwd(const KgMaskedAdjArgs* a, void* stream);
int kg_masked_adj_bwd(const KgMaskedAdjArgs* a, void* stream);

/* ---- label embedding + mapping network of the generator (generator.py:22-37 Mapping_Net, :80-85) -----------------------
 * nn.Linear(D, D) + LeakyReLU(0.2), mlp_dim times, on the (N, D) latents of a step; D = latent + n_classes.
 *   kg_linear_fwd : y[n,o] = act( sum_i xin[n,i] w[o,i] + bias[o] )
 *                   xin[n, :] = [ emb[labels[n], 0:J) | x[n, 0:Din-J) ]: the nn.Embedding lookup and torch.cat of
 *                   generator.py:80-82 folded into the FIRST layer's operand load (J = 0: xin = x); a label outside [0, L)
 *                   makes its sample NaN (nn.Embedding raises)
 *   kg_linear_bwd : with g' = g * act'(y) (the LeakyReLU derivative expressed on the layer's output y):
 *                   gx[n, i] = sum_o g'[n,o] w[o,i]  for i < gx_cols (the first layer needs the J embedding columns only,
 *                   0 = no input gradient);  dw[o,i] (+)= sum_n g'[n,o] xin[n,i];  db[o] (+)= sum_n g'[n,o]   (NULL = skip);
 *                   one launch
 *   kg_embed_bwd  : demb[l, j] (+)= sum_{n: labels[n] = l} gx[n, j], j < J  (aten::embedding_dense_backward; samples in
 *                   index order)
 * w: (Dout, Din) contiguous (nn.Linear.weight); x, y, g, gx: row-major with leading dimensions *_ld (floats).
 * Replaces per layer: aten::addmm + leaky_relu forward; leaky_relu_backward + mm (input) + mm (weight) + sum (bias) + the
 * accumulation adds backward; plus embedding, cat and embedding_dense_backward once.                                   */
typedef struct KgLinearArgs {
    int32_t N, Din, Dout;
    int32_t L, J;                           /* embedding table rows / columns folded into xin (J = 0: none)              */
    const float* x;  int64_t x_ld;          /* (N, Din - J)                                                              */
    const float* emb;                       /* (L, J) label_emb.weight or NULL                                           */
    const int64_t* labels;                  /* (N) or NULL                                                               */
    const float* w;  const float* bias;     /* (Dout, Din), (Dout) or NULL                                               */
    float* y;  int64_t y_ld;                /* fwd: output (N, Dout); bwd: the forward output (read)                     */
    int32_t act;  float slope;              /* KG_ACT_NONE or KG_ACT_LRELU                                               */
    const float* g;  int64_t g_ld;          /* bwd: (N, Dout) gradient of y                                              */
    float* gx;  int64_t gx_ld;  int32_t gx_cols;   /* bwd: (N, gx_cols) or NULL / 0; kg_embed_bwd: its input             */
    float* dw;  float* db;                  /* bwd: (Dout, Din), (Dout) or NULL                                          */
    float* demb;                            /* kg_embed_bwd: (L, J)                                                      */
    int32_t accumulate;                     /* 0: dw / db / demb = result, 1: += (flat gradient bucket)                  */
} KgLinearArgs;
int kg_linear_fwd(const KgLinearArgs* a, void* stream);
int kg_linear_bwd(const KgLinearArgs* a, void* stream);
int kg_embed_bwd(const KgLinearArgs* a, void* stream);

/* ---- data-parallel gradient exchange over RCCL / xGMI -------------------------------------------------
 * One process per GPU; every rank holds full replicas and, per optimiser step (kinetic-gan.py:155,174), the ranks' flat
 * fp32 gradient buckets are summed in place by ONE all-reduce; kg_adam_step's grad_scale = 1 / world averages them.
 * The reference has no distributed code: this is what a data-parallel launcher of its loop binds instead of
 * torch.distributed.  RCCL is dlopen'ed on first use (no link-time dependency).
 *   kg_comm_unique_id : rank 0 fills `id` (KG_COMM_ID_BYTES bytes) and ships it to the other ranks by any side channel
 *   kg_comm_init      : collective over all ranks; binds the communicator to HIP device `device`; *comm = opaque handle
 *   kg_allreduce_flat : buf[0..n) <- sum over ranks, in place, enqueued on `stream` (capturable into a hipGraph)
 *   kg_comm_destroy   : releases the communicator (NULL is a no-op)                                                    */
#define KG_COMM_ID_BYTES 128
int kg_comm_unique_id(void* id);
int kg_comm_init(void** comm, int32_t rank, int32_t world, const void* id, int32_t device);
int kg_comm_world(const void* comm);
int kg_allreduce_flat(void* comm, float* buf, int64_t n, void* stream);
int kg_comm_destroy(void* comm);

/* ---- measured peaks ("use measured peaks as denominators"; ABI v7) ----------------------------------------
 * Two probe launches bench.py times next to every roofline leg, so that the clock the chip really holds in launches of
 * that length is on the benchmark line and not in prose (the reference has no counterpart):
 *   kg_peak_mfma_f32 : every wave of a full grid (four workgroups of four waves per CU) issues `iters` x 16 independent
 *                      v_mfma_f32_32x32x2_f32 on N(0,1)-like operands; *flops (host, optional) = the flops of the launch.
 *                      `sink` (>= 64 floats, device) receives a value that keeps the loop alive.
 *   kg_peak_copy     : dst[i] = src[i], i < n, 16 bytes per lane, grid-stride (the "float4 copy" of the hardware guide);
 *                      n multiple of 4, both pointers 16-byte aligned; moves 8 n bytes.                                    */
int kg_peak_mfma_f32(float* sink, int32_t iters, double* flops, void* stream);
int kg_peak_copy(const float* src, float* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KGAN_HIP_H */
