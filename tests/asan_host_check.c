/* Host-side sanitizer driver (tests/test_abi_cpu.py::test_host_code_under_asan): calls every libkgan_hip.so entry point
 * that does NOT launch a kernel - argument validation, launch plans, workspace sizing, job tables - with valid and
 * with broken arguments, linked against the -fsanitize=address build of the library (build.py --asan).  No GPU. */
#include <stdio.h>
#include <string.h>

#include "kgan_hip.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s (%s)\n", __LINE__, #c, kg_last_error()); return 1; } } while (0)

static KgConvArgs conv_args(int N, int Cin, int M, int T, int V, int taps, int stride) {
    KgConvArgs a;
    memset(&a, 0, sizeof a);
    a.N = N; a.M = M; a.T_out = T / stride; a.V_out = V;
    a.out = (float*)0x1000; a.o_sN = (long)T / stride * V; a.o_sC = (long)N * T / stride * V;
    a.ngroups = 1;
    a.g[0].x = (const float*)0x1000; a.g[0].x_sN = (long)T * V; a.g[0].x_sC = (long)N * T * V;
    a.g[0].Cin = Cin; a.g[0].T_in = T; a.g[0].V_in = V; a.g[0].x_lead = 32;
    a.g[0].w = (const float*)0x1000; a.g[0].w_sT = 1; a.g[0].w_sO = Cin * taps; a.g[0].w_sI = taps; a.g[0].w_MB = 1 << 30;
    a.g[0].taps = taps; a.g[0].tap_mode = KG_TAP_TIME; a.g[0].t_stride = stride;
    a.slope = 0.2f;
    return a;
}

static KgWgradArgs wgrad_args(int N, int Cin, int M, int T, int V, int taps, int nextra) {
    KgWgradArgs a;
    memset(&a, 0, sizeof a);
    a.N = N; a.M = M; a.T_out = T; a.V_out = V; a.Cin = Cin; a.T_in = T; a.V_in = V;
    a.g = (const float*)0x1000; a.g_sN = (long)T * V; a.g_sC = (long)N * T * V;
    a.x = (const float*)0x1000; a.x_sN = (long)T * V; a.x_sC = (long)N * T * V;
    a.taps = taps; a.tap_mode = KG_TAP_TIME; a.t_stride = 1;
    a.dw = (float*)(0x1000 + 64L * M * Cin * taps); a.w_sT = 1; a.w_sO = Cin * taps; a.w_sI = taps;
    a.nextra = nextra;
    for (int i = 0; i < nextra; ++i) {
        a.extra[i].N = N + i + 1;
        a.extra[i].g = a.g; a.extra[i].g_sN = a.g_sN; a.extra[i].g_sC = (long)(N + i + 1) * T * V;
        a.extra[i].x = a.x; a.extra[i].x_sN = a.x_sN; a.extra[i].x_sC = (long)(N + i + 1) * T * V;
    }
    return a;
}

int main(void) {
    CHECK(kg_abi_version() == KG_ABI_VERSION);
    CHECK(strcmp(kg_arch(), "gfx950") == 0);
    kg_reload_env();
    /* conv: plans and scratch for every tile class, deep-K split, invalid arguments */
    int shapes[][7] = {{64, 63, 32, 64, 25, 1, 1}, {128, 64, 64, 64, 11, 3, 1}, {64, 512, 512, 8, 1, 3, 2}, {2, 572, 1536, 1, 1, 1, 1},
                       {192, 128, 256, 32, 5, 3, 2}, {1, 5, 3, 7, 16, 3, 1}, {64, 512, 512, 256, 25, 3, 1}};
    for (unsigned i = 0; i < sizeof shapes / sizeof shapes[0]; ++i) {
        int* s = shapes[i];
        KgConvArgs a = conv_args(s[0], s[1], s[2], s[3], s[4], s[5], s[6]);
        int32_t tile = -1, ns = -1;
        CHECK(kg_conv_plan_info(&a, &tile, &ns) == 0 && tile >= 0 && ns >= 1);
        CHECK(kg_conv_workspace_bytes(&a) >= 0);
        if (ns > 1) CHECK(kg_conv_workspace_bytes(&a) == (int64_t)ns * a.M * a.N * a.T_out * a.V_out * 4);
    }
    KgConvArgs bad = conv_args(4, 8, 8, 8, 5, 3, 1);
    bad.ngroups = 3;
    CHECK(kg_conv_workspace_bytes(&bad) < 0);
    bad = conv_args(4, 8, 8, 8, 5, 2, 1);
    CHECK(kg_conv(&bad, 0) < 0 && strstr(kg_last_error(), "taps") != 0);
    CHECK(kg_conv(0, 0) < 0);
    /* conv_many: the shared-launch decision (host only) - two full-slice problems share a launch, a ragged one does not */
    {
        KgConvArgs mj[3] = {conv_args(64, 64, 64, 64, 11, 3, 1), conv_args(64, 64, 32, 64, 11, 1, 1), conv_args(64, 40, 32, 64, 11, 1, 1)};
        int32_t mt = -2;
        CHECK(kg_conv_many_plan(mj, 2, &mt) == 0 && mt >= 0 && mt <= 2);
        CHECK(kg_conv_many_plan(mj, 3, &mt) == 0 && mt == -1);
        CHECK(kg_conv_many_plan(mj, 1, &mt) == 0 && mt == -1);
        CHECK(kg_conv_many_plan(0, 2, &mt) < 0);
        CHECK(kg_conv_many(mj, 0, 0) < 0);
    }
    /* wgrad: single layer, several layers, duplicate destinations */
    KgWgradArgs jobs[12];
    for (int i = 0; i < 12; ++i) jobs[i] = wgrad_args(64 + i, 32 << (i % 4), 64 << (i % 3), 16, 5, (i % 2) ? 3 : 1, i % 3);
    for (int i = 0; i < 12; ++i) CHECK(kg_wgrad_workspace_bytes(&jobs[i]) > 0);
    int64_t wsb = kg_wgrad_many_workspace_bytes(jobs, 12);
    CHECK(wsb > 0);
    CHECK(kg_wgrad_many(jobs, 12, 0, 0, 0) < 0);                 /* no workspace: rejected before any launch */
    jobs[5].dw = jobs[2].dw;
    CHECK(kg_wgrad_many(jobs, 12, (float*)0x1000, wsb, 0) < 0 && strstr(kg_last_error(), "same dw") != 0);
    jobs[3].taps = 2;
    CHECK(kg_wgrad_many_workspace_bytes(jobs, 12) < 0);
    KgWgradReduceJobs rj;
    memset(&rj, 0, sizeof rj);
    rj.njobs = KG_WGRAD_REDUCE_MAX_JOBS + 1;
    CHECK(kg_wgrad_reduce_many(&rj, 0) < 0);
    rj.njobs = 2;
    for (int i = 0; i < 2; ++i) { rj.job[i].ws = (const float*)0x1000; rj.job[i].dw = (float*)0x2000; rj.job[i].taps = 1; rj.job[i].M = 8; rj.job[i].Cin = 8; rj.job[i].splits = 2; }
    CHECK(kg_wgrad_reduce_many(&rj, 0) < 0 && strstr(kg_last_error(), "same dw") != 0);
    /* fused aggregation + gcn: geometry check */
    KgAggConvArgs ac;
    memset(&ac, 0, sizeof ac);
    ac.N = 64; ac.Cin = 32; ac.M = 64; ac.T = 64; ac.V = 11; ac.W = 11; ac.K = 3;
    ac.x = ac.a = ac.w = (const float*)0x1000; ac.nbr = (const int32_t*)0x1000; ac.out = (float*)0x1000;
    ac.x_sN = 704; ac.x_sC = 64 * 704; ac.w_sT = 64 * 32; ac.w_sO = 32; ac.w_sI = 1; ac.o_sN = 704; ac.o_sC = 64 * 704;
    ac.pcount[0] = 1; ac.pcount[1] = 4; ac.pcount[2] = 1;
    CHECK(kg_aggconv_supported(&ac) == 1);
    ac.V = 25; ac.W = 1;
    CHECK(kg_aggconv_supported(&ac) == 0);
    ac.pcount[0] = 3;
    CHECK(kg_aggconv(&ac, 0) < 0);
    /* aggregation / reductions: scratch sizing and rejection of broken arguments */
    KgAggArgs ag;
    memset(&ag, 0, sizeof ag);
    ag.N = 64; ag.C = 32; ag.K = 3; ag.V = 11; ag.W = 11; ag.T = 64; ag.rep = 1;
    ag.x = ag.y = (const float*)0x1000; ag.x_sN = 704; ag.x_sC = 64 * 704; ag.y_sN = 704; ag.y_sC = 64 * 704;
    CHECK(kg_agg_outer_workspace_bytes(&ag) > 0);
    ag.K = 0;
    CHECK(kg_agg_expand(&ag, 0) < 0);
    KgRowsumArgs rs;
    memset(&rs, 0, sizeof rs);
    rs.N = 64; rs.C = 32; rs.T = 64; rs.V = 11; rs.x = (const float*)0x1000; rs.x_sN = 704; rs.x_sC = 64 * 704;
    CHECK(kg_rowsum_workspace_bytes(&rs) >= 0);
    /* fused generator block (ABI v9): eligibility / LDS sizing on the host, rejection before any launch */
    KgGenBlockArgs gb;
    memset(&gb, 0, sizeof gb);
    gb.N = 128; gb.groups = 2; gb.Cin = 64; gb.C = 32; gb.K = 3; gb.Kp = 3; gb.Tc = 8; gb.Vc = 5; gb.T = 16; gb.V = 11; gb.rep = 2;
    gb.res_kind = 2; gb.bn_t = 0; gb.act = 1;
    gb.wg = gb.wr = gb.wt = (const float*)0x1000;
    CHECK(kg_genblock_lds_bytes(&gb) > 0 && kg_genblock_lds_bytes(&gb) <= 150 * 1024);
    CHECK(kg_genblock_workspace_bytes(&gb) == (int64_t)2 * 128 * 32 * 2 * 4);
    CHECK(kg_genblock_fwd(&gb, 0) < 0);                          /* null adjacency / tape tensors */
    gb.wg = (const float*)0x1004;
    CHECK(kg_genblock_lds_bytes(&gb) == -1);                     /* weight rows not 16-byte aligned: staged form */
    gb.wg = (const float*)0x1000; gb.Cin = 572; gb.C = 512; gb.Tc = gb.T = 1; gb.Vc = gb.V = 1; gb.rep = 1; gb.res_kind = 0;
    CHECK(kg_genblock_lds_bytes(&gb) == -1);                     /* generator block 0: does not fit this form */
    gb.T = 3;
    CHECK(kg_genblock_lds_bytes(&gb) == -2 && strstr(kg_last_error(), "bad dims") != 0);
    KgGenBlockBwdArgs gbb;
    memset(&gbb, 0, sizeof gbb);
    gbb.N = 64; gbb.Cin = 3; gbb.C = 3; gbb.K = 3; gbb.Kp = 3; gbb.Tc = 32; gbb.Vc = 11; gbb.T = 64; gbb.V = 25; gbb.rep = 2;
    gbb.res_kind = 1; gbb.act = 2;
    CHECK(kg_genblock_bwd_lds_bytes(&gbb) > 0);
    CHECK(kg_genblock_bwd_workspace_bytes(&gbb) == (int64_t)64 * 3 * 4 * 4);
    CHECK(kg_genblock_bwd(&gbb, 0) < 0);
    gbb.Tc = 128; gbb.T = 256;
    CHECK(kg_genblock_bwd_lds_bytes(&gbb) == -1);                /* t_size = 256: the sample does not fit LDS */
    printf("asan host check ok\n");
    return 0;
}
