#!/usr/bin/env python3
"""G+D train-step samples/sec of the st_gcn hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one G+D iteration of kinetic-gan.py:137-174 (D step + G step, the i % n_critic == 0
branch) on one synthetic batch per GPU: NTU-60 xsub shapes (N,3,64,25), mlp4, 64 samples per GPU
(BASELINE configs[1]); inputs are resident in HBM before the timed region.  One process per GPU,
gradients summed by one RCCL all-reduce of a flat bucket per optimiser step (weak scaling).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel,
measured live with HIP events on the launch stream) and `cpu_baseline` (the oracle's WGAN-GP
iteration timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    "ntu": dict(channels=3, n_classes=60, t_size=64, v=25, latent=512, mlp=4, dataset="ntu"),          # C1 / C2
    "ntu120": dict(channels=3, n_classes=120, t_size=64, v=25, latent=512, mlp=8, dataset="ntu"),      # C3 (32 / GPU)
    "h36m": dict(channels=2, n_classes=10, t_size=32, v=16, latent=512, mlp=4, dataset="h36m"),        # C4
    "stress": dict(channels=3, n_classes=60, t_size=256, v=25, latent=512, mlp=4, dataset="ntu"),      # C5b (64 / GPU)
}
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: total samples, split evenly over the GPUs (overrides --batch)")
    ap.add_argument("--config", default="ntu", choices=sorted(CONFIGS))
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the D-only / n_critic=5 figures of SURVEY 8(d)")
    ap.add_argument("--no-c5a", action="store_true", help="skip the C5a stress-block leg of the roofline")
    ap.add_argument("--segmented", action="store_true",
                    help="force the data-parallel launch structure (two graphs + eager all-reduce/Adam) on one GPU")
    ap.add_argument("--overlap", action="store_true",
                    help="data parallel: D's all-reduce + Adam on a side stream under the G forward (generator step "
                         "captured as two graphs, no paired synthesis).  Off by default (DESIGN.md 7)")
    ap.add_argument("--no-overlap", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--exact-bn", action="store_true",
                    help="data parallel only: the generator's BatchNorm statistics over the GLOBAL batch (one small all-reduce "
                         "per BatchNorm layer and direction, Generator.exact_bn); eager launches.  Default: per-rank statistics")
    ap.add_argument("--comm", default="torch", choices=["torch", "kg"],
                    help="gradient all-reduce: torch.distributed (backend nccl = RCCL) or the library's own RCCL "
                         "communicator behind the C ABI (kg_comm_init / kg_allreduce_flat)")
    ap.add_argument("--dp-graph", action="store_true",
                    help="with --comm kg: capture the whole data-parallel iteration INCLUDING both all-reduces in one "
                         "hipGraph (RCCL records its kernels into the capture); default: two compute graphs with eager "
                         "all-reduce + Adam between them")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the roofline leg (used under rocprofv3 so that kg_conv_kernel's stats are this launch's)")
    return ap.parse_args()


def metric_name(args, cfg):
    """BASELINE.json's metric string for its own configuration (NTU-60 shapes, 64 samples per GPU); any other
    --config / --batch names itself so that a line is never mistaken for the headline figure."""
    if args.config == "ntu" and args.batch == 64 and not args.global_batch:
        return "G+D train-step samples/sec, NTU (N,3,64,25) bs=64 at 1/2/4/8 MI355X"
    return "G+D train-step samples/sec, %s (N,%d,%d,%d) bs=%d per GPU (not the BASELINE headline configuration)" % (
        args.config, cfg["channels"], cfg["t_size"], cfg["v"], args.batch)


def synth_batch(cfg, n, rank, dev):
    g = torch.Generator().manual_seed(1000 * rank)
    real = (torch.rand(n, cfg["channels"], cfg["t_size"], cfg["v"], generator=g) * 2 - 1).to(dev)
    labels = torch.randint(0, cfg["n_classes"], (n,), generator=g).to(dev)
    z = torch.randn(n, cfg["latent"], generator=g).to(dev)
    alpha = torch.rand(n, 1, 1, 1, generator=g).to(dev)
    return real, labels, z, alpha


def build_models(cfg, dev):
    import kinetic_gan_amd  # noqa: F401
    from kinetic_gan_amd.discriminator import Discriminator
    from kinetic_gan_amd.generator import Generator
    torch.manual_seed(1234)    # same random init on every rank (and rank 0 broadcasts anyway)
    G = Generator(cfg["latent"], cfg["channels"], cfg["n_classes"], cfg["t_size"], cfg["mlp"], dataset=cfg["dataset"])
    D = Discriminator(cfg["channels"], cfg["n_classes"], cfg["t_size"], cfg["latent"], dataset=cfg["dataset"])
    return G.to(dev), D.to(dev)


def _capture(fn, before_capture=None):
    """Capture fn() into a hipGraph (after an allocator warm-up on a side stream) and return its replay.
    ``before_capture`` runs once right before the captured call (not before the warm-up calls)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    if before_capture is not None:
        before_capture()
    # thread_local: other threads of the process (the RCCL watchdog of torch.distributed polls events) must not be
    # able to invalidate the capture
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        fn()
    torch.cuda.synchronize()
    return graph.replay


def _capture_pair(first, second):
    """Capture v = first(); second(v) as two hipGraphs sharing one memory pool; returns their replays."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            second(first())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga, capture_error_mode="thread_local"):
        v = first()
    with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
        second(v)
    del v
    torch.cuda.synchronize()
    return ga.replay, gb.replay


def make_step(tr, batch, use_graph, segmented):
    """Returns a zero-argument callable running one G+D iteration (noise drawn in-step like generator.py:179).

    hipGraph modes: one graph for the whole iteration (1 GPU), or - with data parallelism - one graph per compute
    half (forward + backward + gradient gather) with the RCCL all-reduce + Adam launched eagerly in between, so
    that no collective is ever part of a captured graph."""
    real, labels, z, alpha = batch

    def eager():
        tr.iteration(real, labels, z, alpha, None, None, with_g=True)

    if not use_graph:
        return eager, "eager"
    try:
        if not segmented:
            return _capture(eager), "hipgraph"
        # the warm-up replays inside _capture run the compute halves without their apply halves: harmless for
        # timing (gradients are recomputed from scratch every time), parameters only move in the real steps
        def d_half():           # the generator half that follows reuses this half's mapping-network result:
            if tr.overlap:
                # overlap mode hides D's all-reduce + Adam under the generator step's OWN forward pass, so that pass
                # must stay in the generator step: no paired synthesis here (round-2 ADVICE: with the pairing on, the
                # sample was synthesised here AND again in g_forward, with a third running-statistics update)
                tr.d_compute(real, labels, z, alpha, None)
                return
            with tr.sharing_mapping():      # tr._w of the CAPTURED call (graph memory, rewritten by every replay)
                tr.d_compute(real, labels, z, alpha, None)
        def drop_warmup_graph():
            # the last warm-up call's generator sample still holds its autograd graph: while it lives, G's parameters
            # keep the AccumulateGrad nodes created on the warm-up stream and the captured backward would sync with it
            tr._w = tr._fake_g = None
        d_replay = _capture(d_half, before_capture=drop_warmup_graph)
        # the generator step's sample (or, without pairing, the mapping result) lives in the critic graph's memory;
        # its autograd graph can be consumed ONCE: by the captured call, not by the warm-up calls
        w_cap, f_cap = tr._w, tr._fake_g
        tr._w = tr._fake_g = None

        def hand_over():
            tr._w, tr._fake_g = w_cap, f_cap
        if not tr.overlap:
            g_replay = _capture(lambda: tr.g_compute(labels, z, None), before_capture=hand_over)

            def step():
                d_replay()
                tr.d_apply()
                g_replay()
                tr.g_apply()
            return step, "hipgraph-segmented"
        # overlap: the generator step is captured as TWO graphs in one Python pass (the autograd graph built while the
        # first is captured is consumed while the second is; they share a memory pool and replay in capture order):
        # G forward - which runs while D's all-reduce + Adam are still in flight on the side stream - and the rest
        ga_replay, gb_replay = _capture_pair(lambda: tr.g_forward(labels, z, None), lambda fake: tr.g_backward(fake, labels))

        def step():
            d_replay()
            tr.d_apply_async()
            ga_replay()
            tr.wait_d_apply()
            gb_replay()
            tr.g_apply()
        return step, "hipgraph-segmented-overlap"
    except Exception as e:   # capture is an optimisation of the launch path, not of the arithmetic
        sys.stderr.write(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager\n")
        torch.cuda.synchronize()
        return eager, "eager"


_PEAKS = {}
_PEAKS_ERROR = []


def peaks_or_error(dev):
    """measured_peaks for the record: a failing probe (e.g. no room for the 256 MiB copy) is reported, and tried once only."""
    try:
        return measured_peaks(dev)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def measured_peaks(dev):
    """"use measured peaks as denominators".  Times the library's two probe launches with HIP events on the
    launch stream, replayed from a hipGraph like the legs themselves: the fp32 matrix-core loop at three lengths (the clock
    the chip holds depends on how long a launch is: DVFS has not ramped inside a 20-us launch and throttles in a 1-ms one)
    and a float4 copy of 256 MiB.  Returns TFLOP/s per launch length and GB/s."""
    if _PEAKS:
        return _PEAKS
    if _PEAKS_ERROR:            # the probes failed before: every later leg gets the same error instead of re-running them
        raise _PEAKS_ERROR[0]
    try:
        return _measure_peaks(dev)
    except Exception as e:
        _PEAKS_ERROR.append(e)
        raise


def _measure_peaks(dev):
    from kinetic_gan_amd import _native as nv
    sink = torch.zeros(64, device=dev)

    def time_graph(fn, reps):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (3 * reps) * 1e-3        # seconds per launch

    mf = {}
    for name, iters in (("20us", 10), ("50us", 25), ("1ms", 500)):
        fl = [0.0]

        def fn():
            fl[0] = nv.peak_mfma_f32(sink, iters)
        sec = time_graph(fn, 10 if iters < 100 else 2)
        mf[name] = {"tflops": round(fl[0] / sec / 1e12, 1), "launch_us": round(sec * 1e6, 1)}
    n = 32 * 1024 * 1024                        # 128 MiB read + 128 MiB written
    src = torch.empty(n, device=dev).normal_()
    dst = torch.empty(n, device=dev)
    nbytes = [0]

    def cp():
        nbytes[0] = nv.peak_copy(src, dst)
    sec = time_graph(cp, 5)
    _PEAKS.update({"mfma_f32": mf, "hbm_copy_gbs": round(nbytes[0] / sec / 1e9, 1), "hbm_copy_launch_us": round(sec * 1e6, 1),
                   "how": "kg_peak_mfma_f32 (1024 workgroups x 4 waves, independent v_mfma_f32_32x32x2_f32 on random operands) and "
                          "kg_peak_copy (float4 copy, 256 MiB moved), HIP events around hipGraph replays"})
    del src, dst
    return _PEAKS


def with_measured_peak(leg, dev):
    """peak_measured / frac_measured next to the spec peak of a roofline leg: the probe launch whose length is closest to
    the leg's own launch (matrix-core legs) or the measured copy rate (HBM legs)."""
    try:
        pk = measured_peaks(dev)
    except Exception as e:      # (a probe must never take the benchmark line down)
        leg["peak_measured"] = None
        leg["peak_measured_error"] = "%s: %s" % (type(e).__name__, e)
        return leg
    if leg.get("bound") == "mfma":
        us = leg.get("avg_launch_us") or (leg["avg_launch_ms"] * 1e3 if leg.get("avg_launch_ms") else 50.0)
        key = min(pk["mfma_f32"], key=lambda k: abs(pk["mfma_f32"][k]["launch_us"] - us))
        leg["peak_measured"] = pk["mfma_f32"][key]["tflops"]
        leg["peak_measured_probe"] = "fp32 MFMA loop, %s launch (%.0f us)" % (key, pk["mfma_f32"][key]["launch_us"])
    else:
        leg["peak_measured"] = pk["hbm_copy_gbs"]
        leg["peak_measured_probe"] = "float4 copy of 256 MiB (%.0f us)" % pk["hbm_copy_launch_us"]
    if leg.get("achieved") and leg["peak_measured"]:
        leg["frac_measured"] = round(leg["achieved"] / leg["peak_measured"], 4)
    return leg


def roofline_leg(batch_n, dev):
    """Dominant kernel: kg_conv (tap GEMM on the fp32 matrix cores; 40% of the iteration's GPU time in
    profiles/).  Timed at the tail of discriminator block 1 at NTU bs=64:
    out = lrelu(tcn3(z) + conv1x1(x) + b), C 32->64, T=64, V=11, stride 1, no vertex drop - so the ALGORITHMIC
    flops of SURVEY.md 8d (reference's dense formulation: 2*T*V*(3*Cout^2 + Cin*Cout) per sample) equal the
    executed flops of this launch."""
    from kinetic_gan_amd import _native as nv
    from kinetic_gan_amd._native import TAP_TIME, Group, WView
    n, cin, cout, T, V = batch_n, 32, 64, 64, 11
    z = nv.new_plane(n, cout, T, V, dev).normal_()
    x = nv.new_plane(n, cin, T, V, dev).normal_()
    wt = torch.randn(cout, cout, 3, 1, device=dev) * 0.05
    wr = torch.randn(cout, cin, 1, 1, device=dev) * 0.1
    bt, br = torch.randn(cout, device=dev), torch.randn(cout, device=dev)
    groups = [Group(z, wt, WView(1, cout * 3, 3), cout, 3, TAP_TIME, 1, False, None),
              Group(x, wr, WView(0, cin, 1), cin, 1, TAP_TIME, 1, False, None)]

    def launch(wpack=None):
        return nv.conv(groups, n, cout, T, V, bias0=bt, bias1=br, act=nv.ACT_LRELU, wpack=wpack)

    def timed(fn, reps=20):
        # HIP events on the launch stream (torch's current stream is the stream kg_conv is enqueued on); the
        # launches are replayed from a hipGraph so that host launch overhead is not part of the kernel time
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(reps):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (5 * reps)

    algo = 2.0 * T * V * (3 * cout * cout + cin * cout) * n
    nv.last_conv_plan = []               # which kernel the launcher's plan picks for the call as the step issues it
    try:
        launch()
        code = nv.last_conv_plan[0]
    finally:
        nv.last_conv_plan = None
    ms_call = timed(launch)
    extra = {}
    if code >= 40:
        # Only with KG_CONV_BS=1 / 2 (round 6: the plan no longer takes the bf16-split form by itself): a 3-us weight-pack
        # launch + the tile kernel (fp32 operands and results, six bf16 products per fp32 product, fp32 accumulation).  The
        # leg is the TILE kernel - timed alone on weights packed once (kg_conv_pack + KgConvArgs.wpack) - with the whole
        # call and the direct fp32 kernel on the same operands next to it.
        pack = nv.conv_pack(groups, n, cout, T, V)
        ms = timed(lambda: launch(pack))
        kname = "kg_conv_bsw_kernel<%s> (bf16-split tile kernel)" % {40: "64x128", 41: "32x128", 42: "128x64"}[code]
        extra["call_us_pack_plus_tiles"] = round(ms_call * 1e3, 2)
        extra["call_frac"] = round(algo / (ms_call * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
        prev_bs = os.environ.get("KG_CONV_BS")
        os.environ["KG_CONV_BS"] = "0"; nv.reload_env()
        try:
            ms_d = timed(launch)
        finally:
            if prev_bs is None:
                os.environ.pop("KG_CONV_BS", None)
            else:
                os.environ["KG_CONV_BS"] = prev_bs
            nv.reload_env()
        extra["direct_fp32_kernel_us"] = round(ms_d * 1e3, 2)
        extra["direct_fp32_kernel_frac"] = round(algo / (ms_d * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
        extra["note"] = ("fp32-accurate, not bit-identical to the fp32 MFMA chain: three bf16 terms per operand element, six of "
                         "the nine partial products, fp32 accumulation")
        # the pipe this kernel really runs on: six v_mfma_f32_32x32x16_bf16 per fp32 product against the 2.5 PFLOP/s dense bf16 peak
        extra["pipe"] = "bf16x3"
        extra["frac_of_bf16_peak"] = round(6 * algo / (ms_call * 1e-3) / 1e12 / 2500.0, 4)
    else:
        ms = ms_call
        kname = "kg_conv_kernel<%d,4>" % {0: 128, 1: 64, 2: 32, 3: 64, 4: 32, 9: 32}.get(code, 0)
        pack = nv.conv_pack(groups, n, cout, T, V)
        if pack is not None:        # what the bf16-split tile kernel would do on this launch (not what the plan runs: with its
            ms_b = timed(lambda: launch(pack))       # weight-pack launch it is level with the direct kernel at this size)
            extra["bf16_split_tile_kernel_us"] = round(ms_b * 1e3, 2)
            extra["bf16_split_tile_kernel_frac"] = round(algo / (ms_b * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)
    ach = algo / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "kernel": "%s (disc block 1 tail, 32->64 ch, 3 taps + 1x1 residual, bs=%d)" % (kname, n),
           "achieved": round(ach, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
           "flops_per_launch": algo, "avg_launch_us": round(ms * 1e3, 2)}
    out.update(extra)
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (tools/roofline_pmc.sh) of
    # `bench.py --roofline-only`; the committed summary is quoted here, it cannot be collected in-process
    pmc = os.path.join(ROOT, "profiles", "roofline_pmc.json" if n == 64 else "roofline_pmc_bs%d.json" % n)
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            out["traffic"] = rec["hbm_bytes_per_launch"]
            out["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, commit %s)" % (
                os.path.basename(pmc), rec.get("commit", "?"))
        except (OSError, ValueError, KeyError):
            pass
    return out


def canonical_flops(G, D, cfg):
    """ALGORITHMIC forward flops per sample in the reference's dense formulation: per block
    2*T*V*(Cin*K*Cout + K*Cout*V + 3*Cout^2 [+ Cin*Cout with a conv residual]) at the block's internal resolution,
    plus the mapping net; one G+D iteration = 12 d + 4 g."""
    K = 3
    T, V = cfg["t_size"], cfg["v"]
    nn_ = D.graph.num_node
    d = 0.0
    t, v = T, nn_[0]
    for blk in D.st_gcn_networks:
        ci, co = blk.in_channels, blk.out_channels
        d += 2.0 * t * v * (ci * K * co + K * co * v + 3 * co * co + (ci * co if blk.res_kind == "conv" else 0))
        t = blk.dw_t
        v = nn_[blk.lvl + 1] if blk.dw_s else v
    g = 0.0
    for blk in G.st_gcn_networks:
        ci, co = blk.in_channels, blk.out_channels
        t, v = blk.up_t, nn_[blk.lvl]
        g += 2.0 * t * v * (ci * K * co + K * co * v + 3 * co * co + (ci * co if blk.res_kind == "conv" else 0))
    lat = cfg["latent"] + cfg["n_classes"]
    g += 2.0 * cfg["mlp"] * lat * lat
    return d, g


def work_leg(tr, batch, args, cfg, ms_per_step):
    """What one iteration computes: the canonical figure of SURVEY 8d and the flops the launches really execute
    (the build skips dropped vertices / frames, the label channels of block 0, backward passes nobody consumes),
    counted per kernel family by a hook in _native during one eager iteration."""
    from kinetic_gan_amd import _native as nv
    real, labels, z, alpha = batch
    d, g = canonical_flops(tr.G, tr.D, cfg)
    algo = (12 * d + 4 * g) * args.batch
    nv.flop_count = {}
    exact = getattr(tr.G, "exact_bn", False)
    try:
        # the two compute halves only (rank 0 runs this alone: no collective, no optimiser step - the exact-BatchNorm
        # mode's statistics all-reduces are switched off for it: the flop count does not depend on them)
        tr.G.exact_bn = False
        with tr.sharing_mapping():
            tr.d_compute(real, labels, z, alpha, None)
        tr.g_compute(labels, z, None)
        torch.cuda.synchronize()
        ex = dict(nv.flop_count)
    finally:
        nv.flop_count = None
        tr.G.exact_bn = exact
    tot = sum(ex.values())
    out = {"algorithmic_gflop_per_step": round(algo / 1e9, 2), "executed_gflop_per_step": round(tot / 1e9, 2),
           "executed_by_family_gflop": {k: round(v / 1e9, 2) for k, v in sorted(ex.items())},
           "achieved_tflops_algorithmic": round(algo / (ms_per_step * 1e-3) / 1e12, 2),
           "achieved_tflops_executed": round(tot / (ms_per_step * 1e-3) / 1e12, 2),
           "d_mflop_per_sample": round(d / 1e6, 1), "g_mflop_per_sample": round(g / 1e6, 1)}
    # per-family rate = executed flops / kernel time of the family in the committed rocprofv3 summary of this code
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_eager_kernel_stats.json")))
    prof = cands[-1] if cands else ""
    if prof and args.config == "ntu" and args.batch == 64:
        try:
            rec = json.load(open(prof))
            fam = {}
            for k, us in rec["kernel_us_per_step"].items():
                if k in ex and us > 0:
                    fam[k] = {"us_per_step": us, "tflops": round(ex[k] / (us * 1e-6) / 1e12, 1)}
            out["family_rates"] = fam
            out["family_rates_source"] = "profiles/%s (rocprofv3 --kernel-trace --stats, commit %s)" % (os.path.basename(prof), rec.get("commit", "?"))
        except (OSError, ValueError, KeyError):
            pass
    return out


# the 16 weight-gradient products of the critic step's backward pass through D at NTU shapes, in the order the pass emits
# them (block 5 first): (M, Cin, taps, T_out, V, stride).  Each takes two operand pairs: real + fake (2 x batch) and the
# gradient penalty's double backward (batch).
D_WGRAD_LAYERS = [(512, 512, 1, 8, 1, 1), (512, 512, 3, 4, 1, 2), (512, 256, 3, 16, 1, 1), (512, 512, 3, 8, 1, 2),
                  (512, 256, 1, 8, 1, 2), (256, 128, 3, 32, 5, 1), (256, 256, 3, 16, 5, 2), (256, 128, 1, 16, 5, 2),
                  (128, 64, 3, 64, 5, 1), (128, 128, 3, 32, 5, 2), (128, 64, 1, 32, 5, 2), (64, 32, 3, 64, 11, 1),
                  (64, 64, 3, 64, 11, 1), (64, 32, 1, 64, 11, 1), (32, 3, 3, 64, 11, 1), (32, 32, 3, 64, 11, 1)]


def wgrad_leg(dev, batch_n=64):
    """Second MFMA-bound family: kg_wgrad (weight half of aten::convolution_backward).  Timed as it runs in the step: ONE
    kg_wgrad_many call over the 16 temporal / residual conv weights of D with the critic step's two operand pairs each
    (one tile launch + one slab reduction since round 4).  Algorithmic flops = 2 * taps * M * Cin * columns per layer
    (executed: more - the 3-channel input layer multiplies a 32-wide tile).  `layer`: the round-3 form of this leg, the
    tcn weight of block 3 alone (a 720-workgroup launch: 0.7 of one round of resident workgroups)."""
    from kinetic_gan_amd import _native as nv
    from kinetic_gan_amd._native import TAP_TIME, WView

    def job(M, Cin, taps, t_out, V, s):
        prs = [(nv.new_plane(n, M, t_out, V, dev).normal_(), nv.new_plane(n, Cin, t_out * s, V, dev).normal_())
               for n in (2 * batch_n, batch_n)]
        return dict(g=prs[0][0], x=prs[0][1], Cin=Cin, taps=taps, tap_mode=TAP_TIME, t_stride=s, vmap=None,
                    wv=WView(1, Cin * taps, taps), out=torch.zeros(M * Cin * taps, device=dev), accumulate=True, extra=[prs[1]])

    jobs = [job(*l) for l in D_WGRAD_LAYERS]
    ms = _time_launch(lambda: nv.wgrad_many(jobs), reps=10)
    algo = sum(2.0 * taps * M * Cin * (3 * batch_n * t_out * V) for M, Cin, taps, t_out, V, s in D_WGRAD_LAYERS)
    ach = algo / (ms * 1e-3) / 1e12
    one = [job(256, 256, 3, 16, 5, 2)]
    ms1 = _time_launch(lambda: nv.wgrad_many(one), reps=20)
    algo1 = 2.0 * 3 * 256 * 256 * (3 * batch_n * 16 * 5)
    return {"bound": "mfma", "kernel": "kg_wgrad_many_kernel + reduce (the 16 conv weights of D, critic step: %d samples in two operand pairs)" % (3 * batch_n),
            "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
            "flops_per_launch": algo, "avg_launch_us": round(ms * 1e3, 2),
            "layer": {"kernel": "disc block 3 tcn weight alone (256x256x3)", "achieved": round(algo1 / (ms1 * 1e-3) / 1e12, 2),
                      "frac": round(algo1 / (ms1 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "flops_per_launch": algo1,
                      "avg_launch_us": round(ms1 * 1e3, 2)}}


def _time_launch(launch, reps=20):
    """ms per call, HIP events on the launch stream, the calls replayed from a hipGraph (no host launch overhead)"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            launch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            launch()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps)


def extras_leg(tr, batch, args, gd_ms):
    """SURVEY.md 8(d) side figures, measured after (outside) the timed region: the D-only iteration
    (kinetic-gan.py:137-155, the 4 of 5 iterations without a generator step) and the n_critic=5 amortised rate."""
    real, labels, z, alpha = batch

    def d_only():
        tr.iteration(real, labels, z, alpha, None, None, with_g=False)

    step = d_only
    if not args.no_graph:
        try:
            step = _capture(d_only)
        except Exception:
            torch.cuda.synchronize()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    d_ms = (time.perf_counter() - t0) / reps * 1e3
    return {"d_only_ms_per_step": round(d_ms, 3),
            "n_critic5_samples_per_s": round(args.batch * 5 / ((4 * d_ms + gd_ms) * 1e-3), 1),
            "note": "n_critic=5 schedule of kinetic-gan.py: 4 D-only iterations + 1 G+D iteration"}


def stress_leg(dev):
    """SURVEY.md 8(d) C5a, the kernel-roofline configuration: the tail launch of ONE D-style block with 512
    channels, V=25 (NTU level 0), T=256, identity residual, 64 samples - out = lrelu(tcn3(z) + x + b).  Large
    enough for kg_conv to run in steady state (many workgroups per CU), unlike the 20-us launches of the bs=64
    training step.  Algorithmic flops = executed flops = 2*T*V*3*C^2 per sample."""
    from kinetic_gan_amd import _native as nv
    from kinetic_gan_amd._native import TAP_TIME, Group, WView
    n, c, T, V = 64, 512, 256, 25
    z = nv.new_plane(n, c, T, V, dev).normal_()
    x = nv.new_plane(n, c, T, V, dev).normal_()
    wt = torch.randn(c, c, 3, 1, device=dev) * 0.02
    bt = torch.randn(c, device=dev)
    groups = [Group(z, wt, WView(1, c * 3, 3), c, 3, TAP_TIME, 1, False, None)]

    def launch():
        return nv.conv(groups, n, c, T, V, bias0=bt, add=x, act=nv.ACT_LRELU)

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    algo = 2.0 * T * V * 3 * c * c * n
    ach = algo / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "kg_conv_kernel<128,4> (C5a block tail: 512->512 ch, 3 taps + identity residual, "
                                       "T=256, V=25, 64 samples)",
            "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "flops_per_launch": algo, "avg_launch_ms": round(ms, 3)}


def agg_leg(dev):
    """SURVEY.md 8(d) C5a, the HBM-bound half: the standalone spatial aggregation
    einsum('nkctv,kvw->nctw') on (64, 3*512, 256, 25) -> (64, 512, 256, 25) (tgcn.py:66) as one kg_agg_reduce launch.
    ALGORITHMIC bytes = 4*(K+1)*C*T*V per sample (K input planes read, one output plane written) = 3.36 GB."""
    from kinetic_gan_amd import _native as nv
    n, c, T, V, K = 64, 512, 256, 25, 3
    y = nv.new_plane(n, K * c, T, V, dev).normal_()
    A = torch.rand(K, V, V, device=dev)

    def launch():
        return nv.agg_reduce(y, A, 1)

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    algo = 4.0 * (K + 1) * c * T * V * n
    gbs = algo / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "kg_agg_mfma_kernel<3,1> (C5a aggregation: 3x512 -> 512 planes, T=256, V=25, 64 samples)",
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "bytes_per_launch": algo, "avg_launch_ms": round(ms, 3)}


def agg_train_leg(batch_n, dev):
    """The standalone aggregation launches that are left in the bs=64 training step, at their shapes: the adjoint
    aggregation of the discriminator's backward pass, gx = sum_k gxa_k A_k^T, block 1 (32 channels, 11 kept -> 25
    vertices, T = 64) and block 3 (128 channels, 5 -> 5 vertices, T = 32), over the critic's 3 x batch samples; both
    replayed from one hipGraph.  ALGORITHMIC bytes per launch = 4 * (K * W + V) * C * T per sample (K * C planes at the
    kept vertices read, C planes written; SURVEY.md 8d's 4 (K + 1) C T V with the vertex counts the launch really has)."""
    from kinetic_gan_amd import _native as nv
    n, K = 3 * batch_n, 3
    shapes = [(32, 64, 25, 11), (128, 32, 5, 5)]        # C, T, V (full), W (kept)
    ys = [nv.new_plane(n, K * c, T, W, dev).normal_() for c, T, V, W in shapes]
    As = [torch.rand(K, V, W, device=dev) for c, T, V, W in shapes]

    def launch():
        for y, A in zip(ys, As):
            nv.agg_reduce(y, A.transpose(1, 2), 1)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            launch()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    reps = 20
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            launch()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (5 * reps)                # both launches
    algo = sum(4.0 * (K * W + V) * c * T * n for c, T, V, W in shapes)
    gbs = algo / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "kg_agg_reduce (adjoint aggregation of disc blocks 1 and 3 in the critic's backward pass, %d samples, two launches)" % n,
            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "bytes_per_launch_pair": algo, "avg_pair_us": round(ms * 1e3, 2)}


def cpu_baseline_leg(cfg, n=16, timed=15):
    """The oracle (CPU restatement of the reference's modules + WGAN-GP step, pinned to the reference by
    tests/test_oracle_golden.py) timed on the host cores: bs=16 (BASELINE configs[0]) and bs=64 (configs[1]),
    2 warm-up + `timed` G+D iterations with torch.optim.Adam, median."""
    from oracle import modules_ref as M
    from oracle.host import cpu_model, usable_cores
    cores = usable_cores()
    torch.set_num_threads(cores)
    G = M.Generator(cfg["latent"], cfg["channels"], cfg["n_classes"], cfg["t_size"], cfg["mlp"], dataset=cfg["dataset"])
    D = M.Discriminator(cfg["channels"], cfg["n_classes"], cfg["t_size"], cfg["latent"], dataset=cfg["dataset"])
    oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    oD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    real, labels, z, alpha = synth_batch(cfg, n, 0, "cpu")

    def it():
        oD.zero_grad()
        M.d_step_losses(G, D, real, labels, z, alpha)["d_loss"].backward()
        oD.step()
        oG.zero_grad()
        M.g_step_loss(G, D, labels, z)["g_loss"].backward()
        oG.step()

    it()
    it()
    times = []
    for _ in range(timed):
        t0 = time.perf_counter()
        it()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(n / med, 2), "unit": "samples/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "oracle G+D iteration, %s shapes, bs=%d, 2 warm-up + %d timed (median %.0f ms)" % (cfg["dataset"], n, timed, med * 1e3)}


def main():
    args = parse()
    cfg = CONFIGS[args.config]
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if "WORLD_SIZE" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        # `python bench.py --gpus N` without a launcher: start one rank per GPU as CHILD processes (nothing in this
        # process has touched the GPU yet; never exec over an initialised process) and pass their exit code on
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of {world} GPUs")
        args.batch = args.global_batch // world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the st_gcn path)")
    # test hooks: KG_BENCH_DEVICE pins every rank to one device and KG_BENCH_BACKEND=gloo replaces RCCL, so that the
    # multi-rank control flow (broadcast, segmented graphs, eager all-reduce + Adam) can be exercised on a 1-GPU box
    if os.environ.get("KG_BENCH_DEVICE") is not None:
        local = int(os.environ["KG_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("KG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from kinetic_gan_amd.wgan_gp import Trainer

    if args.roofline_only:
        # (no roofline_critic leg here: the rocprofv3 / PMC passes of this mode average over the launches of ONE size)
        rec = {"roofline": roofline_leg(args.batch, dev), "roofline_wgrad": wgrad_leg(dev)}
        if not args.no_c5a:
            rec["roofline_c5a"] = stress_leg(dev)
            rec["roofline_agg"] = agg_leg(dev)
        for k in list(rec):
            with_measured_peak(rec[k], dev)
        rec["peaks_measured"] = peaks_or_error(dev)
        print(json.dumps(rec), flush=True)
        return
    G, D = build_models(cfg, dev)
    comm = None
    if args.comm == "kg":
        from kinetic_gan_amd import _native as nv
        comm = nv.Comm(rank, world, local, exchange=nv.torch_dist_exchange(0) if world > 1 else None)
    if args.exact_bn and world > 1:
        G.exact_bn = True               # collectives inside the forward / backward pass: not captured into a hipGraph
        args.no_graph = True
    tr = Trainer(G, D, world_size=world, overlap=bool(args.overlap) and not args.no_overlap, comm=comm)
    batch = synth_batch(cfg, args.batch, rank, dev)
    step, mode = make_step(tr, batch, use_graph=not args.no_graph,
                           segmented=((world > 1 and not (args.dp_graph and comm is not None)) or args.segmented))
    if world > 1 and mode == "hipgraph":
        mode = "hipgraph (all-reduce captured)"

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    out = None
    if rank == 0:
        gb = args.batch * world
        out = {
            "metric": metric_name(args, cfg),
            "value": round(gb * args.steps / elapsed, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s shapes (N,%d,%d,%d), %d classes, mlp%d, G+D WGAN-GP iteration, %d samples/GPU"
                                   % (args.config, cfg["channels"], cfg["t_size"], cfg["v"], cfg["n_classes"], cfg["mlp"], args.batch),
                       "global_batch": gb, "parallelism": "dp%d" % world, "launch": mode,
                       "batchnorm": ("global-batch statistics (exact mode)" if (args.exact_bn and world > 1) else "per-rank statistics") if world > 1 else None,
                       "allreduce": ("kg_allreduce_flat (RCCL, C ABI)" if comm is not None else
                                     ("torch.distributed nccl (RCCL)" if dist.get_backend() == "nccl" else
                                      "torch.distributed %s (test hook KG_BENCH_BACKEND: NOT RCCL)" % dist.get_backend())) if world > 1 else None},
        }
        out["work"] = work_leg(tr, batch, args, cfg, out["ms_per_step"])
        if world == 1 and not args.no_extras:
            out["extras"] = extras_leg(tr, batch, args, out["ms_per_step"])
        if world == 1 and not args.no_roofline:
            out["roofline"] = roofline_leg(args.batch, dev)
            # the same layer as the critic's forward / merged backward really launch it: real + fake + interpolates
            out["roofline_critic"] = roofline_leg(3 * args.batch, dev)
            out["roofline_critic"]["where"] = "the launch of the same layer inside the critic step: 3 x batch samples"
            out["roofline_wgrad"] = wgrad_leg(dev)
            if not args.no_c5a:
                out["roofline_c5a"] = stress_leg(dev)
                out["roofline_agg"] = agg_leg(dev)
            out["roofline_agg_train"] = agg_train_leg(args.batch, dev)
            # the MEASURED peak next to the spec one on every leg (fp32 MFMA loop of the leg's launch length /
            # float4 copy), and the probes' own figures
            for k in [k for k in out if k.startswith("roofline")]:
                with_measured_peak(out[k], dev)
            out["peaks_measured"] = peaks_or_error(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_leg(cfg, 16, 15)
            out["cpu_baseline_bs64"] = cpu_baseline_leg(cfg, 64, 5)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
