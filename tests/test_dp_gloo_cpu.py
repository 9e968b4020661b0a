"""Data-parallel path on CPU: 2 ranks over gloo, kernels emulated (oracle/prim_ref.py).  Checks that
(i) rank 0's weights are broadcast, (ii) after one G+D iteration every rank holds bit-identical
parameters, (iii) they equal the hand-computed result: per-shard gradients averaged, one Adam step."""
import os
import socket
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_iteration_matches_manual_average(tmp_path):
    port = _free_port()
    world = 2
    env = dict(os.environ, OMP_NUM_THREADS="2", PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world),
                               str(port), str(tmp_path)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert res["same"] and res["same_grads"], "ranks diverged"
        # the all-reduced gradient buckets, as the optimiser read them, against the hand-made sum over the shards: D's
        # is the same arithmetic up to the summation order; G's went through a critic that has made one Adam step (a
        # weight with a near-zero gradient may sit +-lr apart, which a LeakyReLU kink can amplify)
        assert res["grad_d_l2"] <= 1e-5, res
        assert res["grad_g_l2"] <= 2e-3, res
        assert res["mean"] <= 2e-6, res
        # (round-4 ADVICE) and a bounded maximum, so that a localised error in a few parameters cannot hide behind the mean:
        # Adam moves a parameter by at most lr = 2e-4 per step; two updates that took different signs sit 2 lr apart
        assert res["max"] <= 4 * 2e-4 + 1e-7, res


def test_exact_batchnorm_mode_equals_single_process_global_batch(tmp_path):
    """Generator.exact_bn (optional exact mode): with the BatchNorm statistics all-reduced over the ranks
    (ops.SyncBatchNorm2dFn: 2 C floats per layer and direction) every rank's synthesis equals its rows of ONE process
    running the global batch, the rank-averaged parameter gradients equal the global-batch gradients, and the running
    statistics after two forward passes equal the global run's; the default mode (per-rank statistics) does not."""
    port = _free_port()
    world = 2
    env = dict(os.environ, OMP_NUM_THREADS="2", PYTHONDONTWRITEBYTECODE="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_bn_worker.py"), str(r), str(world),
                               str(port), str(tmp_path)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"bn_rank{r}.pt"))
        assert res["out1"] < 1e-5 and res["out"] < 1e-5, res
        assert res["grads"] < 1e-4, res
        assert res["bufs"] < 1e-5 and res["nbufs"] == 24 and all(v == 2 for v in res["nbt"]), res
        assert res["plain_out"] > 1e-3, res
