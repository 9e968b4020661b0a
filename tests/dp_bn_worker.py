"""Worker of tests/test_dp_gloo_cpu.py::test_exact_batchnorm_mode_equals_single_process_global_batch: one process per
rank (gloo, CPU, kernels emulated by oracle/prim_ref.py).  Every rank runs the generator in the exact data-parallel
BatchNorm mode (Generator.exact_bn) on ITS shard and, for reference, the plain generator on the concatenated global batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def run(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import kinetic_gan_amd  # noqa: F401
    from kinetic_gan_amd import _native
    from kinetic_gan_amd.generator import Generator
    from oracle import prim_ref
    from oracle.fill import fill_module, rand_inputs, rand_noise
    prim_ref.install(_native)

    def gen(exact):
        G = Generator(512, 2, 10, 32, 4, dataset="h36m")
        fill_module(G, seed=1)
        G.exact_bn = exact
        return G

    n = 3
    nodes = [16, 7, 2, 1]
    shards = [rand_inputs(n, 2, 32, 16, 10, 512, seed=100 + r) for r in range(world)]
    noises = [rand_noise(n, 32, nodes, seed=200 + r) for r in range(world)]
    wts = [torch.randn(n, 2, 32, 16, generator=torch.Generator().manual_seed(300 + r)) for r in range(world)]

    # (a) the exact mode on this rank's shard: two forward passes (running statistics move twice), loss = shard mean
    G = gen(True)
    _, labels, z, _ = shards[rank]
    out1 = G(z, labels, noise=noises[rank])
    out = G(z, labels, noise=noises[rank])
    (out * wts[rank]).mean().backward()
    grads = torch.cat([p.grad.reshape(-1) for p in G.parameters()])
    dist.all_reduce(grads)
    grads /= world                                  # what the gradient bucket's all-reduce + 1 / world does
    bufs = {k: v.clone() for k, v in G.state_dict().items() if "running_" in k or "num_batches" in k}

    # (b) reference in this process: the plain generator on the GLOBAL batch, loss = global mean
    R = gen(False)
    zc = torch.cat([s[2] for s in shards]); lc = torch.cat([s[1] for s in shards])
    nc = [torch.cat([noises[r][i] for r in range(world)]) for i in range(len(noises[0]))]
    wc = torch.cat(wts)
    ref1 = R(zc, lc, noise=nc)
    ref = R(zc, lc, noise=nc)
    (ref * wc).mean().backward()
    rgrads = torch.cat([p.grad.reshape(-1) for p in R.parameters()])
    rbufs = {k: v for k, v in R.state_dict().items() if "running_" in k or "num_batches" in k}

    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    mine = slice(rank * n, (rank + 1) * n)
    res = {"out1": l2(out1, ref1[mine]), "out": l2(out, ref[mine]), "grads": l2(grads, rgrads),
           "bufs": max(l2(bufs[k].float(), rbufs[k].float()) for k in bufs), "nbufs": len(bufs),
           "nbt": [int(v) for k, v in bufs.items() if "num_batches" in k]}
    # and the default mode (per-rank statistics) must NOT agree with the global-batch run: the test would otherwise pass
    # with the mode switched off
    P = gen(False)
    res["plain_out"] = l2(P(z, labels, noise=noises[rank]), ref1[mine])
    torch.save(res, os.path.join(out_dir, f"bn_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
