"""Worker for the data-parallel tests: one process per rank.

  tests/test_dp_gloo_cpu.py     python dp_worker.py <rank> <world> <port> <out_dir>   gloo, CPU, emulated kernels
  tests/test_parity_gpu.py      torchrun ... dp_worker.py  with KG_DP_BACKEND=nccl    RCCL, one GPU per rank, HIP kernels
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def run(rank, world, port, out_dir, backend="gloo"):
    if backend == "gloo":
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        dev = torch.device("cpu")
    else:       # RCCL: launched by torchrun, one GPU per rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local = int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=dev)
    import kinetic_gan_amd  # noqa: F401
    from kinetic_gan_amd import _native
    from kinetic_gan_amd.discriminator import Discriminator
    from kinetic_gan_amd.generator import Generator
    from kinetic_gan_amd.wgan_gp import Trainer
    from oracle import prim_ref
    from oracle.fill import fill_module, rand_inputs, rand_noise
    if backend == "gloo":
        prim_ref.install(_native)

    def models(seed_shift):
        G = Generator(512, 2, 10, 32, 4, dataset="h36m")
        D = Discriminator(2, 10, 32, 512, dataset="h36m")
        fill_module(G, seed=1 + seed_shift)
        fill_module(D, seed=2 + seed_shift)
        return G.to(dev), D.to(dev)

    n = 2
    shards = [tuple(t.to(dev) for t in rand_inputs(n, 2, 32, 16, 10, 512, seed=100 + r)) for r in range(world)]
    noises = [[t.to(dev) for t in rand_noise(n, 32, [16, 7, 2, 1], seed=200 + r)] for r in range(world)]

    # rank r starts from DIFFERENT weights; the trainer must broadcast rank 0's
    G, D = models(seed_shift=10 * rank)
    comm = None
    if backend != "gloo" and os.environ.get("KG_DP_COMM") == "kg":      # the library's own RCCL communicator (C ABI)
        comm = _native.Comm(rank, world, dev.index, exchange=_native.torch_dist_exchange(0))
    tr = Trainer(G, D, world_size=world, comm=comm, fused_step=False)      # (the buckets are compared as the optimiser read them)
    real, labels, z, alpha = shards[rank]
    tr.iteration(real, labels, z, alpha, noises[rank], noises[rank], with_g=True)
    got = torch.cat([tr.fD.flat, tr.fG.flat]).clone()
    # the gradient buckets still hold what the optimiser read: the all-reduced SUM over ranks (1 / world is folded into
    # the Adam kernel) - compared below with the hand-made sum, BEFORE any Adam normalisation blurs magnitudes
    got_gd, got_gg = tr.fD.grad.clone(), tr.fG.grad.clone()

    # expected: the same iteration done by hand in one process - per-shard gradients averaged, one Adam step each
    G2, D2 = models(seed_shift=0)
    t2 = Trainer(G2, D2, world_size=1, fused_step=False)
    gsum = torch.zeros_like(t2.fD.grad)
    for r in range(world):
        t2.fD.zero_grad()
        real, labels, z, alpha = shards[r]
        Gr, _ = models(seed_shift=0)          # every rank runs G from the same (pre-step) weights and BN state
        t2.G = Gr
        t2.d_losses(real, labels, z, alpha, noises[r])["d_loss"].backward()
        t2.fD.gather_grads()
        gsum += t2.fD.grad
    want_gd = gsum.clone()
    t2.fD.grad.copy_(gsum / world)
    t2.fD.allreduce_and_step(t2.lr, t2.b1, t2.b2, world=1, gather=False)
    t2.G = G2
    gsumG = torch.zeros_like(t2.fG.grad)
    base_state = {k: v.clone() for k, v in G2.state_dict().items()}
    t2.fD.set_requires_grad(False)
    for r in range(world):
        # each rank's generator has already run one forward in its D step (BN running stats moved); those are
        # buffers, not parameters, and do not enter the train-mode forward - so only gradients are compared
        t2.fG.zero_grad()
        real, labels, z, alpha = shards[r]
        t2.g_losses(labels, z, noises[r])["g_loss"].backward()
        t2.fG.gather_grads()
        gsumG += t2.fG.grad
    t2.fD.set_requires_grad(True)
    want_gg = gsumG.clone()
    t2.fG.grad.copy_(gsumG / world)
    t2.fG.allreduce_and_step(t2.lr, t2.b1, t2.b2, world=1, gather=False)
    want = torch.cat([t2.fD.flat, t2.fG.flat])

    gathered = [torch.zeros_like(got) for _ in range(world)]
    dist.all_gather(gathered, got)
    same_across_ranks = all(torch.equal(gathered[0], g) for g in gathered)
    diff = (got - want).abs()
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    gg = [torch.zeros_like(got_gd) for _ in range(world)]
    dist.all_gather(gg, got_gd)
    same_grads = all(torch.equal(gg[0], g) for g in gg)
    res = {"same": same_across_ranks, "same_grads": same_grads, "max": diff.max().item(), "mean": diff.mean().item(),
           "grad_d_l2": l2(got_gd, want_gd), "grad_g_l2": l2(got_gg, want_gg)}
    if out_dir:
        torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    return res


if __name__ == "__main__":
    if os.environ.get("KG_DP_BACKEND") == "nccl":
        r = run(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), 0, None, backend="nccl")
        # D's bucket: the same arithmetic on both sides up to the summation order of the all-reduce; G's bucket is taken
        # through a critic that already made one Adam step (weights with a near-zero gradient may sit +-lr apart)
        ok = r["same"] and r["same_grads"] and r["grad_d_l2"] <= 1e-5 and r["grad_g_l2"] <= 2e-3 and r["mean"] <= 2e-6
        print("rank", os.environ["RANK"], r, "OK" if ok else "MISMATCH", flush=True)
        sys.exit(0 if ok else 1)
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
