"""WGAN-GP iteration of the reference's training script on the HIP path (row T of SURVEY.md 8a).

``Trainer.d_step`` / ``Trainer.g_step`` reproduce kinetic-gan.py:137-155 and :167-174:

  D step: fake = G(z, labels); D(real), D(fake); gradient penalty on interpolates with
          ``create_graph=True`` (kinetic-gan.py:94-114); d_loss = -E[D(real)] + E[D(fake)] + lambda*GP;
          Adam(lr, (b1, b2)) on D.
  G step: fake = G(z, labels) (same z / labels); g_loss = -E[D(fake)]; Adam on G.

Result-neutral departures from the script: the D step does not back-propagate
into G (the reference does, then discards those gradients at :157); the G step does not compute
D's weight gradients (discarded by the next zero_grad at :137); D(real) and D(fake) run as one
2N batch (D has no batch-coupled op); the penalty's first-order backward skips parameter gradients
(only d/d(interpolates) is consumed).

Data parallelism (one process per GPU): parameters and gradients of each network live in ONE flat
fp32 buffer each; a step's gradients are summed across ranks by a single RCCL all-reduce of the
flat gradient buffer (torch.distributed backend "nccl" == RCCL over xGMI) and the 1/world scaling
is folded into the flat-buffer Adam kernel (kg_adam_step).  BatchNorm batch statistics in G stay per rank
(as under DistributedDataParallel without SyncBatchNorm); the running statistics / batch counters are broadcast
from rank 0 once at start like DDP's buffers (``Trainer.broadcast_buffers``) and then evolve per rank.
"""
from __future__ import annotations

from typing import List, Optional

import contextlib
import os
import weakref

import torch
import torch.distributed as dist

from . import _native as nv
from . import ops


class FlatParams:
    """Re-points a module's parameters (and .grad) at slices of two flat fp32 buffers."""

    def __init__(self, module: torch.nn.Module):
        named = list(module.named_parameters())
        self.params = [p for _, p in named]
        dev = self.params[0].device
        # Large tensors (the mapping network's Linear weights, the embedding table: read with 16-byte loads by
        # kg_linear_*; conv weights: 16-byte LDS-DMA of the ring form of kg_conv) start on a 16-byte boundary; small ones stay
        # packed.  A ParameterList's entries (the edge importances, "edge_importance.<i>") must remain ADJACENT - their
        # gradients are added by ONE launch (disc_trunk.MaskedAdjacencyFn, disc_trunk._pack) - so padding goes in front of
        # a list's first entry, never inside it (round-4 ADVICE: the NTU generator's 1875-element entry used to get a
        # 2-float pad in front of it, and the block-by-block path fell back to torch.cat).
        def _list_key(name):
            head, _, tail = name.rpartition(".")
            return head if tail.isdigit() else None
        self.offsets, total = [], 0
        prev_list = None
        for name, p in named:
            lk = _list_key(name)
            inside = lk is not None and lk == prev_list
            prev_list = lk
            if p.numel() >= 1024 and not inside:
                total = (total + 3) // 4 * 4
            self.offsets.append(total)
            total += p.numel()
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        # fused_step: the optimiser launch clears the gradient bucket it has consumed (kg_adam_step_fused) - the next
        # zero_grad needs no fill launch (Trainer(fused_step=...)).
        # `_clean`: the bucket is known to be all zero (nothing but that launch has written it since)
        self.fused_step = False
        self._clean = False
        self.views = []
        keys = []
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            self.views.append(self.grad[off:off + n])
            p.grad = None
            # conv weights / biases: the kernels accumulate straight into the bucket slice (ops._ParamSink)
            keys.append(ops.register_param_sink(p, self.views[-1]))
        weakref.finalize(self, ops.unregister_param_sinks, keys)

    def zero_grad(self):
        """Zero the bucket (one fill) and drop the per-parameter .grad tensors.  The convolution kernels accumulate
        their weight / bias gradients straight into the bucket slices (ops parameter sink); autograd hands over the
        remaining parameters' gradients as fresh tensors (no accumulation add), which gather_grads folds in."""
        ops.reset_param_sink(self.grad)      # this bucket's records a failed backward pass may have left behind
        if not self._clean:
            self.grad.zero_()
        self._clean = False                  # (the backward pass that follows writes into it)
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """Complete the bucket: wait for sink launches on the side stream (if that option is on), add the gradients
        autograd produced for the non-convolution parameters with ONE multi-tensor add, and point every .grad at
        its bucket slice."""
        ops.join_param_sink()
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is not None and g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g.reshape(-1))
        if dst:
            torch._foreach_add_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v.view(p.shape)

    def set_requires_grad(self, flag: bool):
        for p in self.params:
            p.requires_grad_(flag)

    def broadcast(self, src: int = 0):
        dist.broadcast(self.flat, src)

    def allreduce_and_step(self, lr, b1, b2, eps=1e-8, world: int = 1, gather: bool = True, comm=None):
        """``comm``: a ``_native.Comm`` - the gradient bucket is summed by kg_allreduce_flat (RCCL behind the C ABI, on
        torch's current stream: capturable) instead of torch.distributed."""
        if gather:
            self.gather_grads()
        if comm is not None:
            if comm.world > 1 or comm.force:
                comm.allreduce_(self.grad)
        elif world > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        self.step += 1
        nv.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, lr, b1, b2, eps, self.step,
                     1.0 / world, zero_grad=self.fused_step)
        self._clean = self.fused_step


def penalty_of(d_inter, inter, keep: Optional[dict] = None):
    """kinetic-gan.py:103-113: ((|d D(inter) / d inter|_2 - 1)^2).mean() for an already evaluated D(inter).
    ``keep``: a dict that receives the gradient itself under "gp_grads" (parity tests compare it element-wise)."""
    ones = _const_like(d_inter, 1.0)
    with ops.no_param_grads():
        (grads,) = torch.autograd.grad(outputs=d_inter, inputs=inter, grad_outputs=ones,
                                       create_graph=True, retain_graph=True, only_inputs=True)
    if keep is not None:
        keep["gp_grads"] = grads.detach()
    return ops.GradPenalty.apply(grads)


_CONSTS: dict = {}


def _const_like(t: torch.Tensor, value: float) -> torch.Tensor:
    """A cached constant tensor of t's shape / dtype / device (the seeds of the step's backward passes: ones for the
    penalty's autograd.grad, -1/n for the generator loss): torch.ones_like / neg / div launched a kernel each per step."""
    key = (tuple(t.shape), t.dtype, str(t.device), float(value))
    c = _CONSTS.get(key)
    if c is None:
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            # (round-4 ADVICE) a fill recorded into a graph runs at REPLAY time: such a tensor must not enter the cache,
            # eager callers would read it uninitialised.  The eager warm-up step in front of every capture fills the cache.
            return torch.full(tuple(t.shape), float(value), dtype=t.dtype, device=t.device)
        if len(_CONSTS) > 64:
            _CONSTS.clear()
        c = _CONSTS[key] = torch.full(tuple(t.shape), float(value), dtype=t.dtype, device=t.device)
        if float(value) == 1.0:
            c._kg_unit_seed = True      # ops.CriticLoss.backward: a backward pass seeded with this tensor has constant gradients
    return c        # READ-ONLY: handed to autograd as a gradient seed on every step


def gradient_penalty(D, real, fake, labels, alpha, keep: Optional[dict] = None):
    """kinetic-gan.py:94-114 with alpha passed in (the script draws it with numpy).  ``keep``: as in penalty_of."""
    inter = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    d_inter = D(inter, labels)
    ones = torch.ones_like(d_inter)
    with ops.no_param_grads():
        (grads,) = torch.autograd.grad(outputs=d_inter, inputs=inter, grad_outputs=ones,
                                       create_graph=True, retain_graph=True, only_inputs=True)
    if keep is not None:
        keep["gp_grads"] = grads.detach()
    grads = grads.reshape(grads.size(0), -1)
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


class Trainer:
    def __init__(self, G, D, lr=2e-4, b1=0.5, b2=0.999, lambda_gp=10.0, n_critic=5,
                 world_size: int = 1, flatten: bool = True, overlap: Optional[bool] = None, comm=None,
                 fused_step: bool = True):
        """``fused_step``: the optimiser launches clear the gradient buckets they have consumed (kg_adam_step_fused): after ``d_apply`` / ``g_apply`` the bucket reads zero - pass False to inspect the
        gradients the optimiser used (the data-parallel tests do).
        ``overlap`` (opt-in): the critic's all-reduce + Adam run on a side stream underneath the generator step's G
        forward, which does not read D (kinetic-gan.py:167 needs the updated D only at :170); the two generator
        syntheses of the iteration are then NOT paired (the generator step keeps its own forward pass to hide the
        collective under).  Off by default."""
        self.G, self.D = G, D
        # ``comm``: a _native.Comm (kg_comm_init): the two gradient all-reduces of an iteration go through the C ABI
        # (kg_allreduce_flat) instead of torch.distributed.all_reduce; torch.distributed is then only used for the
        # one-off parameter / buffer broadcast at construction
        self.comm = comm
        if comm is not None and comm.world != world_size:
            raise ValueError("Trainer: comm.world=%d but world_size=%d" % (comm.world, world_size))
        self.lr, self.b1, self.b2 = lr, b1, b2
        self.lambda_gp, self.n_critic = lambda_gp, n_critic
        self.world = world_size
        self.fG = FlatParams(G) if flatten else None
        self.fD = FlatParams(D) if flatten else None
        for f in (self.fG, self.fD):
            if f is not None:
                f.fused_step = bool(fused_step)
        dev = next(D.parameters()).device
        self.overlap = bool(overlap) and dev.type == "cuda" and flatten
        self._side = None
        self._wvec = None
        # fold the real+fake backward pass into the gradient penalty's (d_losses); needs the bucket sinks
        self._promise = bool(flatten) and os.environ.get("KG_TRUNK_MERGE", "1") != "0"
        self._share_mapping = False      # set by iteration(with_g=True) around the critic step
        self._w = None
        self._fake_g = None              # the generator step's sample when it was synthesised with the critic's
        self._noise_g = None             # injected noise of that sample (parity tests)
        if self.world > 1:
            self.fG.broadcast(0)
            self.fD.broadcast(0)
            self.broadcast_buffers()

    def broadcast_buffers(self, src: int = 0):
        """Rank `src`'s module buffers (the generator's BatchNorm running statistics / batch counters) to every rank
        - what DistributedDataParallel does at construction; call again before writing a checkpoint if every rank's
        statistics should agree (they are per-rank batch statistics during training)."""
        for m in (self.G, self.D):
            for b in m.buffers():
                dist.broadcast(b, src)

    # ---- losses (also used un-stepped by the parity tests) -------------------------------------------------
    def d_losses(self, real, labels, z, alpha, noise: Optional[List[torch.Tensor]] = None, fake=None,
                 promise: bool = False):
        """``fake`` (optional) replaces G(z, labels): lets tests feed both implementations the same batch.
        ``promise``: the caller will run exactly ``d_loss.backward()`` next (d_compute does): the critic loss is linear
        in D(real) and D(fake), so their gradient (-1/n, +1/n) is known now and the discriminator may fold that
        backward pass into the gradient penalty's (Discriminator.forward_parts)."""
        n = real.shape[0]
        if fake is None:
            exact_bn = getattr(self.G, "_exact_bn_active", None) is not None and self.G._exact_bn_active()
            if self._share_mapping and hasattr(self.G, "synthesis_pair") and self.G.training and not exact_bn and \
                    ((noise is None) == (self._noise_g is None)):
                # The generator step of this iteration runs G on the same (z, labels) with the same parameters
                # (kinetic-gan.py:143,167): the mapping network runs once and BOTH syntheses - this critic sample and
                # the generator step's sample, which keeps its autograd graph until g_backward - go through the
                # blocks as one 2n batch with per-half BatchNorm statistics (Generator.synthesis_pair).
                w = self.G.mapping(z, labels)
                fake, self._fake_g = self.G.synthesis_pair(w, noise, self._noise_g)
            elif self._share_mapping and hasattr(self.G, "synthesis"):
                self._w = self.G.mapping(z, labels)
                with torch.no_grad():
                    fake = self.G.synthesis(self._w.detach(), noise)
            else:
                with torch.no_grad():
                    fake = self.G(z, labels, noise=noise)
        if getattr(self.D, "use_trunk", False):
            # HIP path: real+fake and the penalty's interpolates go through D as ONE launch sequence over 3n
            # samples (same parameters; kinetic-gan.py:146-150 evaluates them one after the other)
            labels3 = torch.cat((labels, labels, labels), 0)
            buf = nv.mix3(real, fake, alpha)        # [real | fake | alpha real + (1 - alpha) fake] in one launch
            inter = buf[2 * n:].requires_grad_(True)
            key = (n, str(buf.device))
            if self._wvec is None or self._wvec[0] != key:
                w = torch.cat((torch.full((n,), -1.0 / n), torch.full((n,), 1.0 / n))).to(buf.device)
                self._wvec = (key, w)
            both, d_inter = self.D.forward_parts([(buf[:2 * n], labels3[:2 * n]), (inter, labels3[2 * n:])],
                                                 promised_grad=self._wvec[1] if promise else None)
            real_v, fake_v = both[:n], both[n:]
            out = {"fake": fake, "real_validity": real_v, "fake_validity": fake_v}
            gp = penalty_of(d_inter, inter, out)
            out["gradient_penalty"] = gp
            out["d_loss"] = ops.CriticLoss.apply(both, gp, self._wvec[1], float(self.lambda_gp))
            return out
        share = getattr(self.D, "shared_adjacency", None)       # the oracle's modules do not have it
        with (share() if share is not None else contextlib.nullcontext()):
            both = self.D(torch.cat((real, fake), 0), torch.cat((labels, labels), 0))
            real_v, fake_v = both[:n], both[n:]
            out = {"fake": fake, "real_validity": real_v, "fake_validity": fake_v}
            gp = gradient_penalty(self.D, real, fake, labels, alpha, out)
        out["gradient_penalty"] = gp
        out["d_loss"] = -real_v.mean() + fake_v.mean() + self.lambda_gp * gp
        return out

    def g_losses(self, labels, z, noise: Optional[List[torch.Tensor]] = None):
        fake = self.G(z, labels, noise=noise)
        fake_v = self.D(fake, labels)
        return {"fake": fake, "fake_validity": fake_v, "g_loss": -fake_v.mean()}

    # ---- optimisation steps ----------------------------------------------------------------------------------
    # Each step is split into a compute half (forward + backward + gradient gather: pure GPU work, no
    # communication - capturable in a hipGraph) and an apply half (RCCL all-reduce of the flat bucket + Adam).
    def d_compute(self, real, labels, z, alpha, noise=None, fake=None, keep: Optional[dict] = None):
        """``fake`` / ``keep`` (parity tests): feed D this sample instead of G(z, labels); receive the step's losses
        (detached) - the arithmetic, launches and gradient bucket are those of the plain call."""
        self.fD.zero_grad()
        r = self.d_losses(real, labels, z, alpha, noise, fake=fake, promise=self._promise)
        r["d_loss"].backward(_const_like(r["d_loss"], 1.0))        # (a cached seed: no ones_like launch)
        self.fD.gather_grads()
        if keep is not None:
            keep.update({k: v.detach() for k, v in r.items()})
        return r["d_loss"].detach()

    def d_apply(self):
        self.fD.allreduce_and_step(self.lr, self.b1, self.b2, world=self.world, gather=False, comm=self.comm)

    def d_apply_async(self):
        """d_apply on the side stream, ordered after everything queued so far; `wait_d_apply` joins it."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.fD.flat.device)
        cur = torch.cuda.current_stream(self.fD.flat.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self.d_apply()

    def wait_d_apply(self):
        if self._side is not None:
            torch.cuda.current_stream(self.fD.flat.device).wait_stream(self._side)

    def g_forward(self, labels, z, noise=None):
        """First half of the generator step: fake = G(z, labels) with its autograd graph (does not touch D)."""
        self.fG.zero_grad()
        fake, self._fake_g = self._fake_g, None
        if fake is not None:        # already synthesised next to the critic step's sample (d_losses)
            return fake
        w, self._w = self._w, None
        if w is not None:
            return self.G.synthesis(w, noise)
        return self.G(z, labels, noise=noise)

    def g_backward(self, fake, labels):
        """Second half: g_loss = -E[D(fake)], backward into G, gradient gather."""
        self.fD.set_requires_grad(False)
        try:
            # g_loss = -E[D(fake)] (kinetic-gan.py:171): its gradient w.r.t. the validities is the constant -1/n, handed to
            # backward() directly; the value is one dot product (mean / neg / ones_like / neg / div were five launches)
            v = self.D(fake, labels)
            seed = _const_like(v, -1.0 / v.numel())
            g_loss = torch.dot(v.detach().reshape(-1), seed.reshape(-1))
            v.backward(seed)
        finally:
            self.fD.set_requires_grad(True)
        self.fG.gather_grads()
        return g_loss.detach()

    def g_compute(self, labels, z, noise=None):
        return self.g_backward(self.g_forward(labels, z, noise), labels)

    def g_apply(self):
        self.fG.allreduce_and_step(self.lr, self.b1, self.b2, world=self.world, gather=False, comm=self.comm)

    def d_step(self, real, labels, z, alpha, noise=None):
        loss = self.d_compute(real, labels, z, alpha, noise)
        self.d_apply()
        return loss

    def g_step(self, labels, z, noise=None):
        loss = self.g_compute(labels, z, noise)
        self.g_apply()
        return loss

    @contextlib.contextmanager
    def sharing_mapping(self, noise_g=None):
        """Inside: d_compute also prepares the generator step that follows with the same z / labels (mapping network
        once, both syntheses as one batch) - the g_forward after it just picks the sample up.  ``noise_g``: the
        generator step's injected noise, if the critic step's is injected too.  `iteration(with_g=True)` does this
        itself."""
        prev, self._share_mapping = self._share_mapping, self.fG is not None
        self._noise_g = noise_g
        try:
            yield
        finally:
            self._share_mapping = prev
            self._noise_g = None

    def iteration(self, real, labels, z, alpha, noise_d=None, noise_g=None, with_g: bool = True):
        """One loop body of kinetic-gan.py:123-174 (``with_g`` = the i % n_critic == 0 branch)."""
        # (overlap mode keeps the generator step's forward pass IN the generator step: that pass is what D's
        # all-reduce + Adam hide under, so the two syntheses are not paired then)
        self._share_mapping = bool(with_g) and self.fG is not None and not self.overlap
        self._noise_g = noise_g if (with_g and not self.overlap) else None
        try:
            if not (self.overlap and with_g):
                d_loss = self.d_step(real, labels, z, alpha, noise_d)
                g_loss = self.g_step(labels, z, noise_g) if with_g else None
                return d_loss, g_loss
            d_loss = self.d_compute(real, labels, z, alpha, noise_d)
        finally:
            self._share_mapping = False
            self._noise_g = None
        self.d_apply_async()                       # RCCL all-reduce + Adam of D on the side stream ...
        fake = self.g_forward(labels, z, noise_g)  # ... under the generator's forward
        self.wait_d_apply()
        g_loss = self.g_backward(fake, labels)
        self.g_apply()
        return d_loss, g_loss
