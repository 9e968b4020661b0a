"""torch.autograd glue over the C ABI: every Function below is ONE libkgan_hip.so launch (or a
fused launch) in forward, and its backward is written in terms of the other Functions, so the
family is closed under differentiation.  That is what the WGAN-GP gradient penalty needs
(kinetic-gan.py:104-111 differentiates THROUGH the backward of every discriminator op, w.r.t.
weights and edge_importance):

    Conv / ConvT / WGrad          channel contraction, its adjoint, its weight gradient
    AggExpand / AggReduce / AggOuter   spatial aggregation, its adjoint, its adjacency gradient
    ActBwd                        g * act'(out)   (LeakyReLU: piecewise linear, self-adjoint)
    RowSum                        bias gradients
    DiscTail                      fused  lrelu(tcn(z) + residual(x) + bias)[kept t]  of a D block
    GenTail                       fused  act(BN(u) + BN(r) + noise)  of a G block (first order only)

All tensors are "plane tensors" (see _native.py); the native entry points are looked up through
the ``_native`` module at call time.

Every Function opts out of gradient materialisation and returns ``None`` for an absent incoming gradient.  That
matters for the gradient penalty: the double-backward graph references the penalty's FORWARD activations only
through ``ActBwd``'s ``ref`` input, whose gradient is ``None`` (LeakyReLU is piecewise linear) - with materialised
zeros autograd would still run the whole backward of that forward graph (every data-gradient convolution, weight
gradient and aggregation adjoint of D at N samples) on tensors of zeros.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native as nv
from ._native import ACT_LRELU, ACT_NONE, ACT_TANH, TAP_CHANBLOCK, TAP_TIME, Group, WView

# When True, backward passes skip parameter / adjacency gradients.  Set (via `no_param_grads`)
# around the gradient penalty's first-order autograd.grad (only d/d(interpolates) is consumed,
# kinetic-gan.py:104-111) and around the generator step's pass through D (kinetic-gan.py:170-173:
# D's weight gradients are discarded by the next zero_grad) - both result-neutral.
_SKIP_PARAM_GRADS = False


class no_param_grads:
    def __enter__(self):
        global _SKIP_PARAM_GRADS
        self.prev = _SKIP_PARAM_GRADS
        _SKIP_PARAM_GRADS = True

    def __exit__(self, *exc):
        global _SKIP_PARAM_GRADS
        _SKIP_PARAM_GRADS = self.prev


# ---- parameter-gradient sink -----------------------------------------------------------------------------------------
# Weight / bias gradients of the convolutions are consumed by nobody before the optimizer step, so in a first-order
# backward pass (create_graph=False) they need not be autograd outputs at all: when a parameter is registered here
# (wgan_gp.FlatParams does that), its gradient is ACCUMULATED BY THE KERNEL into the parameter's slice of the flat
# gradient bucket (kg_wgrad / kg_rowsum with accumulate=1) and autograd sees None.  That removes the per-parameter
# accumulation adds and the gather into the bucket.  Because nothing downstream waits for these launches they CAN
# run on a side stream next to the data-gradient chain (KG_PARAM_SIDE_STREAM=1 / param_sink_options): measured on
# MI355X under hipGraph replay that is 8 % SLOWER (10.7 vs 9.9 ms per iteration, profiles/r01_v7_param_sink_ab.log),
# like the earlier attempt to run the gradient-penalty branch concurrently, so the default keeps one stream.
# `join_param_sink()` makes the current stream wait for the side stream; callers do that once before they read the
# bucket (all-reduce / Adam).
class _ParamSink:
    def __init__(self):
        self.views = {}
        self.enabled = True
        self.use_side_stream = False
        self.side = {}
        self.dirty = set()
        self.defer = True           # collect the weight-gradient operand pairs of a backward pass per weight and
        self.pending = {}           # launch them together (up to three pairs per kg_wgrad call) at join time
        self.pending_rows = []      # deferred bias-gradient reductions (views, g): one kg_rowsum_many at join time

    def stream(self, device):
        st = self.side.get(device)
        if st is None:
            st = self.side[device] = torch.cuda.Stream(device=device)
        return st


_SINK = _ParamSink()
if os.environ.get("KG_PARAM_SINK") == "0":             # A/B switches (bench / debugging)
    _SINK.enabled = False
if os.environ.get("KG_PARAM_SIDE_STREAM") == "1":
    _SINK.use_side_stream = True
if os.environ.get("KG_PARAM_DEFER") == "0":
    _SINK.defer = False


def register_param_sink(param: torch.Tensor, flat_view: torch.Tensor):
    """flat_view: contiguous 1-D fp32 view of the bucket slice that holds d(loss)/d(param), same element order.
    Returns the key for unregister_param_sinks (entries are keyed by the parameter's address: drop them when the
    bucket goes away)."""
    assert flat_view.numel() == param.numel() and flat_view.is_contiguous()
    key = (param.data_ptr(), param.numel())
    _SINK.views[key] = flat_view
    return key


def unregister_param_sinks(keys):
    for k in keys:
        _SINK.views.pop(k, None)


def clear_param_sinks():
    _SINK.views.clear()


def param_sink_options(enabled: Optional[bool] = None, side_stream: Optional[bool] = None,
                       defer: Optional[bool] = None):
    if enabled is not None:
        _SINK.enabled = enabled
    if side_stream is not None:
        _SINK.use_side_stream = side_stream
    if defer is not None:
        _SINK.defer = defer


def _sink_of(t: Optional[torch.Tensor]):
    if t is None or not _SINK.enabled or not _SINK.views:
        return None
    return _SINK.views.get((t.data_ptr(), t.numel()))


class _on_side:
    """Run the enclosed launches on the device's side stream, after everything queued on the current one; the
    tensors they read are kept alive for it (record_stream)."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.ctx = None

    def __enter__(self):
        t0 = self.tensors[0]
        if t0.is_cuda and _SINK.use_side_stream:
            side = _SINK.stream(t0.device)
            side.wait_stream(torch.cuda.current_stream(t0.device))
            for t in self.tensors:
                t.record_stream(side)
            _SINK.dirty.add(t0.device)
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


def _wgrad_job(view, spec, pairs):
    (x0, g0), rest = pairs[0], pairs[1:]
    return dict(g=g0, x=x0, Cin=spec.Cin, taps=spec.taps, tap_mode=spec.tap_mode, t_stride=spec.t_stride,
                vmap=spec.vmap, wv=WView(spec.wv.sT, spec.wv.sO, spec.wv.sI), out=view, accumulate=True,
                extra=[(g, x) for x, g in rest])


def join_param_sink():
    """Launch the deferred weight-gradient products: ALL layers of the pass in shared launches (kg_wgrad_many), up
    to three operand pairs per layer and round (a weight of D receives one contribution from the real+fake batch
    and one from the gradient penalty's double backward), and make the current stream wait for the side stream if
    that option is on."""
    pending, _SINK.pending = _SINK.pending, {}
    flush_outer()               # (normally done by the adjacency pack's backward)
    with torch.no_grad():       # the operands may be graph tensors (the penalty's interpolates require grad)
        # A job takes up to three operand pairs of one weight.  Further pairs of the SAME weight go to a later
        # round: the jobs of one launch add into the bucket without atomics and must not share a destination.
        entries = list(pending.values())
        rnd = 0
        while True:
            jobs = [_wgrad_job(view, spec, pairs[3 * rnd:3 * rnd + 3]) for (view, spec, pairs) in entries
                    if pairs[3 * rnd:3 * rnd + 3]]
            if not jobs:
                break
            with _on_side(*[t for j in jobs for t in (j["g"], j["x"])]):
                nv.wgrad_many(jobs)
            rnd += 1
        # bias gradients: one launch over all of them; a destination that received several contributions (gradient
        # accumulation over backward passes) takes one per round
        rows, _SINK.pending_rows = _SINK.pending_rows, []
        while rows:
            seen, now, later = set(), [], []
            for views, g, y in rows:
                ptrs = [v.data_ptr() for v in views]
                if any(p in seen for p in ptrs):
                    later.append((views, g, y))
                else:
                    seen.update(ptrs)
                    now.append(dict(x=g, y=y, out=views[0], out2=views[1] if len(views) > 1 else None,
                                    accumulate=True))
            with _on_side(*[t for j in now for t in (j["x"], j["y"]) if t is not None]):
                nv.rowsum_many(now)
            rows = later
    for dev in list(_SINK.dirty):
        torch.cuda.current_stream(dev).wait_stream(_SINK.side[dev])
    _SINK.dirty.clear()


def _direct_param_grads() -> bool:
    # first-order backward only: with create_graph=True the gradients must stay differentiable autograd outputs
    return not torch.is_grad_enabled()


def reset_param_sink(bucket: Optional[torch.Tensor] = None):
    """Drop deferred weight- / bias-gradient work whose destination lies inside ``bucket`` (a flat gradient buffer;
    ``None`` = everything, incl. recorded adjacency-gradient problems and un-joined side-stream marks).  A backward pass
    that raised midway leaves such records behind: they would pin their activations and be added into the next
    step's freshly zeroed bucket.  Records of OTHER buckets stay: zeroing one network's gradients between another
    network's backward pass and its gather must not discard that network's deferred gradients (round-2 ADVICE)."""
    if bucket is None:
        _SINK.pending = {}
        _SINK.pending_rows = []
        _SINK.dirty.clear()
        _OUTER_PENDING.clear()
        return
    lo = bucket.data_ptr()
    hi = lo + 4 * bucket.numel()
    inside = lambda v: v is not None and lo <= v.data_ptr() < hi
    _SINK.pending = {k: e for k, e in _SINK.pending.items() if not inside(e[0])}
    _SINK.pending_rows = [r for r in _SINK.pending_rows if not any(inside(v) for v in r[0])]
    # Recorded adjacency-gradient problems (_OUTER_PENDING) and side-stream marks carry no destination bucket: they are
    # consumed inside the backward pass that recorded them (flush_outer / join_param_sink), so whatever is left when NO
    # other bucket has deferred work is the debris of a pass that raised - drop it, or the next flush_outer() launches
    # outer products over the failed pass's (pinned) activations and a stale stream wait survives (round-3 ADVICE).
    if not _SINK.pending and not _SINK.pending_rows:
        _OUTER_PENDING.clear()
        _SINK.dirty.clear()


def _wgrad_into(view, x, g, spec):
    if _SINK.defer:
        # one entry per destination AND layer geometry: pairs of one entry share a launch
        key = (view.data_ptr(), spec.M, spec.Cin, spec.taps, spec.tap_mode, spec.t_stride, spec.T_in, spec.V_in,
               spec.T_out, spec.V_out)
        ent = _SINK.pending.get(key)
        if ent is None:
            _SINK.pending[key] = (view, spec, [(x, g)])
        else:
            ent[2].append((x, g))
        return
    with _on_side(x, g):
        nv.wgrad(g, x, spec.Cin, spec.taps, spec.tap_mode, spec.t_stride, spec.vmap, view.numel(),
                 WView(spec.wv.sT, spec.wv.sO, spec.wv.sI), out=view, accumulate=True)


def _rowsum_into(views, g, y=None):
    """per-channel sum of g (of g*y with ``y``) added into the bucket views"""
    if _SINK.defer:
        _SINK.pending_rows.append((views, g, y))
        return
    with _on_side(g):
        nv.rowsum(g, y, 2 if y is not None else False, out=views[0], accumulate=True,
                  out2=views[1] if len(views) > 1 else None)


@dataclass(frozen=True, eq=False)
class ConvSpec:
    """Geometry of one tap GEMM:  x (N, Cin[*taps], T_in, V_in) -> out (N, M, T_out, V_out)."""
    M: int
    Cin: int
    taps: int
    tap_mode: int
    t_stride: int
    T_in: int
    V_in: int
    T_out: int
    V_out: int
    wv: WView                      # W(d, m, c) addressing into the parameter's storage
    w_shape: Tuple[int, ...]
    vmap: Optional[torch.Tensor] = None       # out vertex -> in vertex   (int32, device)
    inv_vmap: Optional[torch.Tensor] = None   # in vertex -> out vertex or -1

    @property
    def x_channels(self):
        return self.Cin * (self.taps if self.tap_mode == TAP_CHANBLOCK else 1)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


# ---- channel contraction family ------------------------------------------------------------------------

class Conv(Function):
    @staticmethod
    def forward(ctx, x, w, bias, spec: ConvSpec, pre=None):
        """``pre``: the result, already computed by a launch over a larger batch this one is a sample range of
        (``pair_apply``): only the autograd node is set up."""
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.spec = spec
        ctx.has_bias = bias is not None
        ctx.w_sink, ctx.b_sink = _sink_of(w), _sink_of(bias)
        ctx.save_for_backward(x, w)
        if pre is not None:
            return pre
        grp = Group(x, w, spec.wv, spec.Cin, spec.taps, spec.tap_mode, spec.t_stride, False, spec.vmap)
        return nv.conv([grp], x.shape[0], spec.M, spec.T_out, spec.V_out, bias0=bias)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None, None, None, None)
        x, w = ctx.saved_tensors
        spec = ctx.spec
        gx = ConvT.apply(g, w, spec) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if not _SKIP_PARAM_GRADS:
            direct = _direct_param_grads()
            if ctx.needs_input_grad[1]:
                if direct and ctx.w_sink is not None:
                    _wgrad_into(ctx.w_sink, x, g, spec)
                else:
                    gw = WGrad.apply(x, g, spec)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                if direct and ctx.b_sink is not None:
                    _rowsum_into([ctx.b_sink], g)
                else:
                    gb = RowSum.apply(g)
        return gx, gw, gb, None, None


class ConvT(Function):
    """Adjoint of Conv w.r.t. its input: g (N, M, T_out, V_out) -> (N, x_channels, T_in, V_in)."""

    @staticmethod
    def forward(ctx, g, w, spec: ConvSpec):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.spec = spec
        ctx.w_sink = _sink_of(w)
        ctx.save_for_backward(g, w)
        wv = spec.wv
        if spec.tap_mode == TAP_TIME:
            grp = Group(g, w, WView(wv.sT, wv.sI, wv.sO), spec.M, spec.taps, TAP_TIME, spec.t_stride, True,
                        spec.inv_vmap)
            return nv.conv([grp], g.shape[0], spec.Cin, spec.T_in, spec.V_in)
        assert spec.t_stride == 1 and spec.vmap is None
        grp = Group(g, w, WView(0, wv.sI, wv.sO, wv.sT, spec.Cin), spec.M, 1)
        return nv.conv([grp], g.shape[0], spec.Cin * spec.taps, spec.T_in, spec.V_in)

    @staticmethod
    def backward(ctx, gg):
        if gg is None:
            return (None, None, None)
        g, w = ctx.saved_tensors
        spec = ctx.spec
        dg = Conv.apply(gg, w, None, spec) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS:
            if _direct_param_grads() and ctx.w_sink is not None:
                _wgrad_into(ctx.w_sink, gg, g, spec)
            else:
                dw = WGrad.apply(gg, g, spec)
        return dg, dw, None


class WGrad(Function):
    """dW of Conv: x (N, x_channels, T_in, V_in), g (N, M, T_out, V_out) -> tensor shaped like the weight."""

    @staticmethod
    def forward(ctx, x, g, spec: ConvSpec):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.spec = spec
        ctx.save_for_backward(x, g)
        out = None
        if spec.M * spec.Cin * spec.taps < _numel(spec.w_shape):      # the layer uses a subset of the weight's rows
            out = torch.zeros(_numel(spec.w_shape), dtype=torch.float32, device=g.device)
        flat = nv.wgrad(g, x, spec.Cin, spec.taps, spec.tap_mode, spec.t_stride, spec.vmap,
                        _numel(spec.w_shape), WView(spec.wv.sT, spec.wv.sO, spec.wv.sI), out=out)
        return flat.view(spec.w_shape)

    @staticmethod
    def backward(ctx, gw):
        if gw is None:
            return (None, None, None)
        x, g = ctx.saved_tensors
        spec = ctx.spec
        gw = gw.contiguous()
        dx = ConvT.apply(g, gw, spec) if ctx.needs_input_grad[0] else None
        dg = Conv.apply(x, gw, None, spec) if ctx.needs_input_grad[1] else None
        return dx, dg, None


# ---- spatial aggregation family -------------------------------------------------------------------------

def _t12(a):
    """A^T per partition as a VIEW: the aggregation kernels read a transposed adjacency in place (a_transposed)"""
    return a.transpose(1, 2)


class AggExpand(Function):
    """out[k*C+c,(n,t',w)] = sum_v x[c,(n,t'/rep,v)] A[k,v,w]."""

    @staticmethod
    def forward(ctx, x, A, rep: int, pre=None):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.rep = rep
        ctx.save_for_backward(x, A)
        return pre if pre is not None else nv.agg_expand(x, A, rep)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None, None, None)
        x, A = ctx.saved_tensors
        gx = AggReduce.apply(g, _t12(A), ctx.rep) if ctx.needs_input_grad[0] else None
        gA = None
        if ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS:
            gA = AggOuter.apply(x, g, A.shape[0], ctx.rep)
        return gx, gA, None, None


class AggReduce(Function):
    """out[c,(n,t,w)] = sum_{q<fold} sum_k sum_v y[k*C+c,(n,t*fold+q,v)] A[k,v,w]."""

    @staticmethod
    def forward(ctx, y, A, fold: int, pre=None, lazy_outer: bool = False):
        """``lazy_outer``: the consumer of A's gradient calls ``flush_outer()`` before it reads it (the generator's
        packed adjacencies, disc_trunk.MaskedAdjacencyFn): the outer products of all blocks then share one launch."""
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.fold = fold
        ctx.lazy_outer = lazy_outer
        ctx.save_for_backward(y, A)
        return pre if pre is not None else nv.agg_reduce(y, A, fold)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None, None, None)
        y, A = ctx.saved_tensors
        gy = AggExpand.apply(g, _t12(A), ctx.fold) if ctx.needs_input_grad[0] else None
        gA = None
        if ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS:
            if ctx.lazy_outer and not torch.is_grad_enabled():
                gA = nv.agg_outer(g, y, A.shape[0], ctx.fold, defer=_OUTER_PENDING).transpose(1, 2)
            else:
                gA = AggOuter.apply(g, y, A.shape[0], ctx.fold).transpose(1, 2)
        return gy, gA, None, None, None


class SinkLinear(Function):
    """F.linear whose weight / bias gradients are ADDED straight into their flat-bucket slices (`addmm_` / `add_` on
    the registered sink views) in a first-order backward pass: the mapping network's four 1024 x 1024 weights
    (generator.py:21-34) otherwise come back from autograd as fresh 4 MB tensors that gather_grads then adds into
    the bucket with one more pass over 2 x 16 MB.  Without registered sinks it behaves like F.linear."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.set_materialize_grads(False)
        ctx.sinks = (_sink_of(w), _sink_of(b))
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        x, w = ctx.saved_tensors
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = gb = None
        if not _SKIP_PARAM_GRADS:
            direct = _direct_param_grads()
            sw, sb = ctx.sinks
            g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
            if ctx.needs_input_grad[1]:
                if direct and sw is not None:
                    sw.view(w.shape).addmm_(g2.t(), x2)
                else:
                    gw = g2.t() @ x2
            if ctx.needs_input_grad[2]:
                if direct and sb is not None:
                    sb.add_(g2.sum(0))
                else:
                    gb = g2.sum(0)
        return gx, gw, gb


def emulated() -> bool:
    """True while a test has replaced the native entry points (tests/util.emulated_native installs its own definitions on
    `_native`): host logic that would refuse CPU tensors may then take the kernel route on them."""
    return getattr(nv.linear_fwd, "__module__", nv.__name__) != nv.__name__


class SyncBatchNorm2dFn(Function):
    """Train-mode BatchNorm2d over the GLOBAL batch of a data-parallel job - the optional exact mode of SURVEY.md 8(e):
    without it every rank normalises with its own shard's statistics (what DistributedDataParallel does without
    SyncBatchNorm), which differs numerically from a single-GPU run at the global batch.  Forward: ONE all-reduce of the
    packed per-channel [sum, sum of squares] (2 C floats; equal shard sizes); backward: ONE all-reduce of
    [sum g, sum g x_hat]; parameter gradients stay local (the gradient bucket's all-reduce averages them like every
    other parameter's).  Running statistics are updated with the global batch's mean / unbiased variance
    (generator.py:142,160).  Plain torch arithmetic - the exact mode is for comparisons with a single-GPU run, not
    for the benchmark."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, momentum, eps, group):
        import torch.distributed as dist
        C = x.shape[1]
        xs = x.detach()
        # [sum, sum of squares, local element count]: the count travels with the sums, so ranks may hold unequal shards (a
        # last batch without drop_last - round-4 ADVICE); E[x^2] - mean^2 in fp32 is good to ~1e-6 relative for the
        # |mean| <~ std activations of the generator, the comparison mode's tolerance
        cnt = torch.full((1,), float(x.numel() // C), dtype=xs.dtype, device=xs.device)
        packed = torch.cat((xs.sum((0, 2, 3)), (xs * xs).sum((0, 2, 3)), cnt))
        dist.all_reduce(packed, group=group)
        n = float(packed[2 * C].item())
        mean = packed[:C] / n
        var = (packed[C:2 * C] / n - mean * mean).clamp_min_(0.0)
        rstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            with torch.no_grad():
                nbt += 1
                f = momentum if momentum is not None else 1.0 / float(nbt)
                running_mean.mul_(1 - f).add_(mean, alpha=f)
                running_var.mul_(1 - f).add_(var * (n / max(n - 1.0, 1.0)), alpha=f)
        xhat = (xs - mean.view(1, C, 1, 1)) * rstd.view(1, C, 1, 1)
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.n, ctx.group = n, group
        y = xhat
        if gamma is not None:
            y = y * gamma.view(1, C, 1, 1)
        if beta is not None:
            y = y + beta.view(1, C, 1, 1)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        import torch.distributed as dist
        xhat, rstd, gamma = ctx.saved_tensors
        C = xhat.shape[1]
        sg, sgx = g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))
        dgamma = sgx.clone() if gamma is not None and ctx.needs_input_grad[1] else None
        dbeta = sg.clone() if ctx.needs_input_grad[2] else None
        gx = None
        if ctx.needs_input_grad[0]:
            packed = torch.cat((sg, sgx)) if gamma is None else torch.cat((sg * gamma, sgx * gamma))
            dist.all_reduce(packed, group=ctx.group)
            gs = g if gamma is None else g * gamma.view(1, C, 1, 1)
            gx = (gs - (packed[:C] / ctx.n).view(1, C, 1, 1) - xhat * (packed[C:2 * C] / ctx.n).view(1, C, 1, 1)) * rstd.view(1, C, 1, 1)
        return gx, dgamma, dbeta, None, None, None, None, None, None


class MappingFn(Function):
    """Label embedding + cat + mapping network (generator.py:80-85 with Mapping_Net :22-37) as ONE autograd node over the
    library's own kernels: ``mlp`` launches forward (kg_linear_fwd; the embedding lookup and the cat are the first
    layer's operand load), ``mlp + 1`` backward (kg_linear_bwd per layer: LeakyReLU derivative, input gradient, weight
    and bias gradient in one launch, added straight into the flat-bucket slices in a first-order pass; kg_embed_bwd).
    Stock ops took 57 launches / 0.25 ms per iteration for this.
    apply(z, labels, emb_weight, slope, w0, b0, w1, b1, ...)."""

    @staticmethod
    def forward(ctx, z, labels, emb, slope, *wb):
        ctx.set_materialize_grads(False)
        ws, bs = wb[0::2], wb[1::2]
        ys = []
        y = nv.linear_fwd(z, ws[0], bs[0], nv.ACT_LRELU, slope, emb=emb, labels=labels)
        ys.append(y)
        for w, b in zip(ws[1:], bs[1:]):
            y = nv.linear_fwd(y, w, b, nv.ACT_LRELU, slope)
            ys.append(y)
        ctx.slope, ctx.nl = slope, len(ws)
        ctx.sinks = [(_sink_of(w), _sink_of(b)) for w, b in zip(ws, bs)] + [(_sink_of(emb), None)]
        ctx.save_for_backward(z, labels, emb, *ws, *ys)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        nl = ctx.nl
        if g is None:
            return (None,) * (4 + 2 * nl)
        z, labels, emb = ctx.saved_tensors[:3]
        ws = ctx.saved_tensors[3:3 + nl]
        ys = ctx.saved_tensors[3 + nl:]
        want_params = not _SKIP_PARAM_GRADS
        direct = want_params and _direct_param_grads()
        J = emb.shape[1]
        need_z = ctx.needs_input_grad[0]
        need_e = want_params and ctx.needs_input_grad[2]
        grads = [None] * (2 * nl)
        gz = gemb = None
        g = g.contiguous()
        for l in range(nl - 1, -1, -1):
            w, first = ws[l], l == 0
            x = z if first else ys[l - 1]
            sw, sb = ctx.sinks[l]
            need_w = want_params and ctx.needs_input_grad[4 + 2 * l]
            need_b = want_params and ctx.needs_input_grad[5 + 2 * l]
            dw = db = None
            acc = False
            if direct and (not need_w or sw is not None) and (not need_b or sb is not None):
                dw, db, acc = (sw if need_w else None), (sb if need_b else None), True      # straight into the bucket
            else:
                dw = torch.empty_like(w) if need_w else None
                db = torch.empty(w.shape[0], dtype=w.dtype, device=w.device) if need_b else None
                grads[2 * l], grads[2 * l + 1] = dw, db
            cols = None if not first else (w.shape[1] if need_z else (J if need_e else 0))
            if first and cols == 0 and dw is None and db is None:
                break
            gx = nv.linear_bwd(g, ys[l], x, w, nv.ACT_LRELU, ctx.slope, emb=emb if first else None,
                               labels=labels if first else None, gx_cols=cols, dw=dw, db=db, accumulate=acc)
            if not first:
                g = gx
                continue
            if need_e:
                se = ctx.sinks[nl][0]
                if direct and se is not None:
                    nv.embed_bwd(gx, labels, se.view(emb.shape), accumulate=True)
                else:
                    gemb = torch.empty_like(emb)
                    nv.embed_bwd(gx, labels, gemb, accumulate=False)
            if need_z:
                gz = gx[:, J:]
        return (gz, None, gemb, None) + tuple(grads)


_OUTER_PENDING: list = []       # adjacency-gradient problems recorded by AggReduce.backward(lazy_outer=True)


def flush_outer():
    """Compute the recorded adjacency gradients (one launch + one for the slab sums)."""
    if _OUTER_PENDING:
        nv.agg_outer_finish(_OUTER_PENDING)


def pair_apply(F, xf, xb, *rest):
    """``F`` on a batch ``xf`` whose LAST ``len(xb)`` samples are differentiated: one launch over the whole batch
    without autograd, then the autograd node of the differentiated samples alone (``xb`` carries their history and
    holds the same values as ``xf``'s tail), whose backward then works on that sample range only.
    Returns (result for the whole batch, its differentiated tail)."""
    with torch.no_grad():
        of = F.apply(xf, *rest)
    ob = F.apply(xb, *rest, of[of.shape[0] - xb.shape[0]:])
    return of, ob


class AggOuter(Function):
    """dA[k,v,w] = sum_{c,n,t'} x[c,(n,t'/rep,v)] y[k*C+c,(n,t',w)]."""

    @staticmethod
    def forward(ctx, x, y, K: int, rep: int):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.K, ctx.rep = K, rep
        ctx.save_for_backward(x, y)
        return nv.agg_outer(x, y, K, rep)

    @staticmethod
    def backward(ctx, gA):
        if gA is None:
            return (None, None, None, None)
        x, y = ctx.saved_tensors
        gx = AggReduce.apply(y, _t12(gA), ctx.rep) if ctx.needs_input_grad[0] else None
        gy = AggExpand.apply(x, gA, ctx.rep) if ctx.needs_input_grad[1] else None
        return gx, gy, None, None


# ---- pointwise / reductions ---------------------------------------------------------------------------------

class ActBwd(Function):
    """g * act'(out) with the derivative expressed on the activation output `ref`."""

    @staticmethod
    def forward(ctx, g, ref, act: int):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.act = act
        ctx.save_for_backward(ref)
        return nv.act_bwd(g, ref, act)

    @staticmethod
    def backward(ctx, gg):
        if gg is None:
            return (None, None, None)
        (ref,) = ctx.saved_tensors
        if ctx.act == ACT_TANH:
            raise NotImplementedError("second derivative through tanh is not on the hot path "
                                      "(the generator is only differentiated once)")
        return ActBwd.apply(gg, ref, ctx.act), None, None


class RowSum(Function):
    """(N,C,T,V) -> (C,) sum over n,t,v."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.shape = tuple(x.shape)
        return nv.rowsum(x)[0]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,)
        return g.view(1, -1, 1, 1).expand(ctx.shape)


class GradPenalty(Function):
    """gp = mean_n (|g_n|_2 - 1)^2 (kinetic-gan.py:112-113) in two launches, its backward in one."""

    @staticmethod
    def forward(ctx, g):
        nrm, gp = nv.gp_fwd(g)
        ctx.save_for_backward(g, nrm)
        return gp

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        g, nrm = ctx.saved_tensors
        return nv.gp_bwd(g, nrm, gout)


class CriticLoss(Function):
    """d_loss = -E[D(real)] + E[D(fake)] + lambda * gp (kinetic-gan.py:152) for the stacked validities `both`
    (2n, 1) = [real; fake]: one dot product with a constant +-1/n vector."""

    @staticmethod
    def forward(ctx, both, gp, wvec, lam: float):
        ctx.lam = lam
        ctx.save_for_backward(wvec)
        ctx.shape = both.shape
        return torch.add(torch.dot(both.reshape(-1), wvec), gp, alpha=lam)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (wvec,) = ctx.saved_tensors
        if getattr(gout, "_kg_unit_seed", False):
            # d_loss.backward() seeded with the cached constant 1 (Trainer.d_compute): both gradients are constants too - the
            # weight vector itself and a cached lambda (READ-ONLY, like the seed) - instead of two multiply launches
            key = (float(ctx.lam), str(gout.device), gout.dtype)
            lam = _LAM_CONSTS.get(key)
            if lam is None and not (gout.is_cuda and torch.cuda.is_current_stream_capturing()):
                lam = _LAM_CONSTS[key] = torch.full((), float(ctx.lam), dtype=gout.dtype, device=gout.device)
            if lam is not None:
                return wvec.view(ctx.shape), lam, None, None
        return (wvec * gout).view(ctx.shape), gout * ctx.lam, None, None


_LAM_CONSTS: dict = {}


def time_scatter(g, stride: int, T_in: int):
    """Adjoint of x[:, :, ::stride][:, :, :T_out] (the identity-residual path of a strided block)."""
    if stride == 1:
        return g
    n, c, t, v = g.shape
    out = nv.new_plane(n, c, T_in, v, g.device, zero=True)
    out[:, :, 0:t * stride:stride] = g
    return out


# ---- fused discriminator-block tail -----------------------------------------------------------------------------

class DiscTail(Function):
    """out = LeakyReLU_0.2( tcn(z; wt, bt) + residual(x) ) evaluated only at the frames the block keeps.

    residual: 'conv' (1x1 conv wr, br on x at kept vertices), 'identity' (x itself) or 'none'
    (discriminator.py:108-120,128-136).  One kg_conv launch with two K-slice groups.
    """

    @staticmethod
    def forward(ctx, z, x, wt, bt, wr, br, spec_t: ConvSpec, spec_r: Optional[ConvSpec], res: str):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        ctx.spec_t, ctx.spec_r, ctx.res = spec_t, spec_r, res
        ctx.sinks = (_sink_of(wt), _sink_of(bt), _sink_of(wr) if res == "conv" else None,
                     _sink_of(br) if res == "conv" else None)
        groups = [Group(z, wt, spec_t.wv, spec_t.Cin, spec_t.taps, TAP_TIME, spec_t.t_stride, False, None)]
        add = None
        if res == "conv":
            groups.append(Group(x, wr, spec_r.wv, spec_r.Cin, 1, TAP_TIME, spec_r.t_stride, False, spec_r.vmap))
        elif res == "identity":
            add = x
        out = nv.conv(groups, z.shape[0], spec_t.M, spec_t.T_out, spec_t.V_out,
                      bias0=bt, bias1=br if res == "conv" else None,
                      add=add, add_tstride=spec_t.t_stride, act=ACT_LRELU, slope=0.2)
        ctx.save_for_backward(z, x, wt, wr, out)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None, None, None, None, None, None, None, None)
        z, x, wt, wr, out = ctx.saved_tensors
        st, sr, res = ctx.spec_t, ctx.spec_r, ctx.res
        need = ctx.needs_input_grad
        gm = ActBwd.apply(g, out, ACT_LRELU)
        gz = ConvT.apply(gm, wt, st) if need[0] else None
        gx = gwt = gbt = gwr = gbr = None
        params = not _SKIP_PARAM_GRADS
        direct = _direct_param_grads()
        s_wt, s_bt, s_wr, s_br = ctx.sinks
        if params and need[2]:
            if direct and s_wt is not None:
                _wgrad_into(s_wt, z, gm, st)
            else:
                gwt = WGrad.apply(z, gm, st)
        want_bt, want_br = need[3], res == "conv" and need[5]
        if params and (want_bt or want_br):
            sunk = [v for v, want in ((s_bt, want_bt), (s_br, want_br)) if want and direct and v is not None]
            if sunk:
                _rowsum_into(sunk, gm)
            if (want_bt and not (direct and s_bt is not None)) or (want_br and not (direct and s_br is not None)):
                gb = RowSum.apply(gm)
                gbt = gb if want_bt and not (direct and s_bt is not None) else None
                gbr = gb if want_br and not (direct and s_br is not None) else None
        if res == "conv":
            if need[1]:
                gx = ConvT.apply(gm, wr, sr)
            if params and need[4]:
                if direct and s_wr is not None:
                    _wgrad_into(s_wr, x, gm, sr)
                else:
                    gwr = WGrad.apply(x, gm, sr)
        elif res == "identity" and need[1]:
            gx = time_scatter(gm, st.t_stride, x.shape[2])
        return gz, gx, gwt, gbt, gwr, gbr, None, None, None


# ---- fused generator-block tail -------------------------------------------------------------------------------------

def _bn_coeffs(u, gamma, beta, rm, rv, nbt, training, momentum, eps):
    """Per-channel (scale, shift, mean, rstd) of BatchNorm2d on u - one kg_bn_fwd launch, which also updates the
    running statistics in training mode exactly as torch does (biased variance to normalise, unbiased to track)."""
    # momentum=None (torch: cumulative moving average, factor 1 / num_batches_tracked): the factor is formed ON THE
    # DEVICE from the live counter (kg_bn_fwd, momentum < 0) - no .item() sync, hipGraph-capturable
    cma = training and rm is not None and momentum is None
    coef = nv.bn_fwd(u, gamma, beta, rm, rv, nbt, training, -1.0 if cma else (0.0 if momentum is None else momentum), eps)
    return coef[0], coef[1], coef[2], coef[3]


class GenTail(Function):
    """out = act( BN_t(u) + BN_r(r) + nw * noise )     (generator.py:142,160,176,179-182)

    u: tcn conv output (bias already added); r: residual branch input to its BN (or identity
    residual, or None); bn_t / bn_r: (running_mean, running_var, num_batches_tracked, training,
    momentum, eps) or None; gt/bt_/gr/br_ the affine terms.  One kg_affine_act launch after the statistics; first-order backward.

    ``groups`` = 2: the batch holds TWO independent forward passes back to back (the two generator syntheses of a
    WGAN-GP iteration, wgan_gp.Trainer): BatchNorm statistics, running-statistics updates and the normalisation are
    taken per half, in order, exactly as two separate forwards would; only the SECOND half is differentiated (the
    first is the critic step's no-grad sample).  ``u`` / ``r`` are then the whole batch WITHOUT history and ``u_b`` /
    ``r_b`` the second half's tensors with it (same values as the tail of u / r, ``pair_apply``); the result is
    (whole batch without history, its second half with history), and the backward pass works on that half alone.
    """

    @staticmethod
    def forward(ctx, u, r, noise, nw, gt, bt_, gr, br_, bn_t, bn_r, act: int, groups: int = 1, u_b=None, r_b=None):
        ctx.set_materialize_grads(False)      # an absent gradient arrives as None, not as a zero tensor
        n = u.shape[0]
        h = n // groups
        if groups > 1 and u_b is None:
            raise ValueError("GenTail: groups > 1 needs the differentiated half's tensors (u_b, r_b)")
        use_t = bn_t is not None
        use_r = r is not None and bn_r is not None
        out = nv.new_plane(*u.shape, u.device) if groups > 1 else None
        coef_t = coef_r = None
        if groups > 1 and (use_t or use_r) and all(b is None or (b[3] and b[4] is not None) for b in (bn_t, bn_r if use_r else None)):
            # training mode, plain momentum: the statistics of both layers and all stacked batches in ONE launch
            jobs = []
            for xin, gam, bet, b in ((u, gt, bt_, bn_t if use_t else None), (r, gr, br_, bn_r if use_r else None)):
                if b is not None:
                    jobs.append(dict(x=xin, gamma=gam, beta=bet, running_mean=b[0], running_var=b[1],
                                     num_batches_tracked=b[2], momentum=b[4], eps=b[5], groups=groups))
            res = nv.bn_fwd_many(jobs)
            coef_t = res[0] if use_t else None
            coef_r = res[-1] if use_r else None
        sx = bx = sr = br = None
        mt = rt = mr = rr = None
        one_launch = groups > 1 and (coef_t is not None or not use_t) and (coef_r is not None or not use_r)
        if one_launch:
            # both batches in ONE launch: batch q reads its coefficients q * 4C floats behind the first batch's
            C_ = u.shape[1]
            if use_t:
                sx, bx, mt, rt = coef_t[groups - 1]
            if use_r:
                sr, br, mr, rr = coef_r[groups - 1]
            nv.affine_act(u, coef_t[0, 0] if use_t else None, coef_t[0, 1] if use_t else None, r,
                          coef_r[0, 0] if use_r else None, coef_r[0, 1] if use_r else None, noise, nw.reshape(-1), act, 0.2,
                          out=out, groups=groups if (use_t or use_r) else 1, coef_gs=4 * C_)
        for gi in range(0 if one_launch else groups):
            sl = slice(gi * h, (gi + 1) * h)
            ug = u[sl] if groups > 1 else u
            rg = (r[sl] if groups > 1 else r) if r is not None else None
            ng = (noise[sl] if groups > 1 else noise) if noise is not None else None
            sx = bx = sr = br = None
            mt = rt = mr = rr = None
            if use_t:
                sx, bx, mt, rt = coef_t[gi] if coef_t is not None else _bn_coeffs(ug, gt, bt_, *bn_t)
            if use_r:
                sr, br, mr, rr = coef_r[gi] if coef_r is not None else _bn_coeffs(rg, gr, br_, *bn_r)
            if groups > 1:
                nv.affine_act(ug, sx, bx, rg, sr, br, ng, nw.reshape(-1), act, 0.2, out=out[sl])
            else:
                out = nv.affine_act(ug, sx, bx, rg, sr, br, ng, nw.reshape(-1), act, 0.2)
        ctx.act, ctx.groups = act, groups
        ctx.nw_sink = _sink_of(nw)
        ctx.train_t = bn_t is not None and bool(bn_t[3])
        ctx.train_r = bn_r is not None and r is not None and bool(bn_r[3])
        ctx.has = (bn_t is not None, r is not None, bn_r is not None and r is not None)
        ctx.save_for_backward(u, r, noise, out, gt, gr, mt, rt, mr, rr, sx, sr)      # statistics: the LAST group's
        if groups > 1:
            ctx.mark_non_differentiable(out)
            return out, out[(groups - 1) * h:]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g, g_b=None):
        groups = ctx.groups
        if groups > 1:
            g = g_b
        if g is None:
            return (None,) * 14
        u, r, noise, out, gt, gr, mt, rt, mr, rr, sx, sr = ctx.saved_tensors
        has_bn_t, has_r, has_bn_r = ctx.has
        if groups > 1:          # only the last group carries a gradient
            h = u.shape[0] // groups
            sl = slice((groups - 1) * h, groups * h)
            u, out = u[sl], out[sl]
            r = r[sl] if r is not None else None
            noise = noise[sl] if noise is not None else None
        gpre = nv.act_bwd(g, out, ctx.act)
        g_nw = None
        if not _SKIP_PARAM_GRADS and ctx.needs_input_grad[3]:
            if ctx.nw_sink is not None:
                _rowsum_into([ctx.nw_sink], gpre, noise)        # joins the pass's other per-channel sums
            else:
                g_nw = nv.rowsum(gpre, noise, 2).view(1, -1, 1, 1)

        # the backward sums of both BatchNorm layers in ONE launch with many workgroups per channel (kg_bn_bwd_many)
        jobs = []
        if has_bn_t:
            jobs.append(dict(g=gpre, x=u, gamma=gt, mean=mt, rstd=rt, training=ctx.train_t))
        if has_r and has_bn_r:
            jobs.append(dict(g=gpre, x=r, gamma=gr, mean=mr, rstd=rr, training=ctx.train_r))
        ks = nv.bn_bwd_many(jobs) if jobs else []

        def bn_dx(k, xin, training):      # [a, b, c, dgamma, dbeta]: dL/dx = a g + b x + c
            if not training:   # eval-mode BN is a fixed per-channel affine map
                return nv.affine_act(gpre, k[0])
            return nv.affine_act(gpre, k[0], k[2], xin, k[1])

        if has_bn_t:
            k = ks[0]
            du, dgt, dbt = bn_dx(k, u, ctx.train_t), k[3], k[4]
        else:
            du, dgt, dbt = gpre, None, None
        dr = dgr = dbr = None
        if has_r:
            if has_bn_r:
                k = ks[-1]
                dr, dgr, dbr = bn_dx(k, r, ctx.train_r), k[3], k[4]
            else:
                dr = gpre
        if groups > 1:
            return None, None, None, g_nw, dgt, dbt, dgr, dbr, None, None, None, None, du, dr
        return du, dr, None, g_nw, dgt, dbt, dgr, dbr, None, None, None, None, None, None
