"""ConvTemporalGraphical on the HIP path - same constructor / forward surface and state_dict keys
as the reference's models/init_gan/tgcn.py:36-68 (``conv.weight`` of shape (K*C_out, C_in, kt, 1)).

forward(x, A): y = conv(x) on the fp32 matrix cores (kg_conv), then
out[n,c,t,w] = sum_{k,v} y[n,k*C_out+c,t,v] A[k,v,w] (kg_agg_reduce, A staged in LDS).
The (N, K*C_out, T, V) intermediate is produced channel-major and never re-laid-out
(the reference's permute-copy + bmm + .contiguous() disappear).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._native import TAP_TIME, WView


class ConvTemporalGraphical(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1,
                 t_padding=0, t_dilation=1, bias=False):
        super().__init__()
        if t_dilation != 1 or t_kernel_size not in (1, 3) or t_padding != (t_kernel_size - 1) // 2:
            raise NotImplementedError("HIP ConvTemporalGraphical supports t_kernel_size 1 or 3 with 'same' "
                                      "padding and no dilation (all the reference ever instantiates, "
                                      "generator.py:129 / discriminator.py:96)")
        self.kernel_size = kernel_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.t_kernel_size, self.t_stride = t_kernel_size, t_stride
        # parameter container with the reference's key names / shapes / default init
        self.conv = nn.Conv2d(in_channels, out_channels * kernel_size, kernel_size=(t_kernel_size, 1),
                              padding=(t_padding, 0), stride=(t_stride, 1), dilation=(t_dilation, 1), bias=bias)
        self._specs = {}
        # set by the owning block when the level has ONE vertex and only the first partition of its adjacency is
        # non-zero (A = [[1], [0], [0]] at the coarsest level): the other two partitions' rows of the conv output are
        # multiplied by 0 in the aggregation, so only the first C_out rows are computed
        self.single_partition = False

    def spec1(self, T):
        sp = self._specs.get(("single", T))
        if sp is None:
            cin, co = self.in_channels, self.out_channels
            sp = ops.ConvSpec(M=co, Cin=cin, taps=1, tap_mode=TAP_TIME, t_stride=1, T_in=T, V_in=1, T_out=T, V_out=1,
                              wv=WView(sT=0, sO=cin, sI=1), w_shape=(co * self.kernel_size, cin, 1, 1))
            self._specs[("single", T)] = sp
        return sp

    def spec(self, T, V):
        key = (T, V)
        sp = self._specs.get(key)
        if sp is None:
            kt, cin, m = self.t_kernel_size, self.in_channels, self.out_channels * self.kernel_size
            t_out = (T + 2 * ((kt - 1) // 2) - kt) // self.t_stride + 1
            sp = ops.ConvSpec(M=m, Cin=cin, taps=kt, tap_mode=TAP_TIME, t_stride=self.t_stride,
                              T_in=T, V_in=V, T_out=t_out, V_out=V,
                              wv=WView(sT=1, sO=cin * kt, sI=kt), w_shape=(m, cin, kt, 1))
            self._specs[key] = sp
        return sp

    def forward(self, x, A, x_b=None):
        """``x_b``: ``x`` is a batch without history whose last ``len(x_b)`` samples are differentiated through
        ``x_b`` (``ops.pair_apply``); the result is then the pair (whole batch, differentiated tail)."""
        assert A.size(0) == self.kernel_size
        if self.single_partition and x.shape[3] == 1 and self.t_kernel_size == 1 and self.conv.bias is None:
            w, b, sp, Ak = self.conv.weight, None, self.spec1(x.shape[2]), A[:1]
        else:
            w, b, sp, Ak = self.conv.weight, self.conv.bias, self.spec(x.shape[2], x.shape[3]), A
        # set by the owner whose adjacency pack flushes (Generator).  Not for a sliced adjacency: autograd's slice
        # backward copies the gradient into a zero tensor right away, before the pack's backward computed it
        lazy = bool(getattr(self, "lazy_outer", False)) and Ak is A
        if x_b is not None:
            y, y_b = ops.pair_apply(ops.Conv, x, x_b, w, b, sp)
            with torch.no_grad():
                of = ops.AggReduce.apply(y, Ak, 1)
            return (of, ops.AggReduce.apply(y_b, Ak, 1, of[of.shape[0] - y_b.shape[0]:], lazy)), A
        y = ops.Conv.apply(x, w, b, sp)
        return ops.AggReduce.apply(y, Ak, 1, None, lazy), A
