"""ctypes binding of libkgan_hip.so (C ABI in include/kgan_hip.h) for torch tensors.

Every function here takes torch CUDA tensors, borrows their device pointers, enqueues HIP kernels
on torch's current stream and returns torch tensors allocated by torch's caching allocator.  There
is deliberately NO fallback: if the library is missing or a tensor is not on the GPU the call
raises.  (tests/ swap these functions for torch emulations to exercise the autograd composition
on a CPU-only box; the product never does.)

"Plane tensor" = logical (N, C, T, V) tensor whose (t, v) plane is contiguous; see kgan_hip.h.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional, Sequence

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# KG_LIB: an experiment build of the library (tools/build_variant.sh, A/B timing); the default is the in-tree build
LIB_PATH = os.environ.get("KG_LIB") or os.path.join(_PKG, "libkgan_hip.so")

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
ABI_VERSION = 9
TAP_TIME, TAP_CHANBLOCK = 0, 1

c_f32p = C.c_void_p
c_i32p = C.c_void_p


class _ConvGroup(C.Structure):
    _fields_ = [("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64),
                ("Cin", C.c_int32), ("T_in", C.c_int32), ("V_in", C.c_int32),
                ("x_lead", C.c_int32),
                ("vmap", c_i32p),
                ("w", c_f32p), ("w_sT", C.c_int64), ("w_sO", C.c_int64), ("w_sI", C.c_int64),
                ("w_sMB", C.c_int64), ("w_MB", C.c_int32),
                ("taps", C.c_int32), ("tap_mode", C.c_int32), ("t_stride", C.c_int32),
                ("transposed", C.c_int32)]


class _ConvArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("T_out", C.c_int32), ("V_out", C.c_int32),
                ("out", c_f32p), ("o_sN", C.c_int64), ("o_sC", C.c_int64),
                ("ngroups", C.c_int32),
                ("g", _ConvGroup * 2),
                ("bias0", c_f32p), ("bias1", c_f32p),
                ("add", c_f32p), ("a_sN", C.c_int64), ("a_sC", C.c_int64), ("a_tstride", C.c_int32),
                ("act", C.c_int32), ("slope", C.c_float),
                ("ws", c_f32p), ("ws_bytes", C.c_int64),
                ("mask", c_f32p), ("m_sN", C.c_int64), ("m_sC", C.c_int64),
                ("sync", C.c_void_p), ("sync_len", C.c_int32),
                ("o_tstride", C.c_int32),
                ("wpack", C.c_void_p), ("wpack_bytes", C.c_int64)]


class _WgradPair(C.Structure):
    _fields_ = [("N", C.c_int32),
                ("g", c_f32p), ("g_sN", C.c_int64), ("g_sC", C.c_int64),
                ("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64)]


class _WgradArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("T_out", C.c_int32), ("V_out", C.c_int32),
                ("g", c_f32p), ("g_sN", C.c_int64), ("g_sC", C.c_int64),
                ("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64),
                ("Cin", C.c_int32), ("T_in", C.c_int32), ("V_in", C.c_int32),
                ("vmap", c_i32p),
                ("taps", C.c_int32), ("tap_mode", C.c_int32), ("t_stride", C.c_int32),
                ("dw", c_f32p), ("w_sT", C.c_int64), ("w_sO", C.c_int64), ("w_sI", C.c_int64),
                ("ws", c_f32p), ("ws_bytes", C.c_int64), ("accumulate", C.c_int32),
                ("nextra", C.c_int32), ("extra", _WgradPair * 2), ("defer_reduce", C.c_int32)]


class _WgradReduceJob(C.Structure):
    _fields_ = [("ws", c_f32p), ("dw", c_f32p), ("w_sT", C.c_int64), ("w_sO", C.c_int64), ("w_sI", C.c_int64),
                ("taps", C.c_int32), ("M", C.c_int32), ("Cin", C.c_int32), ("splits", C.c_int32),
                ("accumulate", C.c_int32)]


WGRAD_REDUCE_MAX_JOBS = 24


class _WgradReduceJobs(C.Structure):
    _fields_ = [("njobs", C.c_int32), ("job", _WgradReduceJob * WGRAD_REDUCE_MAX_JOBS)]


class _AggArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("V", C.c_int32), ("W", C.c_int32),
                ("T", C.c_int32), ("rep", C.c_int32),
                ("a", c_f32p),
                ("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64),
                ("y", c_f32p), ("y_sN", C.c_int64), ("y_sC", C.c_int64),
                ("out", c_f32p), ("o_sN", C.c_int64), ("o_sC", C.c_int64),
                ("ws", c_f32p), ("ws_bytes", C.c_int64), ("a_transposed", C.c_int32), ("defer_sum", C.c_int32),
                ("res", c_f32p), ("r_sN", C.c_int64), ("r_sC", C.c_int64), ("r_T", C.c_int32), ("r_V", C.c_int32),
                ("r_tstride", C.c_int32), ("r_inv", c_i32p),
                ("mask", c_f32p), ("m_sN", C.c_int64), ("m_sC", C.c_int64), ("slope", C.c_float)]


class _OuterSumJob(C.Structure):
    _fields_ = [("ws", c_f32p), ("out", c_f32p), ("nout", C.c_int32), ("slabs", C.c_int32)]


OUTER_SUM_MAX_JOBS = 16


class _OuterSumJobs(C.Structure):
    _fields_ = [("njobs", C.c_int32), ("job", _OuterSumJob * OUTER_SUM_MAX_JOBS)]


class _AggConvArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("Cin", C.c_int32), ("M", C.c_int32), ("T", C.c_int32), ("V", C.c_int32),
                ("W", C.c_int32), ("K", C.c_int32),
                ("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64),
                ("a", c_f32p), ("a_transposed", C.c_int32),
                ("nbr", c_i32p), ("pcount", C.c_int32 * 3),
                ("w", c_f32p), ("w_sT", C.c_int64), ("w_sO", C.c_int64), ("w_sI", C.c_int64),
                ("out", c_f32p), ("o_sN", C.c_int64), ("o_sC", C.c_int64),
                ("add", c_f32p), ("a_sN", C.c_int64), ("a_sC", C.c_int64), ("a_tstride", C.c_int32),
                ("xa", c_f32p), ("xa_sN", C.c_int64), ("xa_sC", C.c_int64)]


class _RowsumArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("V", C.c_int32),
                ("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64),
                ("y", c_f32p), ("y_sN", C.c_int64), ("y_sC", C.c_int64),
                ("shift", c_f32p),
                ("want_second", C.c_int32),
                ("out", c_f32p),
                ("ws", c_f32p), ("ws_bytes", C.c_int64), ("accumulate", C.c_int32), ("out2", c_f32p)]


class _BnArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("V", C.c_int32),
                ("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64),
                ("g", c_f32p), ("g_sN", C.c_int64), ("g_sC", C.c_int64),
                ("gamma", c_f32p), ("beta", c_f32p),
                ("running_mean", c_f32p), ("running_var", c_f32p),
                ("num_batches_tracked", C.c_void_p),
                ("mean", c_f32p), ("rstd", c_f32p),
                ("momentum", C.c_float), ("eps", C.c_float),
                ("training", C.c_int32),
                ("coef", c_f32p)]


class _BnJob(C.Structure):
    _fields_ = [("a", _BnArgs), ("groups", C.c_int32)]


class _GpArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("V", C.c_int32),
                ("g", c_f32p), ("g_sN", C.c_int64), ("g_sC", C.c_int64),
                ("nrm", c_f32p), ("gp", c_f32p), ("gout", c_f32p),
                ("out", c_f32p), ("o_sN", C.c_int64), ("o_sC", C.c_int64)]


class _EltArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("V", C.c_int32),
                ("x", c_f32p), ("x_sN", C.c_int64), ("x_sC", C.c_int64),
                ("r", c_f32p), ("r_sN", C.c_int64), ("r_sC", C.c_int64),
                ("noise", c_f32p),
                ("sx", c_f32p), ("bx", c_f32p), ("sr", c_f32p), ("br", c_f32p), ("nw", c_f32p),
                ("out", c_f32p), ("o_sN", C.c_int64), ("o_sC", C.c_int64),
                ("act", C.c_int32), ("slope", C.c_float),
                ("groups", C.c_int32), ("coef_gs", C.c_int64)]


class _GenArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("Cr", C.c_int32), ("Tc", C.c_int32),
                ("Vc", C.c_int32), ("V", C.c_int32), ("rep", C.c_int32),
                ("a", c_f32p), ("u", c_f32p), ("b", c_f32p),
                ("y", c_f32p), ("y_out", c_f32p), ("y_sN", C.c_int64), ("y_sC", C.c_int64),
                ("z", c_f32p), ("z_sN", C.c_int64), ("z_sC", C.c_int64),
                ("zf", c_f32p), ("zf_sN", C.c_int64), ("zf_sC", C.c_int64),
                ("rs", c_f32p), ("rs_out", c_f32p), ("rs_sN", C.c_int64), ("rs_sC", C.c_int64),
                ("rbias", c_f32p),
                ("r", c_f32p), ("r_sN", C.c_int64), ("r_sC", C.c_int64)]


class _GenAdjJob(C.Structure):
    _fields_ = [("dbt", c_f32p), ("u", c_f32p), ("a", c_f32p), ("out", c_f32p),
                ("K", C.c_int32), ("Kd", C.c_int32), ("V", C.c_int32), ("Vc", C.c_int32), ("accumulate", C.c_int32)]


class _HeadArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("V", C.c_int32),
                ("h", c_f32p), ("h_sN", C.c_int64), ("h_sC", C.c_int64),
                ("w", c_f32p), ("b", c_f32p), ("v", c_f32p), ("gv", c_f32p),
                ("g", c_f32p), ("g_sN", C.c_int64), ("g_sC", C.c_int64),
                ("slope", C.c_float), ("masked", C.c_int32),
                ("dw", c_f32p), ("db", c_f32p), ("accumulate", C.c_int32)]


class _LinearArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("Din", C.c_int32), ("Dout", C.c_int32), ("L", C.c_int32), ("J", C.c_int32),
                ("x", c_f32p), ("x_ld", C.c_int64), ("emb", c_f32p), ("labels", C.c_void_p),
                ("w", c_f32p), ("bias", c_f32p), ("y", c_f32p), ("y_ld", C.c_int64),
                ("act", C.c_int32), ("slope", C.c_float),
                ("g", c_f32p), ("g_ld", C.c_int64),
                ("gx", c_f32p), ("gx_ld", C.c_int64), ("gx_cols", C.c_int32),
                ("dw", c_f32p), ("db", c_f32p), ("demb", c_f32p), ("accumulate", C.c_int32)]


class _LabelBiasArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("L", C.c_int32), ("J", C.c_int32), ("K", C.c_int32), ("C", C.c_int32),
                ("V", C.c_int32), ("W", C.c_int32), ("T", C.c_int32),
                ("labels", C.c_void_p), ("emb", c_f32p),
                ("w", c_f32p), ("w_sK", C.c_int64), ("w_sC", C.c_int64),
                ("ak", c_f32p), ("zl", c_f32p),
                ("gz", c_f32p), ("gz_sN", C.c_int64), ("gz_sC", C.c_int64),
                ("demb", c_f32p), ("dw", c_f32p), ("accumulate", C.c_int32),
                ("dak", c_f32p), ("dak_accumulate", C.c_int32),
                ("ws", c_f32p), ("ws_bytes", C.c_int64)]


class _MixArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("V", C.c_int32),
                ("real", c_f32p), ("r_sN", C.c_int64), ("r_sC", C.c_int64),
                ("fake", c_f32p), ("f_sN", C.c_int64), ("f_sC", C.c_int64),
                ("alpha", c_f32p),
                ("out", c_f32p), ("o_sN", C.c_int64), ("o_sC", C.c_int64)]


class _MaskedAdjArgs(C.Structure):
    _fields_ = [("n", C.c_int32), ("a", c_f32p), ("imp", c_f32p), ("sel", C.c_void_p), ("ak", c_f32p),
                ("g", c_f32p), ("dimp", c_f32p), ("accumulate", C.c_int32)]


class _GenTailArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("V", C.c_int32), ("act", C.c_int32), ("slope", C.c_float),
                ("g", c_f32p), ("g_sN", C.c_int64), ("g_sC", C.c_int64),
                ("out", c_f32p), ("o_sN", C.c_int64), ("o_sC", C.c_int64),
                ("u", c_f32p), ("u_sN", C.c_int64), ("u_sC", C.c_int64), ("mean_t", c_f32p), ("rstd_t", c_f32p), ("gamma_t", c_f32p),
                ("r", c_f32p), ("r_sN", C.c_int64), ("r_sC", C.c_int64), ("mean_r", c_f32p), ("rstd_r", c_f32p), ("gamma_r", c_f32p),
                ("noise", c_f32p), ("coef", c_f32p),
                ("dgamma_t", c_f32p), ("dbeta_t", c_f32p), ("dgamma_r", c_f32p), ("dbeta_r", c_f32p), ("dnw", c_f32p),
                ("ws", c_f32p), ("ws_bytes", C.c_int64), ("counters", C.c_void_p), ("counters_len", C.c_int32),
                ("du", c_f32p), ("du_sN", C.c_int64), ("du_sC", C.c_int64),
                ("dr", c_f32p), ("dr_sN", C.c_int64), ("dr_sC", C.c_int64)]


class _GenPrepJob(C.Structure):
    _fields_ = [("a", c_f32p), ("imp", c_f32p), ("u", c_f32p), ("aeff", c_f32p), ("b", c_f32p),
                ("K", C.c_int32), ("V", C.c_int32), ("Vc", C.c_int32)]


class _Plane(C.Structure):
    _fields_ = [("p", c_f32p), ("sN", C.c_int64), ("sC", C.c_int64)]


class _GenBnLayer(C.Structure):
    _fields_ = [("gamma", c_f32p), ("beta", c_f32p), ("running_mean", c_f32p), ("running_var", c_f32p),
                ("num_batches_tracked", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float), ("coef", c_f32p)]


class _GenBlockArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("groups", C.c_int32),
                ("Cin", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("Kp", C.c_int32),
                ("Tc", C.c_int32), ("Vc", C.c_int32), ("T", C.c_int32), ("V", C.c_int32), ("rep", C.c_int32),
                ("res_kind", C.c_int32), ("bn_t", C.c_int32), ("act", C.c_int32), ("slope", C.c_float),
                ("x", _Plane),
                ("pu", _Plane), ("pr", _Plane), ("pcoef_t", c_f32p), ("pcoef_r", c_f32p), ("pnoise", c_f32p), ("pnw", c_f32p),
                ("pact", C.c_int32), ("xout", _Plane),
                ("wg", c_f32p), ("wr", c_f32p), ("br", c_f32p), ("wt", c_f32p), ("bt", c_f32p),
                ("b", c_f32p), ("u", c_f32p),
                ("yc", _Plane), ("z", _Plane), ("r", _Plane), ("uo", _Plane),
                ("bt_", _GenBnLayer), ("br_", _GenBnLayer),
                ("noise", c_f32p), ("nw", c_f32p), ("out", _Plane),
                ("ws", c_f32p), ("ws_bytes", C.c_int64), ("counters", C.c_void_p), ("counters_len", C.c_int32)]


class _GenBlockBwdArgs(C.Structure):
    _fields_ = [("N", C.c_int32),
                ("Cin", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("Kp", C.c_int32),
                ("Tc", C.c_int32), ("Vc", C.c_int32), ("T", C.c_int32), ("V", C.c_int32), ("rep", C.c_int32),
                ("res_kind", C.c_int32), ("bn_t", C.c_int32), ("act", C.c_int32), ("slope", C.c_float),
                ("g", _Plane), ("out", _Plane), ("uo", _Plane), ("r", _Plane),
                ("coef", c_f32p),
                ("wg", c_f32p), ("wr", c_f32p), ("wt", c_f32p), ("b", c_f32p), ("u", c_f32p),
                ("du", _Plane), ("dr", _Plane), ("gyc", _Plane), ("zf", _Plane), ("gx", _Plane),
                ("px", _Plane), ("pu", _Plane), ("pr", _Plane), ("pnoise", c_f32p), ("pact", C.c_int32),
                ("pmean_t", c_f32p), ("prstd_t", c_f32p), ("pgamma_t", c_f32p),
                ("pmean_r", c_f32p), ("prstd_r", c_f32p), ("pgamma_r", c_f32p),
                ("pcoef", c_f32p), ("dgamma_t", c_f32p), ("dbeta_t", c_f32p), ("dgamma_r", c_f32p), ("dbeta_r", c_f32p),
                ("dnw", c_f32p),
                ("ws", c_f32p), ("ws_bytes", C.c_int64), ("counters", C.c_void_p), ("counters_len", C.c_int32)]


GEN_ADJ_MAX_JOBS = 8

EXPORTS = {
    "kg_abi_version": (C.c_int, []),
    "kg_arch": (C.c_char_p, []),
    "kg_last_error": (C.c_char_p, []),
    "kg_reload_env": (None, []),
    "kg_peak_mfma_f32": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.c_void_p]),
    "kg_peak_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "kg_conv_workspace_bytes": (C.c_int64, [C.POINTER(_ConvArgs)]),
    "kg_conv_plan_info": (C.c_int, [C.POINTER(_ConvArgs), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "kg_conv": (C.c_int, [C.POINTER(_ConvArgs), C.c_void_p]),
    "kg_conv_pack_bytes": (C.c_int64, [C.POINTER(_ConvArgs)]),
    "kg_conv_pack": (C.c_int, [C.POINTER(_ConvArgs), C.c_void_p, C.c_int64, C.c_void_p]),
    "kg_conv_many": (C.c_int, [C.POINTER(_ConvArgs), C.c_int32, C.c_void_p]),
    "kg_conv_many_plan": (C.c_int, [C.POINTER(_ConvArgs), C.c_int32, C.POINTER(C.c_int32)]),
    "kg_wgrad_workspace_bytes": (C.c_int64, [C.POINTER(_WgradArgs)]),
    "kg_wgrad": (C.c_int, [C.POINTER(_WgradArgs), C.c_void_p]),
    "kg_wgrad_reduce_many": (C.c_int, [C.POINTER(_WgradReduceJobs), C.c_void_p]),
    "kg_wgrad_many_workspace_bytes": (C.c_int64, [C.POINTER(_WgradArgs), C.c_int32]),
    "kg_wgrad_many": (C.c_int, [C.POINTER(_WgradArgs), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "kg_aggconv_supported": (C.c_int, [C.POINTER(_AggConvArgs)]),
    "kg_aggconv": (C.c_int, [C.POINTER(_AggConvArgs), C.c_void_p]),
    "kg_agg_expand": (C.c_int, [C.POINTER(_AggArgs), C.c_void_p]),
    "kg_agg_reduce": (C.c_int, [C.POINTER(_AggArgs), C.c_void_p]),
    "kg_agg_outer_workspace_bytes": (C.c_int64, [C.POINTER(_AggArgs)]),
    "kg_agg_outer": (C.c_int, [C.POINTER(_AggArgs), C.c_void_p]),
    "kg_agg_outer_slabs": (C.c_int, [C.POINTER(_AggArgs)]),
    "kg_agg_outer_many": (C.c_int, [C.POINTER(_AggArgs), C.c_int32, C.c_void_p]),
    "kg_agg_outer_sum_many": (C.c_int, [C.POINTER(_OuterSumJobs), C.c_void_p]),
    "kg_gen_expand": (C.c_int, [C.POINTER(_GenArgs), C.c_void_p]),
    "kg_gen_fold": (C.c_int, [C.POINTER(_GenArgs), C.c_void_p]),
    "kg_gen_adj_finish": (C.c_int, [C.POINTER(_GenAdjJob), C.c_int32, C.c_void_p]),
    "kg_gen_tail_workspace_bytes": (C.c_int64, [C.POINTER(_GenTailArgs)]),
    "kg_gen_tail_stats": (C.c_int, [C.POINTER(_GenTailArgs), C.c_void_p]),
    "kg_gen_tail_apply": (C.c_int, [C.POINTER(_GenTailArgs), C.c_void_p]),
    "kg_gen_adj_prepare": (C.c_int, [C.POINTER(_GenPrepJob), C.c_int32, C.c_void_p]),
    "kg_genblock_lds_bytes": (C.c_int64, [C.POINTER(_GenBlockArgs)]),
    "kg_genblock_workspace_bytes": (C.c_int64, [C.POINTER(_GenBlockArgs)]),
    "kg_genblock_fwd": (C.c_int, [C.POINTER(_GenBlockArgs), C.c_void_p]),
    "kg_genblock_bwd_lds_bytes": (C.c_int64, [C.POINTER(_GenBlockBwdArgs)]),
    "kg_genblock_bwd_workspace_bytes": (C.c_int64, [C.POINTER(_GenBlockBwdArgs)]),
    "kg_genblock_bwd": (C.c_int, [C.POINTER(_GenBlockBwdArgs), C.c_void_p]),
    "kg_head_fwd": (C.c_int, [C.POINTER(_HeadArgs), C.c_void_p]),
    "kg_head_bwd": (C.c_int, [C.POINTER(_HeadArgs), C.c_void_p]),
    "kg_head_wgrad": (C.c_int, [C.POINTER(_HeadArgs), C.c_void_p]),
    "kg_linear_fwd": (C.c_int, [C.POINTER(_LinearArgs), C.c_void_p]),
    "kg_linear_bwd": (C.c_int, [C.POINTER(_LinearArgs), C.c_void_p]),
    "kg_embed_bwd": (C.c_int, [C.POINTER(_LinearArgs), C.c_void_p]),
    "kg_label_bias_fwd": (C.c_int, [C.POINTER(_LabelBiasArgs), C.c_void_p]),
    "kg_label_bias_workspace_bytes": (C.c_int64, [C.POINTER(_LabelBiasArgs)]),
    "kg_label_bias_bwd": (C.c_int, [C.POINTER(_LabelBiasArgs), C.c_void_p]),
    "kg_mix3": (C.c_int, [C.POINTER(_MixArgs), C.c_void_p]),
    "kg_masked_adj_fwd": (C.c_int, [C.POINTER(_MaskedAdjArgs), C.c_void_p]),
    "kg_masked_adj_bwd": (C.c_int, [C.POINTER(_MaskedAdjArgs), C.c_void_p]),
    "kg_rowsum_workspace_bytes": (C.c_int64, [C.POINTER(_RowsumArgs)]),
    "kg_rowsum": (C.c_int, [C.POINTER(_RowsumArgs), C.c_void_p]),
    "kg_rowsum_many_workspace_bytes": (C.c_int64, [C.POINTER(_RowsumArgs), C.c_int32]),
    "kg_rowsum_many": (C.c_int, [C.POINTER(_RowsumArgs), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "kg_bn_fwd": (C.c_int, [C.POINTER(_BnArgs), C.c_void_p]),
    "kg_bn_bwd_many_workspace_bytes": (C.c_int64, [C.POINTER(_BnArgs), C.c_int32]),
    "kg_bn_bwd_many": (C.c_int, [C.POINTER(_BnArgs), C.c_int32, c_f32p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "kg_bn_fwd_many_workspace_bytes": (C.c_int64, [C.POINTER(_BnJob), C.c_int32]),
    "kg_bn_fwd_many": (C.c_int, [C.POINTER(_BnJob), C.c_int32, c_f32p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "kg_bn_bwd": (C.c_int, [C.POINTER(_BnArgs), C.c_void_p]),
    "kg_gp_fwd": (C.c_int, [C.POINTER(_GpArgs), C.c_void_p]),
    "kg_gp_bwd": (C.c_int, [C.POINTER(_GpArgs), C.c_void_p]),
    "kg_act_bwd": (C.c_int, [C.POINTER(_EltArgs), C.c_void_p]),
    "kg_affine_act": (C.c_int, [C.POINTER(_EltArgs), C.c_void_p]),
    "kg_comm_unique_id": (C.c_int, [C.c_void_p]),
    "kg_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "kg_comm_world": (C.c_int, [C.c_void_p]),
    "kg_allreduce_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "kg_comm_destroy": (C.c_int, [C.c_void_p]),
    "kg_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    "kg_adam_step_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_int32, C.c_void_p]),
}

_lib = None


def load_library():
    """dlopen libkgan_hip.so (built in-tree by build.py) and declare every exported symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the st_gcn path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.kg_abi_version() != ABI_VERSION:
        raise RuntimeError("libkgan_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().kg_last_error().decode()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def peak_mfma_f32(sink: torch.Tensor, iters: int) -> float:
    """Enqueue the fp32-MFMA peak probe (kg_peak_mfma_f32); returns the flops of the launch."""
    fl = C.c_double(0.0)
    _check(load_library().kg_peak_mfma_f32(sink.data_ptr(), int(iters), C.byref(fl), _stream()), "kg_peak_mfma_f32")
    return fl.value


def peak_copy(src: torch.Tensor, dst: torch.Tensor) -> int:
    """Enqueue the float4 copy probe (kg_peak_copy); returns the bytes it moves (read + write)."""
    n = src.numel()
    _check(load_library().kg_peak_copy(src.data_ptr(), dst.data_ptr(), n, _stream()), "kg_peak_copy")
    return 8 * n


def reload_env():
    """Tests / tuning: make the library read its KG_* environment switches again (it reads them once at load)."""
    load_library().kg_reload_env()


SYNC_LEN = 8192
_sync_bufs = {}


def _sync_buffer(device) -> torch.Tensor:
    """Zeroed ticket counters of the last-arriver kernels (kg_bn_*_many, kg_agg_outer_many, ...): one buffer per (device, stream) -
    launches of one stream run one after the other, and every launch leaves its counters at zero again."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _sync_bufs.get(key)
    if buf is None:
        buf = _sync_bufs[key] = torch.zeros(SYNC_LEN, dtype=torch.int32, device=device)
    return buf


# ---- plane tensor helpers -------------------------------------------------------------------------

def is_plane(x: torch.Tensor) -> bool:
    if x.dim() != 4:
        return False
    n, c, t, v = x.shape
    return (v == 1 or x.stride(3) == 1) and (t == 1 or x.stride(2) == v)


def as_plane(x: torch.Tensor) -> torch.Tensor:
    """Return x itself if its (t,v) plane is contiguous, else a channel-major copy."""
    if x.dtype != torch.float32:
        raise TypeError(f"st_gcn path is fp32 only, got {x.dtype}")
    if is_plane(x):
        return x
    out = new_plane(x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.device)
    out.copy_(x)
    return out


PLANE_LEAD = 32     # floats of slack in front of every plane tensor we allocate (see KgConvGroup.x_lead)


def new_plane(n, c, t, v, device, zero=False) -> torch.Tensor:
    """(N,C,T,V) tensor stored channel-major (C,N,T,V): every channel row is one contiguous run.  The
    storage starts PLANE_LEAD floats early so that kg_conv's 128-bit loads may touch the frame in front of
    a row without leaving the allocation."""
    numel = c * n * t * v
    buf = (torch.zeros if zero else torch.empty)(numel + PLANE_LEAD, dtype=torch.float32, device=device)
    return buf[PLANE_LEAD:].view(c, n, t, v).permute(1, 0, 2, 3)


def _sn_sc(x: torch.Tensor):
    n, c, t, v = x.shape
    sn = x.stride(0) if n > 1 else t * v
    sc = x.stride(1) if c > 1 else t * v
    return sn, sc


def _need_cuda(*ts):
    """Every operand on the GPU, and on the CURRENT device: the launches go to the current device's current stream
    (`_stream`), so a tensor of another device would be touched by wrong-device, unordered work (round-1 ADVICE)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("kinetic_gan_amd: the st_gcn hot path runs on the GPU only "
                               "(tensor on %s); there is no CPU fallback" % t.device)
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError("kinetic_gan_amd: tensor on %s but the current device is cuda:%d - select the device "
                               "first (torch.cuda.set_device / `with torch.cuda.device(...)`)" % (t.device, cur))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ---- weight addressing -------------------------------------------------------------------------------

class WView(NamedTuple):
    """W(d, m, c) = base + d*sT + (m // MB)*sMB + (m % MB)*sO + c*sI (elements)."""
    sT: int
    sO: int
    sI: int
    sMB: int = 0
    MB: int = 1 << 30


class Group(NamedTuple):
    x: torch.Tensor          # plane tensor (N, Cin or taps*Cin, T_in, V_in)
    w: torch.Tensor          # contiguous storage of the weights
    wv: WView
    Cin: int
    taps: int = 1
    tap_mode: int = TAP_TIME
    t_stride: int = 1
    transposed: bool = False
    vmap: Optional[torch.Tensor] = None    # int32 device tensor, V_out entries


last_conv_plan = None     # set to a list to have conv() report (tile, nsplit) of its last launch

# Executed-work accounting (bench.py: SURVEY 8d "if the build prunes ... it must also report executed FLOPs"): set
# to a dict and every launch adds the flops it really performs (2 per multiply-add, per kernel family).
flop_count = None


def _count(kind: str, flops: float):
    if flop_count is not None:
        flop_count[kind] = flop_count.get(kind, 0.0) + float(flops)


def _conv_args(groups: Sequence[Group], N: int, M: int, T_out: int, V_out: int,
               bias0=None, bias1=None, add=None, add_tstride: int = 1,
               act: int = ACT_NONE, slope: float = 0.2, mask=None, out: Optional[torch.Tensor] = None, out_t0: int = 0,
               out_tstride: int = 1, wpack: Optional[torch.Tensor] = None, alloc_out: bool = True):
    """Fill a KgConvArgs (without workspace); returns (args, out, tensors that must outlive the launch)."""
    a = _ConvArgs()
    if wpack is not None:
        _need_cuda(wpack)
        a.wpack, a.wpack_bytes = wpack.data_ptr(), wpack.numel() * wpack.element_size()
    a.N, a.M, a.T_out, a.V_out = N, M, T_out, V_out
    keep = []
    dev = groups[0].x.device
    a.ngroups = len(groups)
    for i, g in enumerate(groups):
        x = as_plane(g.x)
        w = g.w if g.w.is_contiguous() else g.w.contiguous()
        _need_cuda(x, w, g.vmap)
        keep += [x, w]
        cg = a.g[i]
        cg.x = x.data_ptr()
        cg.x_sN, cg.x_sC = _sn_sc(x)
        cg.Cin, cg.T_in, cg.V_in = g.Cin, x.shape[2], x.shape[3]
        cg.x_lead = min(int(x.storage_offset()), 1 << 20)
        cg.vmap = _ptr(g.vmap)
        cg.w = w.data_ptr()
        cg.w_sT, cg.w_sO, cg.w_sI, cg.w_sMB = g.wv.sT, g.wv.sO, g.wv.sI, g.wv.sMB
        cg.w_MB = min(g.wv.MB, 1 << 30)
        cg.taps, cg.tap_mode, cg.t_stride, cg.transposed = g.taps, g.tap_mode, g.t_stride, int(g.transposed)
    if out is None and not alloc_out:
        pass                        # (kg_conv_pack / kg_conv_pack_bytes never touch the output)
    elif out is None:
        out = new_plane(N, M, T_out, V_out, dev)
        a.out = out.data_ptr()
    else:
        if (not is_plane(out) or out.shape[0] != N or out.shape[1] != M or out.shape[3] != V_out
                or out_t0 + (T_out - 1) * out_tstride >= out.shape[2] or out_tstride < 1 or out_t0 < 0):
            raise ValueError("conv: out must be a plane tensor (N, M, T, V_out) that holds frames out_t0 + to * out_tstride")
        _need_cuda(out)
        a.out = out.data_ptr() + 4 * out_t0 * V_out
        a.o_tstride = out_tstride
    if out is not None:
        a.o_sN, a.o_sC = _sn_sc(out)
    _need_cuda(bias0, bias1, add)
    a.bias0, a.bias1 = _ptr(bias0), _ptr(bias1)
    if add is not None:
        add = as_plane(add)
        keep.append(add)
        a.add = add.data_ptr()
        a.a_sN, a.a_sC = _sn_sc(add)
    a.a_tstride = add_tstride
    a.act, a.slope = act, slope
    if mask is not None:
        mask = as_plane(mask)
        _need_cuda(mask)
        if tuple(mask.shape) != (N, M, T_out, V_out):
            raise ValueError(f"conv: mask shape {tuple(mask.shape)} != output shape {(N, M, T_out, V_out)}")
        keep.append(mask)
        a.mask = mask.data_ptr()
        a.m_sN, a.m_sC = _sn_sc(mask)
    _count("kg_conv", 2.0 * M * sum(g.taps * g.Cin for g in groups) * N * T_out * V_out)
    return a, out, keep


def conv_pack(groups: Sequence[Group], N: int, M: int, T_out: int, V_out: int, out: Optional[torch.Tensor] = None):
    """The groups' weights in the packed layout of kg_conv's bf16-split form (kg_conv_pack), or None when
    that form cannot run the launch.  Pass the result as ``wpack`` to conv() calls with the same groups / M (any batch
    size) until the weights change; ``out``: a buffer of an earlier call to refill in place."""
    lib = load_library()
    a, _, keep = _conv_args(groups, N, M, T_out, V_out, alloc_out=False)
    nbytes = lib.kg_conv_pack_bytes(C.byref(a))
    if nbytes < 0:
        _check(-1, "kg_conv_pack_bytes")
    if nbytes == 0:
        return None
    if out is None:
        out = torch.empty(nbytes // 4, dtype=torch.int32, device=groups[0].x.device)
    _check(lib.kg_conv_pack(C.byref(a), out.data_ptr(), out.numel() * 4, _stream()), "kg_conv_pack")
    return out


def conv(groups: Sequence[Group], N: int, M: int, T_out: int, V_out: int,
         bias0=None, bias1=None, add=None, add_tstride: int = 1,
         act: int = ACT_NONE, slope: float = 0.2, mask=None, out: Optional[torch.Tensor] = None, out_t0: int = 0,
         out_tstride: int = 1, wpack: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``mask``: optional (N, M, T_out, V_out) activation output; the result is multiplied by its LeakyReLU
    derivative (slope where mask <= 0) - "g * act'(out)" of the consumer folded into this launch.
    ``out`` (a plane tensor (N, M, T, V_out)), ``out_t0``, ``out_tstride``: write output frame `to` to frame
    out_t0 + to * out_tstride of `out` instead of allocating the result (returns `out`)."""
    lib = load_library()
    a, out, keep = _conv_args(groups, N, M, T_out, V_out, bias0, bias1, add, add_tstride, act, slope, mask, out, out_t0, out_tstride,
                              wpack)
    if last_conv_plan is not None:       # tests / tuning: record which kernel configuration ran
        t, ns = C.c_int32(), C.c_int32()
        lib.kg_conv_plan_info(C.byref(a), C.byref(t), C.byref(ns))
        last_conv_plan[:] = [t.value, ns.value]
    nbytes = lib.kg_conv_workspace_bytes(C.byref(a))
    if nbytes < 0:
        _check(-1, "kg_conv_workspace_bytes")
    if nbytes > 0:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=groups[0].x.device)
        a.ws, a.ws_bytes = ws.data_ptr(), nbytes
        sync = _sync_buffer(groups[0].x.device)     # a K-split launch completes its tiles itself (no epilogue launch)
        a.sync, a.sync_len = sync.data_ptr(), sync.numel()
    _check(lib.kg_conv(C.byref(a), _stream()), "kg_conv")
    return out


CONV_MANY_MAX = 4


def conv_many(jobs: Sequence[dict]) -> list:
    """Several independent kg_conv problems (each a dict of conv()'s arguments) in one launch where the launcher's plans
    allow it (kg_conv_many), one launch each otherwise; returns the outputs in order.  No job may write what another
    job of the call reads."""
    lib = load_library()
    if len(jobs) == 1:
        return [conv(**jobs[0])]
    outs, keep = [], []
    for i0 in range(0, len(jobs), CONV_MANY_MAX):
        chunk = jobs[i0:i0 + CONV_MANY_MAX]
        arr = (_ConvArgs * len(chunk))()
        for i, j in enumerate(chunk):
            a, out, kp = _conv_args(**j)
            nbytes = lib.kg_conv_workspace_bytes(C.byref(a))
            if nbytes < 0:
                _check(-1, "kg_conv_workspace_bytes")
            if nbytes > 0:          # (only used when the call falls back to one launch per job)
                ws = torch.empty(nbytes // 4, dtype=torch.float32, device=out.device)
                a.ws, a.ws_bytes = ws.data_ptr(), nbytes
                sync = _sync_buffer(out.device)
                a.sync, a.sync_len = sync.data_ptr(), sync.numel()
                kp.append(ws)
            arr[i] = a
            outs.append(out)
            keep += kp
        if last_conv_plan is not None:       # tests: the shared launch's tile, or -1 (one launch per job)
            t = C.c_int32()
            lib.kg_conv_many_plan(arr, len(chunk), C.byref(t))
            last_conv_plan[:] = [t.value, 1]
        _check(lib.kg_conv_many(arr, len(chunk), _stream()), "kg_conv_many")
    return outs


def _wgrad_args(g, x, Cin, taps, tap_mode, t_stride, vmap, wv, dw, accumulate, extra, keep):
    """Fill a KgWgradArgs (without workspace) for one layer; tensors that must outlive the launch go to `keep`."""
    g = as_plane(g)
    x = as_plane(x)
    _need_cuda(g, x, vmap, dw)
    a = _WgradArgs()
    a.N, a.M, a.T_out, a.V_out = g.shape
    a.g = g.data_ptr()
    a.g_sN, a.g_sC = _sn_sc(g)
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    a.Cin, a.T_in, a.V_in = Cin, x.shape[2], x.shape[3]
    a.vmap = _ptr(vmap)
    a.taps, a.tap_mode, a.t_stride = taps, tap_mode, t_stride
    if len(extra) > 2:
        raise ValueError("wgrad: at most two extra operand pairs")
    keep += [g, x]
    for i, (ge, xe) in enumerate(extra):
        ge, xe = as_plane(ge), as_plane(xe)
        _need_cuda(ge, xe)
        if tuple(ge.shape[1:]) != tuple(g.shape[1:]) or tuple(xe.shape[1:]) != tuple(x.shape[1:]) or ge.shape[0] != xe.shape[0]:
            raise ValueError(f"wgrad: extra pair {i} has another geometry: {tuple(ge.shape)} / {tuple(xe.shape)}")
        keep += [ge, xe]
        e = a.extra[i]
        e.N = ge.shape[0]
        e.g = ge.data_ptr()
        e.g_sN, e.g_sC = _sn_sc(ge)
        e.x = xe.data_ptr()
        e.x_sN, e.x_sC = _sn_sc(xe)
    a.nextra = len(extra)
    a.dw = dw.data_ptr()
    a.accumulate = int(accumulate)
    a.w_sT, a.w_sO, a.w_sI = wv.sT, wv.sO, wv.sI
    _count("kg_wgrad", 2.0 * taps * a.M * Cin * a.T_out * a.V_out * (a.N + sum(ge.shape[0] for ge, _ in extra)))
    return a


def wgrad(g: torch.Tensor, x: torch.Tensor, Cin: int, taps: int, tap_mode: int, t_stride: int,
          vmap: Optional[torch.Tensor], w_numel: int, wv: WView, out: Optional[torch.Tensor] = None,
          accumulate: bool = False, extra=(), defer: Optional[list] = None) -> torch.Tensor:
    """Returns the flat (w_numel,) gradient buffer written with the weight's own addressing.  out: write (or, with
    accumulate, add) into this contiguous (w_numel,) fp32 tensor instead of a new one.  extra: up to two more
    (g, x) pairs of the same layer geometry (batch size may differ) whose products are summed into the same result.
    defer: a list - only the partial slabs are computed now and a job record (which keeps the workspace alive) is
    appended; wgrad_reduce_many(defer) later finishes all of them in one launch."""
    lib = load_library()
    if out is None:
        if accumulate:
            raise ValueError("wgrad: accumulate needs out")
        dw = torch.empty(w_numel, dtype=torch.float32, device=g.device)
    else:
        dw = out
        if dw.numel() != w_numel or not dw.is_contiguous() or dw.dtype != torch.float32:
            raise ValueError("wgrad: out must be a contiguous fp32 tensor of the weight's size")
    keep = []
    a = _wgrad_args(g, x, Cin, taps, tap_mode, t_stride, vmap, wv, dw, accumulate, extra, keep)
    nbytes = lib.kg_wgrad_workspace_bytes(C.byref(a))
    if nbytes < 0:
        _check(-1, "kg_wgrad_workspace_bytes")
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=g.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    if defer is not None:
        a.defer_reduce = 1
        per = taps * a.M * Cin
        defer.append(dict(ws=ws, dw=dw, w_sT=wv.sT, w_sO=wv.sO, w_sI=wv.sI, taps=taps, M=a.M, Cin=Cin,
                          splits=max(1, nbytes // (4 * per)), accumulate=int(accumulate)))
    _check(lib.kg_wgrad(C.byref(a), _stream()), "kg_wgrad")
    return dw


def wgrad_many(jobs: Sequence[dict]):
    """The weight gradients of several layers in shared launches (kg_wgrad_many).  Each job: dict(g, x, Cin, taps,
    tap_mode, t_stride, vmap, wv, out (contiguous fp32 destination), accumulate, extra=[(g, x), ...])."""
    lib = load_library()
    if not jobs:
        return
    arr = (_WgradArgs * len(jobs))()
    keep = []
    for i, j in enumerate(jobs):
        dw = j["out"]
        if not dw.is_contiguous() or dw.dtype != torch.float32:
            raise ValueError("wgrad_many: out must be a contiguous fp32 tensor")
        arr[i] = _wgrad_args(j["g"], j["x"], j["Cin"], j["taps"], j["tap_mode"], j["t_stride"], j.get("vmap"),
                             j["wv"], dw, j.get("accumulate", False), j.get("extra", ()), keep)
    nbytes = lib.kg_wgrad_many_workspace_bytes(arr, len(jobs))
    if nbytes < 0:
        _check(-1, "kg_wgrad_many_workspace_bytes")
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=jobs[0]["g"].device)
    _check(lib.kg_wgrad_many(arr, len(jobs), ws.data_ptr(), ws.numel() * 4, _stream()), "kg_wgrad_many")


def wgrad_reduce_many(jobs: list):
    """Finish the deferred kg_wgrad calls recorded in `jobs` (see wgrad(defer=...)): one launch per 24 jobs."""
    lib = load_library()
    for i in range(0, len(jobs), WGRAD_REDUCE_MAX_JOBS):
        chunk = jobs[i:i + WGRAD_REDUCE_MAX_JOBS]
        js = _WgradReduceJobs()
        js.njobs = len(chunk)
        for k, j in enumerate(chunk):
            r = js.job[k]
            r.ws, r.dw = j["ws"].data_ptr(), j["dw"].data_ptr()
            r.w_sT, r.w_sO, r.w_sI = j["w_sT"], j["w_sO"], j["w_sI"]
            r.taps, r.M, r.Cin, r.splits, r.accumulate = j["taps"], j["M"], j["Cin"], j["splits"], j["accumulate"]
        _check(lib.kg_wgrad_reduce_many(C.byref(js), _stream()), "kg_wgrad_reduce_many")
    jobs.clear()


def _agg_args(N, Cc, K, V, W, T, rep, A):
    a = _AggArgs()
    a.N, a.C, a.K, a.V, a.W, a.T, a.rep = N, Cc, K, V, W, T, rep
    a.a = A.data_ptr()
    return a


def _adjacency(A: torch.Tensor):
    """(storage tensor, transposed flag): a (K,V,W) view of a contiguous (K,W,V) tensor is passed as it is stored"""
    if A.is_contiguous():
        return A, 0
    if A.dim() == 3 and A.transpose(1, 2).is_contiguous():
        return A.transpose(1, 2), 1
    return A.contiguous(), 0


def agg_expand(x: torch.Tensor, A: torch.Tensor, rep: int = 1) -> torch.Tensor:
    lib = load_library()
    x = as_plane(x)
    k, va, w = A.shape
    A, tr = _adjacency(A)
    _need_cuda(x, A)
    n, c, t, v = x.shape
    assert va == v, (A.shape, x.shape)
    a = _agg_args(n, c, k, v, w, t, rep, A)
    a.a_transposed = tr
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    out = new_plane(n, k * c, t * rep, w, x.device)
    a.out = out.data_ptr()
    a.o_sN, a.o_sC = _sn_sc(out)
    _count("kg_agg", 2.0 * k * v * w * c * n * t * rep)
    _check(lib.kg_agg_expand(C.byref(a), _stream()), "kg_agg_expand")
    return out


AGGCONV_P = 4          # width of the neighbour table (kgan_hip.h)


def aggconv_supported(V: int, W: int, pcount, ncols: int) -> bool:
    """Launch geometries the fused aggregation + gcn kernel takes (mirrors kg_aggconv_supported); below ~8k columns
    the unfused pair wins (kg_conv can split K across workgroups there)."""
    span = (127 // W + 2) * V
    return span <= 384 and pcount[0] <= 1 and pcount[1] <= AGGCONV_P and pcount[2] <= 1 and ncols >= 8192


def aggconv(x: torch.Tensor, A: torch.Tensor, nbr: torch.Tensor, pcount, w: torch.Tensor, wv: WView, M: int,
            add: Optional[torch.Tensor] = None, add_tstride: int = 1, want_xa: bool = False):
    """out = sum_k W_k (x A_k) (+ add) in one launch (kg_aggconv); returns (out, xa | None), xa = the aggregated
    planes (N, K*Cin, T, W) when want_xa.  nbr: (K, W, 4) int32 neighbour table of A's fixed sparsity pattern."""
    lib = load_library()
    x = as_plane(x)
    k, va, wd = A.shape
    A, tr = _adjacency(A)
    w = w if w.is_contiguous() else w.contiguous()
    _need_cuda(x, A, nbr, w, add)
    n, c, t, v = x.shape
    assert va == v and nbr.dtype == torch.int32 and tuple(nbr.shape) == (k, wd, AGGCONV_P), (A.shape, x.shape, nbr.shape)
    a = _AggConvArgs()
    a.N, a.Cin, a.M, a.T, a.V, a.W, a.K = n, c, M, t, v, wd, k
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    a.a, a.a_transposed = A.data_ptr(), tr
    a.nbr = nbr.data_ptr()
    for i in range(3):
        a.pcount[i] = int(pcount[i]) if i < k else 0
    a.w = w.data_ptr()
    a.w_sT, a.w_sO, a.w_sI = wv.sT, wv.sO, wv.sI
    out = new_plane(n, M, t, wd, x.device)
    a.out = out.data_ptr()
    a.o_sN, a.o_sC = _sn_sc(out)
    if add is not None:
        add = as_plane(add)
        a.add = add.data_ptr()
        a.a_sN, a.a_sC = _sn_sc(add)
    a.a_tstride = add_tstride
    xa = None
    if want_xa:
        xa = new_plane(n, k * c, t, wd, x.device)
        a.xa = xa.data_ptr()
        a.xa_sN, a.xa_sC = _sn_sc(xa)
    _count("kg_aggconv", 2.0 * M * k * c * n * t * wd)
    _check(lib.kg_aggconv(C.byref(a), _stream()), "kg_aggconv")
    return out, xa


def agg_reduce(y: torch.Tensor, A: torch.Tensor, fold: int = 1, res: Optional[torch.Tensor] = None, res_tstride: int = 1,
               res_inv: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None, slope: float = 0.2) -> torch.Tensor:
    """out = sum_k y_k A_k (kg_agg_reduce).  Optional epilogue (fold = 1): ``res`` (N, C, Tr, Vr) lands on frames
    t = tr * res_tstride and on the vertices w with res_inv[w] >= 0 before the result is multiplied by the LeakyReLU
    derivative of ``mask`` (N, C, T, W): (aggregate + scatter(res)) * lrelu'(mask)."""
    lib = load_library()
    y = as_plane(y)
    k, va, w = A.shape
    A, tr = _adjacency(A)
    _need_cuda(y, A, res, res_inv, mask)
    n, kc, tin, v = y.shape
    assert va == v and kc % k == 0 and tin % fold == 0, (A.shape, y.shape, fold)
    c = kc // k
    a = _agg_args(n, c, k, v, w, tin // fold, fold, A)
    a.a_transposed = tr
    a.x = y.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(y)
    out = new_plane(n, c, tin // fold, w, y.device)
    a.out = out.data_ptr()
    a.o_sN, a.o_sC = _sn_sc(out)
    if res is not None:
        res = as_plane(res)
        assert fold == 1 and res.shape[0] == n and res.shape[1] == c, (res.shape, out.shape)
        a.res = res.data_ptr()
        a.r_sN, a.r_sC = _sn_sc(res)
        a.r_T, a.r_V, a.r_tstride = res.shape[2], res.shape[3], res_tstride
        a.r_inv = _ptr(res_inv)
    if mask is not None:
        mask = as_plane(mask)
        assert fold == 1 and tuple(mask.shape) == tuple(out.shape), (mask.shape, out.shape)
        a.mask = mask.data_ptr()
        a.m_sN, a.m_sC = _sn_sc(mask)
    a.slope = slope
    _count("kg_agg", 2.0 * k * v * w * c * n * tin)
    _check(lib.kg_agg_reduce(C.byref(a), _stream()), "kg_agg_reduce")
    return out


def agg_outer(x: torch.Tensor, y: torch.Tensor, K: int, rep: int = 1, out: Optional[torch.Tensor] = None,
              defer: Optional[list] = None) -> torch.Tensor:
    """``out``: optional contiguous (K, V, W) fp32 destination (e.g. a slice of a packed adjacency-gradient buffer).
    ``defer``: a list - the problem is only recorded; agg_outer_finish(defer) computes all recorded problems in shared
    launches (``out`` is not valid before that)."""
    lib = load_library()
    x = as_plane(x)
    y = as_plane(y)
    _need_cuda(x, y, out)
    n, c, t, v = x.shape
    w = y.shape[3]
    assert y.shape[1] == K * c and y.shape[2] == t * rep, (x.shape, y.shape, K, rep)
    if out is None:
        out = torch.empty((K, v, w), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (K, v, w) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("agg_outer: out must be a contiguous fp32 (K, V, W) tensor")
    a = _agg_args(n, c, K, v, w, t, rep, out)
    a.a = None
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    a.y = y.data_ptr()
    a.y_sN, a.y_sC = _sn_sc(y)
    a.out = out.data_ptr()
    nbytes = lib.kg_agg_outer_workspace_bytes(C.byref(a))
    if nbytes < 0:
        _check(-1, "kg_agg_outer_workspace_bytes")
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=x.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    _count("kg_agg", 2.0 * K * v * w * c * n * t * rep)
    if defer is not None:
        # nothing is launched now: the record keeps the operands and the scratch alive until agg_outer_finish
        defer.append(dict(args=a, keep=(x, y, ws, out)))
        return out
    _check(lib.kg_agg_outer(C.byref(a), _stream()), "kg_agg_outer")
    return out


def agg_outer_finish(jobs: list):
    """The deferred agg_outer problems recorded in `jobs` (the adjacency gradients of a backward pass): ONE launch
    for the matrix-core form + one for all slab sums (kg_agg_outer_many).  ``out`` of every record is valid after
    this call; x and y must not have been written since they were recorded."""
    lib = load_library()
    if jobs:
        arr = (_AggArgs * len(jobs))()
        for i, j in enumerate(jobs):
            C.memmove(C.byref(arr[i]), C.byref(j["args"]), C.sizeof(_AggArgs))
        _check(lib.kg_agg_outer_many(arr, len(jobs), _stream()), "kg_agg_outer_many")
    jobs.clear()


def _gen_common(a, A, U, N, Tc, Vc, V, rep, B=None):
    a.N, a.Tc, a.Vc, a.V, a.rep = N, Tc, Vc, V, rep
    if A is not None:
        a.a = A.data_ptr()
    if U is not None:
        a.u = U.data_ptr()
    if B is not None:
        if not B.is_contiguous() or B.dim() != 3 or B.shape[1] != Vc or B.shape[2] != V:
            raise ValueError("gen_expand / gen_fold: B must be a contiguous (K, Vc, V) tensor")
        a.b = B.data_ptr()


def gen_adj_prepare(jobs: Sequence[dict]):
    """A[lvl] * importance and U (A * importance) of several generator blocks in one launch (kg_gen_adj_prepare).  Each
    job: dict(a (K, V, V), imp (K, V, V) | None, u (Vc, V) | None, aeff (K, V, V) out, b (K, Vc, V) out), all
    contiguous fp32."""
    lib = load_library()
    for i in range(0, len(jobs), GEN_ADJ_MAX_JOBS):
        chunk = jobs[i:i + GEN_ADJ_MAX_JOBS]
        arr = (_GenPrepJob * len(chunk))()
        for q, j in enumerate(chunk):
            ts = (j["a"], j.get("imp"), j.get("u"), j["aeff"], j["b"])
            _need_cuda(*ts)
            assert all(t is None or (t.is_contiguous() and t.dtype == torch.float32) for t in ts)
            k, v, _ = j["a"].shape
            vc = j["b"].shape[1]
            assert tuple(j["aeff"].shape) == (k, v, v) and tuple(j["b"].shape) == (k, vc, v)
            e = arr[q]
            e.a, e.imp, e.u, e.aeff, e.b = j["a"].data_ptr(), _ptr(j.get("imp")), _ptr(j.get("u")), j["aeff"].data_ptr(), j["b"].data_ptr()
            e.K, e.V, e.Vc = k, v, vc
        _check(lib.kg_gen_adj_prepare(arr, len(chunk), _stream()), "kg_gen_adj_prepare")


def gen_expand(y: Optional[torch.Tensor], A: Optional[torch.Tensor], U: Optional[torch.Tensor], rep: int, C_out: int,
               rs: Optional[torch.Tensor] = None, rbias: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None):
    """Generator block, second half of the head (kg_gen_expand): y (N, K*C_out, Tc, Vc) = the gcn conv on the block's
    input grid, A (K, V, V) the effective adjacency, U (Vc, V) the up-sampling matrix (None: Vc == V), ``rep`` the
    frame repeat -> z (N, C_out, Tc*rep, V) = sum_k y_k (U A_k); rs (N, Cr, Tc, Vc) -> r (N, Cr, Tc*rep, V) =
    rs U + rbias.  Returns (z | None, r | None)."""
    lib = load_library()
    src = y if y is not None else rs
    if A is not None and not A.is_contiguous():
        A = A.contiguous()
    _need_cuda(y, A, U, rs, rbias, B)
    a = _GenArgs()
    n, _, tc, vc = src.shape
    v = U.shape[1] if U is not None else vc
    _gen_common(a, A, U, n, tc, vc, v, rep, B)
    z = r = None
    if y is not None:
        y = as_plane(y)
        k = A.shape[0] if A is not None else B.shape[0]
        assert y.shape[1] == k * C_out and (A is None or tuple(A.shape) == (k, v, v)), (y.shape, C_out)
        a.C, a.K = C_out, k
        a.y = y.data_ptr()
        a.y_sN, a.y_sC = _sn_sc(y)
        z = new_plane(n, C_out, tc * rep, v, y.device)
        a.z = z.data_ptr()
        a.z_sN, a.z_sC = _sn_sc(z)
        _count("kg_agg", 2.0 * k * vc * v * C_out * n * tc)
    if rs is not None:
        rs = as_plane(rs)
        assert tuple(rs.shape[2:]) == (tc, vc) and rs.shape[0] == n
        a.Cr = rs.shape[1]
        a.rs = rs.data_ptr()
        a.rs_sN, a.rs_sC = _sn_sc(rs)
        a.rbias = _ptr(rbias)
        r = new_plane(n, rs.shape[1], tc * rep, v, rs.device)
        a.r = r.data_ptr()
        a.r_sN, a.r_sC = _sn_sc(r)
        _count("kg_agg", 2.0 * vc * v * rs.shape[1] * n * tc)
    _check(lib.kg_gen_expand(C.byref(a), _stream()), "kg_gen_expand")
    return z, r


def gen_fold(gz: Optional[torch.Tensor], A: Optional[torch.Tensor], U: Optional[torch.Tensor], rep: int, K: int,
             gr: Optional[torch.Tensor] = None, want_zf: bool = False, y_out: Optional[torch.Tensor] = None,
             rs_out: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None):
    """Adjoint of gen_expand (kg_gen_fold): gz (N, C, Tc*rep, V) -> gy (N, K*C, Tc, Vc) = fold(gz (U A_k)^T),
    gr (N, Cr, Tc*rep, V) -> grs (N, Cr, Tc, Vc) = fold(gr U^T), zf (N, C, Tc, V) = gz summed over the repeated frames
    (``want_zf``; gz itself when rep == 1).  ``y_out`` / ``rs_out``: plane tensors to write into (e.g. channel ranges
    of one (N, K*C + Cr, Tc, Vc) tensor).  Returns (gy | None, grs | None, zf | None)."""
    lib = load_library()
    src = gz if gz is not None else gr
    if A is not None and not A.is_contiguous():
        A = A.contiguous()
    _need_cuda(gz, A, U, gr, y_out, rs_out, B)
    n, _, tf, v = src.shape
    assert tf % rep == 0
    tc = tf // rep
    vc = U.shape[0] if U is not None else v
    a = _GenArgs()
    _gen_common(a, A, U, n, tc, vc, v, rep, B)
    gy = grs = zf = None
    if gz is not None:
        gz = as_plane(gz)
        c = gz.shape[1]
        a.C, a.K = c, K
        a.z = gz.data_ptr()
        a.z_sN, a.z_sC = _sn_sc(gz)
        gy = y_out if y_out is not None else new_plane(n, K * c, tc, vc, gz.device)
        assert tuple(gy.shape) == (n, K * c, tc, vc) and is_plane(gy)
        a.y_out = gy.data_ptr()
        a.y_sN, a.y_sC = _sn_sc(gy)
        if want_zf:
            if rep == 1:
                zf = gz
            else:
                zf = new_plane(n, c, tc, v, gz.device)
                a.zf = zf.data_ptr()
                a.zf_sN, a.zf_sC = _sn_sc(zf)
        _count("kg_agg", 2.0 * K * vc * v * c * n * tf)
    if gr is not None:
        gr = as_plane(gr)
        a.Cr = gr.shape[1]
        a.r = gr.data_ptr()
        a.r_sN, a.r_sC = _sn_sc(gr)
        grs = rs_out if rs_out is not None else new_plane(n, gr.shape[1], tc, vc, gr.device)
        assert tuple(grs.shape) == (n, gr.shape[1], tc, vc) and is_plane(grs)
        a.rs_out = grs.data_ptr()
        a.rs_sN, a.rs_sC = _sn_sc(grs)
        _count("kg_agg", 2.0 * vc * v * gr.shape[1] * n * tf)
    _check(lib.kg_gen_fold(C.byref(a), _stream()), "kg_gen_fold")
    return gy, grs, zf


def gen_tail_bwd(g, out, act: int, u=None, bn_t=None, r=None, bn_r=None, noise=None, sinks=None, slope: float = 0.2,
                 coef: Optional[torch.Tensor] = None, stats_only: bool = False):
    """Backward of a generator block's tail out = act(BN_t(u) + BN_r(r) + w_noise noise) in two launches
    (kg_gen_tail_stats + kg_gen_tail_apply).  bn_t / bn_r: (gamma, mean, rstd) of the layer's training-mode BatchNorm or
    None; r without bn_r: identity residual.  ``sinks``: dict with optional contiguous (C,) tensors gamma_t, beta_t,
    gamma_r, beta_r, nw that RECEIVE (+=) the parameter gradients.  Returns (du, dr | None); du is dr's tensor when
    neither branch has BatchNorm (both equal g * act'(out)).
    ``coef``: the (6, C) tail coefficients when a fused backward launch of the NEXT block has computed them already
    (genblock_bwd: statistics launch and parameter-gradient adds skipped); ``stats_only``: only the statistics launch,
    returns the coefficients (the head of a fused backward chain)."""
    lib = load_library()
    g, out = as_plane(g), as_plane(out)
    n, c, t, v = g.shape
    sinks = sinks or {}
    a = _GenTailArgs()
    a.N, a.C, a.T, a.V, a.act, a.slope = n, c, t, v, act, slope
    a.g = g.data_ptr()
    a.g_sN, a.g_sC = _sn_sc(g)
    a.out = out.data_ptr()
    a.o_sN, a.o_sC = _sn_sc(out)
    keep = [g, out]
    if bn_t is not None:
        u = as_plane(u)
        gam, mean, rstd = (_vec(q, c, "gen_tail_bwd") for q in bn_t)
        a.u = u.data_ptr()
        a.u_sN, a.u_sC = _sn_sc(u)
        a.gamma_t, a.mean_t, a.rstd_t = _ptr(gam), mean.data_ptr(), rstd.data_ptr()
        keep += [u, gam, mean, rstd]
    if r is not None:
        r = as_plane(r)
        a.r = r.data_ptr()
        a.r_sN, a.r_sC = _sn_sc(r)
        keep.append(r)
        if bn_r is not None:
            gam, mean, rstd = (_vec(q, c, "gen_tail_bwd") for q in bn_r)
            a.gamma_r, a.mean_r, a.rstd_r = _ptr(gam), mean.data_ptr(), rstd.data_ptr()
            keep += [gam, mean, rstd]
    if noise is not None:
        noise = noise.contiguous()
        a.noise = noise.data_ptr()
        keep.append(noise)
    _need_cuda(*keep, *sinks.values())
    for k_, t_ in sinks.items():
        if t_ is not None and (t_.numel() != c or not t_.is_contiguous()):
            raise ValueError("gen_tail_bwd: sink %s must be a contiguous (C,) tensor" % k_)
    a.dgamma_t, a.dbeta_t = _ptr(sinks.get("gamma_t")), _ptr(sinks.get("beta_t"))
    a.dgamma_r, a.dbeta_r, a.dnw = _ptr(sinks.get("gamma_r")), _ptr(sinks.get("beta_r")), _ptr(sinks.get("nw"))
    have_coef = coef is not None
    if have_coef:
        if tuple(coef.shape) != (6, c) or not coef.is_contiguous():
            raise ValueError("gen_tail_bwd: coef must be a contiguous (6, C) tensor")
        _need_cuda(coef)
    else:
        coef = torch.empty((6, c), dtype=torch.float32, device=g.device)
    a.coef = coef.data_ptr()
    if not have_coef:
        if c > SYNC_LEN:
            raise ValueError("gen_tail_bwd: %d channels exceed the %d ticket counters" % (c, SYNC_LEN))
        nbytes = lib.kg_gen_tail_workspace_bytes(C.byref(a))
        if nbytes < 0:
            _check(-1, "kg_gen_tail_workspace_bytes")
        ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=g.device)
        sync = _sync_buffer(g.device)
        a.ws, a.ws_bytes, a.counters, a.counters_len = ws.data_ptr(), ws.numel() * 4, sync.data_ptr(), sync.numel()
        _check(lib.kg_gen_tail_stats(C.byref(a), _stream()), "kg_gen_tail_stats")
    if stats_only:
        return coef
    du = new_plane(n, c, t, v, g.device)
    a.du = du.data_ptr()
    a.du_sN, a.du_sC = _sn_sc(du)
    dr = None
    if r is not None:
        if bn_t is None and bn_r is None:
            dr = du
        else:
            dr = new_plane(n, c, t, v, g.device)
            a.dr = dr.data_ptr()
            a.dr_sN, a.dr_sC = _sn_sc(dr)
    _check(lib.kg_gen_tail_apply(C.byref(a), _stream()), "kg_gen_tail_apply")
    return du, dr


def gen_adj_finish(jobs: Sequence[dict]):
    """d edge_importance of several generator blocks in one launch (kg_gen_adj_finish).  Each job: dict(dbt (Kd, V, Vc)
    contiguous, u (Vc, V) | None, a (K, V, V) | None, out (K, V, V) contiguous view, accumulate)."""
    lib = load_library()
    for i in range(0, len(jobs), GEN_ADJ_MAX_JOBS):
        chunk = jobs[i:i + GEN_ADJ_MAX_JOBS]
        arr = (_GenAdjJob * len(chunk))()
        for q, j in enumerate(chunk):
            dbt, out = j["dbt"], j["out"]
            _need_cuda(dbt, j.get("u"), j.get("a"), out)
            assert dbt.is_contiguous() and out.is_contiguous() and dbt.dim() == 3 and out.dim() == 3
            k, v, _ = out.shape
            kd, v2, vc = dbt.shape
            assert v2 == v and kd <= k, (dbt.shape, out.shape)
            e = arr[q]
            e.dbt, e.u, e.a, e.out = dbt.data_ptr(), _ptr(j.get("u")), _ptr(j.get("a")), out.data_ptr()
            e.K, e.Kd, e.V, e.Vc, e.accumulate = k, kd, v, vc, int(bool(j.get("accumulate", False)))
        _check(lib.kg_gen_adj_finish(arr, len(chunk), _stream()), "kg_gen_adj_finish")


# ---- fused generator block (kg_genblock.hip, ABI v9) -----------------------------------------------------------------

def _plane(t: Optional[torch.Tensor]) -> _Plane:
    p = _Plane()
    if t is not None:
        if not is_plane(t) or t.dtype != torch.float32:
            raise ValueError("genblock: plane tensor expected, got %s / strides %s" % (tuple(t.shape), t.stride()))
        p.p = t.data_ptr()
        p.sN, p.sC = _sn_sc(t)
    return p


class GenBlockDims(NamedTuple):
    """Static geometry of one generator block as the fused kernels take it (gen_trunk.GenBlockGeom)."""
    Cin: int
    C: int
    K: int
    Kp: int
    Tc: int
    Vc: int
    T: int
    V: int
    rep: int
    res_kind: int      # 0 none, 1 identity, 2 conv + BatchNorm
    bn_t: bool
    act: int


def _gb_common(a, d: GenBlockDims, slope):
    a.Cin, a.C, a.K, a.Kp, a.Tc, a.Vc, a.T, a.V, a.rep = d.Cin, d.C, d.K, d.Kp, d.Tc, d.Vc, d.T, d.V, d.rep
    a.res_kind, a.bn_t, a.act, a.slope = d.res_kind, int(d.bn_t), d.act, slope


def _gb_weights(a, wg, wr, wt, B, U):
    for t in (wg, wr, wt, B, U):
        if t is not None and (not t.is_contiguous() or t.dtype != torch.float32):
            raise ValueError("genblock: weights / adjacency must be contiguous fp32")
    a.wg, a.wr, a.wt, a.b, a.u = wg.data_ptr(), _ptr(wr), wt.data_ptr(), B.data_ptr(), _ptr(U)


def genblock_supported(d: GenBlockDims, n: int, wg, wr, wt, backward: bool = False) -> bool:
    """Can the fused kernels run this block (per-sample working set in LDS, contraction shapes, 16-byte aligned weight
    rows)?  Otherwise the staged entry points apply."""
    lib = load_library()
    if backward:
        a = _GenBlockBwdArgs()
        a.N = n
        _gb_common(a, d, 0.2)
        return lib.kg_genblock_bwd_lds_bytes(C.byref(a)) >= 0
    a = _GenBlockArgs()
    a.N, a.groups = n, 1
    _gb_common(a, d, 0.2)
    a.wg, a.wr, a.wt = wg.data_ptr(), _ptr(wr), wt.data_ptr()
    return lib.kg_genblock_lds_bytes(C.byref(a)) >= 0


def _gb_bn(layer: _GenBnLayer, bn: dict, c: int, groups: int, device):
    vecs = [_vec(bn.get(k), c, "genblock_fwd") for k in ("gamma", "beta", "running_mean", "running_var")]
    nbt = bn.get("num_batches_tracked")
    _need_cuda(*vecs, nbt)
    layer.gamma, layer.beta, layer.running_mean, layer.running_var = [_ptr(q) for q in vecs]
    if nbt is not None:
        assert nbt.dtype == torch.int64
        layer.num_batches_tracked = nbt.data_ptr()
    layer.momentum, layer.eps = float(bn["momentum"]), float(bn["eps"])
    coef = torch.empty((groups, 4, c), dtype=torch.float32, device=device)
    layer.coef = coef.data_ptr()
    return coef


def genblock_fwd(d: GenBlockDims, *, x=None, pend=None, wg, wr=None, br=None, wt, bt=None, B, U=None, bn_t=None, bn_r=None,
                 groups: int = 1, noise=None, nw=None, slope: float = 0.2) -> dict:
    """One generator block forward in ONE launch (kg_genblock_fwd).  ``x`` (N, Cin, Tc, Vc): finished input, or ``pend`` =
    dict(u, r | None, ct | None, cr | None, noise | None, nw | None, act): the previous block's pending tail
    act(u * ct.scale + ct.shift + r * cr.scale + cr.shift + nw * noise) with ct / cr its (groups, 4, Cin) BatchNorm
    coefficients - applied here, the result written to the returned ``x``.  bn_t / bn_r: dict(gamma, beta, running_mean,
    running_var, num_batches_tracked, momentum, eps) of the block's training-mode BatchNorm layers.  A block without any
    BatchNorm finishes itself with ``noise`` / ``nw`` (returned ``out``).  Returns dict(x, yc, z, r, u, ct, cr, out)."""
    lib = load_library()
    a = _GenBlockArgs()
    _gb_common(a, d, slope)
    if (bn_t is not None) != bool(d.bn_t) or (bn_r is not None) != (d.res_kind == 2):
        raise ValueError("genblock_fwd: BatchNorm layers do not match the block's geometry")
    keep = []
    if x is not None:
        x = as_plane(x)
        n = x.shape[0]
        a.x = _plane(x)
        dev = x.device
        if tuple(x.shape[1:]) != (d.Cin, d.Tc, d.Vc):
            raise ValueError("genblock_fwd: input %s does not match (%d, %d, %d)" % (tuple(x.shape), d.Cin, d.Tc, d.Vc))
    else:
        pu = as_plane(pend["u"])
        n, dev = pu.shape[0], pu.device
        if tuple(pu.shape[1:]) != (d.Cin, d.Tc, d.Vc):
            raise ValueError("genblock_fwd: pending tail %s does not match (%d, %d, %d)" % (tuple(pu.shape), d.Cin, d.Tc, d.Vc))
        a.pu = _plane(pu)
        pr = pend.get("r")
        if pr is not None:
            pr = as_plane(pr)
            a.pr = _plane(pr)
        for key, fld in (("ct", "pcoef_t"), ("cr", "pcoef_r")):
            cf = pend.get(key)
            if cf is not None:
                if tuple(cf.shape) != (groups, 4, d.Cin) or not cf.is_contiguous():
                    raise ValueError("genblock_fwd: pending %s must be a contiguous (groups, 4, Cin) tensor" % key)
                setattr(a, fld, cf.data_ptr())
        pn, pw = pend.get("noise"), pend.get("nw")
        if pn is not None and pw is not None:
            pn, pw = pn.contiguous(), pw.reshape(-1).contiguous()
            a.pnoise, a.pnw = pn.data_ptr(), pw.data_ptr()
        a.pact = int(pend["act"])
        x = new_plane(n, d.Cin, d.Tc, d.Vc, dev)
        a.xout = _plane(x)
        keep += [pu, pr, pn, pw, pend.get("ct"), pend.get("cr")]
        _need_cuda(*keep)
    a.N, a.groups = n, groups
    _gb_weights(a, wg, wr, wt, B, U)
    a.br, a.bt = _ptr(br), _ptr(bt)
    _need_cuda(x, wg, wr, br, wt, bt, B, U, noise, nw)
    Mh = d.Kp * d.C + (d.C if d.res_kind == 2 else 0)
    yc = new_plane(n, Mh, d.Tc, d.Vc, dev)
    z = new_plane(n, d.C, d.T, d.V, dev)
    u = new_plane(n, d.C, d.T, d.V, dev)
    r = new_plane(n, d.C, d.T, d.V, dev) if d.res_kind != 0 else None
    a.yc, a.z, a.uo, a.r = _plane(yc), _plane(z), _plane(u), _plane(r)
    ct = _gb_bn(a.bt_, bn_t, d.C, groups, dev) if bn_t is not None else None
    cr = _gb_bn(a.br_, bn_r, d.C, groups, dev) if bn_r is not None else None
    out = None
    if ct is None and cr is None:
        out = new_plane(n, d.C, d.T, d.V, dev)
        a.out = _plane(out)
        if noise is not None and nw is not None:
            noise, nw = noise.contiguous(), nw.reshape(-1).contiguous()
            a.noise, a.nw = noise.data_ptr(), nw.data_ptr()
    else:
        nbytes = lib.kg_genblock_workspace_bytes(C.byref(a))
        if nbytes < 0:
            _check(-1, "kg_genblock_workspace_bytes")
        ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=dev)
        sync = _sync_buffer(dev)
        a.ws, a.ws_bytes, a.counters, a.counters_len = ws.data_ptr(), ws.numel() * 4, sync.data_ptr(), sync.numel()
    _count("kg_genblock", 2.0 * n * (Mh * d.Cin * d.Tc * d.Vc + 3 * d.C * d.C * d.T * d.V) +
           2.0 * n * (d.Kp + (1 if d.res_kind else 0)) * d.Vc * d.V * d.C * d.Tc)
    _check(lib.kg_genblock_fwd(C.byref(a), _stream()), "kg_genblock_fwd")
    return dict(x=x, yc=yc, z=z, r=r, u=u, ct=ct, cr=cr, out=out)


def genblock_bwd(d: GenBlockDims, *, g, out, u=None, r=None, coef, wg, wr=None, wt, B, U=None, prev=None, slope: float = 0.2) -> dict:
    """One generator block backward in ONE launch (kg_genblock_bwd): g = d loss / d out, (out, u, r) the block's tape,
    coef (6, C) its tail coefficients (gen_tail_bwd(stats_only=True) or the previous call's ``pcoef``).  ``prev`` =
    dict(x, u | None, r | None, noise | None, act, bn_t = (gamma, mean, rstd) | None, bn_r likewise, sinks = dict of (Cin,)
    tensors gamma_t, beta_t, gamma_r, beta_r, nw that RECEIVE the parameter gradients): the block before this one, whose
    tail statistics are taken on the way out (returned ``pcoef`` (6, Cin)).  Returns dict(du, dr, gyc, zf, gx, pcoef);
    dr is du when the block has a residual but no BatchNorm at all."""
    lib = load_library()
    a = _GenBlockBwdArgs()
    _gb_common(a, d, slope)
    g, out = as_plane(g), as_plane(out)
    n, dev = g.shape[0], g.device
    a.N = n
    a.g, a.out = _plane(g), _plane(out)
    keep = [g, out, coef]
    if d.bn_t:
        u = as_plane(u)
        a.uo = _plane(u)
        keep.append(u)
    if d.res_kind == 2:
        r = as_plane(r)
        a.r = _plane(r)
        keep.append(r)
    if tuple(coef.shape) != (6, d.C) or not coef.is_contiguous():
        raise ValueError("genblock_bwd: coef must be a contiguous (6, C) tensor")
    a.coef = coef.data_ptr()
    _gb_weights(a, wg, wr, wt, B, U)
    _need_cuda(*keep, wg, wr, wt, B, U)
    Mh = d.Kp * d.C + (d.C if d.res_kind == 2 else 0)
    du = new_plane(n, d.C, d.T, d.V, dev)
    dr = None
    if d.res_kind == 2 or (d.res_kind == 1 and d.bn_t):
        dr = new_plane(n, d.C, d.T, d.V, dev)
    gyc = new_plane(n, Mh, d.Tc, d.Vc, dev)
    zf = new_plane(n, d.C, d.Tc, d.V, dev)
    gx = new_plane(n, d.Cin, d.Tc, d.Vc, dev)
    a.du, a.dr, a.gyc, a.zf, a.gx = _plane(du), _plane(dr), _plane(gyc), _plane(zf), _plane(gx)
    pcoef = None
    if prev is not None:
        px = as_plane(prev["x"])
        a.px = _plane(px)
        a.pact = int(prev["act"])
        sinks = prev.get("sinks") or {}
        keep2 = [px]
        if prev.get("bn_t") is not None:
            pu = as_plane(prev["u"])
            gam, mean, rstd = (_vec(q, d.Cin, "genblock_bwd") for q in prev["bn_t"])
            a.pu = _plane(pu)
            a.pgamma_t, a.pmean_t, a.prstd_t = _ptr(gam), mean.data_ptr(), rstd.data_ptr()
            keep2 += [pu, gam, mean, rstd]
        if prev.get("bn_r") is not None:
            pr = as_plane(prev["r"])
            gam, mean, rstd = (_vec(q, d.Cin, "genblock_bwd") for q in prev["bn_r"])
            a.pr = _plane(pr)
            a.pgamma_r, a.pmean_r, a.prstd_r = _ptr(gam), mean.data_ptr(), rstd.data_ptr()
            keep2 += [pr, gam, mean, rstd]
        pn = prev.get("noise")
        if pn is not None:
            pn = pn.contiguous()
            a.pnoise = pn.data_ptr()
            keep2.append(pn)
        for k_, t_ in sinks.items():
            if t_ is not None and (t_.numel() != d.Cin or not t_.is_contiguous()):
                raise ValueError("genblock_bwd: sink %s must be a contiguous (Cin,) tensor" % k_)
        _need_cuda(*keep2, *sinks.values())
        a.dgamma_t, a.dbeta_t = _ptr(sinks.get("gamma_t")), _ptr(sinks.get("beta_t"))
        a.dgamma_r, a.dbeta_r, a.dnw = _ptr(sinks.get("gamma_r")), _ptr(sinks.get("beta_r")), _ptr(sinks.get("nw"))
        pcoef = torch.empty((6, d.Cin), dtype=torch.float32, device=dev)
        a.pcoef = pcoef.data_ptr()
        nbytes = lib.kg_genblock_bwd_workspace_bytes(C.byref(a))
        if nbytes < 0:
            _check(-1, "kg_genblock_bwd_workspace_bytes")
        ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=dev)
        sync = _sync_buffer(dev)
        a.ws, a.ws_bytes, a.counters, a.counters_len = ws.data_ptr(), ws.numel() * 4, sync.data_ptr(), sync.numel()
    _count("kg_genblock", 2.0 * n * (Mh * d.Cin * d.Tc * d.Vc + 3 * d.C * d.C * d.T * d.V) +
           2.0 * n * (d.Kp + (1 if d.res_kind else 0)) * d.Vc * d.V * d.C * d.T)
    _check(lib.kg_genblock_bwd(C.byref(a), _stream()), "kg_genblock_bwd")
    if dr is None and d.res_kind != 0:
        dr = du
    return dict(du=du, dr=dr, gyc=gyc, zf=zf, gx=gx, pcoef=pcoef)


def rowsum(x: torch.Tensor, y: Optional[torch.Tensor] = None, second: bool = False,
           shift: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           accumulate: bool = False, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(1|2, C): sum over (n,t,v) of x, and of x*(y-shift) ((x-shift)^2 when y is None).  y may
    broadcast over C (shape (N,1,T,V)); shift is per channel.  out / accumulate as in wgrad; out2: a second
    destination for the same sums."""
    lib = load_library()
    x = as_plane(x)
    _need_cuda(x, y, shift)
    n, c, t, v = x.shape
    a = _RowsumArgs()
    if shift is not None:
        shift = shift.reshape(-1).contiguous()
        a.shift = shift.data_ptr()
    a.N, a.C, a.T, a.V = n, c, t, v
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    keep = None
    if y is not None:
        if y.shape[1] == 1 and c > 1:
            keep = y.contiguous()
            a.y, a.y_sN, a.y_sC = keep.data_ptr(), t * v, 0
        else:
            keep = as_plane(y)
            a.y = keep.data_ptr()
            a.y_sN, a.y_sC = _sn_sc(keep)
    a.want_second = int(second)      # True: both rows; 2: the product row alone
    rows = 2 if int(second) == 1 else 1
    if out is None:
        if accumulate:
            raise ValueError("rowsum: accumulate needs out")
        out = torch.empty((rows, c), dtype=torch.float32, device=x.device)
    else:
        if out.numel() != rows * c or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("rowsum: out must be a contiguous fp32 tensor of %d elements" % (rows * c))
        _need_cuda(out)
    a.out = out.data_ptr()
    a.accumulate = int(accumulate)
    if out2 is not None:
        if out2.numel() != rows * c or not out2.is_contiguous() or out2.dtype != torch.float32:
            raise ValueError("rowsum: out2 must be a contiguous fp32 tensor of %d elements" % (rows * c))
        _need_cuda(out2)
        a.out2 = out2.data_ptr()
    nbytes = lib.kg_rowsum_workspace_bytes(C.byref(a))
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=x.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    _check(lib.kg_rowsum(C.byref(a), _stream()), "kg_rowsum")
    return out


def rowsum_many(jobs: Sequence[dict]):
    """Per-channel sums over (n, t, v) of several tensors in one launch (+ one finishing launch): each job
    dict(x, out, out2=None, accumulate=False) writes / adds sum(x) per channel to `out` (and `out2`); with ``y``
    (same shape as x, or (N,1,T,V)) the sums are of x*y instead."""
    lib = load_library()
    if not jobs:
        return
    arr = (_RowsumArgs * len(jobs))()
    keep = []
    for i, j in enumerate(jobs):
        x = as_plane(j["x"])
        out, out2 = j["out"], j.get("out2")
        _need_cuda(x, out, out2)
        n, c, t, v = x.shape
        for o in (out, out2):
            if o is not None and (o.numel() != c or not o.is_contiguous() or o.dtype != torch.float32):
                raise ValueError("rowsum_many: destinations must be contiguous fp32 vectors of C elements")
        keep.append(x)
        a = arr[i]
        a.N, a.C, a.T, a.V = n, c, t, v
        a.x = x.data_ptr()
        a.x_sN, a.x_sC = _sn_sc(x)
        y = j.get("y")
        if y is not None:
            _need_cuda(y)
            if y.shape[1] == 1 and c > 1:
                y = y.contiguous()
                a.y, a.y_sN, a.y_sC = y.data_ptr(), t * v, 0
            else:
                y = as_plane(y)
                a.y = y.data_ptr()
                a.y_sN, a.y_sC = _sn_sc(y)
            keep.append(y)
            a.want_second = 2
        a.out, a.out2 = out.data_ptr(), _ptr(out2)
        a.accumulate = int(bool(j.get("accumulate", False)))
    nbytes = lib.kg_rowsum_many_workspace_bytes(arr, len(jobs))
    if nbytes < 0:
        _check(-1, "kg_rowsum_many_workspace_bytes")
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=jobs[0]["x"].device)
    _check(lib.kg_rowsum_many(arr, len(jobs), ws.data_ptr(), ws.numel() * 4, _stream()), "kg_rowsum_many")


def _elt_args(x, out, act, slope):
    a = _EltArgs()
    a.N, a.C, a.T, a.V = x.shape
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    a.out = out.data_ptr()
    a.o_sN, a.o_sC = _sn_sc(out)
    a.act, a.slope = act, slope
    return a


def act_bwd(g: torch.Tensor, ref: torch.Tensor, act: int, slope: float = 0.2) -> torch.Tensor:
    lib = load_library()
    g = as_plane(g)
    ref = as_plane(ref)
    _need_cuda(g, ref)
    out = new_plane(*g.shape, g.device)
    a = _elt_args(g, out, act, slope)
    a.r = ref.data_ptr()
    a.r_sN, a.r_sC = _sn_sc(ref)
    _check(lib.kg_act_bwd(C.byref(a), _stream()), "kg_act_bwd")
    return out


def affine_act(x, sx=None, bx=None, r=None, sr=None, br=None, noise=None, nw=None,
               act: int = ACT_NONE, slope: float = 0.2, out: Optional[torch.Tensor] = None,
               groups: int = 1, coef_gs: int = 0) -> torch.Tensor:
    """``out``: optional plane tensor of x's shape to write into (e.g. a sample range of a larger buffer).
    ``groups`` > 1: x holds that many batches stacked along N; sx / bx / sr / br are the FIRST batch's vectors inside a
    buffer that holds batch q's at + q * coef_gs floats (the (groups, 4, C) result of bn_fwd_many: coef_gs = 4 C)."""
    lib = load_library()
    x = as_plane(x)
    vecs = [None if t is None else t.reshape(-1).contiguous() for t in (sx, bx, sr, br, nw)]
    _need_cuda(x, r, noise, out, *vecs)
    if out is None:
        out = new_plane(*x.shape, x.device)
    elif tuple(out.shape) != tuple(x.shape) or not is_plane(out) or out.dtype != torch.float32:
        raise ValueError("affine_act: out must be a plane tensor of x's shape")
    a = _elt_args(x, out, act, slope)
    if r is not None:
        r = as_plane(r)
        a.r = r.data_ptr()
        a.r_sN, a.r_sC = _sn_sc(r)
    if noise is not None:
        noise = noise.contiguous()
        a.noise = noise.data_ptr()
    a.sx, a.bx, a.sr, a.br, a.nw = [_ptr(t) for t in vecs]
    a.groups, a.coef_gs = int(groups), int(coef_gs)
    _check(lib.kg_affine_act(C.byref(a), _stream()), "kg_affine_act")
    return out


def gp_fwd(g: torch.Tensor):
    """(nrm (N,), gp ()) of the WGAN-GP penalty for per-sample gradients g (N, C, T, V): kg_gp_fwd"""
    lib = load_library()
    g = as_plane(g)
    _need_cuda(g)
    a = _GpArgs()
    a.N, a.C, a.T, a.V = g.shape
    a.g = g.data_ptr()
    a.g_sN, a.g_sC = _sn_sc(g)
    nrm = torch.empty(g.shape[0], dtype=torch.float32, device=g.device)
    gp = torch.empty((), dtype=torch.float32, device=g.device)
    a.nrm, a.gp = nrm.data_ptr(), gp.data_ptr()
    _check(lib.kg_gp_fwd(C.byref(a), _stream()), "kg_gp_fwd")
    return nrm, gp


def gp_bwd(g: torch.Tensor, nrm: torch.Tensor, gout: torch.Tensor) -> torch.Tensor:
    """d gp / d g * gout (gout: 0-dim device tensor): kg_gp_bwd"""
    lib = load_library()
    g = as_plane(g)
    gout = gout.reshape(1).contiguous()
    _need_cuda(g, nrm, gout)
    a = _GpArgs()
    a.N, a.C, a.T, a.V = g.shape
    a.g = g.data_ptr()
    a.g_sN, a.g_sC = _sn_sc(g)
    a.nrm, a.gout = nrm.data_ptr(), gout.data_ptr()
    out = new_plane(*g.shape, g.device)
    a.out = out.data_ptr()
    a.o_sN, a.o_sC = _sn_sc(out)
    _check(lib.kg_gp_bwd(C.byref(a), _stream()), "kg_gp_bwd")
    return out


def _vec(t: Optional[torch.Tensor], c: int, what: str):
    if t is None:
        return None
    if t.numel() != c or not t.is_contiguous() or t.dtype != torch.float32:
        raise ValueError(f"{what}: expected a contiguous fp32 vector of {c} elements")
    return t


def bn_fwd(x: torch.Tensor, gamma, beta, running_mean, running_var, num_batches_tracked, training: bool,
           momentum: float, eps: float) -> torch.Tensor:
    """BatchNorm2d statistics of x and the per-channel coefficients, one launch: returns (4, C) =
    [scale, shift, mean, rstd]; training mode updates the running statistics in place (torch semantics)."""
    lib = load_library()
    x = as_plane(x)
    n, c, t, v = x.shape
    vecs = [_vec(q, c, "bn_fwd") for q in (gamma, beta, running_mean, running_var)]
    _need_cuda(x, *vecs, num_batches_tracked)
    a = _BnArgs()
    a.N, a.C, a.T, a.V = n, c, t, v
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    a.gamma, a.beta, a.running_mean, a.running_var = [_ptr(q) for q in vecs]
    if num_batches_tracked is not None:
        assert num_batches_tracked.dtype == torch.int64
        a.num_batches_tracked = num_batches_tracked.data_ptr()
    a.momentum, a.eps, a.training = float(momentum), float(eps), int(bool(training))
    coef = torch.empty((4, c), dtype=torch.float32, device=x.device)
    a.coef = coef.data_ptr()
    _check(lib.kg_bn_fwd(C.byref(a), _stream()), "kg_bn_fwd")
    return coef


def bn_fwd_many(jobs: Sequence[dict]):
    """Training-mode BatchNorm2d statistics of up to four layers in ONE launch.  Each job: dict(x, gamma, beta,
    running_mean, running_var, num_batches_tracked, momentum, eps, groups): x holds ``groups`` independent batches
    stacked along N; returns one (groups, 4, C) tensor [scale, shift, mean, rstd] per job, the running statistics
    updated batch by batch as ``groups`` bn_fwd calls would."""
    lib = load_library()
    arr = (_BnJob * len(jobs))()
    keep, coefs = [], []
    total_c = 0
    for i, j in enumerate(jobs):
        x = as_plane(j["x"])
        n, c, t, v = x.shape
        groups = int(j.get("groups", 1))
        if n % groups:
            raise ValueError("bn_fwd_many: N=%d is not a multiple of groups=%d" % (n, groups))
        vecs = [_vec(j.get(k), c, "bn_fwd_many") for k in ("gamma", "beta", "running_mean", "running_var")]
        nbt = j.get("num_batches_tracked")
        _need_cuda(x, *vecs, nbt)
        a = arr[i].a
        a.N, a.C, a.T, a.V = n // groups, c, t, v
        a.x = x.data_ptr()
        a.x_sN, a.x_sC = _sn_sc(x)
        a.gamma, a.beta, a.running_mean, a.running_var = [_ptr(q) for q in vecs]
        if nbt is not None:
            assert nbt.dtype == torch.int64
            a.num_batches_tracked = nbt.data_ptr()
        a.momentum, a.eps, a.training = float(j["momentum"]), float(j["eps"]), 1
        coef = torch.empty((groups, 4, c), dtype=torch.float32, device=x.device)
        a.coef = coef.data_ptr()
        arr[i].groups = groups
        keep += [x] + vecs
        coefs.append(coef)
        total_c += c
    if total_c > SYNC_LEN:
        raise ValueError("bn_fwd_many: %d channels exceed the %d ticket counters" % (total_c, SYNC_LEN))
    nbytes = lib.kg_bn_fwd_many_workspace_bytes(arr, len(jobs))
    if nbytes < 0:
        _check(-1, "kg_bn_fwd_many_workspace_bytes")
    dev = coefs[0].device
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=dev)
    sync = _sync_buffer(dev)
    _check(lib.kg_bn_fwd_many(arr, len(jobs), ws.data_ptr(), ws.numel() * 4, sync.data_ptr(), sync.numel(), _stream()),
           "kg_bn_fwd_many")
    return coefs


def bn_bwd(g: torch.Tensor, x: torch.Tensor, gamma, mean: torch.Tensor, rstd: torch.Tensor, training: bool) -> torch.Tensor:
    """(5, C) = [a, b, c, dgamma, dbeta] of the BatchNorm2d backward (dL/dx = a*g + b*x + c), one launch."""
    lib = load_library()
    g, x = as_plane(g), as_plane(x)
    n, c, t, v = x.shape
    gamma, mean, rstd = _vec(gamma, c, "bn_bwd"), _vec(mean, c, "bn_bwd"), _vec(rstd, c, "bn_bwd")
    _need_cuda(g, x, gamma, mean, rstd)
    a = _BnArgs()
    a.N, a.C, a.T, a.V = n, c, t, v
    a.x = x.data_ptr()
    a.x_sN, a.x_sC = _sn_sc(x)
    a.g = g.data_ptr()
    a.g_sN, a.g_sC = _sn_sc(g)
    a.gamma, a.mean, a.rstd = _ptr(gamma), _ptr(mean), _ptr(rstd)
    a.training = int(bool(training))
    coef = torch.empty((5, c), dtype=torch.float32, device=x.device)
    a.coef = coef.data_ptr()
    _check(lib.kg_bn_bwd(C.byref(a), _stream()), "kg_bn_bwd")
    return coef


def bn_bwd_many(jobs: Sequence[dict]):
    """bn_bwd of up to four layers in ONE launch (chunked partial sums, many workgroups per channel): each job
    dict(g, x, gamma, mean, rstd, training) -> its (5, C) coefficients [a, b, c, dgamma, dbeta]."""
    lib = load_library()
    arr = (_BnArgs * len(jobs))()
    keep, coefs = [], []
    total_c = 0
    for i, j in enumerate(jobs):
        g, x = as_plane(j["g"]), as_plane(j["x"])
        n, c, t, v = x.shape
        gamma, mean, rstd = _vec(j.get("gamma"), c, "bn_bwd_many"), _vec(j["mean"], c, "bn_bwd_many"), _vec(j["rstd"], c, "bn_bwd_many")
        _need_cuda(g, x, gamma, mean, rstd)
        a = arr[i]
        a.N, a.C, a.T, a.V = n, c, t, v
        a.x = x.data_ptr()
        a.x_sN, a.x_sC = _sn_sc(x)
        a.g = g.data_ptr()
        a.g_sN, a.g_sC = _sn_sc(g)
        a.gamma, a.mean, a.rstd = _ptr(gamma), _ptr(mean), _ptr(rstd)
        a.training = int(bool(j["training"]))
        coef = torch.empty((5, c), dtype=torch.float32, device=x.device)
        a.coef = coef.data_ptr()
        keep += [g, x, gamma, mean, rstd]
        coefs.append(coef)
        total_c += c
    if total_c > SYNC_LEN:
        raise ValueError("bn_bwd_many: %d channels exceed the %d ticket counters" % (total_c, SYNC_LEN))
    nbytes = lib.kg_bn_bwd_many_workspace_bytes(arr, len(jobs))
    if nbytes < 0:
        _check(-1, "kg_bn_bwd_many_workspace_bytes")
    dev = coefs[0].device
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=dev)
    sync = _sync_buffer(dev)
    _check(lib.kg_bn_bwd_many(arr, len(jobs), ws.data_ptr(), ws.numel() * 4, sync.data_ptr(), sync.numel(), _stream()),
           "kg_bn_bwd_many")
    return coefs


def adam_step(p, g, m, v, lr, b1, b2, eps, step_t: torch.Tensor, grad_scale: float = 1.0, zero_grad: bool = False):
    """Flat-buffer Adam; ``step_t`` holds the 1-based number of THIS step (the caller has incremented it).  ``zero_grad``
    (kg_adam_step_fused): the launch also clears the gradient it has consumed."""
    lib = load_library()
    _need_cuda(p, g, m, v, step_t)
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    assert step_t.dtype == torch.int32
    if zero_grad:
        _check(lib.kg_adam_step_fused(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, b1, b2, eps,
                                      step_t.data_ptr(), grad_scale, 1, _stream()), "kg_adam_step_fused")
        return
    _check(lib.kg_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(),
                            lr, b1, b2, eps, step_t.data_ptr(), grad_scale, _stream()), "kg_adam_step")


# ---- container-level fusions around the discriminator's blocks (kg_disc.hip) ---------------------------------------------

def _head_args(h, w):
    h = as_plane(h)
    a = _HeadArgs()
    a.N, a.C, a.T, a.V = h.shape
    a.h = h.data_ptr()
    a.h_sN, a.h_sC = _sn_sc(h)
    w = w.reshape(-1)
    assert w.is_contiguous() and w.numel() == h.shape[1]
    a.w = w.data_ptr()
    return a, h, w


def head_fwd(h: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """v (N,) = b + mean_{t,v}(h) @ w: global average pool + Linear(latent, 1) (kg_head_fwd)"""
    lib = load_library()
    _need_cuda(h, w, b)
    a, h, w = _head_args(h, w)
    a.b = _ptr(b)
    v = torch.empty(h.shape[0], dtype=torch.float32, device=h.device)
    a.v = v.data_ptr()
    _check(lib.kg_head_fwd(C.byref(a), _stream()), "kg_head_fwd")
    return v


def head_bwd(gv: torch.Tensor, w: torch.Tensor, h: torch.Tensor, masked: bool = True, slope: float = 0.2) -> torch.Tensor:
    """(N, C, T, V) top gradient gv[n] w[c] / (T V), times lrelu'(h) when ``masked`` (kg_head_bwd)"""
    lib = load_library()
    gv = gv.reshape(-1).contiguous()
    _need_cuda(gv, w, h)
    a, h, w = _head_args(h, w)
    assert gv.numel() == h.shape[0]
    a.gv = gv.data_ptr()
    a.masked, a.slope = int(bool(masked)), slope
    g = new_plane(*h.shape, h.device)
    a.g = g.data_ptr()
    a.g_sN, a.g_sC = _sn_sc(g)
    _check(lib.kg_head_bwd(C.byref(a), _stream()), "kg_head_bwd")
    return g


def head_wgrad(x: torch.Tensor, gv: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor], accumulate: bool = True):
    """dw (C,) (+)= sum_n gv[n] mean_{t,v} x[n]; db (1,) (+)= sum_n gv[n] (kg_head_wgrad)"""
    lib = load_library()
    gv = gv.reshape(-1).contiguous()
    _need_cuda(x, gv, dw, db)
    a, x, _ = _head_args(x, dw)
    assert gv.numel() == x.shape[0] and dw.is_contiguous() and (db is None or db.numel() == 1)
    a.w = None
    a.gv = gv.data_ptr()
    a.dw, a.db, a.accumulate = dw.data_ptr(), _ptr(db), int(bool(accumulate))
    _check(lib.kg_head_wgrad(C.byref(a), _stream()), "kg_head_wgrad")


def _label_bias_args(labels, emb, wg, K, C_out, cin, J, ak):
    a = _LabelBiasArgs()
    k, v, w = ak.shape
    assert k == K and ak.is_contiguous() and emb.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous()
    a.N, a.L, a.J, a.K, a.C, a.V, a.W = labels.numel(), emb.shape[0], J, K, C_out, v, w
    a.labels, a.emb = labels.data_ptr(), emb.data_ptr()
    assert wg.is_contiguous()
    a.w, a.w_sK, a.w_sC = wg.data_ptr(), C_out * cin, cin
    a.ak = ak.data_ptr()
    return a


def label_bias_fwd(labels, emb, wg, K: int, C_out: int, cin: int, J: int, ak) -> torch.Tensor:
    """zl (N, C_out, 1, W): the bias the J label channels in front of block 0's input add to its gcn output
    (kg_label_bias_fwd).  wg: the gcn weight (K*C_out, cin, 1, 1) whose first J input columns multiply them."""
    lib = load_library()
    _need_cuda(labels, emb, wg, ak)
    a = _label_bias_args(labels, emb, wg, K, C_out, cin, J, ak)
    zl = torch.empty((labels.numel(), C_out, 1, ak.shape[2]), dtype=torch.float32, device=emb.device)
    a.zl = zl.data_ptr()
    nbytes = lib.kg_label_bias_workspace_bytes(C.byref(a))
    if nbytes < 0:
        _check(-1, "kg_label_bias_workspace_bytes")
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=emb.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    _check(lib.kg_label_bias_fwd(C.byref(a), _stream()), "kg_label_bias_fwd")
    return zl


def label_bias_bwd(gz, labels, emb, wg, K: int, C_out: int, cin: int, J: int, ak, demb=None, dw=None, dak=None,
                   accumulate: bool = True, dak_accumulate: bool = True):
    """First-order gradients of label_bias_fwd from gz (N, C_out, T, W) = d loss / d(gcn output): demb (L, J), dw (the gcn
    weight's gradient buffer, same addressing as wg) and dak (K, V, W) are written or (+)= (kg_label_bias_bwd)."""
    lib = load_library()
    gz = as_plane(gz)
    _need_cuda(gz, labels, emb, wg, ak, demb, dw, dak)
    a = _label_bias_args(labels, emb, wg, K, C_out, cin, J, ak)
    assert gz.shape[0] == labels.numel() and gz.shape[1] == C_out and gz.shape[3] == ak.shape[2]
    a.T = gz.shape[2]
    a.gz = gz.data_ptr()
    a.gz_sN, a.gz_sC = _sn_sc(gz)
    for t in (demb, dw, dak):
        assert t is None or t.is_contiguous()
    a.demb, a.dw, a.dak = _ptr(demb), _ptr(dw), _ptr(dak)
    a.accumulate, a.dak_accumulate = int(bool(accumulate)), int(bool(dak_accumulate))
    nbytes = lib.kg_label_bias_workspace_bytes(C.byref(a))
    if nbytes < 0:
        _check(-1, "kg_label_bias_workspace_bytes")
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=gz.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    _check(lib.kg_label_bias_bwd(C.byref(a), _stream()), "kg_label_bias_bwd")


def mix3(real: torch.Tensor, fake: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """(3n, C, T, V) = [real | fake | alpha real + (1 - alpha) fake], stored NCHW-contiguous (kg_mix3)"""
    lib = load_library()
    real, fake = as_plane(real), as_plane(fake)
    alpha = alpha.reshape(-1).contiguous()
    _need_cuda(real, fake, alpha)
    n, c, t, v = real.shape
    assert tuple(fake.shape) == (n, c, t, v) and alpha.numel() == n
    out = torch.empty((3 * n, c, t, v), dtype=torch.float32, device=real.device)
    a = _MixArgs()
    a.N, a.C, a.T, a.V = n, c, t, v
    a.real, a.fake, a.alpha, a.out = real.data_ptr(), fake.data_ptr(), alpha.data_ptr(), out.data_ptr()
    a.r_sN, a.r_sC = _sn_sc(real)
    a.f_sN, a.f_sC = _sn_sc(fake)
    a.o_sN, a.o_sC = c * t * v, t * v
    _check(lib.kg_mix3(C.byref(a), _stream()), "kg_mix3")
    return out


def masked_adj_fwd(A_all: torch.Tensor, imp_all: Optional[torch.Tensor], sel: Optional[torch.Tensor]) -> torch.Tensor:
    """ak = (A_all * imp_all)[sel] (flat; kg_masked_adj_fwd)"""
    lib = load_library()
    _need_cuda(A_all, imp_all, sel)
    n = sel.numel() if sel is not None else A_all.numel()
    ak = torch.empty(n, dtype=torch.float32, device=A_all.device)
    a = _MaskedAdjArgs()
    a.n = n
    a.a, a.imp, a.sel, a.ak = A_all.data_ptr(), _ptr(imp_all), _ptr(sel), ak.data_ptr()
    assert A_all.is_contiguous() and (imp_all is None or imp_all.is_contiguous()) and (sel is None or (sel.dtype == torch.int64 and sel.is_contiguous()))
    _check(lib.kg_masked_adj_fwd(C.byref(a), _stream()), "kg_masked_adj_fwd")
    return ak


def masked_adj_bwd(g: torch.Tensor, A_all: torch.Tensor, sel: Optional[torch.Tensor], dimp: torch.Tensor, accumulate: bool):
    """dimp[sel] (+)= g * A_all[sel] (kg_masked_adj_bwd); elements outside sel are not touched"""
    lib = load_library()
    g = g.reshape(-1).contiguous()
    _need_cuda(g, A_all, sel, dimp)
    a = _MaskedAdjArgs()
    a.n = g.numel()
    assert dimp.is_contiguous() and dimp.numel() == A_all.numel() and (sel is None or sel.numel() == g.numel())
    a.a, a.sel, a.g, a.dimp, a.accumulate = A_all.data_ptr(), _ptr(sel), g.data_ptr(), dimp.data_ptr(), int(bool(accumulate))
    _check(lib.kg_masked_adj_bwd(C.byref(a), _stream()), "kg_masked_adj_bwd")


# ---- label embedding + mapping network (kg_linear_fwd / kg_linear_bwd / kg_embed_bwd; generator.py:22-37,80-85) -------

def _linear_args(x, w, emb, labels):
    a = _LinearArgs()
    assert w.dim() == 2 and w.is_contiguous()
    a.Dout, a.Din = w.shape
    a.w = w.data_ptr()
    J = 0
    if emb is not None:
        assert emb.dim() == 2 and emb.is_contiguous() and labels is not None and labels.dtype == torch.int64 and labels.is_contiguous()
        a.L, J = emb.shape
        a.emb, a.labels = emb.data_ptr(), labels.data_ptr()
    a.J = J
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] == a.Din - J, (tuple(x.shape), a.Din, J)
    a.N = x.shape[0]
    a.x, a.x_ld = x.data_ptr(), x.stride(0)
    return a


def linear_fwd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int = ACT_LRELU, slope: float = 0.2,
               emb: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(cat(emb[labels], x) @ w.T + b): one layer of the mapping network, the embedding lookup + cat of the first layer
    folded into its operand load (kg_linear_fwd)"""
    lib = load_library()
    _need_cuda(x, w, b, emb, labels)
    a = _linear_args(x, w, emb, labels)
    y = torch.empty(a.N, a.Dout, dtype=torch.float32, device=x.device)
    a.bias = _ptr(b)
    a.y, a.y_ld = y.data_ptr(), a.Dout
    a.act, a.slope = act, slope
    _check(lib.kg_linear_fwd(C.byref(a), _stream()), "kg_linear_fwd")
    return y


def linear_bwd(g: torch.Tensor, y: torch.Tensor, x: torch.Tensor, w: torch.Tensor, act: int = ACT_LRELU, slope: float = 0.2,
               emb: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None, gx_cols: Optional[int] = None,
               dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None, accumulate: bool = False):
    """One layer's backward pass in ONE launch (kg_linear_bwd): returns gx (N, gx_cols) (None for 0 columns; default: all
    Din); writes / adds dw (Dout, Din) and db (Dout) when given."""
    lib = load_library()
    _need_cuda(g, y, x, w, emb, labels, dw, db)
    a = _linear_args(x, w, emb, labels)
    assert g.shape == (a.N, a.Dout) and y.shape == (a.N, a.Dout) and g.stride(1) == 1 and y.stride(1) == 1
    a.g, a.g_ld = g.data_ptr(), g.stride(0)
    a.y, a.y_ld = y.data_ptr(), y.stride(0)
    a.act, a.slope = act, slope
    cols = a.Din if gx_cols is None else gx_cols
    gx = None
    if cols > 0:
        gx = torch.empty(a.N, cols, dtype=torch.float32, device=g.device)
        a.gx, a.gx_ld = gx.data_ptr(), cols
    a.gx_cols = cols
    if dw is not None:
        assert dw.is_contiguous() and dw.numel() == a.Dout * a.Din
    if db is not None:
        assert db.is_contiguous() and db.numel() == a.Dout
    a.dw, a.db, a.accumulate = _ptr(dw), _ptr(db), int(bool(accumulate))
    _check(lib.kg_linear_bwd(C.byref(a), _stream()), "kg_linear_bwd")
    return gx


def embed_bwd(gx: torch.Tensor, labels: torch.Tensor, demb: torch.Tensor, accumulate: bool = False):
    """demb[l] (+)= sum of gx[n] over the samples of class l, in index order (kg_embed_bwd)"""
    lib = load_library()
    _need_cuda(gx, labels, demb)
    a = _LinearArgs()
    assert gx.dim() == 2 and gx.stride(1) == 1 and demb.dim() == 2 and demb.is_contiguous() and demb.shape[1] <= gx.shape[1]
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    a.N = gx.shape[0]
    a.L, a.J = demb.shape
    a.gx, a.gx_ld = gx.data_ptr(), gx.stride(0)
    a.labels, a.demb, a.accumulate = labels.data_ptr(), demb.data_ptr(), int(bool(accumulate))
    _check(lib.kg_embed_bwd(C.byref(a), _stream()), "kg_embed_bwd")


# ---- data-parallel gradient exchange (kg_comm_*: RCCL over xGMI behind the C ABI) -------------------------------------

COMM_ID_BYTES = 128


class Comm:
    """One RCCL communicator owned through the C ABI (kg_comm_init / kg_allreduce_flat / kg_comm_destroy).
    ``exchange(id_bytes | None) -> id_bytes``: ships rank 0's unique id to every rank (any side channel: a
    torch.distributed object broadcast, a file, an MPI bcast ...).  ``world == 1`` needs no exchange."""

    def __init__(self, rank: int, world: int, device: int, exchange=None):
        lib = load_library()
        buf = C.create_string_buffer(COMM_ID_BYTES)
        if rank == 0:
            _check(lib.kg_comm_unique_id(buf), "kg_comm_unique_id")
        if world > 1:
            if exchange is None:
                raise ValueError("Comm: world > 1 needs an `exchange` callable for the unique id")
            raw = exchange(bytes(buf.raw) if rank == 0 else None)
            if not isinstance(raw, (bytes, bytearray)) or len(raw) != COMM_ID_BYTES:
                raise ValueError("Comm: exchange() must return rank 0's %d id bytes" % COMM_ID_BYTES)
            buf = C.create_string_buffer(bytes(raw), COMM_ID_BYTES)
        self._h = C.c_void_p()
        _check(lib.kg_comm_init(C.byref(self._h), rank, world, buf, device), "kg_comm_init")
        self.rank, self.world, self.device = rank, world, device
        self.force = False      # tests: issue the collective even with one rank (RCCL then copies in place)

    def allreduce_(self, flat: torch.Tensor) -> torch.Tensor:
        """flat <- sum over ranks (in place), enqueued on torch's current stream."""
        if self._h is None:
            raise RuntimeError("Comm: destroyed")
        _need_cuda(flat)
        if flat.dtype != torch.float32 or not flat.is_contiguous():
            raise ValueError("Comm.allreduce_: contiguous fp32 buffer expected")
        _check(load_library().kg_allreduce_flat(self._h, flat.data_ptr(), flat.numel(), _stream()), "kg_allreduce_flat")
        return flat

    def destroy(self):
        if self._h is not None:
            h, self._h = self._h, None
            _check(load_library().kg_comm_destroy(h), "kg_comm_destroy")

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def torch_dist_exchange(src: int = 0):
    """`exchange` callable for Comm over an initialised torch.distributed process group (gloo or nccl)."""
    import torch.distributed as dist

    def ex(raw):
        box = [raw]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return ex
