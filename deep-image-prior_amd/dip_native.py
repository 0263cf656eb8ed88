"""ctypes binding of libdip_hip.so (C ABI: include/dip_hip.h).

The product path has NO fallback: if the shared library is missing, `lib()` raises.  Build it
with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950; the one build recipe).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdip_hip.so")
ABI_VERSION = 8

PAD_ZERO, PAD_REFLECT, PAD_REPLICATE = 0, 1, 2
UP_NEAREST, UP_BILINEAR = 0, 1

c_float_p = C.c_void_p  # device pointers travel as raw addresses


class DipTransform(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("slope", C.c_float)]


class DipPackRec(C.Structure):
    _fields_ = [("w_off", C.c_int64), ("fwd_off", C.c_int64), ("dgrad_off", C.c_int64),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KS", C.c_int32),
                ("CinP4", C.c_int32), ("CoutP32", C.c_int32), ("CoutP4", C.c_int32), ("CinP32", C.c_int32)]


class DipPackRec3(C.Structure):
    _fields_ = [("w_off", C.c_int64), ("fwd_off", C.c_int64), ("dgrad_off", C.c_int64),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("KS", C.c_int32),
                ("nchF", C.c_int32), ("CoutP32", C.c_int32), ("nchD", C.c_int32), ("CinP32", C.c_int32)]


class DipBnFin(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float),
                ("state", C.c_void_p), ("Cs", C.c_int32), ("C", C.c_int32), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("ticket", C.c_void_p)]


class DipBnbFin(C.Structure):
    _fields_ = [("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("coef", C.c_void_p), ("C", C.c_int32),
                ("npix", C.c_int32), ("ticket", C.c_void_p)]


class DipConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cx", C.c_int32), ("Cin", C.c_int32),
                ("tr", DipTransform), ("wp", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p),
                ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cy", C.c_int32), ("Cout", C.c_int32),
                ("y_pitch", C.c_int32),
                ("ks", C.c_int32), ("stride", C.c_int32), ("pad_mode", C.c_int32), ("off", C.c_int32),
                ("dil", C.c_int32), ("accumulate", C.c_int32), ("stats", C.c_void_p),
                ("ksplit", C.c_int32), ("ws", C.c_void_p),
                # fused phase 1 of the BatchNorm backward of the conv's INPUT activation (data-gradient launches)
                ("bnb_y", C.c_void_p), ("bnb_state", C.c_void_p), ("bnb_partials", C.c_void_p),
                ("bnb_partials_thin", C.c_void_p), ("bnb_Cy", C.c_int32), ("bnb_Cs", C.c_int32),
                ("bnb_pad", C.c_int32), ("bnb_slope", C.c_float),
                ("wp3", C.c_void_p)]            # three-bf16-plane weights (bf16-pipe convolution), or None


class DipWgradDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cx", C.c_int32), ("Cin", C.c_int32),
                ("tr", DipTransform), ("dy", C.c_void_p),
                ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cdy", C.c_int32), ("Cout", C.c_int32),
                ("ks", C.c_int32), ("stride", C.c_int32), ("pad_mode", C.c_int32), ("off", C.c_int32),
                ("partial", C.c_void_p), ("bias_partial", C.c_void_p), ("nsplit", C.c_int32),
                ("tap_groups", C.c_int32), ("chan_block", C.c_int32)]


class DipGradSrc(C.Structure):
    _fields_ = [("g", C.c_void_p), ("pad", C.c_int32), ("fold", C.c_int32), ("Cg", C.c_int32), ("choff", C.c_int32),
                ("win_y", C.c_int32), ("win_x", C.c_int32), ("win_h", C.c_int32), ("win_w", C.c_int32),
                ("tw", C.c_void_p), ("tn", C.c_int32), ("tcw", C.c_int32)]      # thin 1x1 conv in front (round 6), or None


class DipUpcatDesc(C.Structure):
    _fields_ = [("s", C.c_void_p), ("Cs_s", C.c_int32), ("ns", C.c_int32), ("ts", DipTransform),
                ("d", C.c_void_p), ("Cs_d", C.c_int32), ("nd", C.c_int32), ("td", DipTransform),
                ("H", C.c_int32), ("W", C.c_int32), ("mode", C.c_int32),
                ("cat", C.c_void_p), ("Cs_cat", C.c_int32),
                ("stats", C.c_void_p), ("nblk", C.c_int32),
                ("Hs", C.c_int32), ("Ws", C.c_int32), ("os_y", C.c_int32), ("os_x", C.c_int32),
                ("Hd", C.c_int32), ("Wd", C.c_int32), ("od_y", C.c_int32), ("od_x", C.c_int32)]


class DipCmd(C.Structure):
    _fields_ = [("kind", C.c_int32), ("fn", C.c_int32), ("stream", C.c_int32), ("event", C.c_int32),
                ("slots", C.POINTER(C.c_uint64)), ("nslots", C.c_int32), ("reserved", C.c_int32)]


class DipIterState(C.Structure):
    _fields_ = [("step", C.c_uint64), ("step_size", C.c_float), ("bc2_sqrt", C.c_float)]


class DipLossHeadDesc(C.Structure):
    _fields_ = [("u", C.c_void_p), ("Cu", C.c_int32), ("Cin", C.c_int32), ("tr", DipTransform),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("Cout", C.c_int32), ("HW", C.c_int32),
                ("sigmoid", C.c_int32), ("target", C.c_void_p), ("mask", C.c_void_p), ("mask_c", C.c_int32),
                ("out", C.c_void_p), ("partials", C.c_void_p), ("nblk", C.c_int32), ("loss", C.c_void_p)]


_SIGS = {
    "dip_abi_version": (C.c_int, []),
    "dip_build_id": (C.c_char_p, []),
    "dip_last_error": (C.c_char_p, []),
    "dip_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "dip_list_fn_id": (C.c_int, [C.c_char_p]),
    "dip_list_fn_nargs": (C.c_int, [C.c_int]),
    "dip_list_run": (C.c_int, [C.POINTER(DipCmd), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int,
                               C.POINTER(C.c_int)]),
    "dip_events_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dip_events_destroy": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dip_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dip_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dip_head_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dip_head_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dip_pack_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dip_pack_weights_bf3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]),
    "dip_conv_bf3_eligible": (C.c_int, [C.POINTER(DipConvDesc)]),
    "dip_conv_bf3_cols": (C.c_int, [C.POINTER(DipConvDesc), C.c_int, C.c_int, C.c_void_p]),
    "dip_conv_bf3_terms": (C.c_int, []),
    "dip_conv_bf3_set_terms": (C.c_int, [C.c_int]),
    "dip_conv_igemm": (C.c_int, [C.POINTER(DipConvDesc), C.c_void_p]),
    "dip_conv_ntiles": (C.c_int, [C.c_int, C.c_int]),
    "dip_conv_variant": (C.c_int, [C.POINTER(DipConvDesc)]),
    "dip_conv_bnb_fusable": (C.c_int, [C.POINTER(DipConvDesc)]),
    "dip_conv_thin4_ntiles": (C.c_int, [C.c_int, C.c_int]),
    "dip_conv_splitk_finish": (C.c_int, [C.POINTER(DipConvDesc), C.c_void_p]),
    "dip_conv_thin": (C.c_int, [C.POINTER(DipConvDesc), C.c_void_p]),
    "dip_conv_thin_eligible": (C.c_int, [C.POINTER(DipConvDesc)]),
    "dip_conv_thin_shape_ok": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dip_conv_small": (C.c_int, [C.POINTER(DipConvDesc), C.c_void_p]),
    "dip_conv_small_eligible": (C.c_int, [C.POINTER(DipConvDesc)]),
    "dip_conv_small_rows": (C.c_int, [C.POINTER(DipConvDesc)]),
    "dip_conv_dgrad_ring": (C.c_int, [C.POINTER(DipConvDesc), C.c_void_p]),
    "dip_conv_dgrad_ring_ok": (C.c_int, [C.POINTER(DipConvDesc)]),
    "dip_conv_thin4": (C.c_int, [C.POINTER(DipConvDesc), C.c_int, C.c_void_p]),
    "dip_conv_igemm_dma_cols": (C.c_int, [C.POINTER(DipConvDesc), C.c_int, C.c_void_p]),
    "dip_conv_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "dip_conv_plan_fp32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(C.c_int64)]),
    "dip_conv_plan_dil2": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "dip_conv_wgrad": (C.c_int, [C.POINTER(DipWgradDesc), C.c_void_p]),
    "dip_wgrad_bf3_eligible": (C.c_int, [C.POINTER(DipWgradDesc)]),
    "dip_wgrad_bf3": (C.c_int, [C.POINTER(DipWgradDesc), C.c_void_p]),
    "dip_conv_wgrad_tail": (C.c_int, [C.POINTER(DipWgradDesc), C.c_void_p]),
    "dip_wgrad_tail_stream_ok": (C.c_int, [C.POINTER(DipWgradDesc)]),
    "dip_wgrad_tail_stream": (C.c_int, [C.POINTER(DipWgradDesc), C.c_void_p]),
    "dip_conv_wgrad_ntiles": (C.c_int, [C.c_int, C.c_int]),
    "dip_wgrad_thin": (C.c_int, [C.POINTER(DipWgradDesc), C.c_void_p]),
    "dip_wgrad_thin_eligible": (C.c_int, [C.POINTER(DipWgradDesc)]),
    "dip_wgrad_thin_shape_ok": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dip_wgrad_thin_nsplit": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dip_wgrad_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "dip_wgrad_plan2": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dip_wgrad_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "dip_bn_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                  C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dip_bn_bwd_stats": (C.c_int, [C.POINTER(DipGradSrc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p]),
    "dip_bn_bwd_nblk": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "dip_bn_bwd_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "dip_bn_bwd_finalize2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dip_bn_bwd_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "dip_bn_bwd_apply_src": (C.c_int, [C.POINTER(DipGradSrc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dip_bn_bwd_fin_rows_ok": (C.c_int, [C.c_int, C.c_int]),
    "dip_bn_bwd_apply_fin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dip_bn_bwd_apply_src_fin": (C.c_int, [C.POINTER(DipGradSrc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_void_p]),
    "dip_bn_bwd_one_ok": (C.c_int, [C.c_int, C.c_int]),
    "dip_bn_bwd_one": (C.c_int, [C.POINTER(DipGradSrc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                 C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dip_upsample_bwd_one": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dip_fold_to_nchw": (C.c_int, [C.POINTER(DipGradSrc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dip_fold_to_nhwc": (C.c_int, [C.POINTER(DipGradSrc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dip_upcat_fwd": (C.c_int, [C.POINTER(DipUpcatDesc), C.c_void_p]),
    "dip_upcat_nblk": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "dip_upcat_fwd_fin": (C.c_int, [C.POINTER(DipUpcatDesc), C.POINTER(DipBnFin), C.c_void_p]),
    "dip_fin_rows_ok": (C.c_int, [C.c_int, C.c_int]),
    "dip_bn_bwd_stats_fin": (C.c_int, [C.POINTER(DipGradSrc), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.POINTER(DipBnbFin), C.c_void_p]),
    "dip_upsample_bwd_stats_crop_fin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                  C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                  C.POINTER(DipBnbFin), C.c_void_p]),
    "dip_avgpool2_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p]),
    "dip_avgpool2_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dip_maxpool2_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p]),
    "dip_maxpool2_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p]),
    "dip_upsample_bwd_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.c_void_p]),
    "dip_upsample_bwd_stats_crop": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                              C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dip_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "dip_noise_axpy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p]),
    "dip_adam_tick": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "dip_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                    C.c_double, C.c_void_p, C.c_void_p]),
    "dip_noise_axpy_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "dip_noise_axpy_dev2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "dip_counter_add": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "dip_group_begin": (C.c_int, [C.c_int, C.c_longlong, C.c_void_p, C.c_longlong]),
    "dip_group_end": (C.c_int, []),
    "dip_group_size": (C.c_int, []),
    "dip_group_native": (C.c_int, [C.c_int]),
    "dip_loss_head_nblk": (C.c_int, [C.c_int, C.c_int]),
    "dip_loss_head_fwd": (C.c_int, [C.POINTER(DipLossHeadDesc), C.c_void_p]),
    "dip_loss_head_bwd": (C.c_int, [C.POINTER(DipLossHeadDesc), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dip_fit_monitor_nblk": (C.c_int, [C.c_int64]),
    "dip_fit_monitor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "dip_arena_backtrack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "dip_lanczos_down_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    "dip_lanczos_down_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p]),
    "dip_down_dense_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p]),
    "dip_down_dense_bwd_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p]),
    "dip_down_dense_bwd_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

EXPORTS = tuple(_SIGS.keys())
_lib = None


def lib():
    """Load libdip_hip.so once; fail loudly if it is not built (no CPU / eager fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the deep-image-prior MI355X backend has no fallback path. "
                "Build it with `python __graft_entry__.py build` (needs hipcc, targets gfx950).")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError -> missing symbol: loud
            fn.restype = res
            fn.argtypes = args
        if L.dip_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libdip_hip.so ABI {L.dip_abi_version()} != binding ABI {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().dip_last_error()
        raise RuntimeError(f"libdip_hip {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def conv_plan(Hout, Wout, Cin, Cout, ks, stride):
    """(ksplit, stats_rows, ws_floats) of dip_conv_plan."""
    k, rows, wsf = C.c_int(), C.c_int(), C.c_int64()
    check(lib().dip_conv_plan(Hout, Wout, Cin, Cout, ks, stride, C.byref(k), C.byref(rows), C.byref(wsf)), "conv_plan")
    return k.value, rows.value, wsf.value


def conv_plan_fp32(Hout, Wout, Cin, Cout, ks, stride):
    """(ksplit, stats_rows, ws_floats) of dip_conv_plan_fp32: the plan of a layer the bf16-pipe kernel does not take."""
    k, rows, wsf = C.c_int(), C.c_int(), C.c_int64()
    check(lib().dip_conv_plan_fp32(Hout, Wout, Cin, Cout, ks, stride, C.byref(k), C.byref(rows), C.byref(wsf)), "conv_plan_fp32")
    return k.value, rows.value, wsf.value


def conv_plan_dil2(Hout, Wout, Cin, Cout, ks):
    """(ksplit, stats_rows, ws_floats) of dip_conv_plan_dil2 (data gradient of a stride-2 convolution)."""
    k, rows, wsf = C.c_int(), C.c_int(), C.c_int64()
    check(lib().dip_conv_plan_dil2(Hout, Wout, Cin, Cout, ks, C.byref(k), C.byref(rows), C.byref(wsf)), "conv_plan_dil2")
    return k.value, rows.value, wsf.value


def wgrad_plan(Hout, Wout, Cin, Cout, ks, stride):
    n = C.c_int()
    check(lib().dip_wgrad_plan(Hout, Wout, Cin, Cout, ks, stride, C.byref(n)), "wgrad_plan")
    return n.value


def wgrad_plan2(Hout, Wout, Cin, Cout, ks, stride):
    """(nsplit, tap_groups, chan_block) of dip_wgrad_plan2."""
    n, g, cb = C.c_int(), C.c_int(), C.c_int()
    check(lib().dip_wgrad_plan2(Hout, Wout, Cin, Cout, ks, stride, C.byref(n), C.byref(g), C.byref(cb)), "wgrad_plan2")
    return n.value, g.value, cb.value


# ------------------------------------------------------------------------------------------------------------------
# Command lists (include/dip_hip.h "command lists", csrc/dip_list.hip): a static launch list compiled ONCE into an array of
# DipCmd and issued by one dip_list_run call per direction and iteration.
# ------------------------------------------------------------------------------------------------------------------
CMD_LAUNCH, CMD_RECORD, CMD_WAIT = 0, 1, 2
_INT_TYPES = (C.c_int, C.c_int32, C.c_int64, C.c_longlong, C.c_uint64, C.c_uint32, C.c_long, C.c_ulong)


def _slot(argtype, v) -> int:
    """One argument as the 64-bit pattern of its slot."""
    import struct
    if argtype in _INT_TYPES:
        return int(v) & 0xFFFFFFFFFFFFFFFF
    if argtype is C.c_float:
        return int.from_bytes(struct.pack("<f", float(v)), "little")
    if argtype is C.c_double:
        return int.from_bytes(struct.pack("<d", float(v)), "little")
    # pointers: None, an address, a ctypes.byref(struct) or a ctypes object
    if v is None:
        return 0
    if isinstance(v, int):
        return v & 0xFFFFFFFFFFFFFFFF
    obj = getattr(v, "_obj", None)              # ctypes.byref(x)
    if obj is not None:
        return C.addressof(obj)
    if isinstance(v, (C.c_void_p, C.c_char_p)):
        return int(v.value or 0)
    return C.addressof(v)


class CmdList:
    """A compiled launch list.  cmds: sequence of ("launch", fn, args, stream_index, name) | ("record", event_index,
    stream_index) | ("wait", stream_index, event_index); fn = a function of lib(), args WITHOUT the trailing stream.
    run(stream_ptrs) issues the whole list; one set of HIP events per key (eager / capture runs must not share events)."""

    def __init__(self, cmds):
        L = lib()
        self.names = []
        self._keep = []
        n = len(cmds)
        self.arr = (DipCmd * max(n, 1))()
        self.nevents = 0
        for i, c in enumerate(cmds):
            if c[0] == "launch":
                _, fn, args, st, name = c
                fname = fn.__name__
                fid = L.dip_list_fn_id(fname.encode())
                if fid < 0:
                    raise RuntimeError(f"dip-amd: {fname} is not a command-list entry point")
                argtypes = _SIGS[fname][1]
                if len(args) + 1 != len(argtypes) or L.dip_list_fn_nargs(fid) != len(argtypes):
                    raise RuntimeError(f"dip-amd: {fname}: {len(args)} arguments for {len(argtypes) - 1} parameters")
                slots = (C.c_uint64 * len(argtypes))()
                for k, (t, v) in enumerate(zip(argtypes[:-1], args)):
                    slots[k] = _slot(t, v)
                self._keep.append((slots, args))          # the slots point INTO ctypes structs owned by `args`
                self.arr[i] = DipCmd(CMD_LAUNCH, fid, st, -1, C.cast(slots, C.POINTER(C.c_uint64)), len(argtypes), 0)
                self.names.append(name)
            elif c[0] == "record":
                self.arr[i] = DipCmd(CMD_RECORD, -1, c[2], c[1], None, 0, 0)
                self.nevents = max(self.nevents, c[1] + 1)
                self.names.append("record")
            else:
                self.arr[i] = DipCmd(CMD_WAIT, -1, c[1], c[2], None, 0, 0)
                self.nevents = max(self.nevents, c[2] + 1)
                self.names.append("wait")
        self.n = n
        self._events = {}
        self._failed = C.c_int(-1)

    def _event_set(self, key):
        ev = self._events.get(key)
        if ev is None:
            ev = (C.c_void_p * max(self.nevents, 1))()
            if self.nevents:
                check(lib().dip_events_create(ev, self.nevents), "events_create")
            self._events[key] = ev
        return ev

    def run(self, stream_ptrs, key="eager"):
        ns = len(stream_ptrs)
        st = (C.c_void_p * ns)(*stream_ptrs)
        ev = self._event_set(key)
        rc = lib().dip_list_run(self.arr, self.n, st, ns, ev, self.nevents, C.byref(self._failed))
        if rc:
            k = self._failed.value
            check(rc, self.names[k] if 0 <= k < self.n else "list_run")

    def __del__(self):
        try:
            for ev in self._events.values():
                if self.nevents:
                    lib().dip_events_destroy(ev, self.nevents)
        except Exception:
            pass
