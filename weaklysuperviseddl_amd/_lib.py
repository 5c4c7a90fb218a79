"""ctypes binding of libwsdl_hip.so (C ABI declared in include/wsdl_hip.h).

The product path has no CPU fallback: if the shared library is missing or a tensor is not a device
tensor the call raises.  ``oracle/`` is never imported from here.
"""
import ctypes as C
import os

import torch  # noqa: F401  (first: libwsdl_hip.so must bind to the HIP runtime PyTorch-ROCm has loaded, not to a
#                            second copy of libamdhip64 - two runtimes in one process do not share devices / streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
# WSDL_LIB: another build of the same sources (kernel A/B experiments); the default is the in-tree library
LIB_PATH = os.environ.get("WSDL_LIB") or os.path.join(_HERE, "csrc", "libwsdl_hip.so")

_vp, _i, _ll, _f, _sz, _u64 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.c_ulonglong

# name -> (restype, argtypes); mirrors include/wsdl_hip.h one to one
SIGNATURES = {
    "wsdl_last_error": (C.c_char_p, []),
    "wsdl_launch_trace": (_i, [_i]),
    "wsdl_last_launches": (C.c_char_p, []),
    "wsdl_version": (_i, []),
    "wsdl_target_arch": (C.c_char_p, []),
    "wsdl_set_option": (_i, [C.c_char_p, _i]),
    "wsdl_prof_enable": (_i, [_i]),
    "wsdl_prof_collect": (_i, [_i, C.POINTER(_ll)] + [C.POINTER(C.c_double)] * 4),
    "wsdl_prof_reset": (_i, []),
    "wsdl_prof_class_name": (C.c_char_p, [_i]),
    "wsdl_range_enable": (_i, [_i]),
    "wsdl_range_push": (_i, [C.c_char_p]),
    "wsdl_range_pop": (_i, []),
    "wsdl_conv2d_weight_layout_bytes": (_sz, [_i, _i, _i, _i, _i, C.POINTER(_i)]),
    "wsdl_conv2d_prep_weights": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "wsdl_conv2d_fwd": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_vp, _vp, _vp, _i, _ll, _ll, _ll, _vp, _vp, _vp, _sz, _vp]),
    "wsdl_conv2d_igemm_workspace": (_sz, [_i] * 11),
    "wsdl_conv2d_prep_weights_multi": (_i, [_vp, _i, _i, _vp]),
    "wsdl_conv2d_dgrad": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_i, _vp, _ll, _vp, _vp, _sz, _vp]),
    "wsdl_conv2d_fwd_group_ok": (_i, [_i] * 6),
    "wsdl_conv2d_fwd_group_workspace": (_sz, [_i, C.POINTER(_i), C.POINTER(_i), _i, _i, _i, _i, _i]),
    "wsdl_conv2d_fwd_group": (_i, [_i, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), _i, _i, _i, _i, _i, _ll,
                                   C.POINTER(_ll), _vp, _vp, _sz, _vp]),
    "wsdl_conv2d_dgrad_multi_ok": (_i, [_i] * 6),
    "wsdl_conv2d_dgrad_multi": (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i),
                                     C.POINTER(_ll), _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "wsdl_conv2d_wgrad_workspace": (_sz, [_i] * 10),
    "wsdl_conv2d_wgrad": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_i, _ll, _ll, _vp, _vp, _vp, _sz, _vp]),
    "wsdl_conv2d_wgrad_deferred": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_i, _ll, _ll, _vp, _vp, _vp, _sz, _vp, _vp]),
    "wsdl_wgrad_reduce_multi": (_i, [_vp, _i, _i, _vp]),
    "wsdl_conv2d_wgrad_presplit_bytes": (_sz, [_i] * 10),
    "wsdl_conv2d_wgrad_ex": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_i, _ll, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "wsdl_amax": (_i, [_vp, _i, _ll, _ll, _vp, _i, _vp]),
    "wsdl_multi_amax": (_i, [_vp, _vp, _i, _vp, _vp]),
    "wsdl_bias_grad": (_i, [_vp, _vp, _i, _i, _i, _ll, _i, _vp]),
    "wsdl_bn_workspace": (_sz, [_i]),
    "wsdl_bn_train_fwd": (_i, [_vp] * 8 + [_f, _f, _i, _i, _i, _vp, _i, _ll, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "wsdl_bn_train_bwd": (_i, [_vp] * 11 + [_i, _i, _i, _i, _i, _ll, _ll, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "wsdl_bn_channel_resident": (_i, [_i, _i, _i, _i]),
    "wsdl_bn_fold": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "wsdl_affine_act_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "wsdl_affine_act_bwd": (_i, [_vp] * 5 + [_i, _i, _i, _i, _vp, _vp]),
    "wsdl_maxpool3x3s2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "wsdl_maxpool3x3s2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "wsdl_global_avgpool_fwd": (_i, [_vp, _vp, _i, _i, _vp]),
    "wsdl_global_avgpool_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "wsdl_class_logit_head": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "wsdl_bilinear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _ll, _vp]),
    "wsdl_bilinear_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _ll, _vp]),
    "wsdl_dropout_fwd": (_i, [_vp, _vp, _vp, _sz, _f, _u64, _i, _vp, _vp]),
    "wsdl_dropout_bwd": (_i, [_vp, _vp, _vp, _sz, _f, _vp]),
    "wsdl_add": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "wsdl_scale_by_device_scalar": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "wsdl_copy_planes": (_i, [_vp, _vp, _i, _i, _i, _ll, _ll, _vp]),
    "wsdl_reduce_workspace": (_sz, []),
    "wsdl_softmax_ce_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _ll, _vp, _sz, _vp]),
    "wsdl_lovasz_softmax_workspace": (_sz, [_i, _i, _i, _i]),
    "wsdl_lovasz_softmax_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _vp, _sz, _vp]),
    "wsdl_pairwise_affinity_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _sz, _vp]),
    "wsdl_pairwise_cache_bytes": (_sz, [_i, _i, _i, _i]),
    "wsdl_pairwise_cache": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "wsdl_pairwise_workspace": (_sz, [_i, _i, _i]),
    "wsdl_compute_affinities": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "wsdl_layercam_workspace": (_sz, [_i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "wsdl_layercam_epilogue": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                                    _i, _i, _i, _i, _f, _i, _vp, _f, _vp, _vp, _sz, _vp]),
    "wsdl_keep_largest_workspace": (_sz, [_i, _i, _i]),
    "wsdl_keep_largest": (_i, [_vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "wsdl_plane_relu_minmax": (_i, [_vp, _vp, _i, _i, _vp]),
    "wsdl_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _i, _vp, _f, _vp]),
    "wsdl_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "wsdl_kl_div_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _vp, _sz, _vp]),
    "wsdl_kl_div_per_image_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _sz, _vp, _sz, _vp]),
    "wsdl_refine_combine": (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp, _i, _sz, _vp]),
    "wsdl_softmax_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "wsdl_softmax_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    # launch plans + the stream ordering / small helpers a plan needs to see (csrc/plan.hip)
    "wsdl_plan_begin": (_i, []),
    "wsdl_plan_recording": (_i, []),
    "wsdl_plan_end": (_i, [C.POINTER(_vp)]),
    "wsdl_plan_abort": (_i, []),
    "wsdl_plan_mark": (_i, [_ll]),
    "wsdl_plan_pause": (_i, []),
    "wsdl_plan_resume": (_i, []),
    "wsdl_plan_poison": (_i, [C.c_char_p]),
    "wsdl_plan_replay": (_i, [_vp]),
    "wsdl_plan_replay_segment": (_i, [_vp, _i]),
    "wsdl_plan_replay_timed": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_ll)]),
    "wsdl_plan_stats": (_i, [_vp] + [C.POINTER(_ll)] * 5),
    "wsdl_plan_mark_tag": (_ll, [_vp, _i]),
    "wsdl_plan_destroy": (_i, [_vp]),
    "wsdl_event_create": (_i, [C.POINTER(_vp)]),
    "wsdl_event_destroy": (_i, [_vp]),
    "wsdl_event_record": (_i, [_vp, _vp]),
    "wsdl_stream_wait_event": (_i, [_vp, _vp]),
    "wsdl_stream_wait_stream": (_i, [_vp, _vp]),
    "wsdl_memset_async": (_i, [_vp, _i, _sz, _vp]),
    "wsdl_add_int": (_i, [_vp, _i, _ll, _vp]),
    "wsdl_mul": (_i, [_vp, _vp, _vp, _i, _vp]),
    "wsdl_clamp_max_i64": (_i, [_vp, _vp, _ll, _ll, _vp]),
    "wsdl_range_check": (_i, [_vp, _i, _i, _vp, _vp]),
    "wsdl_scale_mean": (_i, [_vp, _i, _f, _vp, _vp]),
    "wsdl_scale_fill": (_i, [_vp, _f, _vp, _i, _vp]),
}

_lib = None


class WsdlError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WsdlError(
                f"{LIB_PATH} not found: build it with `python -m weaklysuperviseddl_amd._build` "
                "(or __graft_entry__.build()).  There is no CPU fallback for the HIP path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the .so is stale / symbol missing
            fn.restype, fn.argtypes = res, args
        # this binding's amax slots are (max, ~min piece maximum) pairs (ops.amax_slot): the BatchNorm kernels publish both
        # (WSDL_RANGE_SENTINEL=0: the A/B partner - the second float of a pair is then never written)
        handle.wsdl_set_option(b"range_sentinel", int(os.environ.get("WSDL_RANGE_SENTINEL", "1") != "0"))
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise WsdlError(f"libwsdl_hip error {rc}: {lib().wsdl_last_error().decode()}")
