"""ctypes binding of libloftr_hip.so (include/loftr_hip.h).

The product path has NO fallback: if the shared object is missing or a call fails, this module
raises.  Build it with ``python -m loftr_amd.build`` (hipcc, gfx950).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# LOFTR_HIP_LIB selects another build of the same library (A/B experiments, tools/ab_build.sh)
LIB_PATH = os.environ.get("LOFTR_HIP_LIB") or os.path.join(HERE, "libloftr_hip.so")

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t
_l = C.c_long


class LayerWeights(C.Structure):
    _fields_ = [(n, _p) for n in ("q_proj", "k_proj", "v_proj", "merge", "mlp0", "mlp2",
                                  "norm1_w", "norm1_b", "norm2_w", "norm2_b")]


class CoarseParams(C.Structure):
    _fields_ = [("N", _i), ("h0c", _i), ("w0c", _i), ("h1c", _i), ("w1c", _i), ("C", _i),
                ("thr", _f), ("border_rm", _i), ("scale", _f),
                ("mask0", _p), ("mask1", _p), ("scale0", _p), ("scale1", _p)]


class MatchOut(C.Structure):
    _fields_ = [(n, _p) for n in ("b_ids", "i_ids", "j_ids", "mconf", "mkpts0_c", "mkpts1_c", "counts")]


class SpvsParams(C.Structure):
    _fields_ = [(n, _i) for n in ("N", "H0", "W0", "H1", "W1", "scale", "dh0", "dw0", "dh1", "dw1")] + \
               [(n, _p) for n in ("depth0", "depth1", "T_0to1", "T_1to0", "K0", "K1", "scale0", "scale1", "mask0", "mask1")]


class FMap(C.Structure):
    _fields_ = [("data", _p), ("sn", _l), ("sc", _l), ("sh", _l), ("sw", _l), ("H", _i), ("W", _i)]


# symbol -> (restype, argtypes); must list every function declared in include/loftr_hip.h
SIGNATURES = {
    "loftr_hip_abi_version": (_i, []),
    "loftr_hip_status_string": (C.c_char_p, [_i]),
    "loftr_hip_device_check": (_i, []),
    "loftr_pos_encode_flatten": (_i, [C.POINTER(FMap), _p, _i, _i, _p, _i, _i, _p]),
    "loftr_encoder_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "loftr_encoder_layer_fwd": (_i, [_p, _p, _p, _p, C.POINTER(LayerWeights), _p, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "loftr_encoder_layer_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "loftr_encoder_layer_bwd": (_i, [_p, _p, _p, _p, C.POINTER(LayerWeights), _p, _p, _p, C.POINTER(LayerWeights), _i, _i, _i, _i, _i, _p, _sz, _p]),
    "loftr_transformer_fwd": (_i, [_p, _p, _p, _p, C.POINTER(LayerWeights), C.POINTER(_i), _i, _i, _i, _i, _i, _i,
                                   _p, _sz, _p, _sz, _p]),
    "loftr_transformer_fwd_padded": (_i, [_p, _p, _p, _p, C.POINTER(LayerWeights), C.POINTER(_i), _i, _i, _i, _i, _i, _i,
                                          _p, _sz, _p, _sz, _i, _p]),
    "loftr_coarse_plan_bytes": (_sz, [C.POINTER(_i), _i, _i, _i, _i]),
    "loftr_coarse_plan_build": (_i, [C.POINTER(_i), _i, _i, _i, _i, _i, _p, _sz, _p]),
    "loftr_coarse_plan_signature": (C.c_uint, [_i, _i, _i, _i, _i]),
    "loftr_transformer_fwd_planned": (_i, [_p, _p, _p, _p, C.POINTER(LayerWeights), C.POINTER(_i), _i, _i, _i, _i, _i, _i,
                                           _p, _sz, _p, _sz, _p, _sz, _i, _p, _sz, _p]),
    "loftr_hip_debug_set": (_i, [C.c_char_p, _i]),
    "loftr_hip_debug_get": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i)]),
    "loftr_transformer_prepared_bytes": (_sz, [_i, _i]),
    "loftr_transformer_prepare": (_i, [C.POINTER(LayerWeights), _i, _i, _p, _sz, _p]),
    "loftr_coarse_match_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "loftr_coarse_match_dual_softmax": (_i, [_p, _p, C.POINTER(CoarseParams), _f, _p, C.POINTER(MatchOut), _p, _sz, _p]),
    "loftr_coarse_match_sinkhorn": (_i, [_p, _p, C.POINTER(CoarseParams), _f, _i, _i, _p, _p, C.POINTER(MatchOut),
                                         _p, _sz, _p]),
    "loftr_fine_preprocess_workspace_bytes": (_sz, [_i, _i, _i]),
    "loftr_fine_preprocess": (_i, [C.POINTER(FMap), C.POINTER(FMap), _p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i,
                                   _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "loftr_fine_preprocess_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "loftr_fine_preprocess_bwd": (_i, [C.POINTER(FMap), C.POINTER(FMap), _p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i,
                                       _p, _p, _p, _p, _p, C.POINTER(FMap), C.POINTER(FMap), _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "loftr_fine_match": (_i, [_p, _p, _i, _i, _i, _p, _p, _f, _p, _p, _p, _p]),
    "loftr_conv_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "loftr_conv_bn_act": (_i, [_p, _i, _i, _i, _i, _p, C.POINTER(_l), _i, _i, _i, _i, _i, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _sz, _p, _p]),
    "loftr_stem_conv_bn_relu": (_i, [_p, C.POINTER(_l), _i, _i, _i, _p, C.POINTER(_l), _i, _p, _p, _p, _p, _f, _p, _p]),
    "loftr_upsample2x_add": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "loftr_bn_train_workspace_bytes": (_sz, [_i, _i, _l]),
    "loftr_bn_train_fwd": (_i, [_p, _i, _i, _l, _i, _p, _p, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "loftr_bn_train_bwd": (_i, [_p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "loftr_act_fwd": (_i, [_p, _p, _l, _i, _f, _p, _p]),
    "loftr_act_bwd": (_i, [_p, _p, _l, _i, _f, _p, _p]),
    "loftr_upsample2x_bilinear_fwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "loftr_upsample2x_bilinear_bwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "loftr_resize_linear_u8": (_i, [_p, _i, _i, _l, _p, _i, _i, _l, _p]),
    "loftr_pack_gray_u8": (_i, [_p, _l, _l, _p, _i, _i, _i, _p, _p, _p, _i, _p]),
    "loftr_epipolar_errors": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _p, _p]),
    "loftr_spvs_coarse_workspace_bytes": (_sz, [_i, _i, _i]),
    "loftr_spvs_coarse": (_i, [C.POINTER(SpvsParams), _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "loftr_spvs_fine": (_i, [_p, _p, _i, _i, _p, _p, _p, _l, _f, _f, _p, _p, _p]),
    "loftr_loss_workspace_bytes": (_sz, [_i, _i, _i]),
    "loftr_coarse_loss_sums": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _l, _p, _p, _f, _f, _p, _p, _sz, _p]),
    "loftr_fine_loss_sums": (_i, [_p, _i, _p, _l, _i, _f, _p, _p, _sz, _p]),
    "loftr_coarse_loss_grad": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _l, _p, _p, _f, _f, C.c_double, C.c_double, _p, _p, _sz, _p]),
    "loftr_conv_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "loftr_conv_wgrad": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "loftr_head_feat_grads": (_i, [_p, C.c_long, C.c_long, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p]),
    "loftr_sinkhorn_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "loftr_sinkhorn_bwd": (_i, [_p, _p, C.POINTER(CoarseParams), _f, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "loftr_fine_loss_grad": (_i, [_p, _i, _p, _l, _i, _f, _i, _p, _f, _p, _p]),
    "loftr_dual_softmax_bwd": (_i, [_p, _p, C.POINTER(CoarseParams), _f, _p, _p, _p, _sz, _p]),
    "loftr_fine_match_bwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "loftr_estimate_pose": (_i, [_p, _p, _l, _p, _p, _f, _f, C.c_uint, _p, _p, _p, C.POINTER(_l)]),
    "loftr_five_point": (_i, [_p, _p, _i, _p, C.POINTER(_i)]),
    "loftr_conv_prepare": (_i, [_p, C.POINTER(_l), _i, _i, _i, _i, _p, _p, _p, _p, _f, _p, _sz, _p]),
    "loftr_conv_bn_act_prepared": (_i, [_p, _i, _i, _i, _i, _p, _sz, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "loftr_conv_scratch_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "loftr_conv_bn_act_prepared_scratch": (_i, [_p, _i, _i, _i, _i, _p, _sz, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "loftr_conv1x1_upsample_add": (_i, [_p, _i, _i, _i, _i, _p, C.POINTER(_l), _i, _p, _p, _p, _sz, _p]),
    "loftr_sp_from_f32": (_i, [_p, _p, _l, _i, _p]),
    "loftr_sp_from_f32_scaled": (_i, [_p, _p, _l, _i, _p, _p]),
    "loftr_sp_to_f32": (_i, [_p, _p, _l, _i, _p]),
    "loftr_rccl_unique_id": (_i, [C.c_char_p, _sz]),
    "loftr_rccl_comm_create": (_i, [C.c_char_p, _sz, _i, _i, C.POINTER(_p)]),
    "loftr_rccl_comm_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i)]),
    "loftr_rccl_comm_destroy": (_i, [_p]),
    "loftr_rccl_allgather_counts": (_i, [_p, _p, _p, _i, _p]),
    "loftr_hip_timing_enable": (_i, [C.c_uint]),
    "loftr_hip_range_check_enable": (_i, [_i]),
    "loftr_hip_timing_kernel_count": (_i, []),
    "loftr_hip_timing_kernel_name": (C.c_char_p, [_i]),
    "loftr_hip_timing_read": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_longlong), _i]),
    "loftr_linear_workspace_bytes": (_sz, [_i, _i, _i]),
    "loftr_linear_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _sz, _p]),
}

ABI_VERSION = 24
_lib = None


class LoftrHipError(RuntimeError):
    pass


def load():
    """Load (once) and type the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LoftrHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m loftr_amd.build` "
            "(needs hipcc). There is no CPU / PyTorch fallback for the matching path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.loftr_hip_abi_version() != ABI_VERSION:
        raise LoftrHipError(f"ABI mismatch: library {lib.loftr_hip_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().loftr_hip_status_string(status).decode()
        raise LoftrHipError(f"{what or 'libloftr_hip'} failed: {msg} (status {status})")
