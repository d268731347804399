"""ctypes binding of libnndepth_amd.so (the C-ABI declared in include/nndepth_amd.h).

There is no fallback: if the HIP library is missing the import fails loudly, and every op
refuses non-CUDA (non-HIP) tensors.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NND_LIB") or os.path.join(_HERE, "libnndepth_amd.so")


class NndError(RuntimeError):
    pass


NND_FLAG_CALIBRATE = 1


class _SizedDesc(C.Structure):
    """Descriptors start with struct_size = sizeof(the struct) (include/nndepth_amd.h): filled in here, so the
    constructors keep taking the payload fields only."""

    def __init__(self, *args, **kwargs):
        super().__init__(C.sizeof(type(self)), *args, **kwargs)


class UpdateBlockDesc(_SizedDesc):
    _fields_ = [("struct_size", C.c_int32), ("hidden_dim", C.c_int32), ("context_dim", C.c_int32), ("cor_planes", C.c_int32),
                ("flow_channels", C.c_int32), ("mask_channels", C.c_int32), ("gru_kind", C.c_int32),
                ("arithmetic", C.c_int32), ("split_layers", C.c_int32), ("flags", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("Cout", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32)]


class Conv3dDesc(_SizedDesc):
    _fields_ = [("struct_size", C.c_int32), ("Cout", C.c_int32), ("Cin0", C.c_int32), ("Cin1", C.c_int32), ("stride", C.c_int32),
                ("arithmetic", C.c_int32), ("flags", C.c_int32)]


class EncoderDesc(_SizedDesc):
    _fields_ = [("struct_size", C.c_int32), ("output_dim", C.c_int32), ("norm", C.c_int32), ("cnet_dim", C.c_int32),
                ("arithmetic", C.c_int32), ("flags", C.c_int32)]


# name -> (restype, argtypes); mirrors include/nndepth_amd.h one to one
_P = C.c_void_p
_I = C.c_int
SIGNATURES = {
    "nnd_version": (_I, []),
    "nnd_last_error": (C.c_char_p, []),
    "nnd_device_count": (_I, []),
    "nnd_corr1d_pyramid_layout": (_I, [_I, _I, _I, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "nnd_corr1d_build": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_corr1d_lookup": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_group_corr_build": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nnd_group_corr_build_scaled": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.c_float, _P]),
    "nnd_group_corr1d_lookup": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "nnd_pyramid_from_level0": (_I, [_P, _I, _I, _I, _I, _P]),
    "nnd_igev_lookup": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "nnd_softargmin_disparity": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "nnd_igev_init_disparity": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_convex_upsample": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_bilinear_sample": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "nnd_agcl_corr_iter": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_agcl_corr_offset": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_agcl_corr_offset_nhwc": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "nnd_update_block_num_tensors": (_I, [C.POINTER(UpdateBlockDesc)]),
    "nnd_update_block_packed_floats": (C.c_int64, [C.POINTER(UpdateBlockDesc)]),
    "nnd_update_block_pack": (_I, [C.POINTER(UpdateBlockDesc), C.POINTER(_P), _P]),
    "nnd_update_block_workspace_floats": (C.c_int64, [C.POINTER(UpdateBlockDesc), _I, _I, _I]),
    "nnd_update_block_forward": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "nnd_update_block_calibration_finish": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _P]),
    "nnd_update_block_scale_slots": (_I, [C.POINTER(UpdateBlockDesc), C.POINTER(C.c_int64), _I]),
    "nnd_conv2d_calibrate_ex": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "nnd_encoder_forward2": (_I, [C.POINTER(EncoderDesc), _P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _P]),
    "nnd_pos_enc_sine_add": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_encoder_calibration_finish": (_I, [C.POINTER(EncoderDesc), _P, _P, _P]),
    "nnd_conv3d_calibration_finish": (_I, [C.POINTER(Conv3dDesc), _P, _P, _P]),
    "nnd_conv2d_packed_floats": (C.c_int64, [_I, _I, _I, _I]),
    "nnd_conv2d_pack": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "nnd_conv2d_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nnd_conv2d_packed_floats_ex": (C.c_int64, [_I, _I, _I, _I, _I]),
    "nnd_conv2d_pack_ex": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_conv2d_forward_ex": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nnd_conv2d_offset_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.c_float, _P]),
    "nnd_split_tanh_relu": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_avg_pool_2x_4x": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "nnd_resize_bilinear_ac": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, C.c_float, _P]),
    "nnd_mask_upsample_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_raft_stereo_refine": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _I, _I, _P, _P, _P, _P, C.c_int64, _P, _P, _P,
                                    _I, _I, _I, _I, _I, _P]),
    "nnd_raft_stereo_group_refine": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _I, _I, _I, _P, _P, _P, _P, C.c_int64, _P, _P, _P,
                                          _I, _I, _I, _I, _I, _P]),
    "nnd_igev_interleaved_floats": (C.c_int64, [_I, _I, _I, _I, _I]),
    "nnd_igev_interleave_pyramids": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_igev_interleave_level0_supported": (_I, [_I, _I, _I]),
    "nnd_igev_refine_reads_interleaved": (_I, [_I, _I, _I]),
    "nnd_igev_interleave_level0": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_igev_stereo_refine": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, C.c_int64, _P, _P, _P,
                                    _I, _I, _I, _I, _I, _P]),
    "nnd_cre_stereo_refine": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _P, _I, _P, _P, C.c_int64, _P, _P, _P, _P, C.c_int64, _P, _P, _P,
                                   _I, _I, _I, _I, _I, _P]),
    "nnd_conv_packed_floats": (C.c_int64, [C.POINTER(ConvDesc)]),
    "nnd_conv_pack": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, C.c_float, _P]),
    "nnd_conv_forward": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_encoder_num_tensors": (_I, [C.POINTER(EncoderDesc)]),
    "nnd_encoder_packed_floats": (C.c_int64, [C.POINTER(EncoderDesc)]),
    "nnd_encoder_workspace_floats": (C.c_int64, [C.POINTER(EncoderDesc), _I, _I, _I]),
    "nnd_encoder_pack": (_I, [C.POINTER(EncoderDesc), C.POINTER(_P), C.c_float, _P]),
    "nnd_encoder_forward": (_I, [C.POINTER(EncoderDesc), _P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "nnd_resize_normalize": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, C.c_float, C.c_float, _P]),
    "nnd_replicate_pad": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "nnd_epe_metrics_workspace_bytes": (C.c_int64, []),
    "nnd_epe_metrics": (_I, [_P, _P, _P, _I, _I, _I, _I, C.c_float, C.POINTER(C.c_float), _I, _P, _P, _P]),
    "nnd_loftr_packed_floats": (C.c_int64, [_I, _I]),
    "nnd_loftr_workspace_floats": (C.c_int64, [_I, _I, _I, _I, _I]),
    "nnd_loftr_pack": (_I, [_I, _I, C.POINTER(_P), _P]),
    "nnd_loftr_layer_forward": (_I, [_I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "nnd_conv3d_packed_floats": (C.c_int64, [C.POINTER(Conv3dDesc)]),
    "nnd_conv3d_pack": (_I, [C.POINTER(Conv3dDesc), _P, _P, _P, _P, _P, _P, C.c_float, _P]),
    "nnd_conv3d_forward": (_I, [C.POINTER(Conv3dDesc), _P, _P, _P, _P, _I, _I, _I, _I, C.c_float, _P]),
    "nnd_volume_to_depth_major": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_depth_major_to_volume": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_volume_rows_to_depth_major": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_depth_major_to_volume_rows": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_volume_upsample2x": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_volume_gate": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nnd_profile_conv": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _I, _I, _I, _I, _I, _P, C.POINTER(C.c_float),
                              C.POINTER(C.c_double)]),
    "nnd_profile_loop_conv": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P,
                                   C.POINTER(C.c_float)]),
    "nnd_profile_loop_event_pair": (_I, [C.POINTER(UpdateBlockDesc), _P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P,
                                         C.POINTER(C.c_float)]),
    "nnd_profile_mfma_peak": (_I, [_I, _I, _P, _P, C.POINTER(C.c_float)]),
    "nnd_profile_mfma16_peak": (_I, [_I, _I, C.c_float, _P, _P, _P, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "nnd_reload_switches": (_I, []),
    "nnd_num_convs": (_I, [C.POINTER(UpdateBlockDesc)]),
    "nnd_conv_name": (C.c_char_p, [C.POINTER(UpdateBlockDesc), _I]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise NndError(
            f"{LIB_PATH} not found: build it with `make -C nndepth_amd/csrc` (or __graft_entry__.build()). "
            "nndepth_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib.nnd_last_error().decode("utf-8", "replace")
        raise NndError(f"{what or 'nndepth_amd'} failed (status {rc}): {msg}")
