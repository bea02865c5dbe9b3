"""ctypes binding of libsdmi.so (include/sdmi.h).  No fallback: if the library is missing this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SDMI_LIB_PATH') or os.path.join(_HERE, 'libsdmi.so')     # (override: same-box A/B of two builds)

c_f32p = C.c_void_p     # device pointers travel as integers (tensor.data_ptr())
c_ptr = C.c_void_p


class UNetCfg(C.Structure):
    _fields_ = [('in_channels', C.c_int32), ('out_channels', C.c_int32), ('model_channels', C.c_int32),
                ('num_res_blocks', C.c_int32), ('n_levels', C.c_int32), ('channel_mult', C.c_int32 * 8),
                ('n_attention_resolutions', C.c_int32), ('attention_resolutions', C.c_int32 * 8),
                ('num_heads', C.c_int32), ('transformer_depth', C.c_int32), ('context_dim', C.c_int32)]


class VaeCfg(C.Structure):
    _fields_ = [('ch', C.c_int32), ('out_ch', C.c_int32), ('n_levels', C.c_int32), ('ch_mult', C.c_int32 * 8),
                ('num_res_blocks', C.c_int32), ('in_channels', C.c_int32), ('z_channels', C.c_int32),
                ('embed_dim', C.c_int32)]


class ClipCfg(C.Structure):
    _fields_ = [('vocab_size', C.c_int32), ('hidden_size', C.c_int32), ('intermediate_size', C.c_int32),
                ('num_layers', C.c_int32), ('num_heads', C.c_int32), ('max_positions', C.c_int32)]


class IGemmDesc(C.Structure):
    _fields_ = [('a0', c_ptr), ('a1', c_ptr), ('a2', c_ptr),
                ('c0', C.c_int32), ('c1', C.c_int32), ('c2', C.c_int32),
                ('lda0', C.c_int32), ('lda1', C.c_int32), ('lda2', C.c_int32),
                ('B', C.c_int32), ('Hin', C.c_int32), ('Win', C.c_int32), ('Hout', C.c_int32), ('Wout', C.c_int32),
                ('ksize', C.c_int32), ('stride', C.c_int32), ('up', C.c_int32),
                ('w', c_ptr), ('N', C.c_int32), ('mode', C.c_int32),
                ('bias', c_ptr), ('rowvec', c_ptr), ('ld_rowvec', C.c_int32),
                ('residual', c_ptr), ('ldr', C.c_int32),
                ('out_f32', c_ptr), ('out_f16', c_ptr), ('ldo', C.c_int32),
                ('seg_dst', c_ptr * 3), ('seg_kind', C.c_int32 * 3),
                ('heads', C.c_int32), ('dh', C.c_int32), ('ntok', C.c_int32), ('ntok_pad', C.c_int32),
                ('segC', C.c_int32), ('splitk', C.c_int32), ('splitk_ws', c_ptr), ('splitk_ws_floats', C.c_int64),
                ('tile', C.c_int32), ('dma', C.c_int32), ('asym_pad', C.c_int32),
                ('gn_n', C.c_int32), ('gn_acc', c_ptr * 2), ('gn_cpg', C.c_int32 * 2), ('gn_cbase', C.c_int32 * 2),
                ('splitk_cnt', c_ptr), ('splitk_cnt_ints', C.c_int32), ('split16', C.c_int32),
                ('f16_scale', c_ptr), ('lnp_out', c_ptr), ('lnf_part', c_ptr), ('lnf_npart', C.c_int32),
                ('lnf_eps', C.c_float), ('lnf_cs', c_ptr), ('lnf_d', c_ptr),
                ('pgn_gamma', c_ptr), ('pgn_beta', c_ptr), ('pgn_eps', C.c_float), ('pgn_silu', C.c_int32),
                ('pgn_out', c_ptr), ('pgn_keep_f32', C.c_int32), ('pgn_applied', C.POINTER(C.c_int32)),
                ('out_lo', c_ptr)]


_SIGS = {
    'sdmi_last_error': (C.c_char_p, []),
    'sdmi_abi_version': (C.c_int, []),
    'sdmi_unet_create': (C.c_int, [C.POINTER(UNetCfg), C.POINTER(c_ptr)]),
    'sdmi_unet_destroy': (C.c_int, [c_ptr]),
    'sdmi_unet_num_weights': (C.c_int, [c_ptr]),
    'sdmi_unet_weight_info': (C.c_int, [c_ptr, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    'sdmi_unet_set_weight': (C.c_int, [c_ptr, C.c_char_p, c_ptr, C.POINTER(C.c_int64), C.c_int, c_ptr]),
    'sdmi_unet_finalize': (C.c_int, [c_ptr]),
    'sdmi_unet_packed_bytes': (C.c_int64, [c_ptr]),
    'sdmi_unet_export_packed': (C.c_int, [c_ptr, c_ptr, C.c_int64, c_ptr]),
    'sdmi_unet_import_packed': (C.c_int, [c_ptr, c_ptr, C.c_int64, c_ptr]),
    'sdmi_unet_workspace_bytes': (C.c_int64, [c_ptr, C.c_int, C.c_int, C.c_int, C.c_int]),
    'sdmi_unet_cache_context': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, C.c_int64, c_ptr]),
    'sdmi_unet_reserve_context': (C.c_int, [c_ptr, C.c_int, C.c_int]),
    'sdmi_unet_cache_timesteps': (C.c_int, [c_ptr, C.POINTER(C.c_int64), C.c_int, c_ptr]),
    'sdmi_unet_hint_timestep': (C.c_int, [c_ptr, C.c_int64]),
    'sdmi_unet_tape_stats': (C.c_int, [c_ptr, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'sdmi_unet_forward': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int,
                                    c_ptr, C.c_int64, c_ptr]),
    'sdmi_sampler_step': (C.c_int, [c_ptr, C.c_int, C.c_float, c_ptr, C.c_int, c_ptr, c_ptr, c_ptr, C.c_float,
                                    C.c_float, C.c_float, C.c_float, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int64, c_ptr]),
    'sdmi_dpm_solver_step': (C.c_int, [c_ptr, C.c_int, C.c_float, c_ptr, c_ptr, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_int, c_ptr, c_ptr, C.c_int64, c_ptr]),
    'sdmi_vae_create': (C.c_int, [C.POINTER(VaeCfg), C.c_int, C.POINTER(c_ptr)]),
    'sdmi_vae_destroy': (C.c_int, [c_ptr]),
    'sdmi_vae_num_weights': (C.c_int, [c_ptr]),
    'sdmi_vae_weight_info': (C.c_int, [c_ptr, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    'sdmi_vae_set_weight': (C.c_int, [c_ptr, C.c_char_p, c_ptr, C.POINTER(C.c_int64), C.c_int, c_ptr]),
    'sdmi_vae_finalize': (C.c_int, [c_ptr]),
    'sdmi_vae_decode_workspace_bytes': (C.c_int64, [c_ptr, C.c_int, C.c_int, C.c_int]),
    'sdmi_vae_decode': (C.c_int, [c_ptr, c_ptr, C.c_float, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, C.c_int64, c_ptr]),
    'sdmi_vae_encode_workspace_bytes': (C.c_int64, [c_ptr, C.c_int, C.c_int, C.c_int]),
    'sdmi_vae_encode': (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, C.c_int64, c_ptr]),
    'sdmi_clip_create': (C.c_int, [C.POINTER(ClipCfg), C.POINTER(c_ptr)]),
    'sdmi_clip_destroy': (C.c_int, [c_ptr]),
    'sdmi_clip_num_weights': (C.c_int, [c_ptr]),
    'sdmi_clip_weight_info': (C.c_int, [c_ptr, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    'sdmi_clip_set_weight': (C.c_int, [c_ptr, C.c_char_p, c_ptr, C.POINTER(C.c_int64), C.c_int, c_ptr]),
    'sdmi_clip_finalize': (C.c_int, [c_ptr]),
    'sdmi_clip_workspace_bytes': (C.c_int64, [c_ptr, C.c_int, C.c_int]),
    'sdmi_clip_forward': (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, C.c_int64, c_ptr]),
    'sdmi_has_experiments': (C.c_int, []),
    'sdmi_k_igemm': (C.c_int, [C.POINTER(IGemmDesc), c_ptr]),
    'sdmi_k_ff_tail': (C.c_int, [C.POINTER(IGemmDesc), c_ptr, c_ptr, C.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sdmi_k_gn_conv3': (C.c_int, [C.POINTER(IGemmDesc), c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, C.c_int64, c_ptr, c_ptr, C.c_float, c_ptr]),
    'sdmi_k_st_mid_ctx': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_float, c_ptr,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_st_tail': (C.c_int, [C.POINTER(IGemmDesc), c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sdmi_k_st_mid': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_float, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, c_ptr]),
    'sdmi_k_st_head': (C.c_int, [c_ptr, c_ptr, C.c_int64, c_ptr, c_ptr, C.c_float, c_ptr, c_ptr, c_ptr, c_ptr, C.c_float, c_ptr, c_ptr, c_ptr,
                                 c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_attention_causal': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                          c_ptr]),
    'sdmi_k_pointwise_nchw': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_ptr]),
    'sdmi_k_softmax_rows': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_float, c_ptr]),
    'sdmi_k_attention': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, c_ptr]),
    'sdmi_k_groupnorm': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, C.c_float, C.c_int,
                                   c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int64, c_ptr]),
    'sdmi_k_groupnorm_ws_floats': (C.c_int64, [C.c_int, C.c_int]),
    'sdmi_k_conv3gn': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, C.c_float, c_ptr,
                                 C.c_int, c_ptr, c_ptr, C.c_int, c_ptr, C.c_int, c_ptr, C.c_int, C.c_int, c_ptr, C.c_int64,
                                 c_ptr, C.c_int64, C.c_int, c_ptr, c_ptr, c_ptr]),
    'sdmi_k_layernorm': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_float, c_ptr]),
    'sdmi_k_cast_f16': (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_int64, c_ptr]),
    'sdmi_k_timestep_embedding': (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_small_linear': (C.c_int, [c_ptr, C.c_int, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, c_ptr]),
    'sdmi_k_conv_in': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_conv_out': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_pack_conv_weight': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_pack_conv_out': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_pack_split3': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_pack_geglu': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, c_ptr]),
    'sdmi_k_attention_ctx': (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       c_ptr, C.c_float, c_ptr, c_ptr, c_ptr]),
    'sdmi_k_ln_fold_prep': (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'sdmi_image_to_uint8': (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    'sdmi_range_check': (C.c_int, [C.c_int]),
    'sdmi_range_report': (C.c_int, [C.c_char_p, C.c_int]),
    'sdmi_tune_begin': (C.c_int, []),
    'sdmi_tune_round': (C.c_int, [C.c_int]),
    'sdmi_tune_end': (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    'sdmi_tune_dump': (C.c_int, [C.c_char_p, C.c_int]),
    'sdmi_profile_begin': (C.c_int, []),
    'sdmi_profile_end': (C.c_int, [C.c_char_p, C.c_int]),
    'sdmi_k_prefetch_lines': (C.c_int, [c_ptr, C.c_int64, c_ptr]),
    'sdmi_zero_page': (c_ptr, []),
}

_lib = None


class SdmiError(RuntimeError):
    pass


def load():
    """Load libsdmi.so (once).  Raises if it has not been built -- there is no CPU / PyTorch fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdmiError(f'{LIB_PATH} not found: build it with `python stable-diffusion_amd/build.py` '
                            '(the MI355X path has no fallback implementation)')
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)      # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        if lib.sdmi_abi_version() != 17:
            raise SdmiError('libsdmi ABI version mismatch')
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc):
    if rc != 0:
        raise SdmiError(load().sdmi_last_error().decode('utf-8', 'replace'))


def ptr(t):
    """tensor (or None) -> device pointer for ctypes"""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
