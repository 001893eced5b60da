"""ctypes binding of libsatools_hip.so (include/satools_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, an exception
is raised.  Nothing here imports from `oracle/`."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SATOOLS_AMD_LIB: another build of the same ABI, for A/B measurements of two builds on one box — tools/ab_builds.sh)
LIB_PATH = os.environ.get("SATOOLS_AMD_LIB") or os.path.join(_HERE, "libsatools_hip.so")

c_float_p = C.c_void_p  # raw device pointers travel as integers


class SatError(RuntimeError):
    pass


CONV_F32, CONV_F16X3, CONV_F16F8, CONV_F16F8R = 0, 1, 2, 3
SPLIT_F16, SPLIT_F8 = 0, 1


class ConvDesc(C.Structure):
    """mirror of sat_conv1d_desc"""
    _fields_ = [
        ("B", C.c_int32), ("C_in", C.c_int32), ("T_in", C.c_int32),
        ("C_out", C.c_int32), ("T_q", C.c_int32),
        ("ksize", C.c_int32), ("dilation", C.c_int32), ("stride", C.c_int32), ("pad_left", C.c_int32),
        ("mode", C.c_int32), ("groups", C.c_int32), ("up", C.c_int32),
        ("in_lrelu", C.c_int32), ("in_slope", C.c_float),
        ("relu", C.c_int32), ("gelu", C.c_int32), ("res_after_act", C.c_int32),
        ("accum", C.c_int32), ("accum_div", C.c_float),
        ("res_scale", C.c_float), ("res_toff", C.c_int32), ("res_tstride", C.c_int32),
        ("x_bstride", C.c_int64), ("x_cstride", C.c_int64),
        ("y_bstride", C.c_int64), ("y_cstride", C.c_int64),
        ("res_bstride", C.c_int64), ("res_cstride", C.c_int64),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ch_scale", C.c_void_p), ("ch_shift", C.c_void_p),
        ("x_split", C.c_void_p), ("y_split", C.c_void_p), ("y_split_slope", C.c_float), ("no_y", C.c_int32),
        ("res_split", C.c_void_p), ("res_split_slope", C.c_float), ("y_split_format", C.c_int32),
        ("relu_first", C.c_int32),
        ("x_wrap_channels", C.c_int32),
        ("w_descale", C.c_float),
        ("up_grouped", C.c_int32), ("up_zero_taps", C.c_uint32),
        ("x_split8", C.c_void_p), ("y_split8", C.c_void_p), ("y_split_hi_only", C.c_int32),
        ("accum_no_store", C.c_int32),
    ]


class TdnnfLayerDesc(C.Structure):
    """mirror of sat_tdnnf_layer_desc"""
    _fields_ = [
        ("B", C.c_int32), ("feat_dim", C.c_int32), ("bottleneck_dim", C.c_int32), ("out_dim", C.c_int32), ("T_in", C.c_int32), ("context_len", C.c_int32),
        ("mode", C.c_int32), ("bypass_scale", C.c_float), ("wB_descale", C.c_float), ("wA_descale", C.c_float),
        ("x", C.c_void_p), ("x_split", C.c_void_p), ("wB_packed", C.c_void_p), ("wA_packed", C.c_void_p),
        ("bB", C.c_void_p), ("bA", C.c_void_p), ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p),
        ("y", C.c_void_p), ("y_split", C.c_void_p), ("z", C.c_void_p), ("z_split", C.c_void_p),
    ]


class MrfDesc(C.Structure):
    """mirror of sat_mrf_desc"""
    _fields_ = [
        ("B", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("n_branches", C.c_int32),
        ("ksize", C.c_int32 * 3), ("dilation", (C.c_int32 * 3) * 3),
        ("w", ((C.c_void_p * 2) * 3) * 3), ("bias", ((C.c_void_p * 2) * 3) * 3), ("w_descale", ((C.c_float * 2) * 3) * 3),
        ("slope", C.c_float), ("x_split", C.c_void_p), ("y", C.c_void_p), ("y_split", C.c_void_p),
        ("y_split_slope", C.c_float), ("out_div", C.c_float), ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t),
        ("residual_from_planes", C.c_int32),
    ]


_PROTOS = {
    "sat_abi_version": (C.c_int, []),
    "sat_last_error": (C.c_char_p, []),
    "sat_last_dispatch_name": (C.c_char_p, []),
    "sat_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "sat_conv1d_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sat_conv1d_multi_f32": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "sat_conv1d_packed_dims": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sat_convtranspose_phase_dims": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sat_upsample_grouped_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sat_convtranspose_zero_taps": (C.c_uint32, [C.c_int, C.c_int, C.c_int]),
    "sat_hifigan_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sat_hifigan_num_convs": (C.c_int, [C.c_void_p]),
    "sat_hifigan_set_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "sat_hifigan_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "sat_hifigan_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                          C.c_int, C.c_void_p]),
    "sat_hifigan_destroy": (None, [C.c_void_p]),
    "sat_hifigan_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "sat_hifigan_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "sat_hifigan_set_range_probe": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sat_resblock_pair_f16x3": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "sat_mrf_debug_stamps": (C.c_int, [C.c_void_p]),
    "sat_attention_debug_stamps": (C.c_int, [C.c_void_p]),
    "sat_pair32_debug_stamps": (C.c_int, [C.c_void_p]),
    "sat_convring_debug_stamps": (C.c_int, [C.c_void_p]),
    "sat_resblock_mrf_supported": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sat_resblock_mrf_scratch_bytes": (C.c_size_t, [C.c_int, C.POINTER(C.c_int)]),
    "sat_resblock_mrf_f16x3": (C.c_int, [C.POINTER(MrfDesc), C.c_void_p]),
    "sat_resblock_pair_scaled_f16x3": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]),
    "sat_hifigan_set_conv_descale": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "sat_hifigan_set_conv_f8r": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "sat_conv1d_f8r_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "sat_planes_f8_sidecar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sat_upsample2_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sat_upsample2_f16x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p]),
    "sat_tdnnf_layer_f32": (C.c_int, [C.POINTER(TdnnfLayerDesc), C.c_void_p]),
    "sat_act_split_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "sat_hifigan_convpost_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]),
    "sat_fbank_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sat_fbank_cmvn_pad_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                         C.c_int, C.c_void_p]),
    "sat_vq_argmin_gather_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sat_vq_argmin_gather_tie_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sat_pad_replicate_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    "sat_f0_stats_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "sat_f0_apply_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "sat_f0_mean_reversion_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "sat_yaapt_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "sat_yaapt_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "sat_yaapt_ragged_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "sat_tdnnf_unfold15_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sat_log_softmax_channels_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sat_melspec_logmel_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_int, C.c_float, C.c_void_p]),
    "sat_instnorm_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "sat_row_mean_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sat_add3_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_int64] * 8 + [C.c_void_p]),
    "sat_se_gate_add_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_int64] * 2 + [C.c_void_p]),
    "sat_tanh_inplace_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "sat_attentive_stats_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sat_l2norm_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sat_w2v2_conv0_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]),
    "sat_layernorm_channels_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "sat_layernorm_channels_planes_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                    C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                                    C.c_void_p]),
    "sat_w2v2_conv0_layernorm_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "sat_conv_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "sat_clock_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sat_attention_f16x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, C.c_void_p]),
    "sat_softmax_columns_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "sat_transpose_heads_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p]),
    "sat_res2_chain_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sat_linear_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "sat_pcm16_to_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "sat_pcm16_from_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "sat_assemble_input_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


def exported_symbols():
    """names declared in include/satools_hip.h that the library must export"""
    return sorted(_PROTOS)


def library_path():
    """where the in-tree shared library lives (built by sa-toolkit_amd/build.py)"""
    return LIB_PATH


def lib():
    """load (once) and return the ctypes library; raises if it has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SatError(
                f"{LIB_PATH} is missing: build it with `python sa-toolkit_amd/build.py` "
                "(there is no CPU fallback for the HIP path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.sat_abi_version() != 8:       # 5: SAT_CONV_F16F8R; sat_conv1d_desc grew x_split8 / y_split8 / y_split_hi_only.  6: sat_pcm16_*.  7: VQ near-tie count, hifigan get_option / range probe.  8: sat_tdnnf_layer_f32
            raise SatError("libsatools_hip.so ABI version mismatch")
        # A/B switches of the conv dispatch for whole-program measurements (bench.py under different kernels):
        # SATOOLS_AMD_CONV_OPTIONS="pair32w=0,lean_balance=2" -> sat_conv_set_option(name, value) at load time
        for item in filter(None, os.environ.get("SATOOLS_AMD_CONV_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            try:
                value = int(value)
            except ValueError:
                raise SatError(f"SATOOLS_AMD_CONV_OPTIONS: '{item.strip()}' is not of the form name=integer") from None
            if l.sat_conv_set_option(name.strip().encode(), value) != 0:
                raise SatError(f"SATOOLS_AMD_CONV_OPTIONS: {l.sat_last_error().decode('utf-8', 'replace')}")
        _lib = l
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().sat_last_error().decode("utf-8", "replace")
        raise SatError(f"{what} failed ({status}): {msg}")


def ptr(t, strided=False):
    """device pointer of a contiguous f32/i32 CUDA(HIP) tensor, None -> NULL.  `strided` accepts a view
    whose innermost axis is contiguous (the caller passes the strides explicitly)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise SatError("HIP entry points need device tensors; got a CPU tensor (no CPU fallback)")
    if strided:
        if t.stride(-1) != 1:
            raise SatError("HIP entry points need the innermost axis contiguous")
    elif not t.is_contiguous():
        raise SatError("HIP entry points need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def cache_rebuild_begin(device, had_old):
    """call before a device-side weight cache is (re)built.  The packed buffers about to be dropped go back to the
    allocator pool of the stream that built them while OTHER streams (several convert() jobs in flight, the
    reference's jobs_per_compute_device) may still be reading them: wait for the whole device first.  Rebuilds happen
    at load time or after a parameter edit, never on the steady-state path."""
    import torch
    if had_old and getattr(device, "type", "cuda") == "cuda":
        torch.cuda.synchronize(device)


def cache_rebuild_end(device):
    """call after a device-side weight cache has been built: the fold / pack / BatchNorm-affine kernels ran
    asynchronously on the builder's stream, and nothing else orders them before the first launch of another stream
    that reads the cache — so the builder waits for them once, here."""
    import torch
    if getattr(device, "type", "cuda") == "cuda":
        torch.cuda.current_stream(device).synchronize()
