"""ctypes binding of libmoca_hip.so (the C-ABI declared in include/moca_hip.h).

This is the only place the Python host touches native code.  There is NO fallback:
if the shared library is missing the import fails loudly, and every wrapper raises
on a non-zero status.  PyTorch is used purely as the device-memory / stream owner
(tensor.data_ptr(), torch.cuda.Stream().cuda_stream).
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: it bundles its own libamdhip64 and must be the HIP runtime image libmoca_hip.so binds to.  Loaded the other way round
# (e.g. `import moca_video_amd` before anything imported torch) the process ends up with two runtime images and the first launch
# from this library fails with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOCA_HIP_LIB: diagnostic builds only (e.g. the -DMOCA_STAMPS library used by tools/stamps.py); load() refuses one without MOCA_HIP_DIAG=1
LIB_PATH = os.environ.get("MOCA_HIP_LIB") or os.path.join(_HERE, "libmoca_hip.so")

MOCA_A_LINEAR, MOCA_A_CONV3X3, MOCA_A_TCONV3 = 0, 1, 2
MOCA_EP_GEGLU, MOCA_EP_OUT_F32, MOCA_FORCE_SMALL_TILE, MOCA_EP_GELU, MOCA_EP_COLSUM, MOCA_EP_LN = 1, 2, 4, 8, 16, 32
MOCA_EP_ROWSUM, MOCA_EP_LNFOLD, MOCA_EP_GSTAT, MOCA_EP_TATTN, MOCA_EP_SLABS = 64, 128, 256, 512, 1024
MOCA_TUNE_GEMM_W80, MOCA_TUNE_GEMM_G4, MOCA_TUNE_GEMM_SQ256, MOCA_TUNE_GEMM_WIDE, MOCA_TUNE_GN_SLAB, MOCA_TUNE_GEMM_G4P, MOCA_TUNE_GEMM_MF32, MOCA_TUNE_GEMM_SQP, MOCA_TUNE_SQP_WALK = 0, 1, 2, 3, 4, 5, 6, 7, 8
MOCA_TUNE_SLAB_F16 = 9
MOCA_TUNE_GEMM_WS = 10

_ERR = {0: "ok", -1: "bad argument (shape/alignment contract)", -2: "HIP launch/runtime error",
        -3: "no gfx950 device", -4: "graph capture/replay failed"}


class MocaHipError(RuntimeError):
    pass


class GemmParams(C.Structure):
    """Mirror of `moca_gemm_params` (include/moca_hip.h)."""
    _fields_ = [
        ("a", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p), ("bias", C.c_void_p),
        ("rowadd", C.c_void_p), ("residual", C.c_void_p), ("splitk_ws", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32), ("ld_rowadd", C.c_int32),
        ("rowadd_div", C.c_int32), ("a_mode", C.c_int32), ("C", C.c_int32),
        ("inH", C.c_int32), ("inW", C.c_int32), ("outH", C.c_int32), ("outW", C.c_int32),
        ("stride", C.c_int32), ("up", C.c_int32), ("T", C.c_int32), ("HW", C.c_int32),
        ("flags", C.c_int32), ("splits", C.c_int32), ("nopad_lo", C.c_int32), ("prefetch_kib", C.c_int32),
        ("colsum", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_out", C.c_void_p),
        ("ld_ln", C.c_int32), ("ln_eps", C.c_float),
        ("rowsum", C.c_void_p), ("lnf_part", C.c_void_p), ("lnf_wsum", C.c_void_p),
        ("lnf_nparts", C.c_int32), ("reserved2_", C.c_int32),
        ("gstat", C.c_void_p), ("gstat_rows", C.c_int32), ("tattn_scale", C.c_float),
        ("prefetch", C.c_void_p), ("up_phase", C.c_int32), ("reserved4_", C.c_int32),
        ("a2", C.c_void_p), ("lda2", C.c_int32), ("k1", C.c_int32), ("gstat_cpg", C.c_int32), ("gstat_coff", C.c_int32),
        ("wgroup_rows", C.c_int32), ("wgroup_stride", C.c_int32),
    ]


class FifoState(C.Structure):
    """Mirror of `moca_fifo_state` (32 bytes, device-resident: the host writes it through an int32 tensor)."""
    _fields_ = [("head", C.c_int32), ("iter", C.c_int32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("ext_noise", C.c_int32), ("reserved_", C.c_int32 * 3)]


class FifoStepParams(C.Structure):
    """Mirror of `moca_fifo_step_params` (include/moca_hip.h)."""
    _fields_ = [
        ("state", C.c_void_p), ("x", C.c_void_p), ("eps_c", C.c_void_p), ("eps_u", C.c_void_p), ("noise", C.c_void_p),
        ("momentum", C.c_void_p), ("queue", C.c_void_p), ("x_prev", C.c_void_p), ("pred_x0", C.c_void_p), ("coef", C.c_void_p),
        ("win_start", C.c_void_p), ("mask", C.c_void_p), ("mask_sums", C.c_void_p), ("mask_frame", C.c_void_p),
        ("enh", C.c_void_p), ("cond", C.c_void_p), ("sam_eff", C.c_void_p), ("sam_idx", C.c_void_p),
        ("cfg_scale", C.c_float), ("beta", C.c_float), ("one_minus_beta", C.c_float), ("gamma", C.c_float),
        ("one_minus_gamma", C.c_float),
        ("nW", C.c_int32), ("C", C.c_int32), ("Q", C.c_int32), ("f", C.c_int32), ("HW", C.c_int32), ("wb_from", C.c_int32),
    ]


# name -> (restype, argtypes); must list every symbol of include/moca_hip.h
_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
SIGNATURES = {
    "moca_gemm_f16": (C.c_int, [C.POINTER(GemmParams), _vp]),
    "moca_gemm_splitk_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "moca_gemm_colsum_rows": (C.c_int, [C.POINTER(GemmParams)]),
    "moca_gemm_ln_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "moca_gemm_rowsum_cols": (C.c_int, [C.POINTER(GemmParams)]),
    "moca_gemm_lnfold_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "moca_gemm_tattn_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "moca_gemm_cat_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "moca_gemm_wgroup_ok": (C.c_int, [C.POINTER(GemmParams)]),
    "moca_groupnorm_fold_weights_f16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _f32, _vp]),
    "moca_gemm_splitk_groupnorm_ok": (C.c_int, [C.POINTER(GemmParams), _i32, _i32]),
    "moca_gemm_splitk_groupnorm_f16": (C.c_int, [C.POINTER(GemmParams), _vp, _vp, _vp, _i32, _i32, _f32, _i32, _i32, _vp]),
    "moca_groupnorm_gstat_cat_f16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "moca_gstat_accum_f16": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "moca_groupnorm_colsum_f16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "moca_groupnorm_nhwc_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "moca_groupnorm_ws_bytes": (_i64, [_i32, _i32, _i32]),
    "moca_groupnorm_gstat_f16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "moca_memset_zero": (C.c_int, [_vp, _i64, _vp]),
    "moca_concat_channels_gstat_f16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "moca_layernorm_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "moca_attention_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "moca_attention_causal_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "moca_embed_tokens_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "moca_temporal_attention_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "moca_ncthw_to_nhwc_f16": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "moca_nhwc_to_ncthw": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "moca_concat_channels_f16": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "moca_repeat_f16": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "moca_timestep_embedding_f16": (C.c_int, [_vp, _vp, _i32, _i32, _f32, _vp]),
    "moca_silu_add_rows_f16": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "moca_channel_mix_f16": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "moca_softmax_rows_f16": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _i64, _f32, _vp]),
    "moca_gaussian_sample_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "moca_cfg_combine_f32": (C.c_int, [_vp, _vp, _vp, _f32, _i64, _vp]),
    "moca_ddim_update_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _i64, _vp]),
    "moca_fifo_ddim_step_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp]),
    "moca_fifo_randn_f32": (C.c_int, [_vp, _vp, _i64, _vp]),
    "moca_fifo_gather_windows_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "moca_fifo_step_windows_f32": (C.c_int, [C.POINTER(FifoStepParams), _vp]),
    "moca_sam_select_masks_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "moca_fifo_advance_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    "moca_fifo_prepare_queue_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "moca_base_set_timestep": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp]),
    "moca_base_ddim_step_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _i32, _i64, _vp]),
    "moca_mask_frame_sums_f32": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "moca_freq_mix_3d_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "moca_freq_mix_ws_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "moca_freq_filter_f32": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _f64, _f64, _vp]),
    "moca_graph_begin": (C.c_int, [_vp]),
    "moca_graph_end": (C.c_int, [_vp, C.POINTER(_vp)]),
    "moca_graph_launch": (C.c_int, [_vp, _vp]),
    "moca_graph_destroy": (C.c_int, [_vp]),
    "moca_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "moca_stream_destroy": (C.c_int, [_vp]),
    "moca_stream_sync": (C.c_int, [_vp]),
    "moca_event_create": (C.c_int, [C.POINTER(_vp)]),
    "moca_event_record": (C.c_int, [_vp, _vp]),
    "moca_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(_f32)]),
    "moca_event_destroy": (C.c_int, [_vp]),
    "moca_set_tuning": (C.c_int, [_i32, _i32]),
    "moca_debug_clock_sampler": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "moca_device_info": (C.c_int, [C.c_char_p, _i32, C.POINTER(_i32)]),
    "moca_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libmoca_hip.so.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C moca_video_amd/csrc`). "
            "moca_video_amd has no CPU fallback by design.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # A diagnostic build (stamps, timing-only variants that compute WRONG results; csrc/Makefile `stamps` / `gndiag` / `diagx`) exports the
    # whole API and reports "... DIAG:<name>": never load one by accident through a stray MOCA_HIP_LIB.
    ver = lib.moca_version().decode()
    if "DIAG:" in ver and os.environ.get("MOCA_HIP_DIAG") != "1":
        raise ImportError(f"{LIB_PATH} is a diagnostic build ({ver}); set MOCA_HIP_DIAG=1 to load it on purpose")
    _lib = lib
    # MOCA_TUNE="knob:value,...": kernel-choice knobs for same-box A/B runs of whole programs (bench.py); never results
    for kv in os.environ.get("MOCA_TUNE", "").split(","):
        if kv:
            k, v = kv.split(":")
            if lib.moca_set_tuning(int(k), int(v)) < 0:
                raise MocaHipError(f"MOCA_TUNE: bad knob setting {kv!r}")
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        raise MocaHipError(f"{what or 'moca_hip call'} failed: {_ERR.get(rc, rc)} (rc={rc})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def set_tuning(knob: int, value: int) -> int:
    """kernel-choice knob for tests / A-B runs (include/moca_hip.h MOCA_TUNE_*); returns the previous value"""
    old = load().moca_set_tuning(knob, value)
    if old < 0:
        raise MocaHipError(f"moca_set_tuning({knob}, {value}): bad argument")
    return old


def version() -> str:
    return load().moca_version().decode()


def device_info():
    lib = load()
    name = C.create_string_buffer(64)
    cus = C.c_int32(0)
    check(lib.moca_device_info(name, 64, C.byref(cus)), "moca_device_info")
    return name.value.decode(), cus.value
