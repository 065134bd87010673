"""ctypes loader of the in-tree HIP shared library (fhestring_amd/libfhestring_hip.so).

There is no CPU fallback: if the library is missing this raises, and every
entry point fails loudly when no MI355X is visible (fhs_ctx_create returns an
error that FhsError carries).
"""
import ctypes as C
import importlib
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# FHS_LIB_PATH: load another build of the same library (kernel experiments: tools/ablate_fft.py); never a fallback
LIB_PATH = os.environ.get("FHS_LIB_PATH") or os.path.join(_HERE, "libfhestring_hip.so")

_lib = None


class FhsError(RuntimeError):
    pass


def hip_runtimes():
    """Paths of the HIP runtimes (libamdhip64) mapped into this process."""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if os.path.basename(path).startswith("libamdhip64") and path not in seen:
                    seen.append(path)
    except OSError:
        pass
    return seen


def check_single_hip_runtime():
    """Raise FhsError when two HIP runtimes are mapped into the process.

    PyTorch ships its own libamdhip64 / libhsa-runtime64 next to /opt/rocm's.  The dynamic loader shares ONE copy only
    when torch's is already there (this library's DT_NEEDED then resolves to it by SONAME); loaded in the other order,
    torch's RPATH maps a second runtime beside /opt/rocm's and the first GPU call of either side can crash the process
    (the segmentation faults of round 2's `call_b.log`: two pytest processes that loaded this library, then torch).
    """
    rts = hip_runtimes()
    if len(rts) > 1:
        raise FhsError(
            "two HIP runtimes are mapped into this process (%s): `import torch` BEFORE fhestring_amd "
            "(or not at all) so that both sides share one; refusing to touch the GPU" % ", ".join(rts))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FhsError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # Load order (see check_single_hip_runtime): if torch is installed but not imported yet, import it first so
        # that a later `import torch` of the host program cannot map a second runtime.  FHS_SKIP_TORCH_PRELOAD=1
        # opts out (hosts that never use torch); the check below and in Context() still guards the process.
        if (not hip_runtimes() and "torch" not in sys.modules and not os.environ.get("FHS_SKIP_TORCH_PRELOAD")
                and importlib.util.find_spec("torch") is not None):
            importlib.import_module("torch")
        _lib = C.CDLL(LIB_PATH)
        check_single_hip_runtime()
        _declare(_lib)
    return _lib


u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def _declare(L):
    vp, sz, i, u64, u8 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint8
    L.fhs_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.fhs_ctx_create.restype = i
    L.fhs_ctx_create_planner.argtypes = [C.POINTER(vp)]
    L.fhs_ctx_create_planner.restype = i
    L.fhs_ctx_destroy.argtypes = [vp]
    L.fhs_ctx_destroy.restype = None
    L.fhs_last_error.argtypes = [vp]
    L.fhs_last_error.restype = C.c_char_p
    L.fhs_load_server_key.argtypes = [vp, vp, vp]
    L.fhs_load_server_key.restype = i
    L.fhs_pbs_batch.argtypes = [vp, vp, vp, vp, sz, vp, sz]
    L.fhs_pbs_batch.restype = i
    L.fhs_keyswitch_modswitch_batch.argtypes = [vp, vp, vp, sz]
    L.fhs_keyswitch_modswitch_batch.restype = i
    L.fhs_pbs_batch_device.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    L.fhs_pbs_batch_device.restype = i
    L.fhs_kernel_timing.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.fhs_kernel_timing.restype = i
    L.fhs_kernel_timing_kind.argtypes = [vp, i, C.POINTER(C.c_double), u64p, u64p]
    L.fhs_kernel_timing_kind.restype = i
    h = C.c_uint64                     # fhs_char_t
    hp = C.POINTER(C.c_uint64)
    L.fhs_trivial.argtypes = [vp, u8]
    L.fhs_trivial.restype = h
    L.fhs_upload.argtypes = [vp, vp]
    L.fhs_upload.restype = h
    L.fhs_upload_string.argtypes = [vp, vp, sz, vp]
    L.fhs_upload_string.restype = i
    L.fhs_download_string.argtypes = [vp, vp, sz, vp]
    L.fhs_download_string.restype = i
    L.fhs_import_device.argtypes = [vp, vp]
    L.fhs_import_device.restype = h
    for name in ("eq", "ne", "le", "lt", "ge", "gt", "bitand", "bitor", "sub", "add"):
        f = getattr(L, "fhs_" + name)
        f.argtypes = [vp, h, h]
        f.restype = h
    L.fhs_if_then_else.argtypes = [vp, h, h, h]
    L.fhs_if_then_else.restype = h
    for name in ("flip", "is_whitespace", "is_uppercase", "is_lowercase", "clone"):
        f = getattr(L, "fhs_" + name)
        f.argtypes = [vp, h]
        f.restype = h
    L.fhs_release.argtypes = [vp, h]
    L.fhs_release.restype = i
    L.fhs_flush.argtypes = [vp]
    L.fhs_flush.restype = i
    L.fhs_flush_async.argtypes = [vp]
    L.fhs_flush_async.restype = i
    L.fhs_download.argtypes = [vp, h, vp]
    L.fhs_download.restype = i
    L.fhs_export_device.argtypes = [vp, h, vp]
    L.fhs_export_device.restype = i
    L.fhs_export_device_async.argtypes = [vp, h, vp]
    L.fhs_export_device_async.restype = i
    L.fhs_stream_handle.argtypes = [vp]
    L.fhs_stream_handle.restype = vp
    L.fhs_set_arithmetic.argtypes = [vp, i]
    L.fhs_set_arithmetic.restype = i
    L.fhs_set_fft4_max_batch.argtypes = [vp, i]
    L.fhs_set_fft4_max_batch.restype = i
    L.fhs_get_arithmetic.argtypes = [vp]
    L.fhs_get_arithmetic.restype = i
    dp = C.POINTER(C.c_double)
    L.fhs_fft_tables.argtypes = [dp, dp, dp, dp]
    L.fhs_fft_tables.restype = None
    L.fhs_set_mode.argtypes = [vp, i]
    L.fhs_set_mode.restype = i
    for name in ("contains", "starts_with", "ends_with", "find", "rfind", "eq", "ne", "eq_ignore_case"):
        f = getattr(L, "fhs_str_" + name)
        f.argtypes = [vp, hp, sz, hp, sz, hp]
        f.restype = i
    for name in ("contains_clear", "find_clear"):
        f = getattr(L, "fhs_str_" + name)
        f.argtypes = [vp, hp, sz, C.c_char_p, sz, hp]
        f.restype = i
    for name in ("is_empty", "len"):
        f = getattr(L, "fhs_str_" + name)
        f.argtypes = [vp, hp, sz, hp]
        f.restype = i
    L.fhs_str_compare.argtypes = [vp, hp, sz, hp, sz, i, hp]
    L.fhs_str_compare.restype = i
    L.fhs_str_compare_partial.argtypes = [vp, hp, sz, hp, sz, i, hp, hp]
    L.fhs_str_compare_partial.restype = i
    for name in ("fhs_str_to_upper", "fhs_str_to_lower", "fhs_str_trim_end", "fhs_str_trim_start",
                 "fhs_str_trim", "fhs_bubble_zeroes_right"):
        f = getattr(L, name)
        f.argtypes = [vp, hp, sz, hp]
        f.restype = i
    L.fhs_str_replace_len.argtypes = [sz, sz, sz]
    L.fhs_str_replace_len.restype = sz
    L.fhs_str_replace.argtypes = [vp, hp, sz, hp, sz, hp, sz, hp, sz, C.POINTER(sz)]
    L.fhs_str_replace.restype = i
    L.fhs_str_replacen.argtypes = [vp, hp, sz, hp, sz, hp, sz, h, hp, sz, C.POINTER(sz)]
    L.fhs_str_replacen.restype = i
    L.fhs_str_repeat.argtypes = [vp, hp, sz, h, hp]
    L.fhs_str_repeat.restype = i
    L.fhs_str_repeat_clear.argtypes = [vp, hp, sz, sz, hp]
    L.fhs_str_repeat_clear.restype = i
    L.fhs_str_concatenate.argtypes = [vp, hp, sz, hp, sz, hp]
    L.fhs_str_concatenate.restype = i
    for name in ("strip_prefix", "strip_suffix"):
        f = getattr(L, "fhs_str_" + name)
        f.argtypes = [vp, hp, sz, hp, sz, hp, hp]
        f.restype = i
    for name in ("fhs_flags_or", "fhs_flags_and"):
        f = getattr(L, name)
        f.argtypes = [vp, hp, sz, hp]
        f.restype = i
    L.fhs_flags_first_decides.argtypes = [vp, hp, hp, sz, i, hp]
    L.fhs_flags_first_decides.restype = i
    L.fhs_str_split_dim.argtypes = [i, sz]
    L.fhs_str_split_dim.restype = sz
    L.fhs_str_split.argtypes = [vp, i, hp, sz, hp, sz, h, hp, sz, C.POINTER(sz), hp]
    L.fhs_str_split.restype = i
    L.fhs_dist_available.argtypes = []
    L.fhs_dist_available.restype = i
    L.fhs_dist_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(i)]
    L.fhs_dist_stats.restype = i
    L.fhs_dist_config.argtypes = [vp, i, i]
    L.fhs_dist_config.restype = i
    L.fhs_flush_plan.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.fhs_flush_plan.restype = i
    L.fhs_flush_level_exec.argtypes = [vp, u64, vp, C.POINTER(u64), C.POINTER(u64)]
    L.fhs_flush_level_exec.restype = i
    L.fhs_flush_level_commit.argtypes = [vp, u64, vp]
    L.fhs_flush_level_commit.restype = i
    L.fhs_stream_sync.argtypes = [vp]
    L.fhs_stream_sync.restype = i
    szp = C.POINTER(C.c_size_t)
    L.fhs_dist_unique_id.argtypes = [vp]
    L.fhs_dist_unique_id.restype = i
    L.fhs_dist_init.argtypes = [vp, i, i, C.c_char_p]
    L.fhs_dist_init.restype = i
    L.fhs_dist_init_host_transport.argtypes = [vp, i, i, vp, vp]
    L.fhs_dist_init_host_transport.restype = i
    L.fhs_dist_shutdown.argtypes = [vp]
    L.fhs_dist_shutdown.restype = i
    L.fhs_dist_abort.argtypes = [vp]
    L.fhs_dist_abort.restype = i
    L.fhs_dist_rank.argtypes = [vp]
    L.fhs_dist_rank.restype = i
    L.fhs_dist_world.argtypes = [vp]
    L.fhs_dist_world.restype = i
    L.fhs_dist_level_parallel.argtypes = [vp, i]
    L.fhs_dist_level_parallel.restype = i
    L.fhs_dist_plan_windows.argtypes = [sz, sz, i, i, szp, szp, szp, szp]
    L.fhs_dist_plan_windows.restype = None
    L.fhs_dist_plan_positions.argtypes = [sz, i, i, szp, szp]
    L.fhs_dist_plan_positions.restype = None
    L.fhs_dist_allgather_chars.argtypes = [vp, hp, sz, hp]
    L.fhs_dist_allgather_chars.restype = i
    L.fhs_dist_allgather_flags.argtypes = [vp, hp, sz, hp]
    L.fhs_dist_allgather_flags.restype = i
    L.fhs_dist_str_contains.argtypes = [vp, hp, sz, hp, sz, hp]
    L.fhs_dist_str_contains.restype = i
    L.fhs_dist_str_contains_clear.argtypes = [vp, hp, sz, C.c_char_p, sz, hp]
    L.fhs_dist_str_contains_clear.restype = i
    L.fhs_dist_str_find.argtypes = [vp, hp, sz, hp, sz, sz, sz, hp]
    L.fhs_dist_str_find.restype = i
    L.fhs_dist_str_find_clear.argtypes = [vp, hp, sz, C.c_char_p, sz, sz, sz, hp]
    L.fhs_dist_str_find_clear.restype = i
    L.fhs_dist_str_eq.argtypes = [vp, hp, sz, hp, sz, i, hp]
    L.fhs_dist_str_eq.restype = i
    L.fhs_dist_str_compare.argtypes = [vp, hp, sz, hp, sz, i, hp]
    L.fhs_dist_str_compare.restype = i
    L.fhs_debug_capture_pbs_inputs.argtypes = [vp, C.c_size_t]
    L.fhs_debug_capture_pbs_inputs.restype = i
    L.fhs_debug_plan_trace.argtypes = [vp, i]
    L.fhs_debug_plan_trace.restype = i
    L.fhs_debug_plan_read.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fhs_debug_plan_read.restype = i
    L.fhs_debug_char_terms.argtypes = [vp, C.c_uint64, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fhs_debug_char_terms.restype = i
    L.fhs_debug_lut_poly.argtypes = [i, vp]
    L.fhs_debug_lut_poly.restype = i
    L.fhs_debug_capture_live.argtypes = [vp, i]
    L.fhs_debug_capture_live.restype = i
    L.fhs_debug_capture_read.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fhs_debug_capture_read.restype = i
    L.fhs_set_launch_chunk.argtypes = [vp, i, sz]
    L.fhs_set_launch_chunk.restype = i
    L.fhs_set_tick_balance.argtypes = [vp, sz]
    L.fhs_set_tick_balance.restype = i
    L.fhs_resident_slots.argtypes = [vp]
    L.fhs_resident_slots.restype = i
    L.fhs_set_auto_flush.argtypes = [vp, sz]
    L.fhs_set_auto_flush.restype = i
    L.fhs_submit.argtypes = [vp]
    L.fhs_submit.restype = i
    L.fhs_pump.argtypes = [vp, sz]
    L.fhs_pump.restype = i
    L.fhs_level_widths.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.fhs_level_widths.restype = i
    L.fhs_pbs_batch_shifted.argtypes = [vp, vp, vp, vp, sz, vp, sz, vp, sz]
    L.fhs_pbs_batch_shifted.restype = i
    L.fhs_set_rotation_sharing.argtypes = [vp, i]
    L.fhs_set_rotation_sharing.restype = i
    L.fhs_launch_groups.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.fhs_launch_groups.restype = i
    L.fhs_get_stats.argtypes = [vp, vp]
    L.fhs_get_stats.restype = i
    L.fhs_char_sum_c2.argtypes = [vp, C.c_uint64, vp]
    L.fhs_char_sum_c2.restype = i
    L.fhs_char_set_noise.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.fhs_char_set_noise.restype = i
    L.fhs_trivial_value.argtypes = [vp, C.c_uint64, vp, vp]
    L.fhs_trivial_value.restype = i
    L.fhs_reset_stats.argtypes = [vp]
    L.fhs_reset_stats.restype = i
    L.fhs_client_create.argtypes = [C.POINTER(vp)]
    L.fhs_client_create.restype = i
    L.fhs_client_create_insecure_seeded.argtypes = [u64, C.POINTER(vp)]
    L.fhs_client_create_insecure_seeded.restype = i
    L.fhs_chacha20_block.argtypes = [vp, C.c_uint32, vp, vp]
    L.fhs_chacha20_block.restype = None
    L.fhs_client_destroy.argtypes = [vp]
    L.fhs_client_destroy.restype = None
    L.fhs_client_bsk.argtypes = [vp]
    L.fhs_client_bsk.restype = C.POINTER(C.c_uint64)
    L.fhs_client_ksk.argtypes = [vp]
    L.fhs_client_ksk.restype = C.POINTER(C.c_uint64)
    L.fhs_client_bsk_mb2.argtypes = [vp]
    L.fhs_client_bsk_mb2.restype = C.POINTER(C.c_uint64)
    L.fhs_client_save_multibit_key.argtypes = [vp, C.c_char_p]
    L.fhs_client_save_multibit_key.restype = i
    L.fhs_load_multibit_key_file.argtypes = [vp, C.c_char_p]
    L.fhs_load_multibit_key_file.restype = i
    L.fhs_debug_blind_rotate_batch.argtypes = [vp, vp, vp, vp, sz, vp, sz]
    L.fhs_debug_blind_rotate_batch.restype = i
    L.fhs_load_multibit_key.argtypes = [vp, vp]
    L.fhs_load_multibit_key.restype = i
    L.fhs_client_encrypt_char.argtypes = [vp, u8, vp]
    L.fhs_client_encrypt_char.restype = i
    L.fhs_client_decrypt_char.argtypes = [vp, vp, C.POINTER(u8)]
    L.fhs_client_decrypt_char.restype = i
    L.fhs_client_encrypt_str.argtypes = [vp, C.c_char_p, sz, sz, vp]
    L.fhs_client_encrypt_str.restype = i
    L.fhs_client_decrypt_str.argtypes = [vp, vp, sz, C.c_char_p, C.POINTER(sz)]
    L.fhs_client_decrypt_str.restype = i
    L.fhs_client_save.argtypes = [vp, C.c_char_p, i]
    L.fhs_client_save.restype = i
    L.fhs_client_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fhs_client_load.restype = i
    L.fhs_load_server_key_file.argtypes = [vp, C.c_char_p]
    L.fhs_load_server_key_file.restype = i
    L.fhs_client_secret_keys.argtypes = [vp, vp, vp]
    L.fhs_client_secret_keys.restype = i


class CaptureRec(C.Structure):
    _fields_ = [("level", C.c_uint32), ("index", C.c_uint32), ("lut", C.c_uint32), ("n_terms", C.c_uint32),
                ("sum_c2", C.c_int64), ("konst", C.c_int32), ("width", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("pbs_executed", C.c_uint64), ("pbs_folded", C.c_uint64), ("levels", C.c_uint64),
                ("max_level_width", C.c_uint64), ("blocks_live", C.c_uint64), ("max_input_sum_c2", C.c_uint64), ("pbs_shared", C.c_uint64),
                ("pbs_extracted", C.c_uint64)]


def fft_tables():
    """Host-derived twiddle tables of the F64_FFT arithmetic (diagnostic)."""
    import numpy as np
    w_re, w_im = np.zeros(1024), np.zeros(1024)
    u_re, u_im = np.zeros(16), np.zeros(16)
    dp = C.POINTER(C.c_double)
    lib().fhs_fft_tables(*(a.ctypes.data_as(dp) for a in (w_re, w_im, u_re, u_im)))
    return w_re, w_im, u_re, u_im
