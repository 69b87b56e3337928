"""ctypes loader for the product library libllcomp_mi.so (HIP kernels + C ABI, include/llcomp_mi.h).

There is deliberately no fallback: if the library is missing this raises, and if there is no HIP device the
library's calls return LLCOMP_MI_NO_DEVICE -- nothing in this package can code a single byte on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LLCOMP_MI_LIB: load another build of the same sources instead (tools/clock_probe.py loads the diagnostic library with
# in-kernel clock stamps, csrc/Makefile `probe`); the tests, bench.py and smoke() never set it
LIB_PATH = os.environ.get("LLCOMP_MI_LIB") or os.path.join(_HERE, "libllcomp_mi.so")

ABI_VERSION = 4  # LLCOMP_MI_ABI_VERSION of the header this binding mirrors

# every symbol include/llcomp_mi.h declares (tests/test_abi.py checks the header against this list and the .so)
SYMBOLS = [
    "llcomp_mi_encode", "llcomp_mi_decode", "llcomp_mi_free", "llcomp_mi_strerror", "llcomp_mi_abi_version",
    "llcomp_mi_device_count", "llcomp_mi_probe", "llcomp_mi_slice_count", "llcomp_mi_merge_bands",
    "llcomp_mi_split_band", "llcomp_mi_codec_create", "llcomp_mi_codec_destroy", "llcomp_mi_codec_slices", "llcomp_mi_codec_kernel_family",
    "llcomp_mi_codec_workspace_bytes", "llcomp_mi_codec_max_payload_bytes", "llcomp_mi_codec_encode",
    "llcomp_mi_codec_decode", "llcomp_mi_codec_model", "llcomp_mi_status_from_bits",
    "llcomp_mi_codec_set_profiling", "llcomp_mi_codec_get_profile",
    "llcomp_mi_encode_into", "llcomp_mi_decode_into", "llcomp_mi_host_alloc", "llcomp_mi_host_free",
    "llcomp_mi_reload_tuning", "llcomp_mi_trim", "llcomp_mi_device_copy_segments", "llcomp_mi_decode_flags", "llcomp_mi_codec_create_ex",
    "llcomp_mi_stream_create", "llcomp_mi_stream_create_ex", "llcomp_mi_stream_frames_per_job", "llcomp_mi_stream_submit_decode_batch",
    "llcomp_mi_stream_result_part", "llcomp_mi_stream_destroy", "llcomp_mi_stream_container_capacity",
    "llcomp_mi_stream_submit_encode", "llcomp_mi_stream_submit_decode", "llcomp_mi_stream_pending",
    "llcomp_mi_stream_poll", "llcomp_mi_stream_wait", "llcomp_mi_stream_release",
    "llcomp_mi_set_pool_limit", "llcomp_mi_pool_limit", "llcomp_mi_pool_idle_bytes", "llcomp_mi_fnv1a64", "llcomp_mi_suggest_tile_w", "llcomp_mi_decode_into_flags", "llcomp_mi_device_range_sums",
    "llcomp_mi_decode_devices", "llcomp_mi_decode_into_devices", "llcomp_mi_last_device_error", "llcomp_mi_plan_chunks",
    "llcomp_mi_stream_create_multi", "llcomp_mi_stream_devices", "llcomp_mi_codec_get_counters", "llcomp_mi_codec_prepare",
]

u8p = C.POINTER(C.c_uint8)


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("format", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                ("planar", C.c_uint32), ("device", C.c_int32), ("small_model", C.c_uint32),
                ("n_devices", C.c_uint32), ("devices", C.POINTER(C.c_int32)), ("chunks_per_device", C.c_uint32), ("reserved", C.c_uint32)]


class Info(C.Structure):
    _fields_ = [("format", C.c_uint32), ("channels", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("planar", C.c_uint32), ("n_slices", C.c_uint32),
                ("table_offset", C.c_uint64), ("payload_offset", C.c_uint64), ("small_model", C.c_uint32), ("reserved", C.c_uint32)]


class StreamResult(C.Structure):
    _fields_ = [("slot", C.c_uint32), ("kind", C.c_uint32), ("status", C.c_int32), ("reserved", C.c_uint32),
                ("tag", C.c_uint64), ("data", C.c_void_p), ("len", C.c_uint64)]


_lib = None


def _preload_hip_runtime():
    """libllcomp_mi.so is linked without its own HIP runtime (see csrc/Makefile): exactly one libamdhip64 may live
    in a process.  PyTorch wheels bundle a private copy, so when torch is installed that copy is the one to share
    (streams and device pointers handed over from torch then belong to the same runtime); otherwise /opt/rocm's."""
    import importlib.util

    cands = []
    spec = importlib.util.find_spec("torch")
    if spec is not None and spec.origin:
        cands.append(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so.7"), os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so"]
    errs = []
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:  # keep looking
            errs.append(f"{c}: {e}")
    raise ImportError("no HIP runtime (libamdhip64) found for libllcomp_mi.so: " + "; ".join(errs))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `make -C llcomp_amd/csrc`). llcomp_amd has no CPU fallback.")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.llcomp_mi_encode.restype = C.c_int
    L.llcomp_mi_encode.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Opts), C.POINTER(u8p), C.POINTER(C.c_size_t)]
    L.llcomp_mi_decode.restype = C.c_int
    L.llcomp_mi_decode.argtypes = [u8p, C.c_size_t, C.c_int32, C.POINTER(u8p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.llcomp_mi_free.restype = None
    L.llcomp_mi_free.argtypes = [C.c_void_p]
    L.llcomp_mi_strerror.restype = C.c_char_p
    L.llcomp_mi_strerror.argtypes = [C.c_int]
    L.llcomp_mi_abi_version.restype = C.c_int
    L.llcomp_mi_device_count.restype = C.c_int
    L.llcomp_mi_probe.restype = C.c_int
    L.llcomp_mi_probe.argtypes = [u8p, C.c_size_t, C.POINTER(Info)]
    L.llcomp_mi_slice_count.restype = C.c_uint32
    L.llcomp_mi_slice_count.argtypes = [C.c_uint32] * 6
    L.llcomp_mi_merge_bands.restype = C.c_int
    L.llcomp_mi_merge_bands.argtypes = [C.POINTER(u8p), C.POINTER(C.c_size_t), C.c_uint32, C.POINTER(u8p), C.POINTER(C.c_size_t)]
    L.llcomp_mi_split_band.restype = C.c_int
    L.llcomp_mi_split_band.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(u8p), C.POINTER(C.c_size_t)]
    L.llcomp_mi_codec_create.restype = C.c_int
    L.llcomp_mi_codec_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32] + [C.c_uint32] * 7
    L.llcomp_mi_codec_destroy.restype = None
    L.llcomp_mi_codec_destroy.argtypes = [C.c_void_p]
    L.llcomp_mi_codec_slices.restype = C.c_uint32
    L.llcomp_mi_codec_slices.argtypes = [C.c_void_p]
    if hasattr(L, "llcomp_mi_codec_kernel_family"):  # (a stale library must reach the ABI check below, not an AttributeError)
        L.llcomp_mi_codec_kernel_family.restype = C.c_uint32
        L.llcomp_mi_codec_kernel_family.argtypes = [C.c_void_p]
    L.llcomp_mi_codec_workspace_bytes.restype = C.c_uint64
    L.llcomp_mi_codec_workspace_bytes.argtypes = [C.c_void_p]
    L.llcomp_mi_codec_max_payload_bytes.restype = C.c_uint64
    L.llcomp_mi_codec_max_payload_bytes.argtypes = [C.c_void_p]
    L.llcomp_mi_codec_encode.restype = C.c_int
    L.llcomp_mi_codec_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.llcomp_mi_codec_decode.restype = C.c_int
    L.llcomp_mi_codec_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.llcomp_mi_codec_model.restype = C.c_int
    L.llcomp_mi_codec_model.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.llcomp_mi_status_from_bits.restype = C.c_uint32
    L.llcomp_mi_status_from_bits.argtypes = [C.c_uint32]
    L.llcomp_mi_codec_set_profiling.restype = C.c_int
    L.llcomp_mi_codec_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.llcomp_mi_codec_get_profile.restype = C.c_int
    L.llcomp_mi_codec_get_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.llcomp_mi_encode_into.restype = C.c_int
    L.llcomp_mi_encode_into.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(Opts), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.llcomp_mi_decode_into.restype = C.c_int
    L.llcomp_mi_decode_into.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if "LLCOMP_MI_LIB" not in os.environ or hasattr(L, "llcomp_mi_decode_into_flags"):
        L.llcomp_mi_decode_into_flags.restype = C.c_int
        L.llcomp_mi_decode_into_flags.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.llcomp_mi_host_alloc.restype = C.c_void_p
    L.llcomp_mi_host_alloc.argtypes = [C.c_size_t]
    L.llcomp_mi_host_free.restype = None
    L.llcomp_mi_host_free.argtypes = [C.c_void_p]
    L.llcomp_mi_device_copy_segments.restype = C.c_int
    L.llcomp_mi_device_copy_segments.argtypes = [C.c_void_p] * 5 + [C.c_uint32, C.c_uint64, C.c_void_p]
    if "LLCOMP_MI_LIB" not in os.environ or hasattr(L, "llcomp_mi_device_range_sums"):
        L.llcomp_mi_device_range_sums.restype = C.c_int
        L.llcomp_mi_device_range_sums.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_void_p]
    L.llcomp_mi_decode_flags.restype = C.c_int
    L.llcomp_mi_decode_flags.argtypes = [u8p, C.c_size_t, C.c_int32, C.c_uint32, C.POINTER(u8p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.llcomp_mi_codec_create_ex.restype = C.c_int
    L.llcomp_mi_codec_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int32] + [C.c_uint32] * 8
    L.llcomp_mi_trim.restype = None
    L.llcomp_mi_trim.argtypes = []
    if "LLCOMP_MI_LIB" not in os.environ or hasattr(L, "llcomp_mi_set_pool_limit"):  # (an A/B build of older sources may lack them)
        L.llcomp_mi_set_pool_limit.restype = None
        L.llcomp_mi_set_pool_limit.argtypes = [C.c_uint64]
        L.llcomp_mi_pool_limit.restype = C.c_uint64
        L.llcomp_mi_pool_limit.argtypes = []
        L.llcomp_mi_pool_idle_bytes.restype = C.c_uint64
        L.llcomp_mi_pool_idle_bytes.argtypes = []
    if "LLCOMP_MI_LIB" not in os.environ or hasattr(L, "llcomp_mi_fnv1a64"):
        L.llcomp_mi_fnv1a64.restype = C.c_uint64
        L.llcomp_mi_fnv1a64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    if "LLCOMP_MI_LIB" not in os.environ or hasattr(L, "llcomp_mi_suggest_tile_w"):
        L.llcomp_mi_suggest_tile_w.restype = C.c_uint32
        L.llcomp_mi_suggest_tile_w.argtypes = [C.c_uint32] * 5
    L.llcomp_mi_reload_tuning.restype = None
    L.llcomp_mi_reload_tuning.argtypes = []
    L.llcomp_mi_stream_create.restype = C.c_int
    L.llcomp_mi_stream_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32] + [C.c_uint32] * 7
    L.llcomp_mi_stream_create_ex.restype = C.c_int
    L.llcomp_mi_stream_create_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int32] + [C.c_uint32] * 8
    L.llcomp_mi_stream_frames_per_job.restype = C.c_uint32
    L.llcomp_mi_stream_frames_per_job.argtypes = [C.c_void_p]
    L.llcomp_mi_stream_submit_decode_batch.restype = C.c_int
    L.llcomp_mi_stream_submit_decode_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint64]
    L.llcomp_mi_stream_result_part.restype = C.c_int
    L.llcomp_mi_stream_result_part.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.llcomp_mi_stream_destroy.restype = None
    L.llcomp_mi_stream_destroy.argtypes = [C.c_void_p]
    L.llcomp_mi_stream_container_capacity.restype = C.c_uint64
    L.llcomp_mi_stream_container_capacity.argtypes = [C.c_void_p]
    L.llcomp_mi_stream_submit_encode.restype = C.c_int
    L.llcomp_mi_stream_submit_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.llcomp_mi_stream_submit_decode.restype = C.c_int
    L.llcomp_mi_stream_submit_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
    L.llcomp_mi_stream_pending.restype = C.c_int
    L.llcomp_mi_stream_pending.argtypes = [C.c_void_p]
    L.llcomp_mi_stream_poll.restype = C.c_int
    L.llcomp_mi_stream_poll.argtypes = [C.c_void_p]
    L.llcomp_mi_stream_wait.restype = C.c_int
    L.llcomp_mi_stream_wait.argtypes = [C.c_void_p, C.POINTER(StreamResult)]
    L.llcomp_mi_stream_release.restype = C.c_int
    L.llcomp_mi_stream_release.argtypes = [C.c_void_p, C.c_uint32]
    old_ab_build = "LLCOMP_MI_LIB" in os.environ and L.llcomp_mi_abi_version() != ABI_VERSION  # (tools/lib_ab.sh against earlier commits)
    if not old_ab_build:
        i32p = C.POINTER(C.c_int32)
        L.llcomp_mi_decode_devices.restype = C.c_int
        L.llcomp_mi_decode_devices.argtypes = [u8p, C.c_size_t, i32p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(u8p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.llcomp_mi_decode_into_devices.restype = C.c_int
        L.llcomp_mi_decode_into_devices.argtypes = [C.c_void_p, C.c_size_t, i32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.llcomp_mi_last_device_error.restype = C.c_int
        L.llcomp_mi_last_device_error.argtypes = [i32p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.llcomp_mi_plan_chunks.restype = C.c_int
        L.llcomp_mi_plan_chunks.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
        L.llcomp_mi_stream_create_multi.restype = C.c_int
        L.llcomp_mi_stream_create_multi.argtypes = [C.POINTER(C.c_void_p), i32p, C.c_uint32] + [C.c_uint32] * 8
        L.llcomp_mi_stream_devices.restype = C.c_uint32
        L.llcomp_mi_stream_devices.argtypes = [C.c_void_p]
        L.llcomp_mi_codec_prepare.restype = C.c_int
        L.llcomp_mi_codec_prepare.argtypes = [C.c_void_p, C.c_uint32]
        L.llcomp_mi_codec_get_counters.restype = C.c_int
        L.llcomp_mi_codec_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_int]
    if "LLCOMP_MI_LIB" not in os.environ and L.llcomp_mi_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {L.llcomp_mi_abi_version()}, this binding was written for {ABI_VERSION}: rebuild the library")
    _lib = L
    return L
