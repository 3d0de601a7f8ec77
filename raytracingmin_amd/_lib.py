"""ctypes binding of the C ABI in include/rtm.h (librtm_hip.so, built by csrc/Makefile).

The HIP library is the product path: importing fails loudly when it has not been built, and
nothing here falls back to a CPU implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTM_LIB_OVERRIDE") or os.path.join(_HERE, "librtm_hip.so")  # override: A/B of builds

RTM_OK = 0
MODE_LITERAL, MODE_REPAIRED = 0, 1
MODE_HOST_TRIG = 0x100  # flag: sin/cos exactly as the host libm returns them (include/rtm.h)
MODE_COUNT_TESTS = 0x200  # flag: count Intersect evaluations into rtm_stats.object_tests (diagnostic)
MODE_SURFACE_SAMPLE = 0x400  # flag: the integrator is png::SurfaeSample instead of png::PathTracing
MODES = {"literal": MODE_LITERAL, "repaired": MODE_REPAIRED, 0: 0, 1: 1}


class RtmError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: {detail}" if detail else what)


class rtm_camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3),
                ("fov", C.c_float), ("_pad", C.c_float)]


class rtm_sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("color", C.c_double * 3),
                ("emission", C.c_double * 3), ("radius", C.c_float), ("_pad", C.c_float)]


class rtm_object(C.Structure):
    _fields_ = [("type", C.c_int32), ("size", C.c_float), ("position", C.c_double * 3),
                ("color", C.c_double * 3), ("emission", C.c_double * 3), ("up", C.c_double * 3),
                ("target", C.c_double * 3), ("width", C.c_double)]


OBJECT_SPHERE, OBJECT_PLANE = 1, 2


class rtm_settings(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32),
                ("super_samples", C.c_int32), ("camera", rtm_camera)]


class rtm_options(C.Structure):
    _fields_ = [("mode", C.c_int32), ("max_bounces", C.c_int32), ("seed", C.c_uint64),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("device", C.c_int32),
                ("variant", C.c_int32), ("band_count", C.c_int32), ("band_index", C.c_int32)]


class rtm_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("casts", C.c_uint64), ("bounces", C.c_uint64),
                ("draws", C.c_uint64), ("kernel_ms", C.c_double), ("variant", C.c_int32),
                ("split", C.c_int32), ("object_tests", C.c_uint64)]

    def as_dict(self):
        return {"samples": int(self.samples), "casts": int(self.casts),
                "bounces": int(self.bounces), "draws": int(self.draws),
                "kernel_ms": float(self.kernel_ms), "variant": int(self.variant),
                "split": int(self.split), "object_tests": int(self.object_tests)}


# every symbol include/rtm.h declares: name -> (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "rtm_abi_version": (C.c_int, []),
    "rtm_strerror": (C.c_char_p, [C.c_int]),
    "rtm_last_error_detail": (C.c_char_p, []),
    "rtm_device_count": (C.c_int, [_P(C.c_int)]),
    "rtm_output_rows": (C.c_int, [_P(rtm_options)]),
    "rtm_release_scratch": (C.c_int, [C.c_int]),
    "rtm_stream_release": (C.c_int, [C.c_int, C.c_void_p]),
    "rtm_scratch_bytes": (C.c_int, [_P(rtm_settings), C.c_void_p, _P(rtm_options), _P(C.c_uint64)]),
    "rtm_num_variants": (C.c_int, []),
    "rtm_variant_name": (C.c_char_p, [C.c_int]),
    "rtm_scene_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, _P(C.c_void_p)]),
    "rtm_scene_create_objects": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, _P(C.c_void_p)]),
    "rtm_scene_destroy": (C.c_int, [C.c_void_p]),
    "rtm_scene_size": (C.c_size_t, [C.c_void_p]),
    "rtm_stream_status": (C.c_int, [C.c_int, C.c_void_p]),
    "rtm_render_scene": (C.c_int, [_P(rtm_settings), C.c_void_p, _P(rtm_options), C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, _P(rtm_stats)]),
    "rtm_render_device": (C.c_int, [_P(rtm_settings), C.c_void_p, C.c_size_t, C.c_int,
                                    _P(rtm_options), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, _P(rtm_stats)]),
    "rtm_render": (C.c_int, [_P(rtm_settings), C.c_void_p, C.c_size_t, _P(rtm_options),
                             C.c_void_p, C.c_void_p, C.c_void_p, _P(rtm_stats)]),
    "rtm_render_objects": (C.c_int, [_P(rtm_settings), C.c_void_p, C.c_size_t, _P(rtm_options),
                                     C.c_void_p, C.c_void_p, C.c_void_p, _P(rtm_stats)]),
    "rtm_path_trace_batch": (C.c_int, [C.c_void_p, C.c_size_t, _P(rtm_options), C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtm_surface_sample_batch": (C.c_int, [C.c_void_p, C.c_size_t, _P(rtm_options), C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtm_intersect_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtm_intersect_objects_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtm_rng_u01": (C.c_double, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rtm_rng_batch": (C.c_int, [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_void_p]),
    "rtm_scene_load_json": (C.c_int, [C.c_char_p, C.c_int, _P(rtm_settings), C.c_void_p,
                                      C.c_size_t, _P(C.c_size_t)]),
    "rtm_scene_parse_json": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, _P(rtm_settings),
                                       C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "rtm_scene_load_json_objects": (C.c_int, [C.c_char_p, C.c_int, _P(rtm_settings), C.c_void_p,
                                              C.c_size_t, _P(C.c_size_t)]),
    "rtm_scene_parse_json_objects": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, _P(rtm_settings),
                                               C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "rtm_scene_save_sample_json": (C.c_int, [C.c_char_p]),
    "rtm_scene_make_stress": (C.c_int, [C.c_uint64, C.c_size_t, _P(rtm_settings), C.c_void_p]),
    "rtm_quantise": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtm_write_bmp": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rtm_write_jpg": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
}
# test and diagnostic hooks: include/rtm_debug.h (not part of the drop-in boundary)
DEBUG_SIGNATURES = {
    "rtm_debug_math_probe": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtm_debug_selfcheck": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "rtm_debug_fp64_peak": (C.c_int, [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rtm_debug_wf_nearest": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_void_p]),
    "rtm_debug_grid_nearest": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtm_debug_scene_facts": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtm_debug_grid_build": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.c_size_t]),
    "rtm_debug_component_bench": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_double)]),
}
ABI_VERSION = 5

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 (same SONAME as
    /opt/rocm's); two copies in one process leave the second without a device.  Importing torch
    BEFORE librtm_hip.so is loaded makes the dynamic loader resolve our libamdhip64.so.7
    dependency to the copy torch already mapped, so tensors, streams and RCCL buffers torch owns
    are valid in our kernels.  Without torch (rtm_cli, plain C callers) /opt/rocm's is used."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    """The loaded HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C raytracingmin_amd/csrc`. There is no CPU fallback.")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        if L.rtm_abi_version() != ABI_VERSION:
            raise ImportError("librtm_hip.so has an unexpected ABI version")
        _lib = L
    return _lib


def check(status, what):
    if status != RTM_OK:
        L = lib()
        raise RtmError(status, f"{what}: {L.rtm_strerror(status).decode()}",
                       (L.rtm_last_error_detail() or b"").decode())
    return status
