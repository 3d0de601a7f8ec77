"""ctypes binding to oracle/liboracle_cpu.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Also holds an independent pure-Python reader of the reference's scene JSON schema
(reference src/SettingData.cpp:129-186) so the product's C++ loader can be
cross-checked against it.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SCENES = os.path.join(ROOT, "scenes")

MODE_LITERAL, MODE_REPAIRED = 0, 1


class Camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3),
                ("fov", C.c_float), ("_pad", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("color", C.c_double * 3),
                ("emission", C.c_double * 3), ("radius", C.c_float), ("_pad", C.c_float)]


class Object(C.Structure):  # include/rtm.h: rtm_object
    _fields_ = [("type", C.c_int32), ("size", C.c_float), ("position", C.c_double * 3),
                ("color", C.c_double * 3), ("emission", C.c_double * 3), ("up", C.c_double * 3),
                ("target", C.c_double * 3), ("width", C.c_double)]


class Settings(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32),
                ("super_samples", C.c_int32), ("camera", Camera)]


class Options(C.Structure):
    _fields_ = [("mode", C.c_int32), ("max_bounces", C.c_int32), ("seed", C.c_uint64),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("device", C.c_int32),
                ("variant", C.c_int32), ("band_count", C.c_int32), ("band_index", C.c_int32)]  # include/rtm.h layout


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "casts", "bounces", "draws", "sphere_tests",
                                          "sphere_tests_d4", "libc_rand_calls", "max_depth")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


RNG_FN = C.CFUNCTYPE(C.c_double, C.c_void_p)
_D3 = C.c_double * 3
_lib = None


def build(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle_cpu.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("cpu_ref.c", "cpu_ref.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle_cpu.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.rtmo_intersect.restype = C.c_int
        L.rtmo_intersect.argtypes = [C.POINTER(Sphere), _D3, _D3, C.c_int, C.POINTER(C.c_double), _D3]
        L.rtmo_path_trace.restype = None
        L.rtmo_path_trace.argtypes = [C.POINTER(Sphere), C.c_size_t, C.c_int, C.c_int, _D3, _D3,
                                      RNG_FN, C.c_void_p, _D3, C.POINTER(Counters)]
        L.rtmo_path_trace_stream.restype = None
        L.rtmo_path_trace_stream.argtypes = [C.POINTER(Sphere), C.c_size_t, C.c_int, C.c_int, _D3,
                                             _D3, C.c_uint64, C.c_uint32, C.c_uint32, _D3,
                                             C.POINTER(Counters)]
        L.rtmo_surface_sample_stream.restype = None
        L.rtmo_surface_sample_stream.argtypes = [C.POINTER(Sphere), C.c_size_t, C.c_int, C.c_int, _D3, _D3, C.c_uint64,
                                                 C.c_uint32, C.c_uint32, _D3, C.POINTER(Counters)]
        L.rtmo_surface_sample.restype = None
        L.rtmo_surface_sample.argtypes = [C.POINTER(Sphere), C.c_size_t, C.c_int, C.c_int, _D3, _D3, RNG_FN, C.c_void_p, _D3,
                                          C.POINTER(Counters)]
        L.rtmo_render.restype = C.c_int
        L.rtmo_render.argtypes = [C.POINTER(Settings), C.POINTER(Sphere), C.c_size_t,
                                  C.POINTER(Options), C.c_void_p, C.POINTER(Counters), C.c_int,
                                  C.c_int]
        L.rtmo_render_pixels.restype = C.c_int
        L.rtmo_render_pixels.argtypes = [C.POINTER(Settings), C.POINTER(Sphere), C.c_size_t, C.POINTER(Options),
                                         C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Counters), C.c_int]
        L.rtmo_render_objects.restype = C.c_int
        L.rtmo_render_objects.argtypes = [C.POINTER(Settings), C.POINTER(Object), C.c_size_t, C.POINTER(Options),
                                          C.c_void_p, C.POINTER(Counters), C.c_int]
        L.rtmo_intersect_object.restype = C.c_int
        L.rtmo_intersect_object.argtypes = [C.POINTER(Object), _D3, _D3, C.c_int, C.POINTER(C.c_double), _D3]
        L.rtmo_sample_radiance.restype = None
        L.rtmo_sample_radiance.argtypes = [C.POINTER(Settings), C.POINTER(Sphere), C.c_size_t,
                                           C.POINTER(Options), C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, _D3, C.POINTER(Counters)]
        L.rtmo_primary_dir.restype = None
        L.rtmo_primary_dir.argtypes = [C.POINTER(Settings), C.c_int, C.c_int, C.c_int, C.c_int, _D3]
        L.rtmo_camera_basis.restype = None
        L.rtmo_camera_basis.argtypes = [C.POINTER(Settings), _D3, _D3, _D3,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rtmo_rng_u01.restype = C.c_double
        L.rtmo_rng_u01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.rtmo_sin_cos_array.restype = None
        L.rtmo_sin_cos_array.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rtmo_normalize.restype = None
        L.rtmo_normalize.argtypes = [_D3, _D3]
        L.rtmo_magnitude.restype = C.c_double
        L.rtmo_magnitude.argtypes = [_D3]
        L.rtmo_kd.restype = C.c_float
        L.rtmo_kd.argtypes = [C.POINTER(Sphere)]
        L.rtmo_color_kd.restype = None
        L.rtmo_color_kd.argtypes = [C.POINTER(Sphere), _D3]
        L.rtmo_quantise.restype = None
        L.rtmo_quantise.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.rtmo_fnv1a64_f64.restype = C.c_uint64
        L.rtmo_fnv1a64_f64.argtypes = [C.c_void_p, C.c_size_t]
        L.rtmo_make_stress_scene.restype = None
        L.rtmo_make_stress_scene.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(Settings),
                                             C.POINTER(Sphere)]
        L.rtmo_max_threads.restype = C.c_int
        _lib = L
    return _lib


# ---------------------------------------------------------------- scene reading (pure Python)
def load_scene(path, literal_loader=False, width=None, height=None, samples=None,
               super_samples=None):
    """Independent reader of the reference schema; returns (Settings, Sphere array)."""
    with open(path) as f:
        j = json.load(f)
    st = Settings()
    st.width = int(j["00 width"]) if width is None else width
    st.height = int(j["00 height"]) if height is None else height
    s = j.get("00 samples", j.get("00 sample", 10))
    st.samples = int(s) if samples is None else samples
    st.super_samples = int(j.get("00 superSamples", 1)) if super_samples is None else super_samples
    cam = j["01 camera"]
    for k in range(3):
        st.camera.origin[k] = float(cam["origin"][k])
        st.camera.target[k] = float(cam["target"][k])
        st.camera.up[k] = float(cam["upVec"][k])
    st.camera.fov = float(cam["fov"])
    objs = [o for o in j.get("02 scene", {}).get("00 object", []) if "00 position" in o]
    arr = (Sphere * max(len(objs), 1))()
    for i, o in enumerate(objs):
        if int(o.get("00 objectType", 1)) != 1:
            raise ValueError("unsupported objectType")
        p = [float(v) for v in o["00 position"]]
        if literal_loader:  # reference src/SettingData.cpp:165-167 (D1): posi.x assigned 3 times
            p = [p[2], 0.0, 0.0]
        for k in range(3):
            arr[i].center[k] = p[k]
            arr[i].color[k] = float(o["02 material"]["color"][k])
            arr[i].emission[k] = float(o["02 material"]["emission"][k])
        arr[i].radius = float(o["01 size"])  # double -> float, like SphereObject's ctor
    return st, arr, len(objs)


def scene_path(name):
    return os.path.join(SCENES, name)


def make_options(mode=MODE_REPAIRED, max_bounces=-1, seed=0x5EED, row_begin=0, row_end=None,
                 height=None, device=0, variant=0):
    o = Options()
    o.mode, o.max_bounces, o.seed = mode, max_bounces, seed
    o.row_begin = row_begin
    o.row_end = height if row_end is None else row_end
    o.device, o.variant = device, variant
    return o


def render(st, spheres, n, opt, threads=0, structure=0, want_counters=True):
    rows = opt.row_end - opt.row_begin
    out = np.zeros((rows, st.width, 3), dtype=np.float64)
    cnt = Counters()
    rc = lib().rtmo_render(C.byref(st), spheres, n, C.byref(opt), out.ctypes.data,
                           C.byref(cnt) if want_counters else None, threads, structure)
    if rc != 0:
        raise RuntimeError(f"rtmo_render failed: {rc}")
    return out, cnt.as_dict()


def render_pixels(st, spheres, n, opt, xy, threads=0):
    """The oracle's image values of the listed pixels: xy = [(x, y), ...] -> (len(xy), 3) float64."""
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((xy.shape[0], 3), dtype=np.float64)
    cnt = Counters()
    rc = lib().rtmo_render_pixels(C.byref(st), spheres, n, C.byref(opt), xy.ctypes.data, xy.shape[0],
                                  out.ctypes.data, C.byref(cnt), threads)
    if rc != 0:
        raise RuntimeError(f"rtmo_render_pixels failed: {rc}")
    return out, cnt.as_dict()


def render_objects(st, objects, n, opt, threads=0):
    """rtmo_render_objects: a scene of rtm_object entries (spheres and planes)."""
    rows = opt.row_end - opt.row_begin
    out = np.zeros((rows, st.width, 3), dtype=np.float64)
    cnt = Counters()
    rc = lib().rtmo_render_objects(C.byref(st), objects, n, C.byref(opt), out.ctypes.data, C.byref(cnt), threads)
    if rc != 0:
        raise RuntimeError(f"rtmo_render_objects failed: {rc}")
    return out, cnt.as_dict()


def intersect_object(obj, org, direction, mode, t_init=-1.0, n_init=7.0):
    t = C.c_double(t_init)
    nrm = _D3(n_init, n_init, n_init)
    hit = lib().rtmo_intersect_object(C.byref(obj), _D3(*org), _D3(*direction), mode, C.byref(t), nrm)
    return hit, t.value, [nrm[0], nrm[1], nrm[2]]


def path_trace(spheres, n, mode, max_bounces, org, direction, rng):
    """rng: python callable returning floats; returns (radiance[3], counters)."""
    cb = RNG_FN(lambda _ctx: float(rng()))
    out = _D3()
    cnt = Counters()
    lib().rtmo_path_trace(spheres, n, mode, max_bounces, _D3(*org), _D3(*direction), cb, None,
                          out, C.byref(cnt))
    return [out[0], out[1], out[2]], cnt.as_dict()


def path_trace_stream(spheres, n, mode, max_bounces, org, direction, seed, pixel, sample=0):
    out = _D3()
    cnt = Counters()
    lib().rtmo_path_trace_stream(spheres, n, mode, max_bounces, _D3(*org), _D3(*direction), seed,
                                 pixel, sample, out, C.byref(cnt))
    return [out[0], out[1], out[2]], cnt.as_dict()


MODE_SURFACE_SAMPLE = 0x400  # include/rtm.h: the integrator is png::SurfaeSample


def surface_sample_stream(spheres, n, mode, max_bounces, org, direction, seed, pixel, sample=0):
    """png::SurfaeSample (src/Renderer.cpp:119-198) entered at depth 0, drawing from the build RNG stream."""
    out = _D3()
    cnt = Counters()
    lib().rtmo_surface_sample_stream(spheres, n, mode, max_bounces, _D3(*org), _D3(*direction), C.c_uint64(seed),
                                     C.c_uint32(pixel), C.c_uint32(sample), out, C.byref(cnt))
    return [out[0], out[1], out[2]], cnt.as_dict()


def surface_sample(spheres, n, mode, max_bounces, org, direction, rng):
    """The same with a python callable as the generator; returns (radiance[3], counters)."""
    cb = RNG_FN(lambda _ctx: float(rng()))
    out = _D3()
    cnt = Counters()
    lib().rtmo_surface_sample(spheres, n, mode, max_bounces, _D3(*org), _D3(*direction), cb, None, out, C.byref(cnt))
    return [out[0], out[1], out[2]], cnt.as_dict()


def intersect(sphere, org, direction, mode, t_init=-1.0, n_init=7.0):
    t = C.c_double(t_init)
    nrm = _D3(n_init, n_init, n_init)
    hit = lib().rtmo_intersect(C.byref(sphere), _D3(*org), _D3(*direction), mode, C.byref(t), nrm)
    return hit, t.value, [nrm[0], nrm[1], nrm[2]]


def normalize(v):
    out = _D3()
    lib().rtmo_normalize(_D3(*v), out)
    return [out[0], out[1], out[2]]


def sin_cos(x):
    """The host libm's sin, cos (what oracle/cpu_ref.c calls) for an array of arguments."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().rtmo_sin_cos_array(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data)
    return s, c


def fnv(arr):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    return int(lib().rtmo_fnv1a64_f64(a.ctypes.data, a.size))


def quantise(arr):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib().rtmo_quantise(a.ctypes.data, a.size, out.ctypes.data)
    return out
