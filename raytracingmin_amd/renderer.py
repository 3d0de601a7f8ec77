"""png::Renderer on the HIP path (reference src/Renderer.h:8-18, src/Renderer.cpp:20-23,200-258).

    ld = LoadData("cornellBoxSetting.json")
    r = Renderer(ld.data, mode="repaired", max_bounces=8)
    r.Render("result")          # writes result.jpg (q=60) and result.bmp like the reference
    r.image                     # (H, W, 3) float64 — the reference's private Renderer::image

Device memory and streams come from torch (plumbing); every pixel is computed by the HIP kernels
behind rtm_render_device.  There is no CPU fallback.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from ._lib import rtm_options, rtm_stats
from .settings import SettingData


class Renderer:
    def __init__(self, data: SettingData, mode="repaired", max_bounces=-1, seed=0x5EED, device=0,
                 variant=0, host_trig=True, count_tests=False, integrator="PathTracing"):
        # host_trig (default): sin/cos of src/Renderer.cpp:93-94 as the HOST's libm returns them, so a
        # render agrees with a CPU run of the reference bit for bit even on scenes that amplify one-ulp
        # differences over many bounces; host_trig=False is the labelled ~2 % faster device-trig row
        self.data = data  # the reference keeps a reference to the caller's SettingData
        # count_tests: rtm_stats.object_tests also for the uniform-grid kernel (its counting instantiation, RTM_MODE_COUNT_TESTS)
        # integrator: "PathTracing" (what the reference's Render always takes, src/Renderer.cpp:238) or "SurfaeSample" (the
        # branch of :234-236 it never takes; RTM_MODE_SURFACE_SAMPLE, served by a general per-object kernel)
        if integrator not in ("PathTracing", "SurfaeSample"):
            raise ValueError(f"unknown integrator {integrator!r}")
        self.mode = _lib.MODES[mode] | (_lib.MODE_HOST_TRIG if host_trig else 0) | (_lib.MODE_COUNT_TESTS if count_tests else 0) \
            | (_lib.MODE_SURFACE_SAMPLE if integrator == "SurfaeSample" else 0)
        self.max_bounces = int(max_bounces)
        self.seed = int(seed)
        self.device = int(device)
        self.variant = int(variant)
        self.image = np.zeros((data.height, data.width, 3), dtype=np.float64)  # src/Renderer.cpp:21
        self.stats = None
        self._scene = None      # rtm_scene* on self.device (made on first use, kept for the renderer's life)
        self._scene_key = None

    # ---- the scene on the device: flattened and uploaded once per renderer -------------------
    def _scene_handle(self):
        """rtm_scene_create once; made again whenever the scene may have changed since — the reference reads the live
        SettingData at render time (src/Renderer.h:16), and so must this.  The key is O(1) per render: the edit epoch of
        settings.py counts every write to a scene object's fields AND every assignment or mutation of a SettingData's
        object list (replaced, reordered, added, removed entries); camera edits do not count (the camera is not part of
        the uploaded scene).  Edits to ANOTHER SettingData also advance the epoch: a needless re-flatten, never a stale
        scene."""
        from .settings import edit_epoch
        objs = self.data.object
        key = (edit_epoch(), id(objs))
        if self._scene is None or key != self._scene_key:
            self.invalidate()
            h = C.c_void_p()
            if self.data.has_planes():  # png::PlaneObject entries: the any-type object list
                arr, n = self.data.objects_c()
                _lib.check(_lib.lib().rtm_scene_create_objects(arr, n, self.device, C.byref(h)),
                           "rtm_scene_create_objects")
            else:
                arr, n = self.data.spheres_c()
                _lib.check(_lib.lib().rtm_scene_create(arr, n, 0, self.device, C.byref(h)), "rtm_scene_create")
            self._scene, self._scene_key = h, key
        return self._scene

    def scratch_bytes(self, row_begin=0, row_end=None, band=None):
        """rtm_scratch_bytes: what a render of these rows asks of the library's per-(device, stream) work buffers."""
        row_end = self.data.height if row_end is None else row_end
        opt = self._options(row_begin, row_end, band)
        st = self.data.settings_c()
        out = (C.c_uint64 * 6)()
        _lib.check(_lib.lib().rtm_scratch_bytes(C.byref(st), self._scene_handle(), C.byref(opt), out), "rtm_scratch_bytes")
        return dict(zip(("total", "terms", "records", "pipeline_state", "steal_rows", "primary_table"), (int(v) for v in out)))

    def invalidate(self):
        if self._scene is not None:
            _lib.lib().rtm_scene_destroy(self._scene)
            self._scene = None

    def __del__(self):
        if sys.is_finalizing():  # interpreter shutdown: the HIP runtime may be gone already; the process's memory goes with it
            return
        try:
            self.invalidate()
        except Exception:
            pass

    def stream_status(self, stream=None):
        """rtm_stream_status: raises RtmError if a stats-less render on the stream was truncated."""
        import torch
        dev = torch.device("cuda", self.device)
        hip_stream = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().rtm_stream_status(self.device, C.c_void_p(hip_stream)), "rtm_stream_status")

    def _options(self, row_begin, row_end, band=None):
        o = rtm_options()
        o.mode, o.max_bounces, o.seed = self.mode, self.max_bounces, self.seed
        o.row_begin, o.row_end = int(row_begin), int(row_end)
        o.device, o.variant = self.device, self.variant
        if band is not None:  # (count, index): only the 8-row bands index, index + count, ...
            o.band_count, o.band_index = int(band[0]), int(band[1])
        return o

    # ---- device-resident render: outputs are torch tensors on the GPU ------------------------
    def render_rows_device(self, row_begin=0, row_end=None, want=("f32",), stats=True,
                           stream=None, band=None):
        """Render rows [row_begin, row_end) into torch CUDA tensors; returns (dict, stats).
        band=(count, index) renders only every count-th 8-row band of the range, stored back to back
        (include/rtm.h, rtm_options.band_count)."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("Renderer needs a HIP device; there is no CPU fallback")
        row_end = self.data.height if row_end is None else row_end
        opt = self._options(row_begin, row_end, band)
        rows, W = _lib.lib().rtm_output_rows(C.byref(opt)), self.data.width
        dev = torch.device("cuda", self.device)
        out = {}
        if "f64" in want:
            out["f64"] = torch.empty((rows, W, 3), dtype=torch.float64, device=dev)
        if "f32" in want:
            out["f32"] = torch.empty((rows, W, 3), dtype=torch.float32, device=dev)
        if "u8" in want:
            out["u8"] = torch.empty((rows, W, 3), dtype=torch.uint8, device=dev)
        st = self.data.settings_c()
        s = rtm_stats()
        hip_stream = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        ptr = lambda k: C.c_void_p(out[k].data_ptr()) if k in out and rows > 0 else None
        _lib.check(_lib.lib().rtm_render_scene(C.byref(st), self._scene_handle(), C.byref(opt), ptr("f64"),
                                               ptr("f32"), ptr("u8"), C.c_void_p(hip_stream),
                                               C.byref(s) if stats else None), "rtm_render_scene")
        return out, (s.as_dict() if stats else None)

    # ---- host-buffer render (the blocking C entry point) ------------------------------------
    def render_rows(self, row_begin=0, row_end=None, want=("f64",), band=None):
        row_end = self.data.height if row_end is None else row_end
        opt = self._options(row_begin, row_end, band)
        rows, W = _lib.lib().rtm_output_rows(C.byref(opt)), self.data.width
        out = {}
        if "f64" in want:
            out["f64"] = np.zeros((rows, W, 3), dtype=np.float64)
        if "f32" in want:
            out["f32"] = np.zeros((rows, W, 3), dtype=np.float32)
        if "u8" in want:
            out["u8"] = np.zeros((rows, W, 3), dtype=np.uint8)
        s = rtm_stats()
        ptr = lambda k: out[k].ctypes.data_as(C.c_void_p) if k in out else None
        if self.data.has_planes():  # rtm_render takes sphere arrays; any other object goes through rtm_render_objects
            st = self.data.settings_c()
            arr, n = self.data.objects_c()
            _lib.check(_lib.lib().rtm_render_objects(C.byref(st), arr, n, C.byref(opt), ptr("f64"), ptr("f32"),
                                                     ptr("u8"), C.byref(s)), "rtm_render_objects")
        else:
            st, arr, n = self.data.to_c()
            _lib.check(_lib.lib().rtm_render(C.byref(st), arr, n, C.byref(opt), ptr("f64"), ptr("f32"),
                                             ptr("u8"), C.byref(s)), "rtm_render")
        self.stats = s.as_dict()
        return out, self.stats

    def Render(self, fileName):
        """src/Renderer.cpp:200-258: render, quantise, write <fileName>.jpg and <fileName>.bmp."""
        out, _ = self.render_rows(0, self.data.height, want=("f64", "u8"))
        self.image = out["f64"]
        rgb8 = np.ascontiguousarray(out["u8"])
        L = _lib.lib()
        H, W = self.data.height, self.data.width
        ok_j = L.rtm_write_jpg(os.fsencode(fileName + ".jpg"), W, H, 3, rgb8.ctypes.data, 60)
        ok_b = L.rtm_write_bmp(os.fsencode(fileName + ".bmp"), W, H, 3, rgb8.ctypes.data)
        if not (ok_j and ok_b):
            raise _lib.RtmError(-3, f"could not write {fileName}.jpg/.bmp")
        return rgb8


# ---- seams below the renderer, for parity tests --------------------------------------------------
def path_tracing_batch(data: SettingData, org, direction, mode="repaired", max_bounces=-1,
                       seed=0x5EED, device=0, host_trig=True):
    """png::PathTracing (src/Renderer.cpp:57-117) for n rays; ray i draws from stream (seed, i, 0)."""
    org = np.ascontiguousarray(org, dtype=np.float64).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    n_rays = org.shape[0]
    out = np.zeros((n_rays, 3), dtype=np.float64)
    draws = np.zeros(n_rays, dtype=np.uint32)
    casts = np.zeros(n_rays, dtype=np.uint32)
    _, arr, n = data.to_c()
    o = rtm_options()
    o.mode = _lib.MODES[mode] | (_lib.MODE_HOST_TRIG if host_trig else 0)
    o.max_bounces, o.seed, o.device = int(max_bounces), int(seed), device
    _lib.check(_lib.lib().rtm_path_trace_batch(arr, n, C.byref(o), org.ctypes.data,
                                               direction.ctypes.data, n_rays, out.ctypes.data,
                                               draws.ctypes.data, casts.ctypes.data),
               "rtm_path_trace_batch")
    return out, draws, casts


def surface_sample_batch(data: SettingData, org, direction, mode="repaired", max_bounces=-1, seed=0x5EED, device=0):
    """png::SurfaeSample (src/Renderer.cpp:119-198), entered at depth 0, for n rays; ray i draws from stream (seed, i, 0)."""
    org = np.ascontiguousarray(org, dtype=np.float64).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    n_rays = org.shape[0]
    out = np.zeros((n_rays, 3), dtype=np.float64)
    draws = np.zeros(n_rays, dtype=np.uint32)
    casts = np.zeros(n_rays, dtype=np.uint32)
    _, arr, n = data.to_c()
    o = rtm_options()
    o.mode = _lib.MODES[mode]
    o.max_bounces, o.seed, o.device = int(max_bounces), int(seed), device
    _lib.check(_lib.lib().rtm_surface_sample_batch(arr, n, C.byref(o), org.ctypes.data, direction.ctypes.data, n_rays,
                                                   out.ctypes.data, draws.ctypes.data, casts.ctypes.data),
               "rtm_surface_sample_batch")
    return out, draws, casts


def intersect_objects_batch(objects_c, org, direction, mode="repaired", t_init=-1.0, n_init=7.0):
    """Object::Intersect for rtm_object entries (spheres and planes), pair i = (ray i, object i)."""
    org = np.ascontiguousarray(org, dtype=np.float64).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    n = org.shape[0]
    hit = np.zeros(n, dtype=np.int32)
    t = np.full(n, t_init, dtype=np.float64)
    nrm = np.full((n, 3), n_init, dtype=np.float64)
    _lib.check(_lib.lib().rtm_intersect_objects_batch(objects_c, org.ctypes.data, direction.ctypes.data, n,
                                                      _lib.MODES[mode], hit.ctypes.data, t.ctypes.data,
                                                      nrm.ctypes.data), "rtm_intersect_objects_batch")
    return hit, t, nrm


def intersect_batch(spheres_c, org, direction, mode="repaired", t_init=-1.0, n_init=7.0):
    """SphereObject::Intersect (src/SettingData.cpp:197-226), pair i = (ray i, sphere i)."""
    org = np.ascontiguousarray(org, dtype=np.float64).reshape(-1, 3)
    direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
    n = org.shape[0]
    hit = np.zeros(n, dtype=np.int32)
    t = np.full(n, t_init, dtype=np.float64)
    nrm = np.full((n, 3), n_init, dtype=np.float64)
    _lib.check(_lib.lib().rtm_intersect_batch(spheres_c, org.ctypes.data, direction.ctypes.data, n,
                                              _lib.MODES[mode], hit.ctypes.data, t.ctypes.data,
                                              nrm.ctypes.data), "rtm_intersect_batch")
    return hit, t, nrm
