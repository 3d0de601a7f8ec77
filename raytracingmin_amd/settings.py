"""Host-side mirror of the reference's scene data model and loader.

Names follow the reference (src/SettingData.h:8-61): vec3, Material, SphereObject, Camera,
SettingData, LoadData.  Parsing is done by the C++ loader behind the C ABI
(rtm_scene_load_json); these classes only hold the result and flatten it for the kernels.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List

from . import _lib
from ._lib import rtm_object, rtm_settings, rtm_sphere


_EDITS = [0]  # bumped by every write that can change a scene's OBJECTS (construction included)


def edit_epoch():
    """Changes whenever any vec3 / Material / SphereObject / PlaneObject was written to or constructed, or a SettingData's
    object list was assigned or mutated: it tells a Renderer that its uploaded copy may be stale (renderer.py) in O(1)
    per render.  The reference reads the live SettingData at render time (src/Renderer.h:16); so does this mirror.  The
    CAMERA is not part of the uploaded scene (it travels with every call in rtm_settings): writes to a Camera's vectors do
    not count, so moving the camera over a 100 000-sphere scene does not re-flatten and re-upload it."""
    return _EDITS[0]


class _Tracked:
    __slots__ = ()

    def __post_init__(self):  # (dataclass construction is over: from here on a write is an edit of a live object)
        object.__setattr__(self, "_live", True)

    def __setattr__(self, name, value):
        # a write to a live scene object counts; the writes of an object's own construction do not (a new object changes
        # a scene only when it is attached to one: that write, or the list mutation, counts)
        if getattr(self, "_live", False) and not getattr(self, "_quiet", False):
            _EDITS[0] += 1
        object.__setattr__(self, name, value)


class _TrackedList(list):
    """SettingData.object: a list whose mutations count as scene edits (replacing, reordering, adding or removing an
    entry changes the scene without writing to any object)."""
    __slots__ = ()

    def _bump(name):  # noqa: N805 — builds the wrappers below
        def wrapper(self, *a, **kw):
            _EDITS[0] += 1
            return getattr(list, name)(self, *a, **kw)
        wrapper.__name__ = name
        return wrapper
    for _n in ("append", "extend", "insert", "remove", "pop", "clear", "sort", "reverse", "__setitem__", "__delitem__",
               "__iadd__", "__imul__"):
        locals()[_n] = _bump(_n)
    del _n, _bump


@dataclass
class vec3(_Tracked):  # src/Ray.h:7-12
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0

    def __iter__(self):
        return iter((self.x, self.y, self.z))


@dataclass
class Material(_Tracked):  # src/SettingData.h:8-17
    color: vec3 = field(default_factory=vec3)
    emission: vec3 = field(default_factory=vec3)


@dataclass
class SphereObject(_Tracked):  # src/SettingData.h:25-32; m_size is a float in the reference
    m_position: vec3 = field(default_factory=vec3)
    m_size: float = 1.0
    m_material: Material = field(default_factory=Material)


@dataclass
class PlaneObject(_Tracked):
    """png::PlaneObject(position, up, target, width, mat) — src/SettingData.h:33-42, src/SettingData.cpp:235-242.
    The reference leaves its Intersect unfinished; this build completes it as the finite square the constructor
    describes (include/rtm.h: rtm_object) — a build-defined semantics."""
    m_position: vec3 = field(default_factory=vec3)
    m_up: vec3 = field(default_factory=lambda: vec3(0, 1, 0))
    target: vec3 = field(default_factory=lambda: vec3(0, 0, 1))
    width: float = 1.0
    m_material: Material = field(default_factory=Material)


@dataclass
class Camera:  # src/SettingData.h:43-46
    origin: vec3 = field(default_factory=vec3)
    target: vec3 = field(default_factory=lambda: vec3(0, 0, 1))
    upVec: vec3 = field(default_factory=lambda: vec3(0, 1, 0))
    fov: float = 60.0

    def __setattr__(self, name, value):
        if isinstance(value, vec3):  # a camera's vectors are not scene objects: writes to them are not scene edits
            object.__setattr__(value, "_quiet", True)
        object.__setattr__(self, name, value)


@dataclass
class SettingData:  # src/SettingData.h:47-51
    width: int = 960
    height: int = 540
    samples: int = 10
    superSamples: int = 1
    camera: Camera = field(default_factory=Camera)
    object: List[SphereObject] = field(default_factory=list)

    def __setattr__(self, name, value):
        if name == "object":  # (a plain list handed in is copied into the tracking kind: mutate data.object, not the original)
            _EDITS[0] += 1
            value = value if isinstance(value, _TrackedList) else _TrackedList(value)
        object.__setattr__(self, name, value)

    # ---- flattening to / from the C ABI structs
    def to_c(self):
        st = self.settings_c()
        arr, n = self.spheres_c()
        return st, arr, n

    def settings_c(self):
        st = rtm_settings()
        st.width, st.height = int(self.width), int(self.height)
        st.samples, st.super_samples = int(self.samples), int(self.superSamples)
        for k, v in enumerate(self.camera.origin):
            st.camera.origin[k] = float(v)
        for k, v in enumerate(self.camera.target):
            st.camera.target[k] = float(v)
        for k, v in enumerate(self.camera.upVec):
            st.camera.up[k] = float(v)
        st.camera.fov = float(self.camera.fov)
        return st

    def has_planes(self):
        return any(isinstance(o, PlaneObject) for o in self.object)

    def objects_c(self):
        """The object vector as rtm_object[] (spheres and planes in the reference's vector order)."""
        n = len(self.object)
        arr = (rtm_object * max(n, 1))()
        for i, o in enumerate(self.object):
            plane = isinstance(o, PlaneObject)
            arr[i].type = _lib.OBJECT_PLANE if plane else _lib.OBJECT_SPHERE
            for k, v in enumerate(o.m_position):
                arr[i].position[k] = float(v)
            for k, v in enumerate(o.m_material.color):
                arr[i].color[k] = float(v)
            for k, v in enumerate(o.m_material.emission):
                arr[i].emission[k] = float(v)
            if plane:
                for k, v in enumerate(o.m_up):
                    arr[i].up[k] = float(v)
                for k, v in enumerate(o.target):
                    arr[i].target[k] = float(v)
                arr[i].width = float(o.width)
            else:
                arr[i].size = float(o.m_size)
        return arr, n

    def spheres_c(self):
        if self.has_planes():
            raise ValueError("the scene holds planes: use objects_c()")
        n = len(self.object)
        arr = (rtm_sphere * max(n, 1))()
        for i, o in enumerate(self.object):
            for k, v in enumerate(o.m_position):
                arr[i].center[k] = float(v)
            for k, v in enumerate(o.m_material.color):
                arr[i].color[k] = float(v)
            for k, v in enumerate(o.m_material.emission):
                arr[i].emission[k] = float(v)
            arr[i].radius = float(o.m_size)
        return arr, n

    @staticmethod
    def from_c(st, arr, n):
        cam = Camera(vec3(*st.camera.origin), vec3(*st.camera.target), vec3(*st.camera.up),
                     float(st.camera.fov))
        objs = [SphereObject(vec3(*arr[i].center), float(arr[i].radius),
                             Material(vec3(*arr[i].color), vec3(*arr[i].emission)))
                for i in range(n)]
        return SettingData(int(st.width), int(st.height), int(st.samples), int(st.super_samples),
                           cam, objs)

    @staticmethod
    def from_c_objects(st, arr, n):
        data = SettingData.from_c(st, (rtm_sphere * 1)(), 0)
        for i in range(n):
            mat = Material(vec3(*arr[i].color), vec3(*arr[i].emission))
            if arr[i].type == _lib.OBJECT_PLANE:
                data.object.append(PlaneObject(vec3(*arr[i].position), vec3(*arr[i].up), vec3(*arr[i].target),
                                               float(arr[i].width), mat))
            else:
                data.object.append(SphereObject(vec3(*arr[i].position), float(arr[i].size), mat))
        return data


class LoadData:
    """png::LoadData (src/SettingData.cpp:6-12): LoadData(path).data is the SettingData.

    literal_loader=True reproduces HEAD's position bug (src/SettingData.cpp:165-167).
    """

    def __init__(self, jsonName, literal_loader=False):
        L = _lib.lib()
        st = rtm_settings()
        n = C.c_size_t(0)
        path = os.fsencode(jsonName)
        _lib.check(L.rtm_scene_load_json_objects(path, int(literal_loader), C.byref(st), None, 0,
                                                 C.byref(n)), f"LoadData({jsonName})")
        arr = (rtm_object * max(n.value, 1))()
        _lib.check(L.rtm_scene_load_json_objects(path, int(literal_loader), C.byref(st), arr, n.value,
                                                 C.byref(n)), f"LoadData({jsonName})")
        self.data = SettingData.from_c_objects(st, arr, n.value)

    @staticmethod
    def SaveSampleJson(fileName):  # src/SettingData.cpp:14-24,100-103
        _lib.check(_lib.lib().rtm_scene_save_sample_json(os.fsencode(fileName)), "SaveSampleJson")


def make_stress_scene(n=100_000, seed=12345):
    """BASELINE config 5 scene."""
    st = rtm_settings()
    arr = (rtm_sphere * max(n, 1))()
    _lib.check(_lib.lib().rtm_scene_make_stress(seed, n, C.byref(st), arr), "make_stress_scene")
    return SettingData.from_c(st, arr, n)
