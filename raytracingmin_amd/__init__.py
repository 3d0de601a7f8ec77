"""raytracingmin_amd — MI355X-native hot path for RaytracingMin scenes.

Host-side mirror of the reference interface (png::LoadData, png::Renderer, SettingData) over the
C ABI of include/rtm.h; all rendering happens in hand-written HIP kernels (csrc/).
"""
from ._lib import MODE_LITERAL, MODE_REPAIRED, RtmError, lib  # noqa: F401
from .renderer import (Renderer, intersect_batch, intersect_objects_batch, path_tracing_batch,  # noqa: F401
                       surface_sample_batch)
from .settings import (Camera, LoadData, Material, PlaneObject, SettingData, SphereObject,  # noqa: F401
                       make_stress_scene, vec3)

__all__ = ["LoadData", "Renderer", "SettingData", "Camera", "SphereObject", "Material", "vec3",
           "make_stress_scene", "path_tracing_batch", "surface_sample_batch", "intersect_batch", "intersect_objects_batch", "PlaneObject", "RtmError", "lib",
           "MODE_LITERAL", "MODE_REPAIRED"]
