"""Host-side cost of the scene entry points (round 3: pooled scene tables, deferred destroy): create/destroy of a small
scene, the blocking rtm_render of a tiny frame through the content-addressed cache, a render of a NEW scene every call."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import raytracingmin_amd as rtm
L = rtm.lib()
data = rtm.LoadData(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "scenes", "cornellBoxSetting.json")).data
data.width, data.height, data.samples, data.superSamples = 16, 16, 1, 1
arr, n = data.spheres_c()
r = rtm.Renderer(data, mode="repaired", max_bounces=8)
r.render_rows(want=("f64",))
def timeit(name, fn, reps=100):
    t0 = time.perf_counter()
    for k in range(reps): fn(k)
    print(f"{name:58s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms per call", flush=True)
def create_destroy(k):
    h = C.c_void_p()
    L.rtm_scene_create(arr, n, 0, 0, C.byref(h)); L.rtm_scene_destroy(h)
timeit("rtm_scene_create + rtm_scene_destroy (7 spheres)", create_destroy)
timeit("rtm_render, 16x16x1spp, cached scene", lambda k: r.render_rows(want=("f64",)))
def new_scene(k):
    data.object[0].m_size = 5.0 + 1e-3 * k
    r.render_rows(want=("f64",))
timeit("rtm_render, 16x16x1spp, new scene content every call", new_scene)
big = rtm.make_stress_scene(700, seed=3)
big.width, big.height, big.samples, big.superSamples = 16, 16, 1, 1
rb = rtm.Renderer(big, mode="repaired", max_bounces=8)
rb.render_rows(want=("f64",))
timeit("rtm_render, 700 spheres (wavefront pipeline), 16x16x1spp", lambda k: rb.render_rows(want=("f64",)), 30)
rb2 = rtm.Renderer(big, mode="repaired", max_bounces=40)
timeit("the same with max_bounces 40", lambda k: rb2.render_rows(want=("f64",)), 30)
