"""The uniform-grid nearest-hit search of large scenes (variant 17, csrc/rtm_path.h: nearest_hit_grid): the hit object
and distance of the reference's loop over ALL objects (src/Renderer.cpp:58-73) from a fraction of its Intersect calls.
Rays: the probe against the reference loop on the GPU (rtm_debug_wf_nearest kind 1) and the oracle; frames: against
the oracle and the exhaustive kernels, bit for bit with the counters."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rtm():
    import raytracingmin_amd
    return raytracingmin_amd


def _ref_nearest(rtm, sph, n, org, d):
    from raytracingmin_amd import _lib
    org = np.ascontiguousarray(org, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    ids = np.zeros(len(org), dtype=np.int32)
    t = np.zeros(len(org), dtype=np.float64)
    _lib.check(rtm.lib().rtm_debug_wf_nearest(1, sph, n, org.ctypes.data, d.ctypes.data, len(org), ids.ctypes.data,
                                              t.ctypes.data), "reference loop")
    return ids, t


def _grid_nearest(rtm, sph, n, org, d):
    from raytracingmin_amd import _lib
    org = np.ascontiguousarray(org, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    ids = np.zeros(len(org), dtype=np.int32)
    t = np.zeros(len(org), dtype=np.float64)
    tests = np.zeros(len(org), dtype=np.uint32)
    steps = np.zeros(len(org), dtype=np.uint32)
    info = np.zeros(6, dtype=np.uint64)
    _lib.check(rtm.lib().rtm_debug_grid_nearest(sph, n, org.ctypes.data, d.ctypes.data, len(org), ids.ctypes.data,
                                                t.ctypes.data, tests.ctypes.data, steps.ctypes.data, info.ctypes.data),
               "grid nearest")
    return ids, t, tests, steps, info


def _normalize_like_the_reference(v):
    """src/Ray.h:67-72: a / (double)sqrtf((float)(a.a))"""
    m = np.sqrt((v * v).sum(axis=1).astype(np.float32)).astype(np.float64)
    return v / m[:, None]


def _spheres(rtm, centers, radii):
    from raytracingmin_amd import _lib
    n = len(radii)
    arr = (_lib.rtm_sphere * n)()
    for i in range(n):
        for k in range(3):
            arr[i].center[k] = float(centers[i, k])
            arr[i].color[k] = 0.5
        arr[i].radius = float(radii[i])
    return arr


def _ray_mix(rng, c, r, count, box):
    """Rays of every kind the render produces, and some it does not: from outside towards the scene (primary-like),
    from sphere surfaces outwards (bounce-like, t1 ~ 0), from inside spheres, silhouette grazes with the discriminant
    swept through zero, axis-parallel directions (zero components), rays that miss the scene's box, rays starting on
    cell-boundary-like round coordinates."""
    n = len(r)
    o = np.empty((count, 3))
    d = np.empty((count, 3))
    for k in range(count):
        kind = k % 8
        i = int(rng.integers(n))
        if kind == 0:    # camera-like
            o[k] = rng.uniform(-1.6, 1.6, 3) * box
            d[k] = c[i] + rng.normal(size=3) * r[i] - o[k]
        elif kind == 1:  # bounce-like: from the surface, outwards
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            o[k] = c[i] + nrm * r[i]
            v = rng.normal(size=3)
            d[k] = v if v @ nrm > 0 else -v
        elif kind == 2:  # from inside a sphere
            o[k] = c[i] + rng.uniform(-0.5, 0.5, 3) * r[i]
            d[k] = rng.normal(size=3)
        elif kind == 3:  # silhouette graze
            o[k] = rng.uniform(-1.2, 1.2, 3) * box
            P = c[i] - o[k]
            L = np.linalg.norm(P)
            e = np.cross(P, rng.normal(size=3))
            e /= np.linalg.norm(e)
            delta = (10.0 ** rng.uniform(-17, 0)) * rng.choice([-1.0, 1.0])
            s = math.sqrt(max(r[i] * r[i] + delta, 0.0)) / max(L, 1e-9)
            d[k] = (math.sqrt(1 - s * s) * P / L + s * e) if s < 1 else rng.normal(size=3)
        elif kind == 4:  # axis-parallel
            o[k] = rng.uniform(-1.0, 1.0, 3) * box
            d[k] = 0.0
            d[k, int(rng.integers(3))] = rng.choice([-1.0, 1.0])
        elif kind == 5:  # one zero component, round origin coordinates
            o[k] = np.round(rng.uniform(-1.0, 1.0, 3) * box)
            d[k] = rng.normal(size=3)
            d[k, int(rng.integers(3))] = 0.0
        elif kind == 6:  # pointing away from / past the scene
            o[k] = rng.uniform(1.05, 1.5, 3) * box * rng.choice([-1.0, 1.0], 3)
            d[k] = rng.normal(size=3)
        else:            # anywhere, any direction
            o[k] = rng.uniform(-1.0, 1.0, 3) * box
            d[k] = rng.normal(size=3)
    return o, _normalize_like_the_reference(d)


SPECIAL = [([0, 0, 0], [np.nan, 0, 1]), ([np.inf, 0, 0], [0, 0, 1]), ([0, 0, 0], [np.inf, 0, 0]), ([1e200, 0, 0], [1, 0, 0]),
           ([1e200, 0, 0], [-1, 0, 0]), ([0, 0, 0], [0, 0, 0]), ([0, 0, 0], [1e-200, 0, 0]), ([0, np.nan, 0], [0, 0, 1]),
           ([0, 0, -1e6], [0, 0, 1]), ([3, 4, -5000], [0, 0, 1])]


def _check(rtm, arr, n, o, d, label, max_mean_tests=None):
    ref_id, ref_t = _ref_nearest(rtm, arr, n, o, d)
    ids, t, tests, steps, info = _grid_nearest(rtm, arr, n, o, d)
    bad = np.flatnonzero((ids != ref_id) | (t.view(np.uint64) != ref_t.view(np.uint64)))
    print(f"{label}: {len(o)} rays, {int((ref_id >= 0).sum())} hit; grid {info[3]}x{info[4]}x{info[5]} = {info[0]} cells, "
          f"{info[1]} list entries, {info[2]} tested by every ray; per ray {tests.mean():.1f} sphere tests (max {tests.max()}), "
          f"{steps.mean():.1f} cell steps")
    assert bad.size == 0, (label, bad[:5], ids[bad[:5]], ref_id[bad[:5]], t[bad[:5]], ref_t[bad[:5]], o[bad[:5]], d[bad[:5]])
    if max_mean_tests is not None:
        assert tests.mean() <= max_mean_tests
    return ref_id, ref_t, tests


def test_grid_nearest_is_the_reference_loops_nearest(rtm, oracle):
    """The stress-scene family at three sizes: every kind of ray, hit object and distance bit for bit, and the work per
    ray is tens of sphere tests, not n."""
    rng = np.random.default_rng(171)
    for n, count in ((400, 60_000), (20_000, 120_000), (100_000, 60_000)):
        data = rtm.make_stress_scene(n, seed=1000 + n)
        _, arr, _ = data.to_c()
        c = np.array([[arr[i].center[k] for k in range(3)] for i in range(n)])
        r = np.array([arr[i].radius for i in range(n)], dtype=np.float64)
        o, d = _ray_mix(rng, c, r, count, 50.0)
        # directions that are not unit length and the non-finite rays: the exhaustive loop / no hit
        d[::97] *= rng.uniform(0.5, 3.0, (len(d[::97]), 1))
        o = np.concatenate([o, np.array([s[0] for s in SPECIAL], dtype=np.float64)])
        d = np.concatenate([d, np.array([s[1] for s in SPECIAL], dtype=np.float64)])
        ref_id, ref_t, tests = _check(rtm, arr, n, o, d, f"stress n={n}")
        hit = ref_id >= 0
        assert 0.15 < hit.mean() < 0.995
        unit = np.ones(len(o), dtype=bool)
        unit[:count:97] = False
        unit[count:] = False
        if n >= 20_000:
            assert tests[unit].mean() < 150  # (the exhaustive rays take n each)
        # the oracle's Intersect on a sample: the reported sphere gives the reported distance
        oarr = (oracle.Sphere * n).from_buffer_copy(bytes(arr))
        for k in rng.choice(np.flatnonzero(hit & unit), 200, replace=False):
            h, tt, _ = oracle.intersect(oarr[int(ref_id[k])], o[k], d[k], oracle.MODE_REPAIRED)
            assert h and tt == ref_t[k]


def test_grid_nearest_awkward_scenes(rtm, oracle):
    """Scenes a uniform grid does not like: spheres that span the scene and of radius 0 (tested by every ray), exact
    duplicates and concentric shells (ties: the lowest index must win, src/Renderer.cpp:67), everything in one plane
    (a grid one cell thick), two far-apart clusters, radii over three orders of magnitude, centres on round numbers."""
    rng = np.random.default_rng(172)
    scenes = {}
    # (a) small spheres + walls of radius 1e4 around them + zero-radius spheres + duplicates
    n = 3000
    c = rng.uniform(-20, 20, (n, 3))
    r = rng.uniform(0.1, 1.5, n)
    c[:6] = [[0, 0, 1e4 + 25], [0, 0, -1e4 - 25], [1e4 + 25, 0, 0], [-1e4 - 25, 0, 0], [0, 1e4 + 25, 0], [0, -1e4 - 25, 0]]
    r[:6] = 1e4
    r[6:30] = 0.0
    c[1000:1100] = c[900:1000]   # exact duplicates at higher indices
    r[1000:1100] = r[900:1000]
    c[1100:1150] = c[850:900]    # concentric: same centre, other radius
    scenes["walls+zero+duplicates"] = (c, r, 25.0)
    # (b) flat: all centres in the plane z = 0
    n = 2000
    c = rng.uniform(-30, 30, (n, 3))
    c[:, 2] = 0.0
    scenes["flat"] = (c, rng.uniform(0.2, 0.9, n), 30.0)
    # (c) two clusters far apart + radii from 0.01 to 10
    n = 4000
    c = np.concatenate([rng.normal(size=(n // 2, 3)) * 5 + [-200, 0, 0], rng.normal(size=(n // 2, 3)) * 8 + [300, 50, -40]])
    scenes["clusters"] = (c, 10.0 ** rng.uniform(-2, 1, n), 350.0)
    # (d) lattice: centres on integers, radius 0.25 (rays along cell boundaries)
    g = np.arange(-6, 7)
    c = np.array([[x, y, z] for x in g for y in g for z in g], dtype=np.float64)
    scenes["lattice"] = (c, np.full(len(c), 0.25), 8.0)
    for name, (c, r, box) in scenes.items():
        n = len(r)
        arr = _spheres(rtm, c, r)
        o, d = _ray_mix(rng, c, np.maximum(r, 1e-3), 40_000, box)
        if name == "lattice":  # rays exactly along the lattice planes and through rows of centres
            o[:4000] = np.round(o[:4000] * 2) / 2
            d[:4000] = 0.0
            d[np.arange(4000), rng.integers(0, 3, 4000)] = rng.choice([-1.0, 1.0], 4000)
        o = np.concatenate([o, np.array([s[0] for s in SPECIAL], dtype=np.float64)])
        d = np.concatenate([d, np.array([s[1] for s in SPECIAL], dtype=np.float64)])
        ref_id, ref_t, _ = _check(rtm, arr, n, o, d, name)
        assert (ref_id >= 0).mean() > 0.2
        if name == "walls+zero+duplicates":  # ties really occur, and go to the lower index
            assert not np.isin(ref_id, np.arange(1000, 1100)).any() and np.isin(ref_id, np.arange(900, 1000)).any()


def _oracle_view(oracle, data):
    st, arr, n = data.to_c()
    return oracle.Settings.from_buffer_copy(bytes(st)), (oracle.Sphere * max(n, 1)).from_buffer_copy(bytes(arr)), n


def _image(rtm, data, mode, mb, seed, variant, rows=None, want=("f64",)):
    r = rtm.Renderer(data, mode=mode, max_bounces=mb, seed=seed, variant=variant)
    rb, re = rows if rows else (0, data.height)
    return r.render_rows(rb, re, want=want)


@pytest.mark.parametrize("n,w,h,s", [(64, 80, 48, 8), (300, 64, 40, 4), (3000, 48, 32, 4), (100_000, 32, 16, 2)])
@pytest.mark.parametrize("mb", [8, -1])
def test_grid_frames_vs_oracle(rtm, oracle, n, w, h, s, mb):
    """Whole frames of the stress-scene family through the grid kernel: the oracle's image and counters, in both modes,
    and what variant 0 picks for these sizes."""
    data = rtm.make_stress_scene(n=n, seed=12345)
    data.width, data.height, data.samples, data.superSamples = w, h, s, 1
    ost, oarr, _ = _oracle_view(oracle, data)
    for mode, omode in (("repaired", 1), ("literal", 0)):
        ref, cnt = oracle.render(ost, oarr, n, oracle.make_options(mode=omode, max_bounces=mb, seed=5, height=h))
        for variant in (17, 0):
            out, st = _image(rtm, data, mode, mb, 5, variant, want=("f64", "u8"))
            assert st["variant"] == 17  # by name, and what variant 0 picks for a scene that has a grid
            assert np.array_equal(out["f64"].view(np.uint64), ref.view(np.uint64)), (mode, variant)
            assert np.array_equal(out["u8"], oracle.quantise(ref))
            assert (st["casts"], st["bounces"], st["draws"]) == (cnt["casts"], cnt["bounces"], cnt["draws"])


def test_grid_frame_row_ranges_and_bands(rtm, oracle):
    """Row ranges and the multi-GPU deal of 8-row bands: the parts are the frame."""
    from raytracingmin_amd.distributed import band_row_index
    data = rtm.make_stress_scene(n=5000, seed=3)
    data.width, data.height, data.samples, data.superSamples = 70, 45, 2, 2
    full, st_full = _image(rtm, data, "repaired", 8, 3, 17)
    wf, st_wf = _image(rtm, data, "repaired", 8, 3, 12)
    assert np.array_equal(full["f64"], wf["f64"]) and st_full["casts"] == st_wf["casts"]
    for lo, hi, world in ((0, 45, 3), (8, 42, 2)):
        got = np.full_like(full["f64"], np.nan)
        for rank in range(world):
            r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=3, variant=17)
            out, _ = r.render_rows(lo, hi, want=("f64",), band=(world, rank))
            got[band_row_index(lo, hi, world, rank)] = out["f64"]
        assert np.array_equal(got[lo:hi], full["f64"][lo:hi])


def test_grid_is_refused_where_there_is_none(rtm, oracle):
    small = rtm.make_stress_scene(n=40, seed=1)
    small.width, small.height, small.samples, small.superSamples = 16, 16, 1, 1
    with pytest.raises(rtm.RtmError, match="grid"):
        _image(rtm, small, "repaired", 8, 1, 17)


def test_config5_strip_grid_vs_pipeline_and_spot_pixels(rtm, oracle):
    """BASELINE configs[4] (100 000 spheres, 1920x1080 @ 256 spp, cap 8): a 16-row strip through the grid kernel and
    through the exhaustive large-scene pipeline — the same bits and counters — and pixels against the oracle."""
    data = rtm.make_stress_scene(n=100_000, seed=12345)
    W, H = 1920, 1080
    data.width, data.height, data.samples, data.superSamples = W, H, 256, 1
    rows = (532, 548)
    g, gs = _image(rtm, data, "repaired", 8, 0x5EED, 17, rows=rows)
    p, ps = _image(rtm, data, "repaired", 8, 0x5EED, 12, rows=rows)
    print(f"configs[4] rows {rows}: grid {gs['kernel_ms']:.1f} ms, exhaustive pipeline {ps['kernel_ms']:.1f} ms "
          f"({ps['kernel_ms'] / gs['kernel_ms']:.0f}x), {gs['casts'] / gs['samples']:.3f} casts/sample")
    assert np.array_equal(g["f64"].view(np.uint64), p["f64"].view(np.uint64))
    assert {k: gs[k] for k in ("samples", "casts", "bounces", "draws")} == {k: ps[k] for k in ("samples", "casts", "bounces", "draws")}
    rng = np.random.default_rng(6)
    xy = np.stack([rng.integers(0, W, 24), rng.integers(rows[0], rows[1], 24)], axis=1).astype(np.int32)
    ost, oarr, n = _oracle_view(oracle, data)
    ref, _ = oracle.render_pixels(ost, oarr, n, oracle.make_options(mode=1, max_bounces=8, seed=0x5EED, height=H), xy)
    assert np.array_equal(g["f64"][xy[:, 1] - rows[0], xy[:, 0]].view(np.uint64), ref.view(np.uint64))


def test_grid_camera_out_of_reach_and_chunked_term_buffer(rtm, oracle):
    """(1) A camera farther from the scene than the grid's pads were sized for: variant 0 leaves the grid alone, variant 17
    by name sends those primary rays through its exhaustive loop — the oracle's frame either way.  (2) A term buffer too
    small for the frame (RTM_DEBUG_GRID_BUDGET_MB, read once per process: a child process): the frame is rendered in
    several launches of a few tiles each — the same frame."""
    import os
    import subprocess
    import sys
    data = rtm.make_stress_scene(n=300, seed=5)
    data.width, data.height, data.samples, data.superSamples = 40, 24, 4, 1
    data.camera.origin = rtm.vec3(0.0, 0.0, -5000.0)
    data.camera.fov = 0.02
    ost, oarr, n = _oracle_view(oracle, data)
    ref, cnt = oracle.render(ost, oarr, n, oracle.make_options(mode=1, max_bounces=8, seed=11, height=24))
    assert ref.any()
    auto, st0 = _image(rtm, data, "repaired", 8, 11, 0)
    named, st17 = _image(rtm, data, "repaired", 8, 11, 17)
    assert st0["variant"] != 17 and st17["variant"] == 17
    assert np.array_equal(auto["f64"].view(np.uint64), ref.view(np.uint64))
    assert np.array_equal(named["f64"].view(np.uint64), ref.view(np.uint64)) and st17["casts"] == cnt["casts"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
import raytracingmin_amd as rtm
data = rtm.make_stress_scene(n=2000, seed=9)
data.width, data.height, data.samples, data.superSamples = 200, 120, 8, 2
r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=3, variant=17)
out, st = r.render_rows(0, 120, want=("f64",))
np.save(sys.argv[1], out["f64"])
print(st["casts"])
"""
    frames, casts = [], []
    for budget in ("1", ""):  # 1 MiB: 8 tiles of 32 spp per launch, 47 launches; default: one
        env = dict(os.environ)
        env.pop("RTM_DEBUG_GRID_BUDGET_MB", None)
        if budget:
            env["RTM_DEBUG_GRID_BUDGET_MB"] = budget
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"grid_chunk_{budget or 'whole'}_{os.getpid()}.npy")
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        frames.append(np.load(path))
        casts.append(int(r.stdout.strip().splitlines()[-1]))
        os.remove(path)
    assert np.array_equal(frames[0].view(np.uint64), frames[1].view(np.uint64)) and casts[0] == casts[1]


def test_grid_through_every_entry_point(rtm, oracle):
    """A 2 000-sphere scene through the C ABI's entry points: a scene object made from a HOST array and one made from a
    DEVICE array (its geometry rows come back once for the host-side grid build) both render through the grid
    (rtm_stats.variant 17); so do the array entry points — a host array through the content-addressed scene cache, a device
    array through the cache keyed by a hash of its content taken on the device (round 3 gave it per-call tables and the
    exhaustive pipeline: 118 x the time for BASELINE configs[4]) — one frame, bit for bit, from all of them; the device
    array's content changed in place is noticed."""
    import ctypes as C
    import torch
    L = rtm.lib()
    data = rtm.make_stress_scene(n=2000, seed=21)
    data.width, data.height, data.samples, data.superSamples = 72, 40, 4, 1
    st = data.settings_c()
    arr, n = data.spheres_c()
    ost, oarr, _ = _oracle_view(oracle, data)
    ref, cnt = oracle.render(ost, oarr, n, oracle.make_options(mode=1, max_bounces=8, seed=3, height=40))
    opt = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=3)._options(0, 40)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d_arr = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()

    def frame(call):
        out = torch.empty((40, 72, 3), dtype=torch.float64, device="cuda")
        stats = rtm._lib.rtm_stats()
        rtm._lib.check(call(C.c_void_p(out.data_ptr()), C.byref(stats)), "render")
        return out.cpu().numpy(), stats.variant, stats.casts

    h_host, h_dev = C.c_void_p(), C.c_void_p()
    rtm._lib.check(L.rtm_scene_create(arr, n, 0, 0, C.byref(h_host)), "rtm_scene_create")
    rtm._lib.check(L.rtm_scene_create(C.c_void_p(d_arr.data_ptr()), n, 1, 0, C.byref(h_dev)), "rtm_scene_create(device)")
    results = [
        frame(lambda o, s: L.rtm_render_scene(C.byref(st), h_host, C.byref(opt), o, None, None, stream, s)),
        frame(lambda o, s: L.rtm_render_scene(C.byref(st), h_dev, C.byref(opt), o, None, None, stream, s)),
        frame(lambda o, s: L.rtm_render_device(C.byref(st), arr, n, 0, C.byref(opt), o, None, None, stream, s)),
        frame(lambda o, s: L.rtm_render_device(C.byref(st), C.c_void_p(d_arr.data_ptr()), n, 1, C.byref(opt), o, None, None, stream, s)),
    ]
    assert [v for _, v, _ in results] == [17, 17, 17, 17]
    for f, _, casts in results:
        assert np.array_equal(f.view(np.uint64), ref.view(np.uint64)) and casts == cnt["casts"]
    # the same device pointer, other content: sphere 5 becomes a big bright ball in front of the camera
    arr[5].center[0], arr[5].center[1], arr[5].center[2], arr[5].radius = 0.0, 0.0, -40.0, 6.0
    arr[5].emission[0] = arr[5].emission[1] = arr[5].emission[2] = 3.0
    d_arr.copy_(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8))
    ost2, oarr2 = ost, (oracle.Sphere * n).from_buffer_copy(bytes(arr))
    ref2, cnt2 = oracle.render(ost2, oarr2, n, oracle.make_options(mode=1, max_bounces=8, seed=3, height=40))
    f2, v2, casts2 = frame(lambda o, s: L.rtm_render_device(C.byref(st), C.c_void_p(d_arr.data_ptr()), n, 1, C.byref(opt), o, None, None, stream, s))
    assert v2 == 17 and casts2 == cnt2["casts"] and not np.array_equal(ref2, ref)
    assert np.array_equal(f2.view(np.uint64), ref2.view(np.uint64))
    assert L.rtm_scene_destroy(h_host) == 0 and L.rtm_scene_destroy(h_dev) == 0


def test_environment_sphere_keeps_variant_0_off_the_grid(rtm, oracle):
    """A diffuse sphere that encloses the gridded spheres from beyond the pads' reach (an environment sphere) sends its
    bounces back from origins the walk cannot serve (csrc/rtm_kernels.hip: build_scene_grid): variant 0 leaves such a
    scene to the exhaustive kernels; variant 17 by name sends those rays through its exhaustive loop — the oracle's frame
    either way.  A black emissive environment sphere never bounces (kd = 0): variant 0 keeps the grid."""
    for colour, expect_grid in (((0.6, 0.6, 0.7), False), ((0.0, 0.0, 0.0), True)):
        data = rtm.make_stress_scene(n=2000, seed=17)
        data.object.append(rtm.SphereObject(rtm.vec3(0, 0, 0), 3000.0, rtm.Material(rtm.vec3(*colour), rtm.vec3(0.8, 0.8, 0.8))))
        data.width, data.height, data.samples, data.superSamples = 24, 16, 2, 1
        ost, oarr, n = _oracle_view(oracle, data)
        ref, cnt = oracle.render(ost, oarr, n, oracle.make_options(mode=1, max_bounces=8, seed=13, height=16))
        auto, st0 = _image(rtm, data, "repaired", 8, 13, 0)
        named, st17 = _image(rtm, data, "repaired", 8, 13, 17)
        assert (st0["variant"] == 17) == expect_grid and st17["variant"] == 17
        assert np.array_equal(auto["f64"].view(np.uint64), ref.view(np.uint64)) and st0["casts"] == cnt["casts"]
        assert np.array_equal(named["f64"].view(np.uint64), ref.view(np.uint64)) and st17["casts"] == cnt["casts"]


def test_grid_deep_paths_take_pooled_records_of_the_kernels_own_type(rtm, oracle):
    """Unlimited depth through the grid kernel with MORE lanes going deeper than its 32 on-chip levels than a byte-sized
    pool would hold: the grid kernel is instantiated for 4-byte records whatever the scene's size, and the pooled stack
    must be sized for those (round 3 sized it from n <= 256: slot 16 384's first push landed on the slot counter).  A
    closed box of six wall spheres with kd 0.98 around 100 bright-coloured small spheres: the mean path is ~50 casts
    long, 24 576 lanes, every one of them deeper than 32 levels at some point."""
    rng = np.random.default_rng(77)
    data = rtm.SettingData()
    data.width, data.height, data.samples, data.superSamples = 192, 128, 3, 1
    data.camera = rtm.Camera(rtm.vec3(0, 0, -9.0), rtm.vec3(0, 0, 0), rtm.vec3(0, 1, 0), 1.2)
    wall = rtm.vec3(0.98, 0.98, 0.98)
    R = 1.0e4
    for axis in range(3):
        for sign in (-1.0, 1.0):
            c = [0.0, 0.0, 0.0]
            c[axis] = sign * (R + 10.0)
            data.object.append(rtm.SphereObject(rtm.vec3(*c), R, rtm.Material(wall, rtm.vec3(0, 0, 0))))
    for i in range(100):
        c = rng.uniform(-8.0, 8.0, 3)
        col = rng.uniform(0.9, 0.98, 3)
        em = (4.0, 4.0, 4.0) if i % 25 == 0 else (0.0, 0.0, 0.0)
        data.object.append(rtm.SphereObject(rtm.vec3(*c), float(rng.uniform(0.4, 1.0)), rtm.Material(rtm.vec3(*col), rtm.vec3(*em))))
    ost, oarr, n = _oracle_view(oracle, data)
    ref, cnt = oracle.render(ost, oarr, n, oracle.make_options(mode=1, max_bounces=-1, seed=21, height=data.height))
    assert cnt["max_depth"] > 200 and cnt["casts"] / cnt["samples"] > 30
    for variant in (17, 0):
        out, st = _image(rtm, data, "repaired", -1, 21, variant)
        assert st["variant"] == 17
        assert np.array_equal(out["f64"].view(np.uint64), ref.view(np.uint64)), variant
        assert (st["casts"], st["bounces"], st["draws"]) == (cnt["casts"], cnt["bounces"], cnt["draws"])


def _plane_scene(rtm, n_spheres, seed):
    """A room of six png::PlaneObject walls (kd 0.75, the ceiling emissive) around `n_spheres` random spheres."""
    rng = np.random.default_rng(seed)
    data = rtm.SettingData()
    data.camera = rtm.Camera(rtm.vec3(0, 0, -19.0), rtm.vec3(0, 0, 0), rtm.vec3(0, 1, 0), 1.0)
    grey, light = rtm.vec3(0.75, 0.75, 0.75), rtm.vec3(0.5, 0.5, 0.5)
    half = 20.0
    for axis in range(3):
        for sign in (-1.0, 1.0):
            pos = [0.0, 0.0, 0.0]
            pos[axis] = sign * half
            up = rtm.vec3(0, 0, 1) if axis == 1 else rtm.vec3(0, 1, 0)
            em = rtm.vec3(2.0, 2.0, 2.0) if (axis == 1 and sign > 0) else rtm.vec3(0, 0, 0)
            data.object.append(rtm.PlaneObject(rtm.vec3(*pos), up, rtm.vec3(0, 0, 0), 2.0 * half, rtm.Material(light if em.x else grey, em)))
    for i in range(n_spheres):
        c = rng.uniform(-16.0, 16.0, 3)
        col = rng.uniform(0.2, 0.9, 3)
        em = (5.0, 5.0, 5.0) if i % 40 == 0 else (0.0, 0.0, 0.0)
        data.object.insert(int(rng.integers(0, len(data.object) + 1)),  # planes anywhere in the vector: the index decides ties
                           rtm.SphereObject(rtm.vec3(*c), float(rng.uniform(0.3, 1.2)), rtm.Material(rtm.vec3(*col), rtm.vec3(*em))))
    return data


@pytest.mark.parametrize("n_spheres,mb", [(300, 8), (300, -1), (1500, 8)])
def test_grid_serves_scenes_with_planes(rtm, oracle, n_spheres, mb):
    """Scenes of 256 objects and more that hold png::PlaneObject entries (src/SettingData.cpp:235-249, this build's
    completion) had only the per-object loop (variant 1); now they get the grid over their spheres, with the planes
    among the objects every ray tests, next to the spheres that span the scene — the oracle's frame and counters through
    variant 0 and variant 17, and the per-object loop's."""
    data = _plane_scene(rtm, n_spheres, seed=n_spheres + 7)
    data.width, data.height, data.samples, data.superSamples = 72, 48, 3, 1
    arr, n = data.objects_c()
    st = data.settings_c()
    ost = oracle.Settings.from_buffer_copy(bytes(st))
    oarr = (oracle.Object * n).from_buffer_copy(bytes(arr))
    ref, cnt = oracle.render_objects(ost, oarr, n, oracle.make_options(mode=1, max_bounces=mb, seed=9, height=data.height))
    assert ref.any()
    for variant in (0, 17, 1):
        out, stt = rtm.Renderer(data, mode="repaired", max_bounces=mb, seed=9, variant=variant).render_rows_device(want=("f64", "u8"))
        assert stt["variant"] == (1 if variant == 1 else 17), (variant, stt["variant"])
        assert np.array_equal(out["f64"].cpu().numpy().view(np.uint64), ref.view(np.uint64)), variant
        assert np.array_equal(out["u8"].cpu().numpy(), oracle.quantise(ref))
        assert (stt["casts"], stt["bounces"], stt["draws"]) == (cnt["casts"], cnt["bounces"], cnt["draws"])
    lit, _ = rtm.Renderer(data, mode="literal", max_bounces=mb, seed=9).render_rows_device(want=("f64",))
    lit_ref, _ = oracle.render_objects(ost, oarr, n, oracle.make_options(mode=0, max_bounces=mb, seed=9, height=data.height))
    assert np.array_equal(lit["f64"].cpu().numpy().view(np.uint64), lit_ref.view(np.uint64))


def test_grid_counts_its_sphere_tests_and_scratch_is_announced(rtm, oracle):
    """RTM_MODE_COUNT_TESTS: the grid kernel's counting instantiation reports the Intersect evaluations it made
    (rtm_stats.object_tests) — the same frame and counters as the plain one, a small fraction of casts x n —, the exhaustive
    kernels report casts x n.  rtm_scratch_bytes announces the work buffers of a render before anything is allocated."""
    data = rtm.make_stress_scene(n=20_000, seed=4)
    data.width, data.height, data.samples, data.superSamples = 160, 96, 8, 1
    plain, ps = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=2).render_rows_device(want=("f64",))
    counted, cs = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=2, count_tests=True).render_rows_device(want=("f64",))
    assert ps["variant"] == cs["variant"] == 17 and ps["object_tests"] == 0
    assert np.array_equal(plain["f64"].cpu().numpy().view(np.uint64), counted["f64"].cpu().numpy().view(np.uint64))
    assert cs["casts"] == ps["casts"]
    per_cast = cs["object_tests"] / cs["casts"]
    print(f"20 000 spheres: {per_cast:.1f} sphere tests per cast through the grid (the reference's loop: 20 000)")
    assert 1.0 < per_cast < 200.0
    strip, xs = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=2, variant=12).render_rows_device(40, 48, want=("f64",))
    assert xs["object_tests"] == xs["casts"] * 20_000
    r = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=2)
    sb = r.scratch_bytes()
    tiles = (160 // 8) * (96 // 8)
    assert sb["terms"] == tiles * (8 * 64 * 32 + 8 * 8) and sb["records"] == 0 and sb["total"] == sb["terms"]  # slots + "term stored" bits
    assert sb["primary_table"] == 0  # (the deferred-fold kernels' table: not the grid kernel's)
    deep = rtm.Renderer(data, mode="repaired", max_bounces=-1, seed=2).scratch_bytes()
    assert deep["records"] == 65536 * 960 * 4 + 64  # the grid kernel's records are 4 bytes wide whatever n
    x = rtm.Renderer(data, mode="repaired", max_bounces=8, seed=2, variant=12).scratch_bytes()
    assert x["pipeline_state"] > 160 * 96 * 150 and x["terms"] == 0
    cornell = rtm.LoadData(oracle.scene_path("cornellBoxSetting.json")).data
    cornell.width, cornell.height, cornell.samples, cornell.superSamples = 1920, 1080, 64, 4
    head = rtm.Renderer(cornell, mode="repaired", max_bounces=8, seed=2).scratch_bytes()
    print("headline frame:", {k: f"{v / 2**30:.2f} GiB" for k, v in head.items()})
    assert head["terms"] > 2**30 and head["steal_rows"] > 2**31 and head["records"] == 0
    assert head["primary_table"] == 32400 * 16 * 3 * 64 * 8  # a direction per (tile, sub-pixel, component, pixel)
    assert head["total"] == head["terms"] + head["steal_rows"] + head["primary_table"]
    tol = rtm.Renderer(cornell, mode="repaired", max_bounces=8, seed=2, variant=18).scratch_bytes()
    assert tol["primary_table"] == head["primary_table"] + 32400 * 64 * 8  # + a mask word per pixel
