"""Host side of the boundary: scene loader (reference src/SettingData.cpp:129-186), quantiser and
image writers (src/Renderer.cpp:251-257).  No GPU needed."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import raytracingmin_amd as rtm
from raytracingmin_amd import _lib

SCENES = ["cornellBoxSetting.json", "simpleSetting1.json", "simpleSetting2.json", "settingData.json"]


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("literal", [False, True])
def test_loader_matches_independent_python_reader(oracle, name, literal):
    path = oracle.scene_path(name)
    st, arr, n = oracle.load_scene(path, literal_loader=literal)
    data = rtm.LoadData(path, literal_loader=literal).data
    cst, carr, cn = data.to_c()
    assert cn == n
    assert (cst.width, cst.height, cst.samples, cst.super_samples) == \
        (st.width, st.height, st.samples, st.super_samples)
    assert bytes(cst.camera)[:76] == bytes(st.camera)[:76]
    for i in range(n):
        assert list(carr[i].center) == list(arr[i].center)
        assert list(carr[i].color) == list(arr[i].color)
        assert list(carr[i].emission) == list(arr[i].emission)
        assert carr[i].radius == arr[i].radius


def test_shipped_files_load_unchanged():
    import _oracle
    d = rtm.LoadData(_oracle.scene_path("cornellBoxSetting.json")).data
    assert (d.width, d.height, d.samples, d.superSamples) == (960, 504, 25, 4)
    assert len(d.object) == 7  # 8 entries minus the "{}"
    assert d.camera.fov == 2.0 and tuple(d.camera.origin) == (0.0, 0.0, -10.0)
    s1 = rtm.LoadData(_oracle.scene_path("simpleSetting1.json")).data
    assert (s1.samples, s1.superSamples, len(s1.object)) == (100, 1, 5)  # "00 sample" alias
    s2 = rtm.LoadData(_oracle.scene_path("simpleSetting2.json")).data
    assert s2.samples == 500


def _parse(text, literal=0):
    L = _lib.lib()
    st = _lib.rtm_settings()
    n = C.c_size_t()
    arr = (_lib.rtm_sphere * 8)()
    b = text.encode()
    rc = L.rtm_scene_parse_json(b, len(b), literal, C.byref(st), arr, 8, C.byref(n))
    return rc, st, arr, n.value


def test_loader_edge_cases():
    base = {"00 width": 4, "00 height": 2, "01 camera": {"fov": 1.5, "origin": [0, 0, 0],
                                                         "target": [0, 0, 1], "upVec": [0, 1, 0]}}
    rc, st, _, n = _parse(json.dumps(base))
    assert rc == 0 and n == 0 and (st.samples, st.super_samples) == (10, 1)  # defaults
    rc, *_ = _parse("{ not json")
    assert rc == -4
    rc, *_ = _parse(json.dumps({"00 height": 2}))
    assert rc == -2  # width required
    obj = {"00 position": [1, 2, 3], "01 size": 0.1, "02 material": {"color": [.1, .2, .3], "emission": [0, 0, 0]}}
    doc = dict(base, **{"02 scene": {"00 object": [obj, {}, dict(obj, **{"00 objectType": 1})]}})
    rc, st, arr, n = _parse(json.dumps(doc))
    assert rc == 0 and n == 2 and list(arr[0].center) == [1, 2, 3]
    assert arr[0].radius == np.float32(0.1)
    rc, st, arr, n = _parse(json.dumps(doc), literal=1)
    assert list(arr[0].center) == [3, 0, 0]  # D1
    doc["02 scene"]["00 object"][0]["00 objectType"] = 3
    rc, *_ = _parse(json.dumps(doc))
    assert rc == -2 and b"objectType" in _lib.lib().rtm_last_error_detail()  # the reference would push a null Object*
    # objectType 2 = png::PlaneObject, a build-defined extension of the schema (include/rtm.h: rtm_object)
    doc["02 scene"]["00 object"][0]["00 objectType"] = 2
    rc, *_ = _parse(json.dumps(doc))
    assert rc == -4 and b"plane" in _lib.lib().rtm_last_error_detail()      # needs "03 up" and "04 target"
    doc["02 scene"]["00 object"][0].update({"03 up": [0, 1, 0], "04 target": [1, 2, 4], "01 size": 2.5})
    rc, *_ = _parse(json.dumps(doc))
    assert rc == -2 and b"rtm_scene_load_json_objects" in _lib.lib().rtm_last_error_detail()  # sphere-only loader
    L = _lib.lib()
    b = json.dumps(doc).encode()
    st2, n2 = _lib.rtm_settings(), C.c_size_t()
    objs = (_lib.rtm_object * 4)()
    assert L.rtm_scene_parse_json_objects(b, len(b), 0, C.byref(st2), objs, 4, C.byref(n2)) == 0 and n2.value == 2
    assert (objs[0].type, objs[0].width, list(objs[0].up), list(objs[0].target)) == (2, 2.5, [0, 1, 0], [1, 2, 4])
    assert (objs[1].type, objs[1].size, list(objs[1].position)) == (1, np.float32(0.1), [1, 2, 3])
    doc["02 scene"]["00 object"][0]["00 objectType"] = 1
    for k in ("03 up", "04 target"):
        del doc["02 scene"]["00 object"][0][k]
    doc["02 scene"]["00 object"][0]["01 size"] = "big"
    rc, *_ = _parse(json.dumps(doc))
    assert rc == -4  # nlohmann would throw type_error
    rc, st, *_ = _parse(json.dumps(dict(base, **{"00 width": 7.9, "unknown": [1, {"a": None}]})))
    assert rc == 0 and st.width == 7  # get<int>() truncates; unknown keys ignored
    # capacity query / too small
    L = _lib.lib()
    st = _lib.rtm_settings()
    n = C.c_size_t()
    doc["02 scene"]["00 object"][0]["01 size"] = 1
    b = json.dumps(doc).encode()
    assert L.rtm_scene_parse_json(b, len(b), 0, C.byref(st), None, 0, C.byref(n)) == 0 and n.value == 2
    one = (_lib.rtm_sphere * 1)()
    assert L.rtm_scene_parse_json(b, len(b), 0, C.byref(st), one, 1, C.byref(n)) == -7
    assert L.rtm_scene_load_json(b"/nonexistent/x.json", 0, C.byref(st), None, 0, C.byref(n)) == -3


def test_save_sample_json_roundtrip(tmp_path):
    p = str(tmp_path / "settingData.json")
    rtm.LoadData.SaveSampleJson(p)
    j = json.load(open(p))
    assert j == {"00 height": 540, "00 samples": 10, "00 superSamples": 4, "00 width": 960,
                 "01 camera": {"fov": 60.0, "origin": [0, 0, 0], "target": [0, 0, 1], "upVec": [0, 1, 0]}}
    d = rtm.LoadData(p).data
    assert (d.width, d.height, d.samples, d.superSamples, len(d.object)) == (960, 540, 10, 4, 0)


def test_stress_scene_matches_oracle_generator(oracle):
    n = 1000
    d = rtm.make_stress_scene(n=n, seed=12345)
    st = oracle.Settings()
    arr = (oracle.Sphere * n)()
    oracle.lib().rtmo_make_stress_scene(12345, n, C.byref(st), arr)
    _, carr, _ = d.to_c()
    assert bytes(carr)[:80 * n] == bytes(arr)
    assert (d.width, d.height, d.samples, d.superSamples) == (1920, 1080, 256, 1)
    assert d.object[0].m_material.emission.x == 5.0 and d.object[1].m_material.emission.x == 0.0


def test_quantise_matches_oracle(oracle):
    rng = np.random.default_rng(1)
    v = np.concatenate([rng.uniform(-0.1, 1.3, 5000), [0.0, 1.0, 0.999999999, 1 / 255, 254.9999 / 255,
                                                       5.0, 1e-300, np.nextafter(1.0, 0)]])
    out = np.zeros(v.size, dtype=np.uint8)
    assert _lib.lib().rtm_quantise(v.ctypes.data, v.size, out.ctypes.data) == 0
    assert np.array_equal(out, oracle.quantise(v))
    assert out[5001] == 255 and out[5002] == 254  # truncation, not rounding


def test_bmp_layout_is_stbs(tmp_path):
    w, h = 5, 3  # row = 15 bytes -> 1 byte padding
    img = np.arange(w * h * 3, dtype=np.uint8).reshape(h, w, 3)
    p = str(tmp_path / "a.bmp")
    assert _lib.lib().rtm_write_bmp(p.encode(), w, h, 3, img.ctypes.data) == 1
    raw = open(p, "rb").read()
    assert raw[:2] == b"BM" and len(raw) == 54 + 16 * h
    size, _, _, off, hdr, bw, bh, planes, bpp = struct.unpack("<IHHIIiiHH", raw[2:30])
    assert (size, off, hdr, bw, bh, planes, bpp) == (len(raw), 54, 40, w, h, 1, 24)
    assert raw[30:54] == bytes(24)
    first = raw[54:54 + 16]  # bottom row first, BGR, padded
    assert first[:3] == bytes(img[h - 1, 0, ::-1]) and first[15] == 0
    from PIL import Image
    assert np.array_equal(np.array(Image.open(p)), img)


def test_jpeg_is_decodable_baseline(tmp_path):
    from PIL import Image
    yy, xx = np.mgrid[0:50, 0:70]
    img = np.ascontiguousarray(np.stack([xx * 3, yy * 5, xx + yy], -1).astype(np.uint8))
    for q in (60, 95):
        p = str(tmp_path / f"a{q}.jpg")
        assert _lib.lib().rtm_write_jpg(p.encode(), 70, 50, 3, img.ctypes.data, q) == 1
        im = Image.open(p)
        assert im.format == "JPEG" and im.size == (70, 50)
        err = np.abs(np.array(im.convert("RGB")).astype(int) - img)
        assert err.mean() < 4 and err.max() < 40
    assert _lib.lib().rtm_write_jpg(b"/nonexistent/dir/a.jpg", 70, 50, 3, img.ctypes.data, 60) == 0


def test_plane_objects_in_the_oracle(oracle):
    """png::PlaneObject as this build completes it (include/rtm.h): the oracle's Intersect for a unit square
    facing the ray — inside / outside the extent, grazing, behind, and the literal mode's lost normal."""
    o = oracle.Object()
    o.type = 2
    for k, v in enumerate((0.0, 0.0, 5.0)):
        o.position[k] = v
    for k, v in enumerate((0.0, 1.0, 0.0)):
        o.up[k] = v
    for k, v in enumerate((0.0, 0.0, 4.0)):   # normal (0, 0, -1): towards the camera
        o.target[k] = v
    o.width = 2.0
    hit, t, n = oracle.intersect_object(o, (0.2, -0.3, 0), (0, 0, 1), 1)
    assert hit == 1 and t == 5.0 and n == [0.0, 0.0, -1.0]
    hit, t, n = oracle.intersect_object(o, (0.999, 0.999, 0), (0, 0, 1), 1)
    assert hit == 1
    assert oracle.intersect_object(o, (1.001, 0, 0), (0, 0, 1), 1)[0] == 0       # outside the square
    assert oracle.intersect_object(o, (0, 0, 0), (0, 0, -1), 1)[0] == 0          # behind the ray
    assert oracle.intersect_object(o, (0, 0, 0), (1, 0, 0), 1)[0] == 0           # parallel: the reference's own line
    assert oracle.intersect_object(o, (0, 0, 4.9995), (0, 0, 1), 1)[0] == 0      # t = 0.0005 <= 0.001
    hit, t, n = oracle.intersect_object(o, (0, 0, 0), (0, 0, 1), 0)              # literal: normal never delivered (D2)
    assert hit == 1 and t == 5.0 and n == [7.0, 7.0, 7.0]
    s = oracle.Object()
    s.type, s.size = 1, 2.0
    for k, v in enumerate((0.0, 0.0, 6.0)):
        s.position[k] = v
    sp = oracle.Sphere()
    sp.radius = 2.0
    for k, v in enumerate((0.0, 0.0, 6.0)):
        sp.center[k] = v
    assert oracle.intersect_object(s, (0.1, 0.2, 0), (0, 0, 1), 1) == oracle.intersect(sp, (0.1, 0.2, 0), (0, 0, 1), 1)


def test_edit_epoch_sees_every_write_on_a_scene_object():
    """settings.edit_epoch(): what lets a Renderer notice in-place edits of the caller's SettingData (the reference
    reads the live scene at render time); the tracking object list covers replacement and reordering."""
    import _oracle
    from raytracingmin_amd.settings import edit_epoch
    data = rtm.LoadData(_oracle.scene_path("cornellBoxSetting.json")).data
    e0 = edit_epoch()
    data.object[0].m_position.x = 1.5
    e1 = edit_epoch()
    data.object[0].m_material.emission.z = 2.0
    e2 = edit_epoch()
    data.object[0].m_size = 4.0
    e3 = edit_epoch()
    assert e0 < e1 < e2 < e3
    ids = tuple(map(id, data.object))
    data.object[1], data.object[2] = data.object[2], data.object[1]
    e4 = edit_epoch()
    assert tuple(map(id, data.object)) != ids and e4 > e3  # reordering writes to no object: the tracking list counts it
    rtm.PlaneObject()
    assert edit_epoch() == e4  # constructing an object that is attached to nothing changes no scene (round 4)


def _grid_build(arr, n):
    info = (C.c_uint64 * 12)()
    pads = np.zeros(n, dtype=np.float64)
    rc = _lib.lib().rtm_debug_grid_build(arr, n, info, pads.ctypes.data, None, 0, None, 0, None, 0)
    if rc != 0:
        return rc, None
    cells, recs, nbig = int(info[0]), int(info[1]), int(info[2])
    ranges = np.zeros(cells * 2, dtype=np.uint32)
    items = np.zeros(max(recs, 1), dtype=np.uint32)
    big = np.zeros(max(nbig, 1), dtype=np.int32)
    _lib.check(_lib.lib().rtm_debug_grid_build(arr, n, info, pads.ctypes.data, ranges.ctypes.data, ranges.size, items.ctypes.data,
                                               items.size, big.ctypes.data, big.size), "grid build")
    d = struct.unpack("<6d", bytes(info)[48:96])
    return 0, dict(cells=cells, recs=recs, dim=[int(info[3 + k]) for k in range(3)], lo=np.array(d[:3]), h=d[3], reach=d[4], t_ok=d[5],
                   pads=pads, ranges=ranges.reshape(cells, 2), items=items[:recs], big=big[:nbig])


def test_grid_builder_lists_cover_every_padded_sphere():
    """The uniform grid of large scenes (csrc/rtm_kernels.hip: make_grid; the exactness argument is in csrc/rtm_path.h): host logic,
    no device.  Every point within r + pad of a gridded sphere's centre lies in a cell that lists the sphere — so a hit point,
    which lies within the pad of the sphere's surface, is always found —, a cell's list ascends (ties go to the lower index by
    the walk's rule, but the order keeps the lists deterministic), the ranges tile the item array, the spheres that span the
    scene are in the list every ray tests and nowhere else, and the pads are what the derivation asks for."""
    rng = np.random.default_rng(12)
    n = 3000
    c = rng.uniform(-20, 20, (n, 3))
    r = rng.uniform(0.05, 1.5, n)
    c[:4] = [[0, 0, 1e4 + 25], [0, -1e4 - 25, 0], [1e4 + 25, 0, 0], [0, 0, 0]]
    r[:3] = 1e4
    r[3] = 0.0
    arr = (_lib.rtm_sphere * n)()
    for i in range(n):
        for k in range(3):
            arr[i].center[k] = float(c[i, k])
        arr[i].radius = float(r[i])
    rc, g = _grid_build(arr, n)
    assert rc == 0
    nx, ny, nz = g["dim"]
    assert g["cells"] == nx * ny * nz and list(g["big"]) == [0, 1, 2]
    rg = g["ranges"]
    assert rg[0, 0] == 0 and rg[-1, 1] == g["recs"] and np.array_equal(rg[1:, 0], rg[:-1, 1])  # the ranges tile the items
    R = np.sqrt((r.astype(np.float32) * r.astype(np.float32)).astype(np.float64))  # the radius the tests use: sqrt(float r*r)
    # pads: 0.05 h + (sqrt(r^2 + 4e-7 t_ok^2) - r) + 1e-6 t_ok
    want = 0.05 * g["h"] + (np.sqrt(R * R + 4e-7 * g["t_ok"] ** 2) - R) + 1e-6 * g["t_ok"]
    assert np.allclose(g["pads"][3:], want[3:], rtol=1e-12, atol=0)
    lists = [g["items"][a:b] for a, b in rg]
    for li in lists:
        assert np.all(np.diff(li.astype(np.int64)) > 0)  # ascending, no duplicates
    assert not np.isin(g["items"], [0, 1, 2]).any()
    # coverage: random points within r + pad of the centre (on the inflated surface, and inside) are in a listing cell
    for i in rng.choice(np.arange(3, n), 400, replace=False):
        v = rng.normal(size=(40, 3))
        v /= np.linalg.norm(v, axis=1)[:, None]
        rad = (R[i] + g["pads"][i]) * np.concatenate([np.full(20, 1.0 - 1e-9), rng.uniform(0, 1, 20)])
        p = c[i] + v * rad[:, None]
        cell = np.floor((p - g["lo"]) / g["h"]).astype(int)
        assert (cell >= 0).all() and (cell < [nx, ny, nz]).all()  # the box holds every padded sphere
        for cx, cy, cz in cell:
            assert i in lists[(cz * ny + cy) * nx + cx], (i, cx, cy, cz)
    # too few gridded spheres: no grid
    few = (_lib.rtm_sphere * 40)(*arr[4:44])
    assert _grid_build(few, 40)[0] == _lib.lib().rtm_debug_grid_build(few, 40, (C.c_uint64 * 12)(), None, None, 0, None, 0, None, 0) != 0


def test_scene_edit_epoch_counts_scene_edits_only():
    """renderer.py re-uploads a scene when settings.edit_epoch() has advanced (O(1) per render; ADVICE r3: the key used to
    hash every object's identity per render, and a camera move re-flattened a 100 000-sphere scene).  What counts: writes
    to live scene objects and every assignment or mutation of a SettingData's object list.  What does not: constructing
    objects that are not attached to anything yet, and camera edits (the camera travels with every call)."""
    import raytracingmin_amd as rtm
    from raytracingmin_amd.settings import edit_epoch
    d = rtm.SettingData()
    e = edit_epoch()
    d.camera.origin.x = 3.0
    d.camera.origin = rtm.vec3(1, 2, 3)
    d.camera.origin.y = 5.0
    d.camera.fov = 3.0
    d.width, d.samples = 64, 4
    s = rtm.SphereObject(rtm.vec3(1, 1, 1), 2.0, rtm.Material(rtm.vec3(.5, .5, .5), rtm.vec3()))
    assert edit_epoch() == e
    d.object.append(s)
    assert edit_epoch() == e + 1
    s.m_position.x = 2.0
    s.m_material.emission = rtm.vec3(1, 1, 1)
    s.m_size = 3.0
    assert edit_epoch() == e + 4
    for mutate in (lambda: d.object.reverse(), lambda: d.object.insert(0, rtm.SphereObject()), lambda: d.object.pop(),
                   lambda: d.object.__setitem__(0, s), lambda: setattr(d, "object", [s, rtm.SphereObject()]),
                   lambda: d.object.extend([rtm.SphereObject()]), lambda: d.object.sort(key=id), lambda: d.object.clear()):
        before = edit_epoch()
        mutate()
        assert edit_epoch() > before
    import _oracle
    loaded = rtm.LoadData(_oracle.scene_path("cornellBoxSetting.json")).data
    before = edit_epoch()
    loaded.object.pop()
    assert edit_epoch() > before  # the loader's list is the tracking kind too


def _facts(objs):
    from raytracingmin_amd import Camera, SettingData, vec3
    data = SettingData(width=8, height=8, samples=1, superSamples=1, camera=Camera(vec3(0, 0, -10), vec3(0, 0, 0), vec3(0, 1, 0), 2.0),
                       object=objs)
    _, arr, n = data.to_c()
    out = (C.c_uint64 * 2)()
    _lib.check(_lib.lib().rtm_debug_scene_facts(arr, n, out), "scene facts")
    return int(out[0]), int(out[1])


def test_scene_facts_the_host_finds_when_it_flattens_a_scene():
    """Host logic behind two kernel specialisations (no device): SceneView::axis_pat — which spheres sit on a coordinate axis; the
    shipped Cornell box's signature selects the axis-signature instantiation — and SceneView::fold_flags — a bounce level's
    "+ emission" is an identity where nothing a path can bounce off emits and nothing carries a sign bit."""
    import raytracingmin_amd as rtm
    from raytracingmin_amd import Material, SphereObject, vec3
    cornell = 2 | (1 << 2) | (1 << 4) | (2 << 6) | (2 << 8) | (3 << 10) | (3 << 12)
    scenes = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")
    box = rtm.LoadData(os.path.join(scenes, "cornellBoxSetting.json")).data
    assert _facts(list(box.object)) == (cornell, 3)  # (fold flag | compact)
    assert _facts(list(rtm.LoadData(os.path.join(scenes, "simpleSetting1.json")).data.object))[0] == (1 << 2) | (1 << 4) | (2 << 6) | (2 << 8)
    assert _facts(list(rtm.LoadData(os.path.join(scenes, "settingData.json")).data.object))[0] == 2 | (1 << 4)

    def s(pos, col=(.5, .5, .5), em=(0, 0, 0)):
        return SphereObject(vec3(*pos), 1.0, Material(vec3(*col), vec3(*em)))
    # patterns: x, y, z, the origin, two non-zero coordinates, a negative zero counts as zero, a denormal does not, NaN / inf: none
    pat, _ = _facts([s((3, 0, 0)), s((0, -2, 0)), s((0, 0, 1e-300)), s((0, 0, 0)), s((1, 1, 0)), s((-0.0, 5, -0.0)), s((5e-324, 5, 0)),
                     s((float("nan"), 0, 0)), s((float("inf"), 0, 0))])
    assert [(pat >> (2 * i)) & 3 for i in range(9)] == [1, 2, 3, 0, 0, 2, 0, 0, 0]
    assert _facts([s((1, 0, 0))] * 40)[0] == int("01" * 32, 2)  # the first 32 spheres only
    # fold flags: on for diffuse non-emitters + black emitters; off for a diffuse emitter, a negative or -0 colour or emission
    light = s((0, 9, 0), col=(0, 0, 0), em=(5, 5, 5))
    assert _facts([light, s((1, 0, 0))])[1] == 3
    assert _facts([light, s((1, 0, 0), em=(0, 0.1, 0))])[1] == 2       # a diffuse emitter: its level adds something
    assert _facts([light, s((1, 0, 0), col=(.5, -.2, .5))])[1] == 2     # a negative colour: a product may be -0 ... or negative
    assert _facts([light, s((1, 0, 0), em=(0, -0.0, 0))])[1] == 2       # (x + -0 is an identity too, but the rule keeps to +0)
    assert _facts([s((0, 9, 0), col=(0, 0, 0), em=(5, -1, 5)), s((1, 0, 0))])[1] == 2  # a path may END anywhere: no negative start
    assert _facts([])[1] == 3 and _facts([])[0] == 0
    # compact: every |centre| + radius within 1e7
    big = SphereObject(vec3(0, 0, 2e9), 1.9e9, Material(vec3(.5, .5, .5), vec3(0, 0, 0)))
    assert _facts([light, big])[1] == 1 and _facts([light, s((9.9e6, 0, 0))])[1] == 3 and _facts([light, s((1.1e7, 0, 0))])[1] == 1
