"""Seeded parity fuzz against the CPU oracle (needs a GPU; uses tests/_oracle.py like the GPU tests do).
    python profiles/fuzz_parity.py <seed> <cases> <host_trig 0|1>
Random scene size and kind (open stress scene / closed box packed with small spheres), image shape,
S, SS, bounce cap (0 .. 200 or unlimited), mode and kernel variant; every frame and its counters are
compared with the oracle bit for bit.  Round 3: every fourth case is a scene with png::PlaneObject entries (axis-aligned and
tilted squares among spheres inside a room sphere; variants 0, 1, 2, 9), and a third of the cases go through the
enqueue-only entry point (rtm_render_scene without rtm_stats: sticky status, fixed trip budget of the large-scene
pipeline) instead of the blocking one.  Round 4: plane scenes up to 1 000 objects (the grid with planes among the objects
every ray tests: variants 0 and 17), every eleventh case through the other integrator (png::SurfaeSample,
RTM_MODE_SURFACE_SAMPLE), and — where it serves the scene — every third eligible case through the fp64 TOLERANCE row
(variant 18), which is judged by north_star's bar (max per-pixel |delta| <= 1e-4) and counted apart, with the number of
frames that differ at all; seven-sphere rooms with the Cornell box's axis signature and random numbers (the axis-signature
instantiation of the exact-n kernels, both translation units).  Results: profiles/r1/ .. r4/fuzz_parity.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import _oracle as oracle  # noqa: E402
import raytracingmin_amd as rtm  # noqa: E402

seed0, cases, host_trig = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
rng = np.random.default_rng(seed0)
box = rtm.LoadData(oracle.scene_path("cornellBoxSetting.json")).data


def sphere(pos, r, col):
    return rtm.SphereObject(rtm.vec3(*pos), r, rtm.Material(rtm.vec3(*col), rtm.vec3(0, 0, 0)))


bad = 0
tol_cases = tol_out = tol_differ = 0
for case in range(cases):
    if case and case % 500 == 0:  # (a long run must not look hung: one line per 500 frames)
        print("...", case, "frames so far,", bad, "differ", flush=True)
    n = int(rng.choice([1, 2, 5, 7, 7, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 33, 64, 100, 254, 255, 256, 257, 400, 511, 512,
                        513, 1000, 2500]))
    kind = case % 4
    if kind == 3:  # planes and spheres mixed, in a room
        n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 24, 25, 40, 100, 254, 255, 256, 400, 1000]))
        objs = []
        for k in range(n):
            col = rtm.vec3(*map(float, rng.uniform(0.2, 0.9, 3)))
            em = rtm.vec3(3.0, 3.0, 3.0) if k % 6 == 0 else rtm.vec3(0, 0, 0)
            pos = rtm.vec3(*map(float, rng.uniform(-6, 6, 3)))
            t = int(rng.integers(3))
            if t == 0:
                objs.append(rtm.SphereObject(pos, float(rng.uniform(0.3, 2.0)), rtm.Material(col, em)))
            elif t == 1:  # axis-aligned square facing the middle of the room: exact zeros in its normal
                axis = int(rng.integers(3))
                tgt = [pos.x, pos.y, pos.z]
                tgt[axis] += 1.0 if tgt[axis] < 0 else -1.0
                up = rtm.vec3(0, 0, 1) if axis == 1 else rtm.vec3(0, 1, 0)
                objs.append(rtm.PlaneObject(pos, up, rtm.vec3(*tgt), float(rng.uniform(1, 9)), rtm.Material(col, em)))
            else:
                objs.append(rtm.PlaneObject(pos, rtm.vec3(0.1, 1, 0.2), rtm.vec3(*map(float, rng.uniform(-2, 2, 3))),
                                            float(rng.uniform(1, 9)), rtm.Material(col, em)))
        objs.append(rtm.SphereObject(rtm.vec3(0, 0, 0), 25.0, rtm.Material(rtm.vec3(0.7, 0.7, 0.7), rtm.vec3(0.3, 0.3, 0.3))))
        n = len(objs)
        cam = rtm.Camera(rtm.vec3(0, 0, -11), rtm.vec3(0, 0, 0), rtm.vec3(0, 1, 0), 1.2)
        data = rtm.SettingData(width=8, height=8, samples=1, superSamples=1, camera=cam, object=objs)
    elif kind == 0 and n == 7 and case % 8 != 0:
        # the shipped Cornell box's AXIS SIGNATURE (light on y, walls on +x -x +y -y +z -z) with other numbers: the scenes of the
        # axis-signature instantiation of the exact-n kernels (rtm_path.h: sphere_disc), camera anywhere in the room
        half = float(rng.uniform(3, 12))
        objs = [rtm.SphereObject(rtm.vec3(0, float(rng.uniform(0.3, 1.0)) * half, 0), float(rng.uniform(0.1, 0.5)) * half,
                                 rtm.Material(rtm.vec3(0, 0, 0), rtm.vec3(5, 5, 5)))]
        for k in range(6):
            r = float(10 ** rng.uniform(1, 4.5))
            pos = [0.0, 0.0, 0.0]
            pos[k // 2] = (r + half * float(rng.uniform(0.8, 1.2))) * (1 if k % 2 == 0 else -1)
            objs.append(sphere(pos, r, rng.uniform(0.2, 0.9, 3)))
        cam = rtm.Camera(rtm.vec3(*map(float, rng.uniform(-0.7, 0.7, 3) * half)), rtm.vec3(*map(float, rng.uniform(-0.2, 0.2, 3) * half)),
                         rtm.vec3(0, 1, 0), float(rng.uniform(0.8, 2.0)))
        data = rtm.SettingData(width=8, height=8, samples=1, superSamples=1, camera=cam, object=objs)
    elif kind == 0:
        objs = list(box.object)[: max(1, min(n, 7))]
        while len(objs) < n:
            objs.append(sphere(rng.uniform(-7, 7, 3), float(rng.uniform(0.3, 1.0)), rng.uniform(0.2, 0.9, 3)))
        data = rtm.SettingData(width=8, height=8, samples=1, superSamples=1, camera=box.camera, object=objs)
    else:
        data = rtm.make_stress_scene(n, seed=int(rng.integers(1 << 30)))
    data.width, data.height = int(rng.integers(1, 90)), int(rng.integers(1, 60))
    data.samples, data.superSamples = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 32])), int(rng.choice([1, 2, 3, 4]))
    mb = int(rng.choice([-1, -1, 0, 1, 2, 7, 8, 9, 15, 16, 17, 40, 200]))
    mode = "literal" if case % 5 == 4 else "repaired"
    seed = int(rng.integers(1 << 40))
    if n >= 1000:  # (the oracle's loop over all spheres: keep the frame small)
        data.width, data.height = int(rng.integers(1, 40)), int(rng.integers(1, 24))
        data.samples, data.superSamples = int(rng.choice([1, 2, 3, 4])), int(rng.choice([1, 2]))
    variant = int(rng.choice([0, 0, 0, 2, 9, 14, 3, 12]))
    if n >= 100 and case % 2 == 0:
        variant = 17  # the uniform-grid kernel by name (variant 0 picks it whenever the scene has a grid: 64 gridded spheres)
    if case % 7 == 3 and 1 <= n <= 24 and 0 <= mb <= 8 and mode == "repaired":
        variant = 15  # the labelled primary-hit-reuse row must give the same bits where it applies
    m = oracle.MODE_LITERAL if mode == "literal" else oracle.MODE_REPAIRED
    if data.has_planes():
        variant = int(rng.choice([0, 0, 1, 2, 9])) if n < 256 else int(rng.choice([0, 0, 1, 17]))
    tol = (not data.has_planes()) and 1 <= n <= 24 and data.samples * data.superSamples ** 2 < 65536 and case % 3 == 2  # (any depth since the round's last step)
    if tol:
        variant = 18
    surface = case % 11 == 5 and n >= 1
    if surface:
        variant, tol = 0, False
        m |= oracle.MODE_SURFACE_SAMPLE
        if mb < 0 or mb > 40:
            mb = 40  # (unbounded, a SurfaeSample recursion in a closed bright box can run for thousands of levels)
    only = os.environ.get("FUZZ_ONLY")  # replay ONE case (the random stream is consumed as in the full run) ...
    if only is not None and case != int(only):
        continue
    if only is not None and not data.has_planes():
        # ... through several kernels, against each other and the oracle: is a mismatch one kernel's, or everybody's?
        st, arr, cnt_n = data.to_c()
        ost = oracle.Settings.from_buffer_copy(bytes(st))
        oarr = (oracle.Sphere * max(cnt_n, 1)).from_buffer_copy(bytes(arr))
        ref, cnt = oracle.render(ost, oarr, n, oracle.make_options(mode=m, max_bounces=mb, seed=seed, height=data.height))
        print("case", case, dict(n=n, w=data.width, h=data.height, S=data.samples, SS=data.superSamples, max_bounces=mb, mode=mode),
              "oracle casts", cnt["casts"], "draws", cnt["draws"])
        for v in [int(x) for x in os.environ.get("FUZZ_VARIANTS", "1,3,17").split(",")]:
            out, stats = rtm.Renderer(data, mode=mode, max_bounces=mb, seed=seed, variant=v, host_trig=host_trig).render_rows(
                0, data.height, want=("f64",))
            print("  variant", v, "image == oracle:", bool(np.array_equal(out["f64"], ref, equal_nan=True)), "casts", stats["casts"],
                  "draws", stats["draws"], "max delta", float(np.nanmax(np.abs(out["f64"] - ref))))
        continue
    if data.has_planes():
        oarr_c, cnt_n = data.objects_c()
        oobj = (oracle.Object * max(cnt_n, 1)).from_buffer_copy(bytes(oarr_c))
        ost = oracle.Settings.from_buffer_copy(bytes(data.settings_c()))
        ref, cnt = oracle.render_objects(ost, oobj, n, oracle.make_options(mode=m, max_bounces=mb, seed=seed, height=data.height))
    else:
        st, arr, cnt_n = data.to_c()
        ost = oracle.Settings.from_buffer_copy(bytes(st))
        oarr = (oracle.Sphere * max(cnt_n, 1)).from_buffer_copy(bytes(arr))
        ref, cnt = oracle.render(ost, oarr, n, oracle.make_options(mode=m, max_bounces=mb, seed=seed, height=data.height))
    r = rtm.Renderer(data, mode=mode, max_bounces=mb, seed=seed, variant=variant, host_trig=host_trig,
                     integrator="SurfaeSample" if surface else "PathTracing")
    if tol:  # the tolerance row: north_star's bar, counted apart
        out, stats = r.render_rows(0, data.height, want=("f64",))
        tol_cases += 1
        with np.errstate(invalid="ignore"):
            worst = float(np.nanmax(np.abs(out["f64"] - ref))) if out["f64"].size else 0.0
        same = np.array_equal(out["f64"], ref, equal_nan=True)
        tol_differ += not same
        if not (worst <= 1e-4):
            tol_out += 1
            print("TOLERANCE ROW OUTSIDE 1e-4: case", case, dict(n=n, w=data.width, h=data.height, S=data.samples, SS=data.superSamples,
                                                                 max_bounces=mb, mode=mode), "max pixel delta", worst)
        continue
    if case % 3 == 1:  # the enqueue-only entry point: no rtm_stats, the stream's sticky status afterwards
        import torch
        dev_out, _ = r.render_rows_device(0, data.height, want=("f64",), stats=False)
        torch.cuda.synchronize()
        r.stream_status()
        ok = np.array_equal(dev_out["f64"].cpu().numpy(), ref, equal_nan=True)
        out = {"f64": dev_out["f64"].cpu().numpy()}
    else:
        out, stats = r.render_rows(0, data.height, want=("f64",))
        ok = np.array_equal(out["f64"], ref, equal_nan=True) and \
            (stats["casts"], stats["draws"]) == (cnt["casts"], cnt["draws"])
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(n=n, kind=("closed box", "stress", "stress", "planes")[kind], w=data.width, h=data.height, S=data.samples,
                                          SS=data.superSamples, max_bounces=mb, mode=mode, variant=variant),
              "max pixel delta", float(np.nanmax(np.abs(out["f64"] - ref))))
print("seed", seed0, "cases", cases, "host_trig", host_trig, "frames differing from the oracle:", bad,
      "| tolerance row (variant 18):", tol_cases, "frames,", tol_out, "outside 1e-4,", tol_differ, "differing at all")
