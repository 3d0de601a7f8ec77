#!/usr/bin/env python3
"""Generates tests/golden/frozen/*.npz — FROZEN OUTPUTS OF THE ORACLE under the build's counter RNG.

These are not reference pins (those are survey_8c.json: values recorded from the reference's own
compiled code, which the oracle reproduces bit for bit).  They freeze what the pinned oracle says for
the three seams SURVEY.md §8(c) lists as "fixtures to generate and commit":
  intersect_1k.npz   1024 (ray, sphere) pairs  -> Intersect hit / t / normal, both modes
  pathtrace_1k.npz   1024 (ray, rng stream) on the Cornell box -> radiance, draws, casts (L1, cap 8 and unlimited)
  image_<scene>.npz  64x64, 16 spp (SS 2 x S 4) images of every shipped scene (L1, unlimited depth;
                     Cornell also L0), raw float64
so that (a) a later change to oracle/cpu_ref.c that alters any bit is caught without the reference,
and (b) the GPU tests have committed expected outputs that do not depend on building the oracle.
Run from the repo root:  python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as oracle  # noqa: E402

OUT = os.path.join(HERE, "frozen")
SCENES = ["cornellBoxSetting.json", "simpleSetting1.json", "simpleSetting2.json", "settingData.json"]
SEED = 0x5EED


def intersect_cases():
    rng = np.random.default_rng(20261003)
    n = 1024
    center = np.empty((n, 3))
    radius = np.empty(n, dtype=np.float32)
    org = rng.uniform(-12, 12, (n, 3))
    d = rng.normal(size=(n, 3))
    for i in range(n):
        big = i % 3 == 0
        center[i] = rng.uniform(-1, 1, 3) * (10010 if big else 8)
        radius[i] = 10000.0 if big else rng.uniform(0.1, 6)
        d[i] = oracle.normalize(d[i])
    # rays that start on / just off the surface (the 0.001 and 1e-5f thresholds)
    for i in range(0, 64):
        p = center[i] + radius[i] * np.array(oracle.normalize(rng.normal(size=3)))
        org[i] = p + d[i] * (10.0 ** -(i % 8)) * (1 if i % 2 else -1) * 1e-3
    d[5] = np.nan
    out = {"center": center, "radius": radius, "org": org, "dir": d}
    for name, mode in (("literal", oracle.MODE_LITERAL), ("repaired", oracle.MODE_REPAIRED)):
        hit = np.zeros(n, dtype=np.int32)
        t = np.full(n, -1.0)
        nrm = np.full((n, 3), 7.0)
        for i in range(n):
            s = oracle.Sphere()
            for k in range(3):
                s.center[k] = center[i, k]
            s.radius = float(radius[i])
            h, tt, nn = oracle.intersect(s, org[i], d[i], mode)
            hit[i], t[i], nrm[i] = h, tt, nn
        out[f"hit_{name}"], out[f"t_{name}"], out[f"normal_{name}"] = hit, t, nrm
    return out


def pathtrace_cases():
    st, arr, n = oracle.load_scene(oracle.scene_path("cornellBoxSetting.json"))
    rng = np.random.default_rng(11)
    n_rays = 1024
    cam = np.array([st.camera.origin[k] for k in range(3)])
    org = np.tile(cam, (n_rays, 1)) + rng.uniform(-0.5, 0.5, (n_rays, 3))
    d = np.array([oracle.normalize(v) for v in rng.normal(size=(n_rays, 3))])
    out = {"org": org, "dir": d, "seed": np.uint64(99)}
    for mb in (8, -1):
        L = np.empty((n_rays, 3))
        draws = np.empty(n_rays, dtype=np.uint32)
        casts = np.empty(n_rays, dtype=np.uint32)
        for i in range(n_rays):
            Lo, cnt = oracle.path_trace_stream(arr, n, oracle.MODE_REPAIRED, mb, org[i], d[i], 99, i)
            L[i], draws[i], casts[i] = Lo, cnt["draws"], cnt["casts"]
        tag = "cap8" if mb == 8 else "unlimited"
        out[f"radiance_{tag}"], out[f"draws_{tag}"], out[f"casts_{tag}"] = L, draws, casts
    return out


def image_case(scene):
    out = {}
    for name, mode in (("repaired", oracle.MODE_REPAIRED),) + ((("literal", oracle.MODE_LITERAL),) if scene.startswith("cornell") else ()):
        st, arr, n = oracle.load_scene(oracle.scene_path(scene), literal_loader=(name == "literal"), width=64, height=64,
                                       samples=4, super_samples=2)
        img, cnt = oracle.render(st, arr, n, oracle.make_options(mode=mode, max_bounces=-1, seed=SEED, height=64))
        out[f"image_{name}"] = img
        out[f"casts_{name}"] = np.uint64(cnt["casts"])
        out[f"draws_{name}"] = np.uint64(cnt["draws"])
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "intersect_1k.npz"), **intersect_cases())
    np.savez_compressed(os.path.join(OUT, "pathtrace_1k.npz"), **pathtrace_cases())
    for s in SCENES:
        np.savez_compressed(os.path.join(OUT, "image_" + s.replace(".json", "") + ".npz"), **image_case(s))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
