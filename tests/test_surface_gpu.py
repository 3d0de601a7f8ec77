"""png::SurfaeSample (src/Renderer.cpp:119-198 with SphereObject::ComputeSurfacePoint, src/SettingData.cpp:227-233) — the
reference's second integrator, which its Render never selects (:234 asks a U[0,1) draw to be >= 1.0) — on the device:
the per-ray seam rtm_surface_sample_batch and the integrator switch RTM_MODE_SURFACE_SAMPLE, both against the oracle's
literal restatement (itself cross-checked by a second one on the CPU: tests/test_oracle_golden.py), bit for bit, with
draws and casts exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = ["cornellBoxSetting.json", "simpleSetting1.json", "simpleSetting2.json", "settingData.json"]


@pytest.fixture(scope="module")
def rtm():
    import raytracingmin_amd as m
    return m


@pytest.mark.parametrize("scene", SCENES)
@pytest.mark.parametrize("max_bounces", [-1, 8, 0, 2])
def test_surface_sample_batch_vs_oracle(rtm, oracle, scene, max_bounces):
    data = rtm.LoadData(oracle.scene_path(scene)).data
    st, arr, n = oracle.load_scene(oracle.scene_path(scene))
    rng = np.random.default_rng(12)
    n_rays = 3000
    org = np.tile(list(data.camera.origin), (n_rays, 1)) + rng.uniform(-0.5, 0.5, (n_rays, 3))
    d = np.array([oracle.normalize(v) for v in rng.normal(size=(n_rays, 3))])
    for mode, omode in (("repaired", oracle.MODE_REPAIRED), ("literal", oracle.MODE_LITERAL)):
        L, draws, casts = rtm.surface_sample_batch(data, org, d, mode=mode, max_bounces=max_bounces, seed=99)
        for i in range(n_rays if mode == "repaired" else 300):
            Lo, cnt = oracle.surface_sample_stream(arr, n, omode, max_bounces, org[i], d[i], 99, i)
            assert cnt["draws"] == draws[i] and cnt["casts"] == casts[i], (mode, i, cnt, draws[i], casts[i])
            assert np.array_equal(np.array(Lo).view(np.uint64), L[i].view(np.uint64)), (mode, i, Lo, L[i])
        if mode == "repaired":
            print(f"{scene} max_bounces={max_bounces}: mean casts {casts.mean():.3f}, deepest {casts.max() - 1}, "
                  f"{int((L != 0).any(axis=1).sum())} of {n_rays} rays return light")
        else:  # the normal is lost (D2): level 1's dot1 = Dot((0,0,0), dir) is 0, so every hit ends after one more cast
            assert casts.max() <= 2


@pytest.mark.parametrize("scene,w,h,s,ss,mb", [("cornellBoxSetting.json", 96, 64, 4, 2, -1), ("simpleSetting1.json", 80, 48, 5, 1, 8),
                                               ("settingData.json", 64, 40, 3, 2, 3), ("planeRoom.json", 72, 48, 4, 1, -1)])
def test_surface_sample_integrator_vs_oracle(rtm, oracle, scene, w, h, s, ss, mb):
    """Renderer(integrator="SurfaeSample"): the loop nest of src/Renderer.cpp:215-250 with the branch :234-236 taken — image,
    8-bit view and counters against the oracle's render with the same switch; row ranges and band parts are the frame; a
    scene with planes (PlaneObject::ComputeSurfacePoint returns the origin, :247-249) goes through the object list."""
    path = oracle.scene_path(scene)
    data = rtm.LoadData(path).data
    data.width, data.height, data.samples, data.superSamples = w, h, s, ss
    opt = oracle.make_options(mode=oracle.MODE_REPAIRED | oracle.MODE_SURFACE_SAMPLE, max_bounces=mb, seed=5, height=h)
    if data.has_planes():
        arr, n = data.objects_c()
        ost = oracle.Settings.from_buffer_copy(bytes(data.settings_c()))
        ref, cnt = oracle.render_objects(ost, (oracle.Object * n).from_buffer_copy(bytes(arr)), n, opt)
    else:
        st, arr, n = oracle.load_scene(path, width=w, height=h, samples=s, super_samples=ss)
        ref, cnt = oracle.render(st, arr, n, opt)
    r = rtm.Renderer(data, mode="repaired", max_bounces=mb, seed=5, integrator="SurfaeSample")
    out, stats = r.render_rows_device(want=("f64", "u8"))
    img = out["f64"].cpu().numpy()
    assert stats["variant"] == 19 and b"SurfaeSample" in rtm.lib().rtm_variant_name(19)
    assert np.array_equal(img.view(np.uint64), ref.view(np.uint64))
    assert np.array_equal(out["u8"].cpu().numpy(), oracle.quantise(ref))
    assert (stats["casts"], stats["bounces"], stats["draws"]) == (cnt["casts"], cnt["bounces"], cnt["draws"])
    assert ref.any()
    part, _ = r.render_rows_device(8, h - 8, want=("f64",))
    assert np.array_equal(part["f64"].cpu().numpy().view(np.uint64), ref[8:h - 8].view(np.uint64))
    from raytracingmin_amd.distributed import band_row_index
    band, _ = r.render_rows_device(0, h, want=("f64",), band=(3, 2))
    assert np.array_equal(band["f64"].cpu().numpy().view(np.uint64), ref[band_row_index(0, h, 3, 2)].view(np.uint64))
    host, _ = r.render_rows(0, h, want=("f64",))  # the blocking entry points (rtm_render / rtm_render_objects)
    assert np.array_equal(host["f64"].view(np.uint64), ref.view(np.uint64))
    with pytest.raises(rtm.RtmError, match="SURFACE_SAMPLE"):
        rtm.Renderer(data, mode="repaired", max_bounces=mb, seed=5, integrator="SurfaeSample", variant=2).render_rows_device(want=("f64",))
    with pytest.raises(rtm.RtmError, match="variant 19"):
        rtm.Renderer(data, mode="repaired", max_bounces=mb, seed=5, variant=19).render_rows_device(want=("f64",))
