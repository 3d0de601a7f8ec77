"""rtm_cli — the reference's main() over the C ABI (-m gpu): shipped scene files run unchanged and the
files written match the oracle's quantised image."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "raytracingmin_amd", "rtm_cli")


@pytest.fixture(scope="module", autouse=True)
def _built_cli():
    if not os.path.exists(CLI):  # a tree without build artefacts: build them (hipcc is in the image)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracingmin_amd", "csrc")],
                              stdout=subprocess.DEVNULL)
    assert os.path.exists(CLI)


def test_cli_renders_shipped_scene_and_writes_both_files(tmp_path, oracle):
    from PIL import Image
    scene = oracle.scene_path("cornellBoxSetting.json")
    stem = str(tmp_path / "result")
    w, h, s, ss = 96, 56, 64, 2  # 256 spp: smooth enough for a meaningful JPEG comparison
    r = subprocess.run([CLI, "-json", scene, "--width", str(w), "--height", str(h), "--samples", str(s),
                        "--superSamples", str(ss), "--max-bounces", "8", "--seed", "24301", "--out", stem],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Msamples/s" in r.stdout
    bmp = np.array(Image.open(stem + ".bmp"))
    jpg = np.array(Image.open(stem + ".jpg").convert("RGB"))
    assert bmp.shape == (h, w, 3) and jpg.shape == (h, w, 3)
    st, arr, n = oracle.load_scene(scene, width=w, height=h, samples=s, super_samples=ss)
    ref, _ = oracle.render(st, arr, n, oracle.make_options(mode=1, max_bounces=8, seed=24301, height=h))
    want = oracle.quantise(ref)
    assert np.array_equal(bmp, want)  # rtm_cli's default is host-libm trig: the oracle's bytes exactly
    # quality-60 4:2:0 JPEG of the same pixels: compare 8x8 block means (the frame still has noise)
    bm = lambda a: a[:h // 8 * 8, :w // 8 * 8].astype(float).reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))
    assert np.abs(bm(jpg) - bm(want)).mean() < 4 and np.abs(jpg.astype(int) - want.astype(int)).mean() < 12


def test_cli_literal_mode_matches_reference_golden(tmp_path, oracle):
    """HEAD as shipped (L0) on the shipped Cornell file: the white-disc image of SURVEY §0.3."""
    from PIL import Image
    scene = oracle.scene_path("cornellBoxSetting.json")
    stem = str(tmp_path / "lit")
    r = subprocess.run([CLI, "-json", scene, "--width", "128", "--height", "128", "--samples", "8",
                        "--superSamples", "2", "--mode", "literal", "--out", stem],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    bmp = np.array(Image.open(stem + ".bmp")).reshape(-1, 3)
    white = (bmp == 255).all(axis=1)
    assert int(white.sum()) == 1100 and not bmp[~white].any()  # golden of tests/golden/survey_8c.json


def test_cli_flags_of_the_reference(tmp_path):
    r = subprocess.run([CLI, "-?"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "-sampleJson" in r.stdout and "-json" in r.stdout
    r = subprocess.run([CLI, "-sampleJson"], capture_output=True, text=True, cwd=tmp_path, timeout=30)
    assert r.returncode == 0
    j = json.load(open(tmp_path / "settingData.json"))
    assert (j["00 width"], j["00 height"], j["00 samples"], j["00 superSamples"]) == (960, 540, 10, 4)
    # default run: no -json => settingData.json (just written, no scene) => a black 64x36 frame
    r = subprocess.run([CLI, "--width", "64", "--height", "36", "--samples", "1", "--superSamples", "1"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    from PIL import Image
    assert not np.array(Image.open(tmp_path / "result.bmp")).any()
    assert os.path.exists(tmp_path / "result.jpg")
    bad = subprocess.run([CLI, "-json", "/nonexistent/dir/x.json"], capture_output=True, text=True, timeout=30)
    assert bad.returncode != 0


def test_cli_strip_tiling_equals_single_render(tmp_path, oracle):
    """The host program's N-strip path (one strip per GPU + one RCCL gather on a multi-GPU node; here
    N virtual strips on the one GPU through the same partition/assembly code) writes the same file."""
    scene = oracle.scene_path("cornellBoxSetting.json")
    args = [CLI, "-json", scene, "--width", "88", "--height", "50", "--samples", "2", "--superSamples", "2",
            "--max-bounces", "8", "--seed", "7"]
    a = subprocess.run(args + ["--out", str(tmp_path / "one")], capture_output=True, text=True, timeout=120)
    b = subprocess.run(args + ["--virtual-strips", "3", "--out", str(tmp_path / "three")], capture_output=True,
                       text=True, timeout=120)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert open(tmp_path / "one.bmp", "rb").read() == open(tmp_path / "three.bmp", "rb").read()
    assert open(tmp_path / "one.jpg", "rb").read() == open(tmp_path / "three.jpg", "rb").read()
    many = subprocess.run(args + ["--gpus", "64"], capture_output=True, text=True, timeout=60)
    assert many.returncode != 0 and "HIP device" in many.stderr


def _run_node(cmd, stall=None):
    """rtm_cli's multi-GPU path with every stage bounded: csrc/rtm_node.cpp prints a line per stage on stderr and its
    watchdog ends a stalled run with exit code 3 and the stage's name, long before the limit given here — which is
    only the backstop, and FAILS the test with whatever stage lines were printed."""
    env = dict(os.environ, RTM_NODE_STAGE_TIMEOUT="60", RTM_NODE_RENDER_TIMEOUT="60")
    if stall:
        env.update(RTM_NODE_DEBUG_STALL=stall, RTM_NODE_STAGE_TIMEOUT="3")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        stages = [ln for ln in err.splitlines() if ln.startswith("rtm_node:")]
        pytest.fail(f"rtm_cli neither finished nor was ended by its own watchdog in 240 s; last stage lines: {stages[-3:]}")
    return r


def test_cli_stalled_stage_ends_with_its_name(tmp_path, oracle):
    """A stage that never returns (here: parked by the test hook where ncclCommInitAll would be called) must not be a
    silent hang: the watchdog exits with code 3 and names the stage on stderr."""
    scene = oracle.scene_path("cornellBoxSetting.json")
    r = _run_node([CLI, "-json", scene, "--width", "88", "--height", "50", "--samples", "2", "--superSamples", "2",
                   "--max-bounces", "8", "--gpus", "1", "--force-rccl", "--out", str(tmp_path / "x")], stall="ncclCommInitAll")
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "WATCHDOG: stage 'ncclCommInitAll over 1 device(s)" in r.stderr and "after 3 s" in r.stderr
    assert "loading" in r.stdout  # line-buffered stdout: the progress line is there although the run was cut short
    assert not os.path.exists(str(tmp_path / "x.bmp"))


def test_cli_forced_rccl_gather_equals_plain_render(tmp_path, oracle):
    """rtm_cli --gpus 1 --force-rccl takes csrc/rtm_node.cpp's RCCL branch on the one GPU of this box —
    ncclCommInitAll + the grouped send/recv of the float3 + 8-bit band stack (to itself) + the band-wise
    de-interleave — and must deliver the bytes of the plain render: files and float3 buffer."""
    scene = oracle.scene_path("cornellBoxSetting.json")
    args = [CLI, "-json", scene, "--width", "88", "--height", "50", "--samples", "2", "--superSamples", "2",
            "--max-bounces", "8", "--seed", "7"]
    a = subprocess.run(args + ["--out", str(tmp_path / "one"), "--dump-f32", str(tmp_path / "one.f32")],
                       capture_output=True, text=True, timeout=120)
    b = _run_node(args + ["--gpus", "1", "--force-rccl", "--out", str(tmp_path / "rccl"), "--dump-f32",
                          str(tmp_path / "rccl.f32")])
    assert "stage: ncclCommInitAll over 1 device(s), bootstrap interface lo" in b.stderr  # the loopback bootstrap
    assert "stage: grouped ncclSend/ncclRecv" in b.stderr and "stage: de-interleave" in b.stderr
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    for ext in (".bmp", ".jpg", ".f32"):
        assert open(str(tmp_path / "one") + ext, "rb").read() == open(str(tmp_path / "rccl") + ext, "rb").read(), ext
    f32 = np.fromfile(tmp_path / "rccl.f32", dtype=np.float32).reshape(50, 88, 3)
    st, arr, n = oracle.load_scene(scene, width=88, height=50, samples=2, super_samples=2)
    ref, _ = oracle.render(st, arr, n, oracle.make_options(mode=1, max_bounces=8, seed=7, height=50))
    assert np.array_equal(f32.view(np.uint32), ref.astype(np.float32).view(np.uint32))


def test_cli_renders_a_scene_with_planes(tmp_path, oracle):
    """scenes/planeRoom.json (objectType 2 = png::PlaneObject, the build-defined extension) through the host
    program: the bytes of the oracle's quantised frame."""
    from PIL import Image
    import raytracingmin_amd as rtm
    scene = oracle.scene_path("planeRoom.json")
    stem = str(tmp_path / "room")
    r = subprocess.run([CLI, "-json", scene, "--width", "80", "--height", "48", "--samples", "4", "--superSamples", "2",
                        "--max-bounces", "8", "--seed", "5", "--out", stem], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    data = rtm.LoadData(scene).data
    data.width, data.height, data.samples, data.superSamples = 80, 48, 4, 2
    arr, n = data.objects_c()
    oobj = (oracle.Object * n).from_buffer_copy(bytes(arr))
    ost = oracle.Settings.from_buffer_copy(bytes(data.settings_c()))
    ref, _ = oracle.render_objects(ost, oobj, n, oracle.make_options(mode=1, max_bounces=8, seed=5, height=48))
    assert np.array_equal(np.array(Image.open(stem + ".bmp")), oracle.quantise(ref))
