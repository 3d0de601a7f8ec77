"""The C-ABI library loads and exports every symbol include/rtm.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="rtm.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from raytracingmin_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 24
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/rtm.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().rtm_abi_version() == _lib.ABI_VERSION == 5


def test_exported_symbols_are_exactly_the_two_headers():
    """librtm_hip.so exports the drop-in boundary (include/rtm.h) plus the test/diagnostic hooks of
    include/rtm_debug.h and no other rtm_* C symbol."""
    import subprocess
    from raytracingmin_amd import _lib
    debug = _declared_symbols("rtm_debug.h")
    assert sorted(_lib.DEBUG_SIGNATURES) == debug and all(n.startswith("rtm_debug_") for n in debug)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if re.search(r" T rtm_[a-z0-9_]+$", l))
    assert exported == sorted(_declared_symbols() + debug)


def test_struct_layouts_match_header():
    from raytracingmin_amd import _lib
    assert C.sizeof(_lib.rtm_camera) == 80
    assert C.sizeof(_lib.rtm_sphere) == 80
    assert C.sizeof(_lib.rtm_settings) == 96
    assert C.sizeof(_lib.rtm_options) == 40
    assert C.sizeof(_lib.rtm_stats) == 56
    assert C.sizeof(_lib.rtm_object) == 136
    # the oracle's view of the same PODs
    import _oracle
    assert C.sizeof(_oracle.Sphere) == 80 and C.sizeof(_oracle.Settings) == 96
    assert C.sizeof(_oracle.Options) == 40 and C.sizeof(_oracle.Object) == 136


def test_strerror_and_variants():
    from raytracingmin_amd import _lib
    L = _lib.lib()
    assert L.rtm_strerror(0) == b"ok"
    for code in range(-8, 0):
        assert L.rtm_strerror(code) not in (b"ok", b"unknown status")
    assert L.rtm_num_variants() >= 1
    assert L.rtm_variant_name(0)
    assert L.rtm_variant_name(10_000) is None


def test_invalid_arguments_are_rejected_without_a_gpu():
    from raytracingmin_amd import _lib
    L = _lib.lib()
    st = _lib.rtm_settings()
    opt = _lib.rtm_options()
    assert L.rtm_render(None, None, 0, None, None, None, None, None) == -1
    st.width, st.height, st.samples, st.super_samples = 0, 4, 1, 1
    opt.row_end = 4
    assert L.rtm_render(C.byref(st), None, 0, C.byref(opt), None, None, None, None) == -1
    st.width = 4
    opt.row_end = 5  # outside the image
    assert L.rtm_render(C.byref(st), None, 0, C.byref(opt), None, None, None, None) == -1
    opt.row_end, opt.mode = 4, 7
    assert L.rtm_render(C.byref(st), None, 0, C.byref(opt), None, None, None, None) == -1
    assert b"mode" in L.rtm_last_error_detail()


def test_host_rng_matches_oracle(oracle):
    from raytracingmin_amd import _lib
    L = _lib.lib()
    for seed in (0, 1, 0x5EED, 2 ** 63 + 12345):
        for pixel in (0, 1, 77, 2 ** 23 - 1, 2 ** 32 - 1):
            for sample in (0, 5, 4095):
                for idx in (0, 1, 2, 54, 100000):
                    assert L.rtm_rng_u01(seed, pixel, sample, idx) == \
                        oracle.lib().rtmo_rng_u01(seed, pixel, sample, idx)


def test_product_does_not_link_the_oracle():
    """The shipped library and package never reference oracle/ (grading rule of the tier)."""
    import subprocess
    from raytracingmin_amd import _lib
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rtmo_" not in out
    pkg = os.path.join(ROOT, "raytracingmin_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "cpu_ref" not in src and "rtmo_" not in src, f


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() compiles everything and imports the package
    (it asserts the ABI version, so this catches a bump that forgot it)."""
    import importlib
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_headers_compile_as_c99_and_link(tmp_path):
    """include/rtm.h and include/rtm_debug.h are a C boundary: a strict C99 translation unit that includes both
    compiles without a warning, links against librtm_hip.so and can call the entry points that need no GPU."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text(
        '#include "rtm.h"\n#include "rtm_debug.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        '    printf("%d %d %d %d\\n", (int)sizeof(rtm_settings), (int)sizeof(rtm_sphere), (int)sizeof(rtm_object), (int)sizeof(rtm_options));\n'
        "    if (rtm_abi_version() != RTM_ABI_VERSION) return 1;\n"
        "    if (!rtm_strerror(RTM_ERR_INVALID_ARGUMENT) || rtm_num_variants() < 15) return 2;\n"
        "    return rtm_render(0, 0, 0, 0, 0, 0, 0, 0) == RTM_ERR_INVALID_ARGUMENT ? 0 : 3;\n"
        "}\n")
    exe = tmp_path / "hdr"
    libdir = os.path.join(root, "raytracingmin_amd")
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                         str(src), "-o", str(exe), "-L", libdir, "-lrtm_hip", "-Wl,-rpath," + libdir],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert run.stdout.split() == ["96", "80", "136", "40"]
