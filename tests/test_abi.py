"""The C ABI: libt3d_hip.so loads on a machine without a GPU, exports every function include/t3d.h
declares, and the ctypes mirrors have the layout the C compiler gives the structures."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from raytracing_amd import _native
from raytracing_amd.flatten import Opts, SceneDesc, Stats

HEADER = os.path.join(ROOT, "include", "t3d.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(t3d_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = C.CDLL(_native.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 11
    for name in names:
        assert hasattr(lib, name), f"{name} declared in t3d.h but not exported"
    assert set(names) == set(_native.SYMBOLS)
    lib.t3d_abi_version.restype = C.c_int
    assert lib.t3d_abi_version() == _native.ABI_VERSION
    # the one constant of the header that Python code mirrors (which scenes Photon renders as two bands, what bench.py prices)
    assert int(re.search(r"#define\s+T3D_BRUTE_FORCE_MAX_PRIMS\s+(\d+)", open(HEADER).read()).group(1)) == _native.BRUTE_FORCE_MAX_PRIMS


def test_ctypes_layout_matches_c(tmp_path):
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "t3d.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu ", sizeof(t3d_scene_desc), sizeof(t3d_opts), sizeof(t3d_stats), sizeof(t3d_node));'
                   'printf("%zu %zu %zu %zu %zu ", offsetof(t3d_scene_desc, cam_pos), offsetof(t3d_opts, ambient),'
                   'offsetof(t3d_opts, row_stride), offsetof(t3d_stats, ms_device), offsetof(t3d_node, color));'
                   'printf("%zu %zu\\n", offsetof(t3d_opts, frames), offsetof(t3d_opts, frame_seed_step));return 0;}')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(SceneDesc), C.sizeof(Opts), C.sizeof(Stats), C.sizeof(_native.Node), SceneDesc.cam_pos.offset,
            Opts.ambient.offset, Opts.row_stride.offset, Stats.ms_device.offset, _native.Node.color.offset,
            Opts.frames.offset, Opts.frame_seed_step.offset]
    assert got == want


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the library still loads; creating a context must fail with a message."""
    lib = _native.load()
    if lib.t3d_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.NativeError):
        _native.Context(0)
