"""The C-ABI library without a GPU: it loads, exports every symbol that
include/slamfe.h declares, fails loudly where a device is needed, and the
ctypes views have the layouts of the header's structs."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from slam_amd import abi
from slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "slamfe.h")).read()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(slamfe_[a-z0-9_]+)\s*\(", body)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in slamfe.h but not exported"
    assert sorted(_lib.SIGNATURES) == names  # the Python binding covers the whole header


def test_every_entry_point_cites_the_reference_or_says_it_has_no_counterpart():
    # the comment in front of each declaration names the reference lines it
    # replaces (Init.cc / keyframe.h / main.cc), refers to the call it is a
    # variant of, or states that the reference has no counterpart
    text = re.sub(r"\n#define[^\n]*", "", HEADER)
    for n in declared_functions():
        i = text.index(n + "(")
        before = text[max(0, i - 2500):i]
        last_comment = before[before.rfind("/*"):]
        assert re.search(r"Init\.cc:\d+|keyframe\.h:\d+|main\.cc:\d+|[Nn]o counterpart",
                         last_comment), (n, last_comment[:200])


def test_struct_layouts_match_header():
    assert C.sizeof(abi.Params) == 4 * 16 + 8 + 8
    assert abi.KEYPOINT_DTYPE.itemsize == 28 and abi.MATCH2_DTYPE.itemsize == 16
    lib = _lib.load()
    p = abi.Params()
    assert lib.slamfe_default_params(C.byref(p)) == 0
    q = abi.reference_params()
    for name, _ in abi.Params._fields_:
        assert getattr(p, name) == getattr(q, name), name
    assert (p.fast_threshold, p.nfeatures, p.nlevels, p.edge_threshold, p.patch_size) == (30, 10000, 8, 31, 31)
    assert abs(p.scale_factor - 1.2) < 1e-6 and p.ratio == 0.75
    assert lib.slamfe_version() >= 100


def test_create_fails_loudly_without_a_device_or_with_bad_params():
    import torch
    lib = _lib.load()
    ctx = C.c_void_p()
    bad = abi.reference_params(nlevels=99)
    assert lib.slamfe_create(C.byref(bad), C.byref(ctx)) == abi.SLAMFE_E_INVALID
    bad = abi.reference_params(struct_size=12)
    assert lib.slamfe_create(C.byref(bad), C.byref(ctx)) == abi.SLAMFE_E_INVALID
    if not torch.cuda.is_available():
        from slam_amd.frontend import FrontEnd, SlamfeError
        with pytest.raises(SlamfeError) as e:
            FrontEnd()
        assert e.value.code == abi.SLAMFE_E_DEVICE  # no CPU fallback exists


def test_null_context_calls_are_rejected():
    lib = _lib.load()
    assert lib.slamfe_sync(None) == abi.SLAMFE_E_INVALID
    assert lib.slamfe_match_knn2(None, None, 0, None, 0, None) == abi.SLAMFE_E_INVALID
    assert lib.slamfe_last_error(None) == b"null context"


def test_product_never_touches_the_checker():
    # nothing under slam_amd/ or include/ may reference oracle/
    for base in ("slam_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    assert "oracle" not in text.lower(), os.path.join(dirpath, fn)


def test_host_adapter_compiles_as_cxx14():
    # the reference builds with -std=c++14 (CMakeLists.txt:4)
    src = "#include \"slamfe_host.hpp\"\nint main(){ slamfe::Mat8 m; return m.rows; }\n"
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "slam_amd", "host"), "-x", "c++", "-"],
                       input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
