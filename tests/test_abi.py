"""The C-ABI shared library loads and exports exactly the entry points include/rtx.h declares.  No GPU, no compute."""
import ctypes
import os
import re

import pytest
import torch

import raytracer_amd as rt
from conftest import ROOT
from raytracer_amd import api


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(rt.lib_path())
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/rtx.h but not exported by librtx.so"
    assert sorted(api.EXPORTED_SYMBOLS) == names, "python binding and header disagree on the entry points"


# struct of include/rtx.h -> its ctypes mirror in raytracer_amd/api.py
MIRRORS = {"rtx_scene_view": api._SceneView, "rtx_camera": api._Camera, "rtx_grid_info": api.GridInfo,
           "rtx_render_params": api._RenderParams, "rtx_stats": api.RenderStats}


def test_ctypes_mirrors_match_the_header_as_a_c_compiler_lays_it_out(tmp_path):
    """tests/c/abi_layout.c, compiled as C11 against include/rtx.h (which also proves the header is C), prints the size
    of every struct and the offset and size of every field; the ctypes mirrors the tests reach the library through
    must agree field by field, in the same order."""
    import shutil
    import subprocess
    cc = shutil.which("cc") or shutil.which("gcc")
    assert cc, "no C compiler"
    exe = str(tmp_path / "abi_layout")
    subprocess.run([cc, "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_layout.c"), "-o", exe], check=True)
    lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    sizes, fields = {}, {}
    for ln in lines:
        t = ln.split()
        if len(t) == 3 and t[0] == "struct":
            sizes[t[1]] = int(t[2])
        elif len(t) == 3:
            st, f = t[0].split(".")
            fields.setdefault(st, []).append((f, int(t[1]), int(t[2])))
    assert sorted(sizes) == sorted(MIRRORS)
    # every struct the header declares is printed, and every field of it
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rtx.h")).read(), flags=re.S)
    declared = dict(re.findall(r"typedef struct (rtx_\w+) \{(.*?)\} \1;", text, flags=re.S))
    assert sorted(declared) == sorted(MIRRORS)
    for name, body in declared.items():
        names = [re.sub(r"\[.*", "", d.strip().lstrip("*")) for decl in body.split(";") if decl.strip()
                 for d in re.sub(r"^\s*(const\s+)?\w+\s*\*?", "", decl.strip(), count=1).split(",")]
        assert names == [f for f, _, _ in fields[name]], f"{name}: abi_layout.c does not print every field"
    for name, mirror in MIRRORS.items():
        assert ctypes.sizeof(mirror) == sizes[name], name
        assert [f for f, _ in mirror._fields_] == [f for f, _, _ in fields[name]], f"{name}: field order"
        for f, off, size in fields[name]:
            d = getattr(mirror, f)
            assert (d.offset, d.size) == (off, size), f"{name}.{f}"


def test_error_strings():
    lib = api._load()
    assert lib.rtx_strerror(0) == b"ok"
    for code in range(-6, 0):
        assert lib.rtx_strerror(code) not in (b"ok", b"unknown error code")
    assert lib.rtx_strerror(-99) == b"unknown error code"


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour on a machine without a GPU")
def test_no_gpu_means_a_loud_error_not_a_fallback():
    with pytest.raises(rt.RtxError) as e:
        rt.Renderer(0)
    assert e.value.code == -3 and "no CPU path" in str(e.value)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text and "ref_cpu" not in text, f"{f} refers to the oracle"
