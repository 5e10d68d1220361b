"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, the ctypes table covers the header, and there is no silent CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mdbg_hip.h")


def declared_functions() -> list[str]:
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdbg_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from metamdbg_amd import build, capi
    build.build_lib()
    return capi.LIB_PATH


def test_header_declares_a_c_abi():
    src = open(HEADER).read()
    assert 'extern "C"' in src
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code   # plain C types at the boundary
    assert len(declared_functions()) >= 30


def test_library_exports_every_declared_symbol(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mdbg_[a-z0-9_]+)", out))
    missing = [f for f in declared_functions() if f not in exported]
    assert not missing, f"declared in include/mdbg_hip.h but not exported: {missing}"


def test_ctypes_table_matches_header(lib_path):
    from metamdbg_amd import capi
    assert sorted(capi.SIGNATURES) == declared_functions()
    lib = capi.lib()                      # loads the .so and binds every symbol
    for name in capi.SIGNATURES:
        assert hasattr(lib, name)


def test_no_cpu_fallback_without_gpu(lib_path):
    """Without a GPU (this container) context creation must fail loudly, not fall back."""
    from metamdbg_amd import capi
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(capi.MdbgError) as ei:
        capi.Context(0)
    assert ei.value.code == -2 and "no CPU path" in str(ei.value)


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under metamdbg_amd/ may reference it."""
    pkg = os.path.join(ROOT, "metamdbg_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath).startswith("build"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and "mdbg_oracle.h" not in text, f


def test_c_example_compiles_links_and_fails_loudly_without_gpu(lib_path, tmp_path):
    """examples/first_pass.c is strict C99: the header is a C header, the library links from C, and on a box without a
    gfx950 device the program stops at mdbg_create with the library's message instead of computing anything."""
    exe = str(tmp_path / "first_pass")
    libdir = os.path.dirname(lib_path)
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "first_pass.c"), "-o", exe, "-L" + libdir, "-lmdbg_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU tests run the path")
    r = subprocess.run([exe], input="ACGTACGTACGTACGTACGT\n", capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no CPU path" in r.stderr


def test_test_double_covers_what_the_host_programs_import(tmp_path):
    """tests/host/stub_mdbg_hip.cpp stands in for the library under refdrv_hip and mdbg_tool in the CPU tests: every mdbg_* symbol those
    programs import must be defined by it, or they stop at start-up with a symbol lookup error (an entry point added to the binding
    without its line in the double did exactly that)."""
    stub = str(tmp_path / "libmdbg_hip.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "host", "stub_mdbg_hip.cpp"), "-o", stub, "-lpthread"], check=True)
    out = subprocess.run(["nm", "-D", "--defined-only", stub], capture_output=True, text=True, check=True).stdout
    defined = set(re.findall(r" T (mdbg_[a-z0-9_]+)", out))
    programs = [p for p in (os.path.join(ROOT, "oracle", "_ref", "refdrv_hip"), os.path.join(ROOT, "metamdbg_amd", "bin", "mdbg_tool")) if os.path.exists(p)]
    if not programs:
        pytest.skip("neither refdrv_hip nor mdbg_tool is built here")
    for prog in programs:
        und = subprocess.run(["nm", "-D", "--undefined-only", prog], capture_output=True, text=True, check=True).stdout
        wanted = set(re.findall(r" U (mdbg_[a-z0-9_]+)", und))
        missing = sorted(wanted - defined)
        assert not missing, f"{os.path.basename(prog)} imports {missing}: not defined by tests/host/stub_mdbg_hip.cpp"
