"""Build libmdbg_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmdbg_hip.so")
SOURCES = ["context", "prims", "reads", "scan", "minimizers", "kminmer", "partition", "multigpu"]
HEADERS = ["common.hpp", "murmur.hpp", "objects.hpp", "table.hpp", "kminmer_dev.hpp", "peerlink.hpp", os.path.join("..", "..", "include", "mdbg_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force: bool = False, verbose: bool = False, tag: str = "", extra_flags: list[str] | None = None) -> str:
    """tag/extra_flags build an experimental variant libmdbg_hip_<tag>.so (select it with MDBG_LIB=<path>)."""
    global LIB
    lib_path = LIB if not tag else os.path.join(HERE, f"libmdbg_hip_{tag}.so")
    flags = FLAGS + (extra_flags or [])
    objdir = os.path.join(HERE, "build" + ("_" + tag if tag else ""))
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s + ".hip")
        obj = os.path.join(objdir, s + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + flags + ["-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for cmd, r in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose or r.returncode:
                    print(" ".join(cmd)); print(r.stdout, r.stderr)
                if r.returncode:
                    raise RuntimeError("hipcc failed for " + cmd[-3])
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if force or jobs or _stale(lib_path, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout, r.stderr)
            raise RuntimeError("link of libmdbg_hip.so failed")
    if not tag:
        build_tool()
    return lib_path


def build_tool() -> str:
    """The C++ host tool (drop-in readSelection / graph over the C ABI), in-tree at metamdbg_amd/bin/mdbg_tool."""
    src = os.path.join(HERE, "host", "mdbg_tool.cpp")
    deps = [src, os.path.join(HERE, "host", "fastx.hpp"), os.path.join(HERE, "host", "hostfeed.hpp"),
            os.path.join(HERE, "host", "inflate.hpp"), os.path.join(HERE, "host", "gzip_parallel.hpp"), os.path.join(HERE, "host", "crc32_fast.hpp"), os.path.join(HERE, "..", "include", "mdbg_hip.h"), LIB]
    out = os.path.join(HERE, "bin", "mdbg_tool")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if _stale(out, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", src, "-o", out, "-L" + HERE, "-lmdbg_hip", "-lz", "-lpthread",
               "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout, r.stderr)
            raise RuntimeError("g++ failed for mdbg_tool")
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:   # python build.py <tag> <extra flags...>
        print(build_lib(verbose=True, tag=sys.argv[1], extra_flags=sys.argv[2:]))
    else:
        print(build_lib(verbose=True))
