"""Builds libthr_hip.so (the C-ABI of include/thr_hip.h) in-tree with hipcc for gfx950.

Every ``csrc/*.hip`` is compiled to its own object (in parallel, only when the source, a
shared header or the flags changed) and the objects are linked into one shared library --
no device symbol crosses a file, so no relocatable device code is needed."""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "build")
# THR_LIB_PATH: load another build of the library (A/B experiments of scripts/); never set in tests
LIB_PATH = os.environ.get("THR_LIB_PATH") or os.path.join(PKG_DIR, "libthr_hip.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               # every float64 operation on the parity paths is one IEEE rounding, as in
               # the oracle; the fp32 scan asks for FMAs explicitly
               "-ffp-contract=off"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + \
        [os.path.join(os.path.dirname(PKG_DIR), "include", "thr_hip.h")]


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _stamp() -> str:
    """What an object depends on besides its own source: the flags."""
    return hashlib.sha1(" ".join(HIPCC_FLAGS).encode()).hexdigest()[:12]


def _obj_stale(src: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o) or not os.path.exists(o + "." + _stamp()):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in [src] + _headers())


def build_variant(name: str, defines, only=("bm25",)) -> str:
    """An A/B build: the sources in ``only`` recompiled with extra -D flags, linked with the
    regular objects into build/libthr_<name>.so (load it with THR_LIB_PATH)."""
    build_native()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = []
    for src in sources():
        base = os.path.basename(src)[:-4]
        if base in only:
            o = os.path.join(OBJ_DIR, f"{base}.{name}.o")
            subprocess.check_call([hipcc] + HIPCC_FLAGS + [f"-D{d}" for d in defines] + ["-c", src, "-o", o])
            objs.append(o)
        else:
            objs.append(_obj(src))
    out = os.path.join(OBJ_DIR, f"libthr_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


def stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


import contextlib
import fcntl


@contextlib.contextmanager
def _build_lock():
    """Exclusive lock on build/ for the link step (the objects are each renamed into place)."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into one shared library.  hipcc cross-compiles
    without a GPU, so this also runs in the build container."""
    if os.environ.get("THR_LIB_PATH") or (not force and not stale()):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libthr_hip.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = [s for s in sources() if force or _obj_stale(s)]

    def compile_one(src):
        # to a name of this process's own, then renamed into place: several processes that find
        # the tree stale at once (torchrun ranks, the knob tests' subprocesses) never link a
        # half-written object of another
        tmp = f"{_obj(src)}.tmp{os.getpid()}"
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, _obj(src))
        for old in glob.glob(_obj(src) + ".*"):
            if ".tmp" not in os.path.basename(old):
                try:
                    os.remove(old)
                except FileNotFoundError:
                    pass
        open(_obj(src) + "." + _stamp(), "w").close()

    with ThreadPoolExecutor(max_workers=min(len(todo) or 1, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    tmp_lib = f"{LIB_PATH}.tmp{os.getpid()}"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in sources()] + \
        ["-o", tmp_lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    with _build_lock():
        subprocess.check_call(cmd)
        os.replace(tmp_lib, LIB_PATH)
    return LIB_PATH
