"""Build recipe for libegopose_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun).

Every translation unit is compiled to an object of its own (in parallel, reused while the source and the headers are
older than it) and the objects are linked into the shared library: a one-kernel change costs one file's compile time."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libegopose_hip.so")
SOURCES = ["egp_kernels.hip", "egp_lstm.hip", "egp_policy.hip", "egp_gemm.hip", "egp_dynamics.hip", "egp_engine.hip", "egp_update.hip", "egp_probe.hip",
           "egp_physics.cpp"]
HEADERS = ["egp_internal.hpp", "egp_quat.hpp", "egp_filter_dev.hpp", "egp_dynamics_dev.hpp", "egp_tree58.inc", "egp_pd_grid.hpp", os.path.join("..", "..", "include", "egopose_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wno-unused-result", "-Xarch_host", "-mavx2", "-Xarch_host", "-mfma"]


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _obj_of(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _obj_stale(src, force):
    o = _obj_of(src)
    if force or not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose):
    cmd = [_hipcc()] + FLAGS + os.environ.get("EGP_BUILD_DEFS", "").split() + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", _obj_of(src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into egopose_amd/libegopose_hip.so."""
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in SOURCES if _obj_stale(s, force)]
    if not todo and os.path.exists(LIB) and all(os.path.getmtime(_obj_of(s)) <= os.path.getmtime(LIB) for s in SOURCES):
        return LIB
    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as pool:
        list(pool.map(lambda s: _compile(s, verbose), todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + [_obj_of(s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
