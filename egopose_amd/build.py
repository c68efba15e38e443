"""Build recipe for libegopose_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libegopose_hip.so")
SOURCES = ["egp_kernels.hip", "egp_lstm.hip", "egp_policy.hip", "egp_gemm.hip", "egp_dynamics.hip", "egp_engine.hip", "egp_physics.cpp"]
HEADERS = ["egp_internal.hpp", "egp_quat.hpp", "egp_dynamics_dev.hpp", "egp_tree58.inc", "egp_pd_grid.hpp", os.path.join("..", "..", "include", "egopose_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into egopose_amd/libegopose_hip.so."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-Wno-unused-result", "-Xarch_host", "-mavx2", "-Xarch_host", "-mfma", "-x", "hip"]
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    cmd += ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
