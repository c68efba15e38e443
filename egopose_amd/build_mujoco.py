"""Build recipe for the MuJoCo physics plugin, egopose_amd/libegopose_mujoco.so (csrc/egp_physics_mujoco.cpp).

    MUJOCO_DIR=/path/to/mujoco210 python -m egopose_amd.build_mujoco

MuJoCo is an un-vendored dependency of the reference (README.md:20-21 names mujoco-py) and is not in the build image, so this
is NOT part of `egopose_amd.build` / `__graft_entry__.build()`: it needs a MuJoCo tree with `include/mujoco.h` (2.0 / 2.1.0) or
`include/mujoco/mujoco.h` (>= 2.1.2) and the shared library under `lib/` or `bin/`. The plugin links against libegopose_hip.so
(its public C-ABI only) and MuJoCo; `physics.MujocoPhysics` loads it. mujoco200 and older: add EGP_MUJOCO_ACTIVATE=1."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "egp_physics_mujoco.cpp")
LIB = os.path.join(HERE, "libegopose_mujoco.so")


def find_mujoco(root):
    inc = os.path.join(root, "include")
    if not (os.path.exists(os.path.join(inc, "mujoco.h")) or os.path.exists(os.path.join(inc, "mujoco", "mujoco.h"))):
        raise SystemExit("no mujoco.h under %s/include" % root)
    for sub in ("lib", "bin"):
        for pat in ("libmujoco.so*", "libmujoco210.so", "libmujoco200.so", "libmujoco*.so"):
            hits = sorted(glob.glob(os.path.join(root, sub, pat)))
            if hits:
                return inc, hits[0]
    raise SystemExit("no libmujoco*.so under %s/lib or %s/bin" % (root, root))


def build(verbose=True, out=None):
    """`out`: where to write the plugin (default: egopose_amd/libegopose_mujoco.so, where physics.MujocoPhysics looks for it)."""
    root = os.environ.get("MUJOCO_DIR")
    if not root:
        raise SystemExit("set MUJOCO_DIR to a MuJoCo tree (include/ + lib/ or bin/); MuJoCo is not shipped with this package")
    from .build import build as build_main
    main_lib = build_main()
    inc, mj = find_mujoco(root)
    lib_out = out or LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + inc, SRC, "-o", lib_out,
           main_lib, mj, "-Wl,-rpath," + os.path.dirname(mj), "-Wl,-rpath," + HERE]
    if os.environ.get("EGP_MUJOCO_ACTIVATE") == "1":
        cmd.insert(1, "-DEGP_MUJOCO_ACTIVATE")
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib_out


if __name__ == "__main__":
    print(build())
