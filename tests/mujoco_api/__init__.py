"""Test double for MuJoCo's C API (see include/mujoco.h): builds a fake libmujoco.so on this package's surrogate integrator and the
real plugin (egopose_amd/csrc/egp_physics_mujoco.cpp) against it, in a scratch directory. Pins no physics."""
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def write_fake_model(skel, mjcf_path, damping=1.0, support_k=2000.0, support_c=200.0):
    """The side-car `mj_loadXML` of the fake reads (<mjcf_path>.fakemj): the skeleton tables + the surrogate's constants."""
    M0 = skel.zero_pose_inertia()
    nbody, njoint = len(skel.body_names), len(skel.joint_names)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).tobytes()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64).tobytes()
    with open(mjcf_path + ".fakemj", "wb") as f:
        f.write(struct.pack("8i", skel.nq, skel.nv, skel.nu, nbody, skel.nM, njoint, 0, 0))
        f.write(struct.pack("4d", float(skel.timestep), damping, support_k, support_c))
        for a in (skel.dof_parentid, skel.dof_Madr, skel.body_parent, skel.body_ndof, skel.body_qpos_start):
            f.write(i32(a))
        for a in (skel.sparse_from_full(M0), np.linalg.inv(M0), skel.body_pos.ravel(), skel.joint_axis.ravel(), skel.joint_anchor.ravel()):
            f.write(f64(a))
    if not os.path.exists(mjcf_path):
        open(mjcf_path, "w").write("<!-- placeholder: the fake mj_loadXML reads %s.fakemj -->\n" % os.path.basename(mjcf_path))
    return mjcf_path


def build(out_dir):
    """-> path of the plugin built against the fake: <out_dir>/mujoco/{include,lib/libmujoco.so}, <out_dir>/libegopose_mujoco.so."""
    from egopose_amd.build import build as build_main
    from egopose_amd import build_mujoco
    main_lib = build_main()
    root = os.path.join(out_dir, "mujoco")
    os.makedirs(os.path.join(root, "include"), exist_ok=True)
    os.makedirs(os.path.join(root, "lib"), exist_ok=True)
    inc = os.path.join(HERE, "include")
    with open(os.path.join(inc, "mujoco.h")) as src, open(os.path.join(root, "include", "mujoco.h"), "w") as dst:
        dst.write(src.read())
    fake = os.path.join(root, "lib", "libmujoco.so")
    subprocess.run([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + inc, os.path.join(HERE, "fake_mujoco.cpp"),
                    "-o", fake, main_lib, "-Wl,-rpath," + os.path.dirname(main_lib)], check=True)
    old = os.environ.get("MUJOCO_DIR")
    os.environ["MUJOCO_DIR"] = root
    try:
        return build_mujoco.build(verbose=False, out=os.path.join(out_dir, "libegopose_mujoco.so"))
    finally:
        if old is None:
            os.environ.pop("MUJOCO_DIR", None)
        else:
            os.environ["MUJOCO_DIR"] = old
