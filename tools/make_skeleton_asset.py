#!/usr/bin/env python3
"""Derive egopose_amd/assets/humanoid_1205_v1.json (tree, joint map, mass props) from the
reference MJCF. Run in the build container only (needs /root/reference); the JSON is data."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.skeleton import parse_mjcf, DEFAULT_ASSET

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets/mujoco_models/humanoid_1205_v1.xml"
sk = parse_mjcf(src)
d = sk.to_json()
d["_source"] = "derived from assets/mujoco_models/humanoid_1205_v1.xml:22-192 by tools/make_skeleton_asset.py"
with open(DEFAULT_ASSET, "w") as f:
    json.dump(d, f, indent=1)
print("bodies", len(sk.body_names), "nq", sk.nq, "nv", sk.nv, "nu", sk.nu, "nM", sk.nM)
