#!/usr/bin/env python3
"""Re-serialise the reference's egomimic / egoforecast / statereg YAML configs (data, not code) into egopose_amd/assets/config/.
Run in the build container only."""
import os, sys, yaml
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for task in ("egomimic", "egoforecast", "statereg"):
    src = "/root/reference/config/%s" % task
    dst = os.path.join(REPO, "egopose_amd", "assets", "config", task)
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(src)):
        d = yaml.safe_load(open(os.path.join(src, f)))
        with open(os.path.join(dst, f), "w") as o:
            o.write("# data re-serialised from config/%s/%s of the reference by tools/make_config_assets.py\n" % (task, f))
            yaml.safe_dump(d, o, default_flow_style=None, sort_keys=False, width=120)
        print(task, f, len(d))
