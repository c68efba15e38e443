#!/usr/bin/env python3
"""tests/golden/resnet18_keys.json: the state-dict layout of torchvision's `resnet18` with `fc` -> 128 outputs, i.e. of the
reference's models/resnet.py:6-18 (`self.resnet = models.resnet18(...)`; `self.resnet.fc = nn.Linear(512, out_dim)`).

torchvision is not in the image, so the table is written from the published layout rules (torchvision/models/resnet.py:
stem conv1 7x7/2 + bn1; layer1..4 = two BasicBlocks each with conv1/bn1/conv2/bn2, 64-128-256-512 planes, the first block
of layers 2-4 strides by 2 and carries `downsample.0` (1x1 conv) + `downsample.1` (batch norm); `fc`), NOT from
egopose_amd.nets.ResNet18 -- the test that consumes it (tests/test_nets_golden.py) checks that module against this table, so a
torchvision checkpoint is guaranteed to load. Known totals used as a self-check: 11 176 512 backbone parameters (resnet18's
published 11 689 512 minus its 512 x 1000 + 1000 classifier)."""
import json
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resnet18_keys.json")


def bn(prefix, c):
    return [(prefix + ".weight", [c]), (prefix + ".bias", [c]), (prefix + ".running_mean", [c]), (prefix + ".running_var", [c]),
            (prefix + ".num_batches_tracked", [])]


def table(out_dim=128):
    rows = [("conv1.weight", [64, 3, 7, 7])] + bn("bn1", 64)
    cin = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for b in range(2):
            p = "layer%d.%d" % (li, b)
            rows += [(p + ".conv1.weight", [planes, cin if b == 0 else planes, 3, 3])] + bn(p + ".bn1", planes)
            rows += [(p + ".conv2.weight", [planes, planes, 3, 3])] + bn(p + ".bn2", planes)
            if b == 0 and li > 1:
                rows += [(p + ".downsample.0.weight", [planes, cin, 1, 1])] + bn(p + ".downsample.1", planes)
        cin = planes
    rows += [("fc.weight", [out_dim, 512]), ("fc.bias", [out_dim])]
    return rows


if __name__ == "__main__":
    rows = table()
    n_backbone = 0
    for k, shp in rows:
        if k.startswith("fc.") or "running" in k or "num_batches" in k:
            continue
        n = 1
        for d in shp:
            n *= d
        n_backbone += n
    assert n_backbone == 11176512, n_backbone
    json.dump({"source": "torchvision resnet18 layout rules, fc -> 128 (models/resnet.py:6-18)", "keys": [[k, s] for k, s in rows],
               "trainable_backbone_parameters": n_backbone}, open(OUT, "w"), indent=0)
    print("wrote", OUT, len(rows), "entries")
