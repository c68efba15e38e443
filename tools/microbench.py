#!/usr/bin/env python3
"""Kernel microbenchmarks K1-K6 / K8 at several env counts (HIP events on the launch stream).
Prints one JSON line per (kernel, n): time per launch, algorithmic bytes (float64),
achieved GB/s and fraction of the 8 TB/s HBM peak. (The measurement itself lives in
egopose_amd/bench_support.py: bench.py reports the same table in its `kernels` block.)

    python tools/microbench.py [n ...]        default: 1024 8192 65536
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.bench_support import kernel_microbench


def main():
    sizes = [int(s) for s in (sys.argv[1:] or ["1024", "8192", "65536"])]
    for row in kernel_microbench(sizes, variants=True):
        print(json.dumps(row))


if __name__ == "__main__":
    main()
