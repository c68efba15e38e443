#!/usr/bin/env python3
"""Kernel microbenchmarks K1-K6 / K8 at several env counts (HIP events on the launch stream).
Prints one JSON line per (kernel, n): time per launch, algorithmic bytes (float64),
achieved GB/s and fraction of the 8 TB/s HBM peak. (The measurement itself lives in
egopose_amd/bench_support.py: bench.py reports the same table in its `kernels` block.)

    python tools/microbench.py [n ...] [--only K2_reward[,K8_dynamics...]] [--no-variants] [--rotate R]
default sizes: 1024 8192 65536 1048576. Launches rotate over input sets so that nothing is re-read from the
256 MiB Infinity Cache (`rotating_sets`, `beyond_mall` in every row); --rotate 1 = round 5's cache-resident figures.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egopose_amd.bench_support import kernel_microbench


def main():
    argv, only, variants = list(sys.argv[1:]), None, True
    if "--only" in argv:
        i = argv.index("--only")
        only = set(argv[i + 1].split(","))
        del argv[i:i + 2]
    if "--no-variants" in argv:
        argv.remove("--no-variants")
        variants = False
    rotate = None
    if "--rotate" in argv:
        i = argv.index("--rotate")
        rotate = int(argv[i + 1])
        del argv[i:i + 2]
    sizes = [int(s) for s in (argv or ["1024", "8192", "65536", "1048576"])]
    for row in kernel_microbench(sizes, variants=variants, only=only, rotate=rotate):
        print(json.dumps(row))


if __name__ == "__main__":
    main()
