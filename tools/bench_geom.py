#!/usr/bin/env python3
"""bench.py with extra experimental geometries (kernel experiments only, never the reported line):
   python3 tools/bench_geom.py --workload x1 ...   (same flags as bench.py)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.realpath(__file__))))
import bench  # noqa: E402

S420, S444 = ((2, 2), (1, 1), (1, 1)), ((1, 1),) * 3
bench.WORKLOADS.update({
    "x1": ("2048x1088 4:2:0 (two whole tiles per MCU row: c5's size without linear tiling)", 2048, 1088, S420, 256),
    "x2": ("1920x1080 4:4:4 (linear tiling, 12-byte runs)", 1920, 1080, S444, 128),
    "x4": ("1920x1080 grayscale (gray linear tiling)", 1920, 1080, ((1, 1),), 512),
    "x3": ("2048x1088 4:4:4 (whole tiles: row-pair units)", 2048, 1088, S444, 128),
})
if __name__ == "__main__":
    bench.main()
