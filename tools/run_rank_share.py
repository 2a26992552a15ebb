#!/usr/bin/env python3
"""Config 5 on one GPU: whole reference (both engines) + one rank's share at world = 1/2/4/8 (bench.py's rank_share leg)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pgs = bench.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 250_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
print(json.dumps(bench.extra_config5(pgs, 0, n, m), indent=1))
