#!/usr/bin/env python3
"""Config 4 on one GPU through bench.py's leg: calls, parity against host-built lists, one rank's LPT share at world 1/2/4/8."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pgs = bench.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 561_356
print(json.dumps(bench.extra_config4(pgs, 0, n), indent=1))
