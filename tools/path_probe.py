#!/usr/bin/env python3
"""Prints mi355_sw_last_path for a set of inputs under every switch (what tests/test_gpu_round4.py asserts engagement with)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench

pgs = bench.load_package()
from switch_inputs import INPUTS, SWITCHES, run_input  # noqa: E402

only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
for name in INPUTS:
    if only and name not in only:
        continue
    base = None
    for sw in ["default"] + SWITCHES:
        c = pgs.Context(0)
        try:
            if sw != "default":
                for part in sw.split("+"):
                    k, _, v = part.partition("=")
                    c.set_option(k, v or True)
            try:
                path = run_input(pgs, c, name)
            except Exception as e:
                path = ["ERROR", repr(e)[:120]]
        finally:
            c.close()
        if sw == "default":
            base = path
            print(json.dumps({"input": name, "switch": sw, "path": path}), flush=True)
        elif path != base:
            print(json.dumps({"input": name, "switch": sw, "path": path}), flush=True)
