#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, CSV output) for one kernel:
python tools/pmc_summary.py [--kernel SUBSTRING] <dir> [<dir> ...] > profiles/<name>.json
Sums each counter over the dispatches whose kernel name contains SUBSTRING (default sw_score_kernel; the first matching
name is reported) and divides by its launches."""
import csv
import glob
import json
import os
import sys

tot = {}
launches = {}
kernel = None
args = sys.argv[1:]
want = "sw_score_kernel"
if args and args[0] == "--kernel":
    want = args[1]
    args = args[2:]
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if want in r["Kernel_Name"]]
        if not rows:
            continue
        kernel = kernel or rows[0]["Kernel_Name"]
        seen = {}
        for r in rows:
            c = r["Counter_Name"]
            tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
            seen.setdefault(c, set()).add(r["Dispatch_Id"])
        for c, ids in seen.items():
            launches[c] = launches.get(c, 0) + len(ids)
per = {c: tot[c] / max(1, launches[c]) for c in tot}
out = {"kernel": kernel, "counters_per_launch": per, "launches_seen": launches}
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE x2.0 for this kernel's 1-byte-per-lane load pattern
    # (tools/ubench/fetch_calib.hip, profiles/r01_pmc_score_kernel_v1.json)
    out["hbm_bytes_per_launch"] = per["FETCH_SIZE"] * 1024 * 2.0 + per["WRITE_SIZE"] * 1024
d = {}
if "GRBM_GUI_ACTIVE" in per and "SQ_INSTS_VALU" in per:
    # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles
    simd_cycles = per["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    d["valu_cycles_per_instr_per_simd"] = simd_cycles / per["SQ_INSTS_VALU"]
    d["valu_busy_frac"] = 4.0 * per["SQ_INSTS_VALU"] / simd_cycles
if "SQ_LDS_BANK_CONFLICT" in per and "SQ_LDS_IDX_ACTIVE" in per:
    d["lds_bank_conflict_frac"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
if "SQ_WAVE_CYCLES" in per:
    for c, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac"),
                    ("SQ_ACTIVE_INST_VALU", "active_inst_valu_frac"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds_frac")):
        if c in per:
            d[name] = per[c] / per["SQ_WAVE_CYCLES"]
if "SQ_WAVES" in per and "SQ_WAVE_CYCLES" in per and "GRBM_GUI_ACTIVE" in per:
    # SQ_WAVE_CYCLES counts quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs
    d["mean_resident_waves_per_simd"] = per["SQ_WAVE_CYCLES"] * 4.0 / (per["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
out["derived"] = d
print(json.dumps(out, indent=1))
