#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection CSVs: mean counter value per dispatch for every (kernel, counter).
usage: python tools/pmc_summarize.py <dir with *counter_collection.csv> [...more dirs] > summary.json"""
import csv
import glob
import json
import os
import sys

acc = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
                e = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, 0, set()])
                e[0] += float(row["Counter_Value"])
                e[2].add(row["Dispatch_Id"])
out = {}
for k, cs in acc.items():
    out[k] = {c: v[0] / max(1, len(v[2])) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v[2]) for v in cs.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out[k] and "GRBM_GUI_ACTIVE" in out[k]:
        # busy cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs
        out[k]["mfma_pipe_busy"] = round(out[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (out[k]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    if "SQ_WAIT_ANY" in out[k] and "SQ_WAVE_CYCLES" in out[k]:
        out[k]["wave_parked"] = round(out[k]["SQ_WAIT_ANY"] / out[k]["SQ_WAVE_CYCLES"], 4)
json.dump(out, sys.stdout, indent=1)
