#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE / SQ passes of tools/pmc_traffic.py -> profiles/r6_pmc_traffic.json + r6_pmc_mfma_busy.json, stamped
with the hash of the kernel sources they were collected on (safediffcon_amd.build.source_hash; bench.py refuses a stale record).
usage: python tools/pmc_to_json.py <fetch_dir> <write_dir> <sq_dir> <cases.json> <out_traffic.json> <out_busy.json>"""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd.build import source_hash  # noqa: E402

fetch_dir, write_dir, sq_dir, cases_path, out_t, out_b = sys.argv[1:7]
doc = json.load(open(cases_path))
B, cases = doc["batch"], doc["cases"]


def per_kernel(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
                e = acc.setdefault(k, {}).setdefault(row["Counter_Name"], [0.0, set()])
                e[0] += float(row["Counter_Value"])
                e[1].add(row["Dispatch_Id"])
    return {k: {c: v[0] / max(1, len(v[1])) for c, v in cs.items()} for k, cs in acc.items()}


F, W, S = per_kernel(fetch_dir), per_kernel(write_dir), per_kernel(sq_dir)


def find(tab, sub):
    for k in tab:
        if sub in k:
            return k, tab[k]
    return None, {}


n = cases["act_kernel"]["read"]
_, fa = find(F, "act_kernel")
_, wa = find(W, "act_kernel")
cal = dict(kernel="act_kernel over 64Mi floats (dword per lane)", known_read_bytes=n, FETCH_SIZE_KB=fa.get("FETCH_SIZE"),
           known_write_bytes=n, WRITE_SIZE_KB=wa.get("WRITE_SIZE"),
           fetch_factor=round(n / (fa.get("FETCH_SIZE", 1) * 1024.0), 4), write_factor=round(n / (wa.get("WRITE_SIZE", 1) * 1024.0), 4),
           correction="bytes = FETCH_SIZE*1024*fetch_factor (gfx950 tallies 128-B requests at 64 B: factor 2, calibrated here on a known-size "
                      "stream in the same run), WRITE_SIZE*1024*write_factor")
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/pmc_traffic.py {B}, MI355X, round 6",
       "kernel_source_hash": source_hash(), "calibration": cal}
for sub, c in cases.items():
    if sub == "act_kernel":
        continue
    kname, fc = find(F, sub)
    _, wc = find(W, sub)
    if not kname or "FETCH_SIZE" not in fc or "WRITE_SIZE" not in wc:
        continue
    rd = fc["FETCH_SIZE"] * 1024.0 * cal["fetch_factor"]
    wr = wc["WRITE_SIZE"] * 1024.0 * cal["write_factor"]
    # bench.py's name of the instance (tools/stages.py): conv_wg*_kernel<W>, conv_wg_kernel<tile...>, conv_f43_kernel<...> by prefix
    key = sub + ">" if sub.endswith(("<64", "<32", "<16", "<128", "512")) else (sub.split("<")[0] if "<" in sub else sub)
    key = key.replace(", ", ",")
    if sub.startswith("conv_f43_kernel"):
        key = "conv_f43_kernel<128,128,4,16,F43>"
    out[key] = dict(kernel_symbol=kname, shape=c["shape"], workload=c.get("workload", "c4"), FETCH_SIZE_KB=fc["FETCH_SIZE"], WRITE_SIZE_KB=wc["WRITE_SIZE"],
                    hbm_read_bytes=int(rd), hbm_write_bytes=int(wr), traffic_bytes=int(rd + wr), algorithmic_bytes=c["algorithmic"],
                    traffic_over_algorithmic=round((rd + wr) / c["algorithmic"], 3))
json.dump(out, open(out_t, "w"), indent=1)
busy = {"kernel_source_hash": source_hash()}
for k, cs in S.items():
    if "GRBM_GUI_ACTIVE" in cs and cs.get("SQ_WAVE_CYCLES", 0) > 0:
        e = dict(cs)
        if cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            e["mfma_pipe_busy"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        if "SQ_WAIT_ANY" in cs:
            e["wave_parked"] = round(cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"], 4)
        busy[k[:90]] = e
busy["_note"] = (f"tools/pmc_traffic.py {B}; mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                 "wave_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
json.dump(busy, open(out_b, "w"), indent=1)
print(json.dumps({k: (v.get("traffic_over_algorithmic") if isinstance(v, dict) else v) for k, v in out.items() if k not in ("source", "calibration")}, indent=1))
print(json.dumps({k: (v.get("mfma_pipe_busy") if isinstance(v, dict) else v) for k, v in busy.items()}, indent=1))
