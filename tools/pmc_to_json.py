#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE / SQ passes of tools/pmc_traffic.py -> profiles/r2_pmc_traffic.json + r2_pmc_mfma_busy.json.
usage: python tools/pmc_to_json.py <fetch_dir> <write_dir> <sq_dir> <workload> <batch> <out_traffic.json> <out_busy.json>"""
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, sq_dir, wl, B, out_t, out_b = sys.argv[1:8]
B = int(B)


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


n = 64 * 1024 * 1024 * 4
_, fa = find(F, "act_kernel")
_, wa = find(W, "act_kernel")
cal = dict(kernel="act_kernel over 64Mi floats (dword per lane)", known_read_bytes=n, FETCH_SIZE_KB=fa.get("FETCH_SIZE"),
           known_write_bytes=n, WRITE_SIZE_KB=wa.get("WRITE_SIZE"),
           fetch_factor=round(n / (fa.get("FETCH_SIZE", 1) * 1024.0), 4), write_factor=round(n / (wa.get("WRITE_SIZE", 1) * 1024.0), 4),
           correction="bytes = FETCH_SIZE*1024*fetch_factor (gfx950 tallies 128-B requests at 64 B: factor 2, calibrated here on a known-size "
                      "stream in the same run), WRITE_SIZE*1024*write_factor")
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/pmc_traffic.py {wl} {B}, MI355X, round 2",
       "calibration": cal}
kname, fc = find(F, "conv_wg")
_, wc = find(W, "conv_wg")
if kname:
    if wl == "c4":
        nin = B * 64 * 32 * 64 * 64
        nw = 27 * 64 * 64
        shape = f"64->64 3x3x3 at ({B},64,32,64,64) + GroupNorm statistics in the epilogue"
    else:
        nin = B * 64 * 16 * 128
        nw = 9 * 64 * 64
        shape = f"64->64 3x3 at ({B},64,16,128)"
    alg = 4 * (nin + nin + nw)
    rd = fc["FETCH_SIZE"] * 1024.0 * cal["fetch_factor"]
    wr = wc["WRITE_SIZE"] * 1024.0 * cal["write_factor"]
    # bench.py's name of the instance: template name + first argument (the row width)
    key = kname.split("<")[0] + "<" + kname.split("<")[1].split(",")[0].split(">")[0] + ">" if ("wg2" in kname or "wg3" in kname) else kname
    out[key] = dict(kernel_symbol=kname, shape=shape, workload=wl, FETCH_SIZE_KB=fc["FETCH_SIZE"], WRITE_SIZE_KB=wc["WRITE_SIZE"],
                    hbm_read_bytes=int(rd), hbm_write_bytes=int(wr), traffic_bytes=int(rd + wr), algorithmic_bytes=alg,
                    traffic_over_algorithmic=round((rd + wr) / alg, 3))
json.dump(out, open(out_t, "w"), indent=1)
busy = {}
for k, cs in S.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
        e = dict(cs)
        e["mfma_pipe_busy"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        if "SQ_WAIT_ANY" in cs and "SQ_WAVE_CYCLES" in cs:
            e["wave_parked"] = round(cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"], 4)
        busy[k[:90]] = e
busy["_note"] = (f"tools/pmc_traffic.py {wl} {B}; mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                 "wave_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
json.dump(busy, open(out_b, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "source"}, indent=1)[:1500])
print(json.dumps({k: (v.get("mfma_pipe_busy") if isinstance(v, dict) else v) for k, v in busy.items()}, indent=1))
