#!/usr/bin/env python3
"""Instruction census of the hottest loop of every kernel of a translation unit: what a wave issues per MFMA.
usage: python tools/isa_census.py safediffcon_amd/csrc/sdc_lablock.hip [kernel-name substring]

An fp32 MFMA hides nothing the same wave issues and a wave issues one instruction per four cycles whatever its kind
(3.6), so the non-MFMA instructions per MFMA of a kernel's main loop say how much slack its matrix pipe has: `conv_pw2_kernel` 1.6,
`conv_wg3s_kernel` ~3 (16x16x4 MFMAs of half the length), `la_blk_out<64>` 9 before round 5's diet, the generic `conv_kernel` 9,
the stem's `conv_rh_kernel` 10.  Compiles with hipcc -S (no GPU needed) and reads the listing: the loop of a kernel with the most
MFMAs between a label and a backward branch to it."""
import collections
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")


def census(seg):
    c = collections.Counter()
    for l in seg:
        s = l.strip()
        if not s or s[0] in ".;" or s.endswith(":"):
            continue
        i = s.split()[0]
        if i.startswith("v_mfma"):
            c["mfma"] += 1
        elif re.match(r"v_(exp|rcp|rsq|log|sqrt|sin|cos)", i):
            c["trans"] += 1
        elif i.startswith("v_"):
            c["valu"] += 1
        elif i.startswith("ds_"):
            c["lds"] += 1
        elif i.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif i.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif i.startswith("s_barrier"):
            c["barrier"] += 1
        elif i.startswith("s_"):
            c["salu"] += 1
    return c


starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+: ", l)]
starts.append((len(lines), "end"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        dem = name
    if want and want not in dem and want not in name:
        continue
    seg = lines[a:b]
    labels = {l.split(":")[0]: i for i, l in enumerate(seg) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(seg):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    tot = census(seg)
    if not tot["mfma"]:
        continue
    short = re.sub(r"\(anonymous namespace\)::|sdcconv::|void ", "", dem).split("(")[0]
    if loops:
        x, y = max(loops, key=lambda p: (census(seg[p[0]:p[1]])["mfma"], p[1] - p[0]))     # the outermost loop holding them
        c = census(seg[x:y])
    else:
        c = tot
    other = sum(v for k, v in c.items() if k != "mfma")
    print(f"{short[:70]:70s} loop: {dict(c)}  -> {other / max(1, c['mfma']):.2f} other instructions per MFMA")
