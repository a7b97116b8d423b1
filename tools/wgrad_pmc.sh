#!/bin/bash
# counters of the weight-gradient kernel on one shape: bash tools/wgrad_pmc.sh B Cin Cout D H W kD kH kW
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wgpmc; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/a --output-format csv -- python3 tools/wgrad_probe.py "$@" 2 > $O/a.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS -d $O/b --output-format csv -- python3 tools/wgrad_probe.py "$@" 2 > $O/b.log 2>&1 || exit 4
python3 - $O <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "wgrad" not in r["Kernel_Name"]:
                continue
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            print(k, {c: f"{x:.3g}" for c, x in v.items()})
PY
find $O -name "*kernel_trace.csv" -delete
