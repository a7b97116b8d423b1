set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcw3; mkdir -p $O; rm -rf $O/p1 $O/p2 $O/p3
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- python3 tools/one_wg3.py > $O/p1.log 2>&1 || exit 3
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL -d $O/p2 --output-format csv -- python3 tools/one_wg3.py > $O/p2.log 2>&1 || exit 4
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES -d $O/p3 --output-format csv -- python3 tools/one_wg3.py > $O/p3.log 2>&1 || exit 5
python3 - <<'PY'
import csv, glob, collections
for p in ("p1","p2","p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/pmcw3/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            n[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "conv_wg" in k:
            print(p, k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
