#!/bin/bash
# rocprofv3 kernel statistics of one fine-tuning step workload: bash tools/ft_prof.sh c2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ftprof_$1; rm -rf $O; mkdir -p $O
export FT_NO_EAGER=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 tools/ft_time.py $1 > $O/run.log 2>&1
find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(__import__("os").environ.get("FT_TOP", "22"))]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d}  {r['Name'][:110]}")
PY
tail -1 $O/run.log | cut -c100-330
