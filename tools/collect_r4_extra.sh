#!/bin/bash
# round-4 records beside tools/collect_profiles.sh: the C2 / C3 bench lines (with their own roofline objects), the C3 line at
# precision 5 (F(4,3) Conv1d), rocprofv3 kernel statistics of the smoke rollout, the sustained-MFMA clock / power log.
# usage (GPU box): bash tools/collect_r4_extra.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4x; mkdir -p $O; rm -rf $O/smoke_stats
timeout -k 10 400 python bench.py --workload c2 --extra-workloads none > $O/r4_bench_c2.json.log 2> $O/c2.err || echo "c2 failed"
echo "c2 done"; tail -c 300 $O/r4_bench_c2.json.log; echo
timeout -k 10 400 python bench.py --workload c3 --extra-workloads none > $O/r4_bench_c3.json.log 2> $O/c3.err || echo "c3 failed"
echo "c3 done"; tail -c 300 $O/r4_bench_c3.json.log; echo
SDC_PRECISION=5 timeout -k 10 200 python tools/stage_report.py tokamak 128 256 > $O/r4_stage_roofline_c3_precision5.md 2>/dev/null || echo "c3 p5 failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/smoke_stats --output-format csv -- python3 tools/smoke_solver_probe.py > $O/r4_smoke_rollout_probe.log 2>&1 || echo "smoke stats failed"
find $O/smoke_stats -name "*kernel_trace.csv" -delete
cp $(find $O/smoke_stats -name "*kernel_stats.csv" | head -1) $O/r4_smoke_rollout_kernel_stats.csv 2>/dev/null
timeout -k 10 120 python tools/mfma_sustain.py 5 > $O/r4_mfma_sustain_clock_power.log 2>&1 || echo "mfma sustain failed"
echo "extra done"
