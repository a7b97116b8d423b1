#!/bin/bash
# one GPU call: PMC passes (stamped with the kernel-source hash), then the round's bench line (which reads them), rocprofv3 kernel
# stats and the stage tables -> gpurun_out/r6p (copied into profiles/ afterwards).  usage (GPU box): bash tools/collect_profiles.sh   (round 6: the bench writes its report to a side file)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6p; mkdir -p $O; rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout -k 10 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 tools/pmc_traffic.py 64 $O/pmc_cases.json > $O/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 tools/pmc_traffic.py 64 $O/pmc_cases.json > $O/pmc_write.log 2>&1 || exit 4
timeout -k 10 250 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq --output-format csv -- python3 tools/pmc_traffic.py 64 $O/pmc_cases.json > $O/pmc_sq.log 2>&1 || exit 5
python3 tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_cases.json $O/r6_pmc_traffic.json $O/r6_pmc_mfma_busy.json > $O/pmc_to_json.log || exit 6
cp $O/r6_pmc_traffic.json $O/r6_pmc_mfma_busy.json profiles/
find $O -name "*kernel_trace.csv" -delete
echo "pmc done"
# the default run (what the driver runs: one compact headline on stdout, the report in the side file), then every extra
( time timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --extra-file $O/r6_bench_c4_extra.json > $O/r6_bench_c4.json.log 2> $O/bench_c4.err ) 2> $O/bench_c4.time || exit 1
echo "bench c4 done"; wc -c $O/r6_bench_c4.json.log; tail -3 $O/bench_c4.time
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 --all-extras --extra-file $O/r6_bench_c4_all_extras_extra.json > $O/r6_bench_c4_all_extras.json.log 2> $O/bench_c4_all.err || exit 13
echo "bench c4 --all-extras done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --cal-steps 0 > $O/stats.log 2>&1 || exit 2   # (--cal-steps 0: no calibration-batch launches of the same kernels at B = 25 in the averages)
find $O/stats -name "*kernel_trace.csv" -delete
echo "kernel stats done"
timeout -k 10 200 python tools/stage_report.py smoke 64 > $O/r6_stage_roofline_c4.md 2>$O/stage_c4.err || exit 7
timeout -k 10 100 python tools/stage_report.py burgers 256 > $O/r6_stage_roofline_c2.md 2>/dev/null || exit 8
timeout -k 10 100 python tools/stage_report.py tokamak 128 256 > $O/r6_stage_roofline_c3.md 2>/dev/null || exit 9
# the widths the reference ships besides the BASELINE ones
timeout -k 10 150 python tools/stage_report.py burgers 256 128 > $O/r6_stage_roofline_c2_turbo.md 2>/dev/null || exit 10
timeout -k 10 100 python tools/stage_report.py tokamak 128 128 > $O/r6_stage_roofline_c3_turbo.md 2>/dev/null || exit 11
timeout -k 10 100 python tools/stage_report.py tokamak 128 64 > $O/r6_stage_roofline_c3_small.md 2>/dev/null || exit 12
echo "stage reports done"
