#!/bin/bash
# round 6, second collection: the 1-D configs as the main workload, and one complete 1000-step C4 sample + its score check
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6x; mkdir -p $O
timeout -k 10 300 python bench.py --workload c2 --steps 20 --warmup 5 --extra-workloads "" --extra-file $O/r6_bench_c2_extra.json > $O/r6_bench_c2.json.log 2> $O/c2.err || exit 1
echo "c2 done"
timeout -k 10 300 python bench.py --workload c3 --steps 20 --warmup 5 --extra-workloads "" --extra-file $O/r6_bench_c3_extra.json > $O/r6_bench_c3.json.log 2> $O/c3.err || exit 2
echo "c3 done"
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline --full-sample --extra-file $O/r6_bench_c4_full_sample_extra.json > $O/r6_bench_c4_full_sample.json.log 2> $O/full.err || exit 3
echo "full sample done"
