set -o pipefail
O=gpurun_out/r2l; mkdir -p $O
timeout -k 10 300 python bench.py --workload c2 --full-calibration --full-sample --no-cpu-baseline --extra-workloads none > $O/r2_bench_c2_full_calibration.json.log 2>$O/c2.err || echo "c2 failed"
tail -c 400 $O/r2_bench_c2_full_calibration.json.log; echo
timeout -k 10 200 python bench.py --workload c3 --no-cpu-baseline --extra-workloads none > $O/r2_bench_c3.json.log 2>$O/c3.err || echo "c3 failed"
SDC_DIST_BACKEND=gloo SDC_FORCE_DEVICE=0 timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 --no-extra --no-cpu-baseline > $O/r2_bench_c4_gloo_2proc_1gpu_rehearsal.json.log 2>$O/gloo.err || echo "gloo rehearsal failed"
tail -c 300 $O/r2_bench_c4_gloo_2proc_1gpu_rehearsal.json.log; echo
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-extra --cpu-c1-full > $O/r2_bench_c4_with_cpu_c1_full.json.log 2>$O/c1.err || echo "c1 failed"
tail -c 600 $O/r2_bench_c4_with_cpu_c1_full.json.log
