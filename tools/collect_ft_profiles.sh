#!/bin/bash
# rocprofv3 kernel statistics of the fine-tuning steps and the KSTAR rollout -> gpurun_out/ftp (copied into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ftp; rm -rf $O; mkdir -p $O
export FT_NO_EAGER=1
for wl in c4 c2 c3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$wl --output-format csv -- python3 tools/ft_time.py $wl > $O/$wl.log 2>&1 || exit 2
  find $O/$wl -name "*kernel_trace.csv" -delete
  cp $(find $O/$wl -name "*kernel_stats.csv" | head -1) $O/r4_finetune_${wl}_kernel_stats.csv
  tail -1 $O/$wl.log | cut -c1-160
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kstar --output-format csv -- python3 tools/kstar_time.py 128 512 > $O/kstar.log 2>&1 || exit 3
find $O/kstar -name "*kernel_trace.csv" -delete
cp $(find $O/kstar -name "*kernel_stats.csv" | head -1) $O/r3_kstar_rollout_kernel_stats.csv
tail -2 $O/kstar.log
