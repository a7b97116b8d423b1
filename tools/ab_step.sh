#!/bin/bash
# same-box A/B of the whole C4 step between library builds: bash tools/ab_step.sh "base mid ''" (suffixes of safediffcon_amd/libsdc_hip_<sfx>.so;
# '' = the tree's own library).  Each build runs twice, interleaved; prints ms/step with the clock and power the bench sampled.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for sfx in ${1:-base cur}; do
    if [ "$sfx" = "cur" ]; then unset SDC_LIB_PATH; else export SDC_LIB_PATH=$PWD/safediffcon_amd/libsdc_hip_$sfx.so; fi
    timeout -k 10 200 python bench.py --steps 12 --warmup 3 --no-extra --no-cpu-baseline --cal-steps 0 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['gpu_sensors']; r=d['roofline']
print('$sfx', d['ms_per_step'], 'ms/step  wg3<64>', r['avg_launch_ms'], 'ms  clock', g['sclk_mhz_median'], 'MHz  power', g['power_w_mean'], 'W')"
  done
done
