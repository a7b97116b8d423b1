#!/bin/bash
# kernel experiments on the 3x3x3 conv (experiment build, SDC_WG3_DBG): usage: bash tools/exp_wg3.sh "<dbg values>"
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/exp; mkdir -p $O
export SDC_LIB_PATH=$PWD/safediffcon_amd/libsdc_hip_exp.so
for d in ${1:-0 8 0 8}; do
  echo "== SDC_WG3_DBG=$d"; SDC_WG3_DBG=$d MODES=4 timeout -k 10 120 python tools/wg_probe.py 64 8 2>&1 | grep "c4" | cut -c1-120
done
