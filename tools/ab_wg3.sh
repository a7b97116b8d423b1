#!/bin/bash
# same-box A/B of the 3x3x3 conv shapes: safediffcon_amd/libsdc_hip_base.so (previous kernel) vs the current build, plus the
# parity tests of the kernels touched.  usage (GPU box): bash tools/ab_wg3.sh
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab1; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_kernels.py -m gpu -x -q -s -k "grad or wgrad or backward or winograd3 or precision4 or wg3 or sweep" > $O/tests.log 2>&1; echo "tests rc $?"; grep -c "measured" $O/tests.log; tail -5 $O/tests.log
SDC_LIB_PATH=$PWD/safediffcon_amd/libsdc_hip_base.so MODES=4 timeout -k 10 200 python tools/wg_probe.py 64 64 > $O/probe_base.log 2>&1; echo "base rc $?"
MODES=4 timeout -k 10 200 python tools/wg_probe.py 64 64 > $O/probe_new.log 2>&1; echo "new rc $?"
SDC_LIB_PATH=$PWD/safediffcon_amd/libsdc_hip_base.so MODES=4 timeout -k 10 200 python tools/wg_probe.py 64 64 > $O/probe_base2.log 2>&1
MODES=4 timeout -k 10 200 python tools/wg_probe.py 64 64 > $O/probe_new2.log 2>&1
paste -d'\n' $O/probe_base.log $O/probe_new.log | cut -c1-140 | head -16
