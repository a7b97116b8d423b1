#!/bin/bash
# build experiment variants of libsdc_hip.so: tools/exp/libsdc_exp<N>.so with -DSDC_EXP=<N> (timing only)
set -e
cd "$(dirname "$0")/.."
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -w -DSDC_EXP=${n%%_*} ${EXTRA_FLAGS} -shared \
    safediffcon_amd/csrc/sdc_api.hip safediffcon_amd/csrc/sdc_conv.hip safediffcon_amd/csrc/sdc_norm.hip \
    safediffcon_amd/csrc/sdc_attn.hip safediffcon_amd/csrc/sdc_step.hip -o tools/exp/libsdc_exp$n.so
done
