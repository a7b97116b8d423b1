#!/bin/bash
# usage: tools/run_exp.sh <tag> <variants...>   -> gpurun_out/exp_<tag>_<variant>.log (conv_probe on burgers B=256)
tag=$1; shift
for n in "$@"; do
  if [ "$n" = "base" ]; then unset SDC_LIB_PATH; else export SDC_LIB_PATH=$PWD/tools/exp/libsdc_exp$n.so; fi
  timeout -k 10 200 python tools/conv_probe.py burgers 256 64 > gpurun_out/exp_${tag}_$n.log 2>&1
  echo "== $n: $(tail -1 gpurun_out/exp_${tag}_$n.log)"
done
