#!/bin/bash
# mid-round GPU call: production-width fine-tune parity + the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3mid; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_finetune.py -m gpu -x -q -s -k production > $O/ft_wide.log 2>&1; echo "ft rc $?"; grep "measured\|passed\|failed\|Error" $O/ft_wide.log | cut -c1-400
timeout -k 10 600 python bench.py > $O/bench_c4.json.log 2> $O/bench_c4.err; echo "bench rc $?"; tail -c 3000 $O/bench_c4.err; python tools/show_bench.py $O/bench_c4.json.log 2>/dev/null | head -60
