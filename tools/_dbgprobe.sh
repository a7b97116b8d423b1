timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "softmax_attention" 2>&1 | tail -3
timeout -k 10 300 python tools/stage_report.py smoke 16 2>/dev/null | grep -a "softmax attention\|U-Net forward"
