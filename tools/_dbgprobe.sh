timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -k "winograd" 2>&1 | grep -a "measured\|passed\|failed\|Error\|error" | tail -12
for d in 0 3 1 2; do echo "DBG=$d"; SDC_WG2_DBG=$d MODES=3 timeout -k 10 120 python tools/wg_probe.py 16 256 2>&1 | grep -a "c4 L2 256->256\|c4 L0 64->64 "; done
MODES=2,3 timeout -k 10 120 python tools/wg_probe.py 16 256
