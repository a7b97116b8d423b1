SDC_PRECISION=4 MODES=3,4 timeout -k 10 100 python tools/wg_probe.py 64 1 2>&1 | grep "c4"
for d in 8 4; do echo "DBG=$d"; SDC_WG3_DBG=$d SDC_PRECISION=4 MODES=4 timeout -k 10 100 python tools/wg_probe.py 64 1 2>&1 | grep "L0"; done
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "winograd" 2>&1 | tail -2
