echo BASE; SDC_LIB_PATH=$PWD/safediffcon_amd/libsdc_base.so SDC_PRECISION=4 MODES=4 timeout -k 10 100 python tools/wg_probe.py 64 1 2>&1 | grep "c4"
echo NEW; SDC_PRECISION=4 MODES=4 timeout -k 10 100 python tools/wg_probe.py 64 1 2>&1 | grep "c4"
echo BASE; SDC_LIB_PATH=$PWD/safediffcon_amd/libsdc_base.so SDC_PRECISION=4 MODES=4 timeout -k 10 100 python tools/wg_probe.py 64 1 2>&1 | grep "c4"
echo NEW; SDC_PRECISION=4 MODES=4 timeout -k 10 100 python tools/wg_probe.py 64 1 2>&1 | grep "c4"
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "winograd_3d" 2>&1 | tail -2
