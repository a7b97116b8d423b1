timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -k "winograd or groupnorm_statistics" 2>&1 | grep -a "passed\|failed\|Error\|error" | tail -5
for i in 1 2; do
echo "== prev"; SDC_LIB_PATH=$PWD/tools/_ab/libsdc_prev.so MODES=3 timeout -k 10 120 python tools/wg_probe.py 16 256 2>&1 | grep -a "c4\|c2"
echo "== new"; MODES=3 timeout -k 10 120 python tools/wg_probe.py 16 256 2>&1 | grep -a "c4\|c2"
done
