timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tattn or temporal" 2>&1 | tail -3
for i in 1 2; do
echo "== prev"; SDC_LIB_PATH=$PWD/tools/_ab/libsdc_prev.so timeout -k 10 120 python tools/ta_probe.py 16 64 64 5
echo "== new"; timeout -k 10 120 python tools/ta_probe.py 16 64 64 5
done
