timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv" 2>&1 | tail -2
for i in 1 2; do
echo "== prev"; SDC_LIB_PATH=$PWD/tools/_ab/libsdc_prev.so timeout -k 10 120 python tools/stem_probe.py 16
echo "== new"; timeout -k 10 120 python tools/stem_probe.py 16
done
