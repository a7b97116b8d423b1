for d in 0 4 0 4; do echo "DBG=$d"; SDC_WG3_DBG=$d SDC_PRECISION=4 MODES=4 timeout -k 10 100 python tools/wg_probe.py 64 1 2>&1 | grep "L0"; done
