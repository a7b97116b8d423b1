for t in 0 512 0 512; do echo "== tile $t"; SDC_STEM_TILE=$t timeout -k 10 120 python tools/stem_probe.py 16; done
