#!/usr/bin/env python3
"""tests/golden/kstar_weights.npz: the parameters of the KSTAR surrogate's networks (the first two of each ensemble), read
from the reference's own Keras weight files -- DATA, not source: tokamak/weights/{lstm/v220505,nn,bpw}/best_model* -- with
safediffcon_amd.h5lite, so that the GPU box (which has no /root/reference) can run the rollout on the real simulator.

    python oracle/make_kstar_fixture.py [/root/reference/tokamak/weights]

No outputs of the reference are stored: its simulator needs TensorFlow, which this image lacks (oracle/kstar.py header).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd import kstar  # noqa: E402


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tokamak/weights"
    out = os.environ.get("SDC_GOLDEN_OUT", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    flat = kstar.flatten_weights(kstar.load_weights(src, n_models=2))
    path = os.path.join(out, "kstar_weights.npz")
    np.savez_compressed(path, **flat)
    print(f"{path}: {len(flat)} arrays, {sum(np.asarray(v).size for v in flat.values())} values, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
