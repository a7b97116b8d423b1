#!/usr/bin/env python3
"""tests/golden/kstar_weights.npz + kstar_rl_policy.npz: the parameters of the KSTAR surrogate's networks (the first two of each ensemble), read
from the reference's own Keras weight files -- DATA, not source: tokamak/weights/{lstm/v220505,nn,bpw}/best_model* -- with
safediffcon_amd.h5lite, so that the GPU box (which has no /root/reference) can run the rollout on the real simulator; and the actor network of the
reference's trained controller (tokamak/weights/rl/rt_control/3frame_v220505/best_model.zip), which the reference itself
evaluates in numpy.

    python oracle/make_kstar_fixture.py [/root/reference/tokamak/weights]

No outputs of the reference are stored: its simulator needs TensorFlow, which this image lacks (oracle/kstar.py header).
"""
import io
import json
import os
import sys
import zipfile

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
    # the actor of the reference's trained real-time controller (common/model_structure.py:178-204 SB2_model reads the same zip
    # with numpy): used by tests/test_kstar_host.py to close the loop around the restated simulator
    zf = zipfile.ZipFile(os.path.join(src, "rl", "rt_control", "3frame_v220505", "best_model.zip"))
    layers = json.loads(zf.read("data").decode("utf-8"))["policy_kwargs"].get("layers", [64, 64])
    params = np.load(io.BytesIO(zf.read("parameters")))
    pol = {k.replace("model/pi/", "").replace(":0", "").replace("/", "_"): params[k] for k in params.files if k.startswith("model/pi/")}
    pol["layers"] = np.array(layers)
    path = os.path.join(out, "kstar_rl_policy.npz")
    np.savez_compressed(path, **pol)
    print(f"{path}: {sorted(pol)} {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
