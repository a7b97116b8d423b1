"""tools only: SDC_LIB_PATH=<another build of libsdc_hip.so> selects the library for a same-box A/B (the package itself reads no
environment variable: safediffcon_amd._lib.use_library is the explicit hook).  Import before anything loads the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_p = os.environ.get("SDC_LIB_PATH")
if _p:
    from safediffcon_amd import _lib
    _lib.use_library(_p)
    print(f"[tools] library override: {_p}", file=sys.stderr)
