"""CPU oracle for the SafeDiffCon sampling hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain PyTorch-CPU fp32 restatement of the
reference's sampler algorithm (U-Net epsilon prediction, guidance, DDPM
posterior update, conformal quantile).  It exists so the HIP path can be
checked against *something that was itself checked against the reference*:

* ``oracle/make_goldens.py`` imports the real reference from ``/root/reference``
  (build container only) and writes small input/output fixtures to
  ``tests/golden/``.
* ``tests/test_oracle_*.py`` prove this restatement equals those fixtures.
* the ``-m gpu`` tests and ``__graft_entry__.smoke()`` then use the restatement
  as the on-box checker for the HIP kernels; ``bench.py`` times it as the
  ``cpu_baseline`` ("port").

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import from here.  The product package ``safediffcon_amd`` never does and
has no CPU fallback: it raises if ``libsdc_hip.so`` is missing.

Pinning status: 1D (Burgers) and tokamak are pinned by fixtures produced from
the unmodified reference modules.  2D (smoke) is pinned by fixtures produced
from the unmodified reference modules *plus* shims for four third-party
packages absent from this image (see ``oracle/_shims``); the rotary embedding
arithmetic comes from the ``rotary-embedding-torch`` package (unpinned in the
reference's requirements.txt) and is restated from its published algorithm ->
"parity unpinned" at that one boundary.
"""
