"""Build libsdc_hip.so for gfx950 with hipcc (in-tree, so it travels with gpurun snapshots)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsdc_hip.so")
SOURCES = ["sdc_api.hip", "sdc_conv.hip", "sdc_conv_wino.hip", "sdc_norm.hip", "sdc_attn.hip", "sdc_lablock.hip", "sdc_tablock.hip", "sdc_step.hip", "sdc_solver.hip", "sdc_grad.hip", "sdc_attn_bwd.hip", "sdc_kstar.hip", "sdc_smoke.hip", "sdc_linear.hip"]
HEADERS = ["sdc_common.h", "sdc_conv.h", "sdc_conv_wino3s.inc", "sdc_conv_wino2s.inc", "sdc_conv_pw2.inc"]      # csrc files the translation units include
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and headers: stamps the PMC records under profiles/ so that
    bench.py can tell a record collected on other kernels from a current one (no git on the GPU box)"""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(HERE, "..", "include", "sdc.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_experiments(verbose=True):
    """libsdc_hip_exp.so: the same sources with -DSDC_KERNEL_EXPERIMENTS (dispatch overrides and the result-changing debug
    modes of tools/*_probe.py, read from SDC_* environment variables).  The tools load it through _lib.use_library (SDC_LIB_PATH in tools/ and bench.py);
    never shipped or tested."""
    out = os.path.join(HERE, "libsdc_hip_exp.so")
    cmd = [HIPCC, *FLAGS, "-DSDC_KERNEL_EXPERIMENTS", "-shared", *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True, jobs=4):
    """compile the translation units (up to `jobs` hipcc processes at a time) and link libsdc_hip.so"""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(HERE, "..", "include", "sdc.h")]
    objs, todo = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            todo.append([HIPCC, *FLAGS, "-c", s, "-o", o])
        objs.append(o)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if todo:
        with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(todo)))) as ex:
            list(ex.map(run, todo))
    if force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        build_experiments()
    else:
        build(force="--force" in sys.argv)
