"""RCCL itself, as far as one GPU allows: a world of ONE rank on the "nccl" backend (= RCCL on ROCm) -- communicator creation
bound to the device (device_id, as bench.py does), all-reduce / all-gather / barrier on device tensors, and the sharded conformal
quantile (safediffcon_amd.conformal.weighted_quantile) through it.  Two ranks on one device are refused by RCCL, so the
multi-rank exchange is covered by the gloo world-2 tests and the driver's 8-GPU tier; this test is what can be said about the
RCCL code path from a one-GPU box: the library loads, the calls the sampler makes are accepted, the result equals the local one."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    from safediffcon_amd import conformal
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    assert int(one.item()) == 1
    g = torch.Generator().manual_seed(5)
    s = torch.rand(200, generator=g).to(dev)
    w = (torch.rand(200, generator=g) + 0.1).to(dev)
    gathered = conformal.all_gather_1d(s)
    assert gathered.is_cuda and torch.equal(gathered, s)
    for smoke in (False, True):
        q_dist, nw = conformal.weighted_quantile(s, w, 0.9, smoke=smoke)
        q_loc = conformal.calculate_quantile(conformal.normalize_weights(w, smoke) * s, 0.9, smoke)
        assert float(q_dist) == float(q_loc), (float(q_dist), float(q_loc))
    dist.barrier()
    torch.cuda.synchronize()
    print("RCCL world-1 ok", dist.get_backend(), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
    dist.destroy_process_group()
""") % ROOT


def test_rccl_world_of_one_runs_the_samplers_collectives():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout[-400:], r.stderr[-800:])
    assert r.returncode == 0 and "RCCL world-1 ok" in r.stdout
