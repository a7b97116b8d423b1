#!/usr/bin/env python3
"""bench.py -- sampled control trajectories / second of the SafeDiffCon DDPM hot path on MI355X.

Default workload = the configuration BASELINE.json's north star quotes its targets on ("C4", configs[3]): 2D smoke,
Unet3D_with_Conv3D dim=64 (1,2,4), state (B,32,7,64,64), B=64 per GPU, 1000-step DDPM, closed-form safety guidance on,
conformal quantile on (Q comes from the HIP conformal-score kernel on a calibration shard + all-gather + rank select and
feeds the guidance).  At N GPUs the batch axis is sharded, 64 trajectories per GPU (N=8 is BASELINE configs[4], "C5":
B=512, calibration n=8x25, alpha=0.04).  `--workload c2` (1D Burgers Unet2D dim 64, B=256) and `--workload c3` (tokamak
Unet1D dim 256, B=128) run the other single-GPU configs through the same harness; at N=1 a few steps of each are also
reported under `extra`.

A "step" is ONE denoising step of the whole batch: U-Net epsilon prediction + guidance reduction + fused posterior
update + step counter, replayed from one captured hipGraph.  A trajectory costs exactly `timesteps`=1000 such steps, so
value = global_batch / (1000 * seconds_per_step).

`python bench.py --gpus N` with N > 1 launches itself: the parent touches no GPU and starts N fresh worker processes
through `python -m torch.distributed.run` (one rank per GPU, RCCL); under an external torch.distributed.run (WORLD_SIZE
set) it is a worker directly.  Rank 0 prints the one JSON line.

`roofline`: the dominant kernel (the fp32-MFMA Winograd conv), HIP events on the launch stream.  `achieved` / `frac` are
the MFMA FLOPs the kernel actually issues / second against the 157.3 TFLOP/s fp32 matrix peak (never above 1);
`effective_tflops` is the direct-form (algorithmic) rate the same launch stands for.  `roofline.stages` carries the
stage fractions the north star names: GroupNorm-apply GB/s against 8 TB/s, the Conv3d+GN+SiLU block as a whole, and the
temporal-attention kernels' TFLOP/s against the fp32 matrix peak.  `cpu_baseline` = the CPU oracle (oracle/, kind
"port") on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
import types

# Same-box A/B of two BUILDS of the library (tools/ab_step.sh, tools/_r5_ab_*.sh): SDC_LIB_PATH=<libsdc_hip_exp.so | a previous commit's
# build> is handed to the package's explicit hook before anything loads the library, and the line reports it (`library`).  The
# package itself reads no environment variable; without the variable this is the in-tree library the tests load.
LIBRARY_OVERRIDE = os.environ.get("SDC_LIB_PATH") or None
if LIBRARY_OVERRIDE:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from safediffcon_amd import _lib as _sdc_lib_sel
    _sdc_lib_sel.use_library(LIBRARY_OVERRIDE)

# RCCL / CUDA-tensor sharing between the ranks of one node needs dmabuf IPC on this driver (set before anything touches HIP;
# the GPU boxes export it already, a bare `torchrun bench.py` on another machine may not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

T_DDPM = 1000
DEFAULT_B = {"c2": 256, "c3": 128, "c4": 64}
HEADLINE_MAX_BYTES = 6144                 # the last stdout line stays a headline the driver parses
PMC_FILES = ("r6_pmc_traffic.json",)      # stamped with the kernel-source hash they were collected on (tools/pmc_to_json.py)


# --------------------------------------------------------------------------- launcher (N > 1 without a torchrun parent)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(n, argv):
    """Parent of `bench.py --gpus N`: no GPU call here (a process that has initialised HIP must not spawn the ranks)."""
    if "SDC_FORCE_DEVICE" not in os.environ and "--selftest-launcher" not in argv:
        # the count comes from a throwaway child: torch.cuda.device_count() falls back to hipGetDeviceCount (which initialises
        # the HIP runtime in THIS process) whenever amdsmi discovery is unavailable.  No count -> let each rank report it.
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                               text=True, timeout=300)
            ndev = int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError, subprocess.SubprocessError):
            ndev = None
        if ndev is not None and ndev < n:
            raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible on this node")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


# --------------------------------------------------------------------------- workloads
def shard_config(name, B, world):
    """what the N-GPU line is quoted on (pure arithmetic, no GPU): per-GPU batch B, global batch, the calibration set split
    over the ranks.  N = 8 of the smoke workload is BASELINE configs[4] ("C5": B = 512, calibration n = 8 x 25)."""
    n_tot = 200 if name == "c4" else 1000                  # calibration set: 8x25 (2d/inference_2d.py) / 4x250 (1D) / 1x1000 (tokamak)
    n_cal = max(8, n_tot // world)
    tag = {"c2": "C2", "c3": "C3", "c4": "C5" if world == 8 else "C4"}[name]
    return dict(tag=tag, batch_per_gpu=B, global_batch=B * world, n_cal_per_rank=n_cal, n_cal=n_cal * world,
                alpha={"c2": 0.98, "c3": 0.9, "c4": 0.04}[name])


def workload(name, dim, B, dev, rank, world, precision=4, cal_steps=5, split_small=False):
    """-> dict(desc, gd, prep() -> _Loop, conformal=dict(Q, n_cal, alpha, ms), calib(...) -> calibration-mode _Loop)
    split_small: the net's small-batch plan (net.split_small_grids, the `*_shard8_split` extras)"""
    import torch
    import safediffcon_amd as sdc
    from safediffcon_amd import conformal
    torch.manual_seed(0)                                   # weights: default nn-style init under seed 0
    g1 = torch.Generator().manual_seed(1 + rank)           # conditions: seed 1 (+rank: every shard differs)
    cfgN = shard_config(name, B, world)
    n_cal = cfgN["n_cal_per_rank"]

    def quantile(kind, pred, truth, gpar, alpha, **kw):
        """score kernel on this rank's calibration shard -> all-gather (RCCL) -> normalise / sort / rank select"""
        kw2 = {k: v for k, v in kw.items() if k != "smoke"}
        # untimed first pass: module load of the score kernel, the communicator's first collective, torch's sort / cumsum kernels
        s_, w_ = conformal.scores_and_weights(kind, pred, truth, gpar, **kw2)
        conformal.weighted_quantile(s_, w_, alpha, smoke=kw.get("smoke", False))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s, w = conformal.scores_and_weights(kind, pred, truth, gpar, **kw2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        Q = float(conformal.weighted_quantile(s, w, alpha, smoke=kw.get("smoke", False))[0].item())
        t2 = time.perf_counter()
        return dict(Q=Q, n_cal_per_rank=n_cal, n_cal=n_cal * world, alpha=alpha, score_kernel_ms=round((t1 - t0) * 1e3, 3),
                    allgather_quantile_ms=round((t2 - t1) * 1e3, 3))

    def sampled_pred(calib, cal_B, like):
        """this rank's calibration shard drawn by the calibration-mode sampler itself (a bounded sample: `cal_steps` reverse
        steps per batch, like extra.calibration) -- the tensors the score kernel then reads are sampler output, not synthetic"""
        Bc = min(cal_B, n_cal)
        if cal_steps <= 0:      # profiling runs (--cal-steps 0): no calibration-batch launches at all, Q from plain noise
            return torch.randn(tuple(like.shape), generator=torch.Generator().manual_seed(5)).to(dev)
        out = []
        for _ in range((n_cal + Bc - 1) // Bc):
            Sc = calib(Bc)
            Sc.init()
            for _ in range(cal_steps):
                Sc.step()
            out.append(Sc.x.clone())
            Sc.close()
        return torch.cat(out)[:n_cal]

    if name == "c2":
        dim = dim or 64
        net = sdc.Unet2D(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, resnet_block_groups=1).to(dev)
        gd = sdc.GaussianDiffusionBurgers(net, seq_length=(16, 128), timesteps=T_DDPM, temporal=True, use_conv2d=True,
                                          is_condition_u0=True, is_condition_uT=True, condition_idx=10,
                                          train_on_padded_locations=False).to(dev)
        u0 = (0.1 * torch.randn(B, 128, generator=g1)).clamp(-0.1, 0.3).to(dev)
        uT = (0.1 * torch.randn(B, 128, generator=g1)).clamp(-0.1, 0.3).to(dev)
        truth = (0.1 * torch.randn(n_cal, 3, 16, 128, generator=g1)).to(dev)
        guid = sdc.BurgersGuidance(0.0, 500.0, 0.8, use_max_safety=True)         # 1D/configs/inference_config.py:122; Q set below
        prep = lambda: gd.sample(batch_size=B, clip_denoised=True, u_init=u0, u_final=uT, guidance_u0=True,   # noqa: E731
                                 nablaJ=guid, J_scheduler=None, enable_grad=False, _prepare=True)

        def calib(Bc):      # 1D/inference/conformal.py:53-65: unguided, w_groundtruth imposed, two noise draws per step
            wgt = (0.05 * torch.randn(Bc, 16, 128, generator=g1)).to(dev)
            return gd.sample(batch_size=Bc, clip_denoised=True, guidance_u0=False, u_init=u0[:Bc], u_final=uT[:Bc],
                             w_groundtruth=wgt, nablaJ=None, enable_grad=False, _prepare=True)
        cal_B, cal_batches = 250, 4
        net.precision = precision
        net.split_small_grids = bool(split_small)
        cf = quantile("burgers", sampled_pred(calib, cal_B, truth), truth, [500.0, 0.8 ** 2, 0.0, 10.0], 0.98)
        guid.Q = cf["Q"]
        desc = f"C2: 1D Burgers Unet2D dim={dim} (1,2,4,8) state (B,3,16,128), guided 1000-step DDPM, conformal quantile on"
    elif name == "c3":
        dim = dim or 256
        net = sdc.Unet1D(dim=dim, dim_mults=(1, 2, 4, 8), channels=12, resnet_block_groups=1).to(dev)
        gd = sdc.GaussianDiffusionTokamak(net, seq_length=128, nt=122, timesteps=T_DDPM).to(dev)
        u0 = (0.4 + 0.4 * torch.rand(B, 3, generator=g1)).to(dev)
        uT = (0.6 + 0.02 * torch.randn(B, 2, 122, generator=g1).cumsum(-1)).clamp(0.3, 0.9).to(dev)
        target = (uT.new_zeros(B, 3, 122))
        target[:, 0], target[:, 2] = uT[:, 0] * 2, uT[:, 1] * 2
        truth = (0.5 + 0.3 * torch.randn(n_cal, 12, 128, generator=g1)).to(dev)
        tgt_cal = (1.0 + 0.3 * torch.randn(n_cal, 3, 122, generator=g1)).to(dev)
        # tokamak/scripts/finetune.sh:13, configs/inference_config.py:25,107-111 (the pipeline always uses Q = 0.0)
        guid = sdc.TokamakGuidance(target, 122, w_obj=0.0, w_safe=1.0, guidance_scaler=0.01, Q=0.0, safety_threshold=4.98)
        prep = lambda: gd.sample(batch_size=B, clip_denoised=True, u_init=u0, u_final=uT, guidance_u0=True,   # noqa: E731
                                 nablaJ=guid, J_scheduler=None, enable_grad=False, _prepare=True)

        def calib(Bc):      # tokamak/inference/conformal.py:62-74 (DDPM + w_groundtruth hits the reference's IndexError: unguided only)
            return gd.sample(batch_size=Bc, clip_denoised=True, guidance_u0=False, u_init=u0[:Bc], u_final=uT[:Bc],
                             nablaJ=None, enable_grad=False, _prepare=True)
        cal_B, cal_batches = 125, 8
        net.precision = precision
        net.split_small_grids = bool(split_small)
        cf = quantile("tokamak", sampled_pred(calib, cal_B, truth), truth, [0.0, 1.0, 0.01, 4.98, 0.0], 0.9, target=tgt_cal)
        desc = f"C3: tokamak Unet1D dim={dim} (1,2,4,8) state (B,12,128), guided 1000-step DDPM"
    elif name == "c4":
        dim = dim or 64
        net = sdc.Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2, 4), channels=7).to(dev)
        gd = sdc.GaussianDiffusionSmoke(net, image_size=64, frames=32, timesteps=T_DDPM, loss_type="l2",
                                        standard_fixed_ratio=100.0).to(dev)
        init = (0.5 * torch.rand(B, 64, 64, generator=g1)).to(dev)
        guid = sdc.SmokeGuidance(0.0, w_safe=0.9, safe_bound=0.1)                # 2d/scripts/posttrain.sh:20-21; Q set below
        prep = lambda: gd.sample(batch_size=B, design_fn=guid, enable_grad=False, init=init, _prepare=True)  # noqa: E731

        def calib(Bc):      # 2d/inference_2d.py:129-134: unguided, frame-0 density and the two control channels imposed
            control = (0.3 * torch.randn(Bc, 32, 2, 64, 64, generator=g1)).to(dev)
            return gd.sample(batch_size=Bc, design_fn=None, enable_grad=False, init=init[:Bc], control=control, _prepare=True)
        cal_B, cal_batches = 25, 8
        net.precision = precision
        net.split_small_grids = bool(split_small)
        truth = (0.3 * torch.randn(n_cal, 32, 7, 64, 64, generator=g1)).to(dev)
        cf = quantile("smoke", sampled_pred(calib, cal_B, truth), truth, [0.9, 0.1, 0.0, 100.0], 0.04, smoke=True)
        del truth
        guid.Q = cf["Q"]
        desc = (f"{cfgN['tag']}: 2D smoke Unet3D_with_Conv3D dim={dim} (1,2,4) state (B,32,7,64,64), B={B} per GPU, guided 1000-step DDPM, "
                f"conformal quantile on")
    else:
        raise SystemExit(f"unknown workload {name}")
    cf["source"] = (f"this rank's {n_cal} calibration trajectories drawn by the calibration-mode sampler ({cal_steps} reverse steps each: "
                    f"a bounded sample), scored against synthetic ground truth (no dataset ships with the reference)")
    return dict(desc=desc, gd=gd, prep=prep, conformal=cf, calib=calib, cal_B=cal_B, cal_batches=cal_batches)


class GpuSensors:
    """Shader clock and socket power of the GPU under test, sampled from the amdgpu hwmon files every 50 ms by a thread while
    the timed region runs (no child process, no HIP call: plain sysfs reads).  VERDICT r3: a sustained-clock figure is evidence
    only with the clock log beside it."""

    def __init__(self, dev_index=0):
        import glob
        import torch
        self.dir = None
        try:
            p = torch.cuda.get_device_properties(dev_index)
            slot = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            hw = glob.glob(f"/sys/bus/pci/devices/{slot}/hwmon/hwmon*")
            if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
                self.dir, self.slot = hw[0], slot
        except Exception:                                  # noqa: BLE001  (sensors are optional)
            pass
        self.samples, self._stop, self._th = [], False, None

    def read(self):
        if self.dir is None:
            return None
        try:
            with open(os.path.join(self.dir, "freq1_input")) as f:
                mhz = int(f.read()) / 1e6
            with open(os.path.join(self.dir, "power1_input")) as f:
                w = int(f.read()) / 1e6
            return mhz, w
        except (OSError, ValueError):
            return None

    def start(self):
        import threading
        if self.dir is None:
            return
        self.samples, self._stop = [], False

        def loop():
            while not self._stop:
                r = self.read()
                if r:
                    self.samples.append(r)
                time.sleep(0.05)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        if self._th is None:
            return None
        self._stop = True
        self._th.join()
        self._th = None
        if not self.samples:
            return None
        clk = sorted(s[0] for s in self.samples)
        pw = [s[1] for s in self.samples]
        return {"source": f"hwmon freq1_input / power1_input of {self.slot}, sampled every 50 ms over warm-up + timed region",
                "samples": len(clk), "sclk_mhz_min": round(clk[0]), "sclk_mhz_median": round(clk[len(clk) // 2]),
                "sclk_mhz_max": round(clk[-1]), "sclk_mhz_mean": round(sum(clk) / len(clk)),
                "power_w_mean": round(sum(pw) / len(pw)), "power_w_max": round(max(pw))}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_step_seconds(name, batch, steps, dim, device):
    """seconds per guided p_sample step (U-Net + autograd guidance + posterior update) of the oracle's functional restatement,
    torch fp32 on `device` (cpu: the host cores; cuda: PyTorch-ROCm eager kernels -- the "hipified reference" strawman)"""
    import torch
    from oracle import nets as onets, samplers as osam, schedules as osched
    from oracle.detweights import det_params, det_tensor
    import safediffcon_amd as sdc
    if name == "c2":
        net, fwd, shape = sdc.Unet2D(dim=dim or 64, channels=3, resnet_block_groups=1), onets.unet_burgers, (3, 16, 128)
        nablaJ, sched, k = osam.burgers_guidance(0.01, 500.0, 0.8), "cosine", 1.0
    elif name == "c3":
        net, fwd, shape = sdc.Unet1D(dim=dim or 256, channels=12, resnet_block_groups=1), onets.unet_tokamak, (12, 128)
        nablaJ, sched, k = osam.tokamak_guidance(torch.ones(batch, 3, 122, device=device), 122, 0.0, 4.98, 0.0, 1.0, 0.01), "cosine", 1.0
    else:
        net, fwd, shape = sdc.Unet3D_with_Conv3D(dim=dim or 64, dim_mults=(1, 2, 4), channels=7), onets.unet_smoke, (32, 7, 64, 64)
        nablaJ, sched, k = osam.smoke_guidance(0.01, 0.9, 0.1), "sigmoid", 100.0
    dim = net.dim
    kw = dict(dim=dim) if name != "c4" else dict(dim=dim, dim_mults=(1, 2, 4))
    spec = [(kk, tuple(v.shape)) for kk, v in net.state_dict().items()]
    P = {kk: v.to(device) for kk, v in det_params(spec, 0).items()}
    tabs = {kk: (v.to(device) if torch.is_tensor(v) else v) for kk, v in osched.make_tables(sched, T_DDPM).items()}
    x = det_tensor((batch, *shape), 5).to(device)
    sync = torch.cuda.synchronize if str(device).startswith("cuda") else (lambda: None)
    ts = []
    with torch.no_grad():
        for i in range(steps + 1):
            t = T_DDPM - 1 - i
            sync()
            t0 = time.perf_counter()
            eps = fwd(P, x, torch.full((batch,), t, dtype=torch.long, device=device), **kw)
            g = nablaJ(osam._x0_from_eps(tabs, x, t, eps))
            x, _ = osam._posterior_step(tabs, x, t, eps, g, k, True, torch.randn_like(x))
            sync()
            ts.append(time.perf_counter() - t0)
    return sum(ts[1:]) / steps


def physical_cores():
    """(physical cores of the host, logical CPUs this process may run on)"""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    return (len(cores) or allowed), allowed


def cpu_baseline(name, batch, steps, dim, all_cores=False):
    """the CPU oracle's guided p_sample step on the host cores with the 16 threads of the CPU share a one-GPU box gets (`value`:
    what a user of that box can use).  `all_cores` (--cpu-all-cores) repeats the sample with one thread per physical core of the
    host: on the 128-core boxes of this pool that leg is ten times SLOWER than 16 threads (B = 1 oversubscribes) and
    costs ~100 s, so it is not part of the default run."""
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    s16 = _oracle_step_seconds(name, batch, steps, dim, "cpu")
    n16 = torch.get_num_threads()
    out = dict(value=round(batch / (T_DDPM * s16), 7), unit="trajectories/s", cores=n16, kind="port",
               cpu_model=cpu_model(), logical_cpus_visible=os.cpu_count(),
               sample=f"{steps} guided p_sample steps (after 1 warm-up) at B={batch} of the same workload, "
                      f"{s16 * 1e3:.0f} ms/step, extrapolated x{T_DDPM} steps per trajectory")
    if all_cores:
        phys, allowed = physical_cores()
        nall = max(1, min(phys, allowed))
        if nall != n16:
            torch.set_num_threads(nall)
            sall = _oracle_step_seconds(name, batch, 1, dim, "cpu")
            out["all_physical_cores"] = dict(value=batch / (T_DDPM * sall), unit="trajectories/s", cores=torch.get_num_threads(),
                                             physical_cores_of_host=phys, logical_cpus_allowed=allowed,
                                             ms_per_step=round(sall * 1e3, 1), sample="one step of the same sample")
            torch.set_num_threads(n16)
    return out


def pmc_traffic(kernel, wl):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; collected
    separately with rocprofv3 --pmc by tools/pmc_traffic.py -- bench.py cannot run the profiler on itself).  The file carries
    the hash of the kernel sources it was collected on (safediffcon_amd.build.source_hash): a record taken on other sources
    is stale and is reported as traffic = null."""
    from safediffcon_amd.build import source_hash
    for fn in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as fh:
                doc = json.load(fh)
        except (OSError, ValueError):
            continue
        ent = doc.get(kernel, {})
        if doc.get("kernel_source_hash") != source_hash():
            return None, ent.get("algorithmic_bytes"), f"{fn}: STALE (collected on kernel sources {doc.get('kernel_source_hash')}, running {source_hash()})"
        if ent.get("workload", "c2") == wl and ent.get("traffic_bytes"):
            return ent["traffic_bytes"], ent.get("algorithmic_bytes"), fn
    return None, None, None


def build_roofline(S, lib, stream, step_ms, wl):
    """stage / kernel timings of the plan with HIP events on the launch stream -> the `roofline` object"""
    import stages as stg
    stages, kernels = stg.time_plan(S.ent["plan"], lib, stream)
    convs = {k: v for k, v in kernels.items() if k.startswith("conv")}
    name, g = max(convs.items(), key=lambda kv: kv[1]["ms"])
    sec = g["ms"] * 1e-3
    issued, eff = g["issued"] / sec / 1e12, g["flops"] / sec / 1e12
    traffic, alg_bytes, src = pmc_traffic(name, wl)
    roof = dict(bound="mfma", kernel=name, achieved=round(issued, 2), peak=stg.PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                frac=round(issued / stg.PEAK_F32_MFMA_TFLOPS, 4), traffic=traffic, traffic_source=src,
                algorithmic_bytes_per_launch=alg_bytes, effective_tflops=round(eff, 2),
                mfma_share_of_direct_form=round(g["issued"] / g["flops"], 4),
                launches_per_step=g["launches"], avg_launch_ms=round(g["ms"] / g["launches"], 4),
                share_of_step=round(g["ms"] / step_ms, 3))
    roof["notes"] = dict(
        flops="achieved = MFMA FLOPs the kernel executes per second (Winograd issues 2/3 [F(2,3) along W], 4/9 [F(2x2,3x3)] or 8/27 "
              "[F(2x2x2,3x3x3)] of the direct-form multiply-adds); effective_tflops = direct-form (algorithmic) FLOPs per second",
        # bare v_mfma_f32_32x32x2 loop held for seconds on an MI355X of this pool (tools/mfma_sustain.py,
        # profiles/r4_mfma_sustain_clock_power.log): the pipe holds the datasheet rate with <= 2 waves per SIMD issuing
        mfma_loop_measured={"one_or_two_waves_per_simd": 154.5, "three_or_more_waves_per_simd": 123.5},
        conv_gn_silu_block="conv (GN statistics in its epilogue) + finalize + apply/SiLU; fp32-MFMA bound by >= 40x (SURVEY 8d), "
                           "so its HBM fraction is small by construction; the HBM-bound piece is gn_apply_silu")

    def mfma_stage(v):
        s = v["ms"] * 1e-3
        return dict(bound="mfma", achieved=round(v["issued"] / s / 1e12, 2), unit="TFLOP/s",
                    frac=round(v["issued"] / s / 1e12 / stg.PEAK_F32_MFMA_TFLOPS, 4), launches=v["launches"], ms_per_step=round(v["ms"], 3))

    def hbm_stage(v, ms=None, by=None):
        ms = v["ms"] if ms is None else ms
        by = v["bytes"] if by is None else by
        gbs = by / (ms * 1e-3) / 1e9
        return dict(bound="hbm", achieved=round(gbs, 1), unit="GB/s", frac=round(gbs / stg.PEAK_HBM_GBS, 4),
                    launches=v["launches"], ms_per_step=round(ms, 3))
    st = {}
    for k, v in stages.items():
        if k.startswith("groupnorm apply"):
            st["gn_apply_silu"] = hbm_stage(v)
        elif k.startswith("fused temporal-attention block"):
            st["ta_block_w" + k.split("width ")[1].split(" ")[0]] = mfma_stage(v)
        elif k == "temporal attention core":
            # an MFMA kernel whose time is its q / k / v / o traffic (16 FLOP per byte): both fractions, the larger one is the bound
            m, h = mfma_stage(v), hbm_stage(v)
            st["tattn_core"] = dict(h if h["frac"] >= m["frac"] else m, mfma_frac=m["frac"], hbm_frac=h["frac"])
        elif k.startswith("fused LinearAttention"):
            st["la_block"] = mfma_stage(v)
    # every temporal-attention launch together (fused blocks + unfused cores): the north star's "MFMA utilisation on temporal attention"
    ta = [v for k, v in stages.items() if k.startswith("fused temporal-attention block") or k == "temporal attention core"]
    if ta:
        tot = dict(ms=sum(v["ms"] for v in ta), issued=sum(v["issued"] for v in ta), launches=sum(v["launches"] for v in ta))
        st["temporal_attention_all"] = mfma_stage(tot)
    # the Conv+GN+SiLU block as a whole: two-pass algorithmic bytes 4 (N_in + 3 N_out) + 4 nW over conv + finalize + apply time.
    # It is MFMA-bound by >= 40x in fp32 (SURVEY 8d caveat), so its HBM fraction is small by construction; the HBM-bound piece is
    # gn_apply_silu above.
    blk = [v for k, v in stages.items() if "GroupNorm statistics in the epilogue" in k]
    gna = [v for k, v in stages.items() if k.startswith("groupnorm apply") or k.startswith("groupnorm stats")]
    if blk and gna:
        ms = sum(v["ms"] for v in blk) + sum(v["ms"] for v in gna)
        by = sum(v["bytes"] for v in blk) + sum(v["bytes"] for v in gna if v["bytes"])
        e = hbm_stage(dict(launches=sum(v["launches"] for v in blk)), ms, by)
        e["mfma_issued_tflops"] = round(sum(v["issued"] for v in blk) / (ms * 1e-3) / 1e12, 2)
        st["conv_gn_silu_block"] = e
    roof["stages"] = st
    roof["stage_peaks"] = {"TFLOP/s": stg.PEAK_F32_MFMA_TFLOPS, "GB/s": stg.PEAK_HBM_GBS}
    roof["all_kernels"] = {k: dict(launches=v["launches"], ms_per_step=round(v["ms"], 3),
                                   **({"issued_tflops": round(v["issued"] / (v["ms"] * 1e-3) / 1e12, 2),
                                       "effective_tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} if v["issued"] else
                                      {"gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}))
                           for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}
    roof["stage_sum_ms"] = round(sum(v["ms"] for v in stages.values()), 3)
    # the six heaviest kernels of the step: [name, launches, ms per step, TFLOP/s issued (MFMA kernels) or GB/s]
    roof["top_kernels"] = [[k, v["launches"], v["ms_per_step"], v.get("issued_tflops", v.get("gbs"))]
                           for k, v in list(roof["all_kernels"].items())[:6]]
    return roof


def selftest_worker(a):
    """`--selftest-launcher`: the launcher / rendezvous / relay path without a GPU (CPU tests, gloo): every rank contributes
    a shard of synthetic conformal (score, weight) pairs, the all-gather + quantile epilogue runs, the max-over-ranks timer is
    all-reduced and rank 0 prints a line of the same shape as the real one."""
    import torch
    import torch.distributed as dist
    from safediffcon_amd import conformal
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    backend = os.environ.get("SDC_DIST_BACKEND", "gloo")
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    n = 200
    scores, weights = torch.rand(n, generator=g), torch.rand(n, generator=g) * 3
    per = n // world
    t0 = time.perf_counter()
    Q, _ = conformal.weighted_quantile(scores[rank * per:(rank + 1) * per], weights[rank * per:(rank + 1) * per], 0.04, smoke=True)
    el = torch.tensor([time.perf_counter() - t0])
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        sc = shard_config("c4", DEFAULT_B["c4"], world)
        print(json.dumps({"metric": "selftest (launcher + conformal all-gather, no GPU)", "n_gpus": world, "backend": backend,
                          "config": sc["tag"], "global_batch": sc["global_batch"], "calibration_per_rank": per,
                          "dist_world_size": dist.get_world_size() if world > 1 else 1, "conformal_Q": round(float(Q), 6),
                          "allgather_quantile_ms": round(el.item() * 1e3, 3)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# --------------------------------------------------------------------------- worker
def worker(a):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    local = int(os.environ.get("SDC_FORCE_DEVICE", local))
    ndev = torch.cuda.device_count()
    if local >= ndev:
        raise SystemExit(f"bench.py --gpus {a.gpus}: rank {rank} needs device cuda:{local} but only {ndev} device(s) are visible")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm.  SDC_DIST_BACKEND=gloo + SDC_FORCE_DEVICE=0 exist only to rehearse the multi-process
        # path on a one-GPU box (RCCL refuses two ranks on one device).  device_id binds the communicator to this rank's GPU
        # at creation instead of at the first collective.
        backend = os.environ.get("SDC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from safediffcon_amd import _lib
    lib = _lib.get_lib()
    wl = "c4" if a.workload == "c5" else a.workload
    B = a.batch or DEFAULT_B[wl]
    prec = {"fp32": 4, "fp32-wino2d": 3, "fp32-wino1d": 2, "fp32-direct": 0}[a.precision]
    side = torch.cuda.Stream(device=dev)

    def timed(S, warmup, steps):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
        def run(n):
            for _ in range(n):
                if S.t_host < (0 if S.impose_last else 1):      # ran out of graph-able timesteps: restart at t = T-1
                    S.init()
                S.step()
        run(warmup)
        side.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        return el

    extra = {}
    t_start = time.perf_counter()
    with torch.cuda.stream(side), torch.no_grad():
        W = workload(wl, a.dim, B, dev, rank, world, prec, cal_steps=a.cal_steps, split_small=a.split_small)
        torch.manual_seed(2 + rank)                            # noise: seed 2 (+rank)
        S = W["prep"]()
        S.init()
        sensors = GpuSensors(local) if rank == 0 else None
        idle = sensors.read() if sensors else None
        if sensors:
            sensors.start()
        t_setup = time.perf_counter() - t_start
        dt = timed(S, a.warmup, a.steps)
        clocks = sensors.stop() if sensors else None
        if clocks and idle:
            clocks["sclk_mhz_before"], clocks["power_w_before"] = round(idle[0]), round(idle[1])
        finite = bool(torch.isfinite(S.x).all().item())
        step_ms = dt / a.steps * 1e3
        roof = build_roofline(S, lib, side.cuda_stream, step_ms, wl) if rank == 0 else None
        S.close()
        t_head = time.perf_counter() - t_start

        if world == 1 and not a.no_extra:
            import bench_extras as bx
            ctx = types.SimpleNamespace(workload=workload, timed=timed, rank=rank, world=world, step_seconds=_oracle_step_seconds,
                                        build_roofline=lambda S_, ms_, wl_: build_roofline(S_, lib, side.cuda_stream, ms_, wl_))
            extra = bx.run_extras(ctx, a, W, wl, B, prec, dev, step_ms)
        if a.full_sample and rank == 0:
            import bench_extras as bx
            extra["full_sample"] = bx.full_sample(W, wl, B)
        if a.full_calibration and rank == 0:
            import bench_extras as bx
            extra["full_calibration"] = bx.full_calibration(W, wl, dev)
    if not finite:
        raise SystemExit("non-finite state after the timed steps")

    value = world * B / (T_DDPM * dt / a.steps)
    # ranks the collective library itself reports: every rank contributes a device-resident 1 to an all-reduce
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())
    if rank == 0:
        cf = W["conformal"]
        cpu = None
        if not a.no_cpu_baseline and world == 1:              # reported on rank 0 at N=1 only
            t0 = time.perf_counter()
            cb = a.cpu_batch or {"c2": 32, "c3": 32, "c4": 1}[wl]
            cpu = cpu_baseline(wl, cb, a.cpu_steps or (3 if wl != "c4" else 2), a.dim, all_cores=a.cpu_all_cores)
            extra.setdefault("phases_s", {})["cpu_baseline"] = round(time.perf_counter() - t0, 2)
        if a.cpu_c1_full:
            import bench_extras as bx
            extra["cpu_c1_full"] = bx.cpu_c1_full(cpu_model)
        extra.setdefault("phases_s", {}).update(setup=round(t_setup, 2), headline=round(t_head - t_setup, 2))
        # the report: every stage, every kernel, every extra workload -> the side file; the line below stays a headline
        report = {"roofline_all_kernels": roof.pop("all_kernels"), "roofline_notes": roof.pop("notes"),
                  "conformal": dict(cf, Q=round(cf["Q"], 6)), "gpu_sensors": clocks, "cpu_baseline": cpu, "extra": extra}
        extra_file = None
        try:
            with open(a.extra_file, "w") as fh:
                json.dump(report, fh, indent=1)
            extra_file = os.path.relpath(a.extra_file, ROOT) if os.path.abspath(a.extra_file).startswith(ROOT) else a.extra_file
        except OSError as e:                                  # read-only tree: the headline still prints
            print(f"bench.py: could not write {a.extra_file}: {e}", file=sys.stderr)
        print(f"bench.py: report -> {a.extra_file}; phases (s): {extra.get('phases_s')}", file=sys.stderr, flush=True)
        print(json.dumps(headline(a, world, B, value, step_ms, W, roof, cpu, clocks, cf, backend, ranks_seen, extra, extra_file)),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def headline(a, world, B, value, step_ms, W, roof, cpu, clocks, cf, backend, ranks_seen, extra, extra_file):
    """the ONE stdout line of the contract, kept under HEADLINE_MAX_BYTES (the driver could not parse a 25 KB
    line).  Everything descriptive lives in the side file `extra_file`."""
    out = {
        "metric": "sampled control trajectories/sec (1000-step DDPM)", "value": round(value, 4), "unit": "trajectories/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(step_ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": W["desc"], "batch_per_gpu": B, "global_batch": world * B, "ddpm_timesteps": T_DDPM,
                   "step": "U-Net + guidance + posterior update of the whole batch, one hipGraph replay",
                   "parallelism": f"batch-sharded x{world}, no data-path collective", "conv_precision": a.precision},
        "roofline": roof,
    }
    if cpu:
        out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model")}
    out["conformal"] = {"Q": round(cf["Q"], 6), "n_cal": cf["n_cal"], "alpha": cf["alpha"],
                        "allgather_quantile_ms": cf["allgather_quantile_ms"], "backend": backend or "none (single process)",
                        "rccl_ranks": (ranks_seen if backend == "nccl" else (1 if world == 1 else 0)), "dist_world_size": world}
    if clocks:
        out["gpu_sensors"] = {k: clocks[k] for k in ("sclk_mhz_median", "sclk_mhz_min", "power_w_mean", "power_w_max") if k in clocks}
    if LIBRARY_OVERRIDE:
        out["library"] = f"OVERRIDE (A/B run): {LIBRARY_OVERRIDE}"
    # one number per extra workload; the rest of each block is in the side file
    brief = {k: ({"ms_per_step": v["ms_per_step"], "value": v["value"], "frac": v["roofline"]["frac"]} if "roofline" in v else
                 {kk: v[kk] for kk in ("ms_per_step", "hip_ms", "hip_graph_ms", "hip_speedup_per_trajectory",
                                       "seconds_for_one_1000_step_sample", "trajectories_per_s", "seconds") if kk in v})
             for k, v in extra.items() if isinstance(v, dict) and k != "phases_s"}
    if brief:
        out["extra"] = brief
    out["extra_file"] = extra_file
    line = json.dumps(out)
    if len(line) > HEADLINE_MAX_BYTES:                        # never print a line the driver cannot parse: shed the optional keys
        sheds = [lambda: out.pop("extra", None), lambda: out.pop("gpu_sensors", None), lambda: out.pop("conformal", None),
                 lambda: roof.pop("top_kernels", None),
                 lambda: roof.update(stages={k: v.get("frac") for k, v in list(roof["stages"].items())[:16]})]
        for shed in sheds:
            shed()
            if len(json.dumps(out)) <= HEADLINE_MAX_BYTES:
                break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c4", choices=["c2", "c3", "c4", "c5"],
                    help="c4 (default) = the north-star configuration, 2D smoke B=64 per GPU; c5 = the same, named for N=8")
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (default: 256 / 128 / 64 for c2 / c3 / c4)")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32-wino2d", "fp32-wino1d", "fp32-direct"],
                    help="conv arithmetic: fp32 MFMA with Winograd F(2x2x2,3x3x3) / F(2x2,3x3) / F(2,3) on the 3-tap convs (default), "
                         "without the depth transform, F(2,3) along W only, or the fp32 direct form everywhere")
    ap.add_argument("--no-extra", action="store_true", help="skip the calibration sample and the other workloads at N=1")
    ap.add_argument("--extra-workloads", default="c2,c3",
                    help="other configs timed at N=1 and written to the side file (keys of bench_extras.EXTRA_WORKLOADS)")
    ap.add_argument("--all-extras", action="store_true",
                    help="every extra: the shipped widths and the 8-way-shard batches of the 1-D configs, the PyTorch-ROCm autograd "
                         "comparison of the fine-tuning step, the 1-D fine-tuning steps, the smoke score check "
                         "(tools/collect_profiles.sh passes it; minutes)")
    ap.add_argument("--extra-file", default=os.path.join(ROOT, "bench_extra.json"),
                    help="side file for everything that is not the headline (named in the headline as `extra_file`)")
    ap.add_argument("--cpu-all-cores", action="store_true", help="cpu_baseline: also one step with a thread per physical core")
    ap.add_argument("--split-small", action="store_true", help="the headline workload's net with split_small_grids (small-batch plan)")
    ap.add_argument("--extra-steps", type=int, default=20)
    ap.add_argument("--cal-steps", type=int, default=5)
    ap.add_argument("--other-precisions", action="store_true", help="also time precision 0 / 3 on the headline workload")
    ap.add_argument("--full-sample", action="store_true",
                    help="also time ONE complete 1000-step sample() call (validates value = B / (1000 * step time))")
    ap.add_argument("--full-calibration", action="store_true", help="also run one complete calibration pass (minutes)")
    ap.add_argument("--cpu-c1-full", action="store_true", help="also run BASELINE configs[0] (C1) in full on the host cores (minutes)")
    ap.add_argument("--selftest-launcher", action="store_true", help="CPU-only check of the launcher path (tests)")
    ap.add_argument("--no-finetune", action="store_true", help="skip the fine-tuning step timing (extra.finetune_step)")
    ap.add_argument("--finetune-batch", type=int, default=0)
    ap.add_argument("--no-strawman", action="store_true", help="skip the PyTorch-ROCm eager timing of the oracle (extra.strawman)")
    ap.add_argument("--strawman-batch", type=int, default=0)
    ap.add_argument("--cpu-batch", type=int, default=0)
    ap.add_argument("--cpu-steps", type=int, default=0)
    a = ap.parse_args()
    if a.all_extras:
        import bench_extras as bx
        a.extra_workloads = bx.ALL_EXTRA_WORKLOADS
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(a.gpus, sys.argv[1:]))
    if a.selftest_launcher:
        return selftest_worker(a)
    worker(a)


if __name__ == "__main__":
    main()
