"""Smoke score check: drop-ins for the fluid rollout that follows sampling in the 2-D smoke task.

  init_sim_128 / build_obstacles_pi_128      2d/dataset/apps/evaluate_solver.py:29-65
  init_velocity_                             2d/dataset/apps/evaluate_solver.py:77-79
  get_bucket_mask / get_bucket_mask_safe     2d/dataset/apps/evaluate_solver.py:114-178
  solver                                     2d/dataset/apps/evaluate_solver.py:209-350
  multi_evaluate                             2d/inference_2d.py:407-507   (InferencePipeline method; Q / safe_bound passed in)

The reference starts one Python process per sample, each stepping PhiFlow's numpy CG 255 times; here the whole batch is one
launch of `sdc_smoke_rollout` (csrc/sdc_smoke.hip), one workgroup per sample, on device tensors.  `solver` takes a batch
axis in front of the reference's per-sample arguments; everything else keeps the reference's names, argument meaning, dtypes
of the results (float64) and error behaviour (sizes that do not divide raise ValueError like the reference's reshape).
No CPU fallback: tensors must live on the MI355X.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check

_N = 127


class FluidSimulation:
    """The part of phi.flow.FluidSimulation the score check touches: the obstacle layout of the 127 x 127 open domain
    (`set_obstacle`, phi/flow.py:170-200) -- masks only, the solve happens in the kernel."""

    def __init__(self, dimensions=(_N, _N)):
        if list(dimensions) != [_N, _N]:
            raise ValueError("safediffcon_amd.smoke_solver: the rollout kernel is built for the reference's 127 x 127 domain")
        self.dimensions = list(dimensions)
        self._fluid_mask = np.ones((1, _N, _N, 1), np.int8)
        self._active_mask = self._fluid_mask           # the reference sets both alike
        self._dev = {}

    def set_obstacle(self, mask_or_size, origin=None):
        if isinstance(mask_or_size, np.ndarray):
            raise NotImplementedError()                 # as in the reference (phi/flow.py:181-184)
        if isinstance(mask_or_size, int):
            mask_or_size = [mask_or_size, mask_or_size]
        origin = [0, 0] if origin is None else list(origin)
        self._fluid_mask[0, origin[0]:origin[0] + mask_or_size[0], origin[1]:origin[1] + mask_or_size[1], 0] = 0
        self._dev = {}

    def fluid_mask_device(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(np.ascontiguousarray(self._fluid_mask[0, :, :, 0]).astype(np.uint8)).to(device)
        return self._dev[key]


def build_obstacles_pi_128(sim):
    sim.set_obstacle((1, 96), (16, 16))          # bottom
    sim.set_obstacle((8, 1), (16, 16))           # left
    sim.set_obstacle((16, 1), (40, 16))
    sim.set_obstacle((40, 1), (72, 16))
    sim.set_obstacle((8, 1), (16, 112))          # right
    sim.set_obstacle((16, 1), (40, 112))
    sim.set_obstacle((40, 1), (72, 112))
    sim.set_obstacle((1, 8), (112, 16))          # buckets
    sim.set_obstacle((1, 16), (112, 40))
    sim.set_obstacle((1, 16), (112, 72))
    sim.set_obstacle((1, 8), (112, 104))
    sim.set_obstacle((16, 1), (64, 48))          # vertical bars
    sim.set_obstacle((16, 1), (96, 48))
    sim.set_obstacle((16, 1), (64, 80))
    sim.set_obstacle((16, 1), (96, 80))
    sim.set_obstacle((1, 128 - 40 - 40), (40, 40))


def init_sim_128():
    sim = FluidSimulation([_N] * 2)
    build_obstacles_pi_128(sim)
    return sim


def init_velocity_():
    """(1, 128, 128, 2) float32 staggered field, vx = 0, vy = 0.8."""
    v = np.empty([1, 128, 128, 2], np.float32)
    v[..., 0] = 0
    v[..., 1] = 0.8
    return v


_BUCKETS = [(112, 22, 15, 20), (112, 54, 15, 20), (112, 86, 15, 20),
            (22, 0, 20, 16), (54, 0, 20, 16), (22, 112, 20, 15), (54, 112, 20, 15)]
_SAFE = [(40, 44, 24, 12)]


def _masks(pos):
    each, concat, keep = [], np.zeros((128, 128)), np.ones((128, 128))
    for y, x, ly, lx in pos:
        m = np.zeros((128, 128))
        m[y:y + ly, x:x + lx] = 1
        concat[y:y + ly, x:x + lx] = 1
        keep[y:y + ly, x:x + lx] = 0
        each.append(m)
    return each, concat, keep


def get_bucket_mask():
    return _masks(_BUCKETS)


def get_bucket_mask_safe():
    return _masks(_SAFE + _BUCKETS)


def _labels(each):
    lab = np.zeros((128, 128), np.uint8)
    for k, m in enumerate(each):
        if (lab[m > 0] != 0).any():
            raise ValueError("absorbing areas overlap")
        lab[m > 0] = k + 1
    return lab


_label_cache = {}


def _device_labels(device):
    key = str(device)
    if key not in _label_cache:
        n, s = get_bucket_mask()[0], get_bucket_mask_safe()[0]
        _label_cache[key] = (torch.from_numpy(_labels(n)).to(device), len(n), torch.from_numpy(_labels(s)).to(device), len(s))
    return _label_cache[key]


def _rollout(sim, init_velocity, init_density, c1, c2, per_timelength, dt=1, accuracy=1e-8, max_iterations=500):
    """-> (out (B, nt, 7, nx, nx), zero_densitys (B, nt, nx, nx)), float64 on the device of c1"""
    if dt != 1:
        raise NotImplementedError("the reference calls solver with dt = 1 only (2d/inference_2d.py:398)")
    if not (torch.is_tensor(c1) and c1.is_cuda):
        raise RuntimeError("safediffcon_amd.smoke_solver runs on MI355X only (no CPU fallback)")
    dev = c1.device
    B, nt, nx = c1.shape[0], c1.shape[1], c1.shape[2]
    if c1.shape != c2.shape or c1.shape[3] != nx or tuple(init_density.shape) != (B, nx, nx):
        raise ValueError(f"shapes: c1 {tuple(c1.shape)}, c2 {tuple(c2.shape)}, init_density {tuple(init_density.shape)}")
    if 128 % nx or per_timelength % nt:
        raise ValueError(f"cannot reshape: nx = {nx} must divide 128 and nt = {nt} must divide per_timelength = {per_timelength}")

    def rows_dense(t):
        t = t.detach()
        if t.dtype != torch.float32:
            t = t.float()
        if t.stride(-1) != 1 or t.stride(-2) != nx:
            t = t.contiguous()
        return t
    c1d, c2d = rows_dense(c1), rows_dense(c2)
    if c1d.stride()[:2] != c2d.stride()[:2]:
        c1d, c2d = c1d.contiguous(), c2d.contiguous()
    d0 = init_density.detach().to(dev, torch.float32).contiguous()
    v0 = torch.as_tensor(np.asarray(init_velocity) if not torch.is_tensor(init_velocity) else init_velocity)
    v0 = v0.to(dev, torch.float32).reshape(-1, 128, 128, 2).contiguous()
    if v0.shape[0] not in (1, B):
        raise ValueError(f"init_velocity batch {v0.shape[0]} vs {B}")
    lab_n, nb_n, lab_s, nb_s = _device_labels(dev)
    fluid = sim.fluid_mask_device(dev)
    lib = _lib.get_lib()
    out = torch.empty(B, nt, 7, nx, nx, dtype=torch.float64, device=dev)
    outz = torch.empty(B, nt, nx, nx, dtype=torch.float64, device=dev)
    wbytes = lib.sdc_smoke_rollout_workspace_bytes(B)
    work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    check(lib.sdc_smoke_rollout(c1d.data_ptr(), c2d.data_ptr(), c1d.stride(0), c1d.stride(1), d0.data_ptr(), d0.stride(0),
                                v0.data_ptr(), 0 if v0.shape[0] == 1 else v0.stride(0), fluid.data_ptr(), lab_n.data_ptr(),
                                lab_s.data_ptr(), nb_n, nb_s, out.data_ptr(), outz.data_ptr(), work.data_ptr(), wbytes, B, nt, nx,
                                int(per_timelength), 16, 112, float(accuracy), int(max_iterations), stream),
          "sdc_smoke_rollout")
    return out, outz


def solver(sim, init_velocity, init_density, c1, c2, per_timelength, dt=1, accuracy=1e-8, max_iterations=500):
    """Batched evaluate_solver.solver on device tensors.

      init_velocity (128,128,2) / (1,128,128,2) shared, or (B,128,128,2); numpy or tensor
      init_density  (B, nx, nx);  c1, c2 (B, nt, nx, nx) -- any strided fp32 views with dense rows (e.g. pred[:, :, 3])
    -> (densitys, zero_densitys, velocitys, c1, c2, smoke_out_record, smoke_out_safe_record), float64 device tensors of
       shapes (B,nt,nx,nx) x2, (B,nt,nx,nx,2), (B,nt,nx,nx) x4: the reference's seven arrays with a batch axis in front.
    """
    out, outz = _rollout(sim, init_velocity, init_density, c1, c2, per_timelength, dt, accuracy, max_iterations)
    return (out[:, :, 0], outz, torch.stack((out[:, :, 1], out[:, :, 2]), dim=-1), out[:, :, 3], out[:, :, 4],
            out[:, :, 5], out[:, :, 6])


def solver_out(sim, pred, data, per_timelength=256):
    """The rollout part of multi_evaluate (2d/inference_2d.py:413-456): pred, data (B, 32, 7, 64, 64) un-rescaled device
    tensors -> solver_out (B, 32, 7, 64, 64) float64 on the device.  pred is modified like in the reference (initial
    density imposed); the indirect-control zeroing is applied to a copy of its two control channels."""
    pred[:, 0, 0] = data[:, 0, 0]
    ctrl = pred[:, :, 3:5].detach().float().clone()
    ctrl[:, :, :, 8:56, 8:56] = 0
    # the kernel writes the seven channels in solver_out's own layout
    return _rollout(sim, init_velocity_(), data[:, 0, 0], ctrl[:, :, 0], ctrl[:, :, 1], per_timelength)[0]


def multi_evaluate(pred, data, Q, safe_bound, sim=None, batch_id=0, plot=False, per_timelength=256):
    """InferencePipeline.multi_evaluate (2d/inference_2d.py:407-507) with self.Q / self.args_general.safe_bound as
    arguments; returns the reference's eight numpy arrays."""
    if plot:
        raise NotImplementedError("GIF output is the reference's debugging aid, not part of the score check")
    sim = sim or init_sim_128()
    device = pred.device
    out = solver_out(sim, pred, data, per_timelength)
    data = out.to(device)                                    # float64, as torch.tensor(solver_out) is in the reference
    mask = torch.ones_like(pred, device=device)
    mask[:, 0] = False
    pred = pred * mask
    data = data * mask
    diff = pred - data
    mse = torch.cat((diff[:, :, :3], diff[:, :, -2:]), dim=2).square().mean((1, 2, 3, 4)).detach().cpu().numpy()
    n_l2 = (diff[:, :, :3].square().sum((1, 2, 3, 4)).sqrt() / data[:, :, :3].square().sum((1, 2, 3, 4)).sqrt()).detach().cpu().numpy()
    zero = torch.zeros_like(data[:, -1, 6, 0, 0])
    J_target = -data[:, -1, 5, 0, 0].detach().cpu().numpy()
    safe_target = data[:, -1, 6, 0, 0].detach().cpu().numpy()
    J_safe_target = torch.maximum(data[:, -1, 6, 0, 0] - safe_bound, zero).detach().cpu().numpy()
    J_safe_target_pred = torch.maximum(pred[:, -1, 6, 0, 0] + Q - safe_bound, zero).detach().cpu().numpy()
    zt = torch.zeros_like(data[:, :, 6, 0, 0])
    J_safe_target_time = torch.maximum(data[:, :, 6, 0, 0] - safe_bound, zt).detach().cpu().numpy()
    J_safe_target_pred_time = torch.maximum(pred[:, :, 6, 0, 0] + Q - safe_bound, zt).detach().cpu().numpy()
    return J_target, safe_target, J_safe_target, J_safe_target_pred, J_safe_target_time, J_safe_target_pred_time, mse, n_l2
