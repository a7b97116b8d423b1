"""Oracle: the smoke task's evaluation rollout -- what the reference runs on every sampled control sequence to
score it (2d/inference_2d.py:389-447 -> 2d/dataset/apps/evaluate_solver.py:209-350 `solver`), restated in plain
numpy from the reference's own call chain through its vendored PhiFlow 1.x:

    init_sim_128 / build_obstacles_pi_128   evaluate_solver.py:29-65      obstacle layout on the 127 x 127 grid
    FluidSimulation masks                    phi/flow.py:161-200,455-474   fluid / active / staggered velocity masks
    get_envolve                              evaluate_solver.py:82-111     control ring + last interior velocity
    divergence_free                          phi/flow.py:317-326           mask, divergence, pressure, mask * gradient
    StaggeredGrid.divergence / .gradient     phi/math/nd.py:333-344,581-592
    sparse_pressure_matrix                   phi/solver/sparse.py:27-77    5-point matrix with obstacle / open-border terms
    conjugate_gradient                       phi/solver/base.py:63-103     CG, |r|_max >= 1e-8, at most 500 iterations
    StaggeredGrid.advect (centred field)     phi/math/nd.py:407-428        semi-Lagrangian, linear, REPLICATE clamp
    SciPyBackend.resample -> scipy interpn   phi/math/scipy_backend.py:55-75
    bucket masks and the smoke book-keeping  evaluate_solver.py:114-178,262-336

Arithmetic follows the reference operation for operation (float64 throughout, float32 density fields, scipy's CSC
mat-vec accumulation order, its generic linear interpolant, numpy's pairwise sums, and the CG's first-iteration
aliasing of `residual` and `momentum`), so that on the same inputs the result is bit-identical to the reference run
under `oracle/phi_compat.py` -- tests/test_smoke_solver_oracle.py holds it to tests/golden/smoke_solver_*.npz.

TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import numpy as np

N = 127            # cells per side (evaluate_solver.py:63)
S = 128            # staggered samples per side
CG_ACCURACY = 1e-8      # evaluate_solver.py:108
CG_MAX_ITER = 500       # phi/solver/sparse.py:88 default


# ---------------------------------------------------------------------------------------------- domain

_OBSTACLES = [  # (size_y, size_x), (origin_y, origin_x) -- evaluate_solver.py:36-60
    ((1, 96), (16, 16)),
    ((8, 1), (16, 16)), ((16, 1), (40, 16)), ((40, 1), (72, 16)),
    ((8, 1), (16, 112)), ((16, 1), (40, 112)), ((40, 1), (72, 112)),
    ((1, 8), (112, 16)), ((1, 16), (112, 40)), ((1, 16), (112, 72)), ((1, 8), (112, 104)),
    ((16, 1), (64, 48)), ((16, 1), (96, 48)), ((16, 1), (64, 80)), ((16, 1), (96, 80)),
    ((1, 128 - 40 - 40), (40, 40)),
]


def fluid_mask():
    """1 = fluid, 0 = obstacle, (127, 127) int8.  The reference sets active and fluid masks alike (phi/flow.py:190-191)."""
    m = np.ones((N, N), np.int8)
    for (sy, sx), (oy, ox) in _OBSTACLES:
        m[oy:oy + sy, ox:ox + sx] = 0
    return m


def velocity_mask(fluid):
    """Staggered mask (128, 128, 2): component 0 = x faces, 1 = y faces; open borders pad the fluid mask with ones."""
    p = np.pad(fluid, 1, constant_values=1)
    vm = np.empty((S, S, 2), np.int8)
    vm[..., 0] = np.minimum(p[1:, 1:], p[1:, :-1])
    vm[..., 1] = np.minimum(p[1:, 1:], p[:-1, 1:])
    return vm


def pressure_stencil(fluid):
    """Coefficients of the reference's sparse matrix: (a_im, a_jm, diag, a_jp, a_ip), each (127, 127) float64 --
    neighbour order = ascending column index of row (i, j), the order scipy's CSC mat-vec adds them in."""
    act = np.pad(fluid, 1, constant_values=0).astype(np.int8)      # pad_active: inactive outside
    flu = np.pad(fluid, 1, constant_values=1).astype(np.int8)      # pad_fluid (open): fluid outside
    c = act[1:-1, 1:-1]
    a_ip, a_im = act[2:, 1:-1] * c, act[:-2, 1:-1] * c
    a_jp, a_jm = act[1:-1, 2:] * c, act[1:-1, :-2] * c
    centre = (-flu[2:, 1:-1] - flu[:-2, 1:-1]) + (-flu[1:-1, 2:] - flu[1:-1, :-2])
    diag = np.minimum(centre, -1)
    return tuple(np.asarray(v, np.float64) for v in (a_im, a_jm, diag, a_jp, a_ip))


def bucket_masks():
    """get_bucket_mask (evaluate_solver.py:114-135): seven absorbing areas; the second one is the target bucket."""
    pos = [(112, 22, 15, 20), (112, 54, 15, 20), (112, 86, 15, 20),
           (22, 0, 20, 16), (54, 0, 20, 16), (22, 112, 20, 15), (54, 112, 20, 15)]
    return _masks(pos)


def bucket_masks_safe():
    """get_bucket_mask_safe (evaluate_solver.py:138-178): the hazard area first, then the same seven."""
    pos = [(40, 44, 24, 12),
           (112, 22, 15, 20), (112, 54, 15, 20), (112, 86, 15, 20),
           (22, 0, 20, 16), (54, 0, 20, 16), (22, 112, 20, 15), (54, 112, 20, 15)]
    return _masks(pos)


def _masks(pos):
    each, concat, keep = [], np.zeros((S, S)), np.ones((S, S))
    for y, x, ly, lx in pos:
        m = np.zeros((S, S))
        m[y:y + ly, x:x + lx] = 1
        concat[y:y + ly, x:x + lx] = 1
        keep[y:y + ly, x:x + lx] = 0
        each.append(m)
    return each, concat, keep


# ---------------------------------------------------------------------------------------------- pressure projection

def apply_A(st, p):
    """A @ p on the (127, 127) grid in the CSC accumulation order (columns ascending per row, from zero)."""
    a_im, a_jm, diag, a_jp, a_ip = st
    pp = np.pad(p, 1)
    y = np.zeros_like(p)
    y = y + a_im * pp[:-2, 1:-1]
    y = y + a_jm * pp[1:-1, :-2]
    y = y + diag * p
    y = y + a_jp * pp[1:-1, 2:]
    y = y + a_ip * pp[2:, 1:-1]
    return y


def conjugate_gradient(st, k, accuracy=CG_ACCURACY, max_iterations=CG_MAX_ITER, dot=None):
    """phi/solver/base.py:63-103 with initial_x None.  `residual` and `momentum` start as the same array there and
    `residual -= ...` works in place, so the first direction update reads the NEW residual on both sides."""
    if dot is None:
        dot = lambda a, b: np.sum((a * b).reshape(1, -1))
    x = np.zeros_like(k)
    momentum = k.copy()
    residual = momentum                       # the reference's alias
    Ap = apply_A(st, momentum)
    it = 0
    while np.max(np.abs(residual)) >= accuracy:
        if it == max_iterations:
            break
        tmp = dot(momentum, Ap)
        a = dot(momentum, residual) / tmp
        x += a * momentum
        residual -= a * Ap
        b = -dot(residual, Ap) / tmp
        momentum = residual + b * momentum
        Ap = apply_A(st, momentum)
        it += 1
    return x, it


def divergence(v):
    """StaggeredGrid.divergence: y component first, then x (phi/math/nd.py:333-344)."""
    return (v[1:, :-1, 1] - v[:-1, :-1, 1]) + (v[:-1, 1:, 0] - v[:-1, :-1, 0])


def gradient(p):
    """StaggeredGrid.gradient with symmetric padding: (128, 128, 2), component 0 = d/dx."""
    f = np.pad(p, 1, mode="symmetric")
    g = np.empty((S, S, 2))
    g[..., 0] = f[1:, 1:] - f[1:, :-1]
    g[..., 1] = f[1:, 1:] - f[:-1, 1:]
    return g


def evolve(dom, prev, c1f, c2f, dot=None):
    """get_envolve: prev (128, 128, 2) staggered velocity, c1f / c2f (128, 128) control frame -> next velocity."""
    vm, st = dom["vmask"], dom["stencil"]
    cur = np.zeros((S, S, 2))
    cur[..., 0] = c1f
    cur[..., 1] = c2f
    cur[16:112, 16:112, :] = prev[16:112, 16:112, :]
    v = cur * vm
    p, it = conjugate_gradient(st, divergence(v), dot=dot)
    v = v - gradient(p) * vm
    return v * vm, it


# ---------------------------------------------------------------------------------------------- advection

def advect(v, fields):
    """Semi-Lagrangian step of centred (127, 127) float32 fields through the staggered velocity v."""
    cy = (v[1:, :-1, 1] + v[:-1, :-1, 1]) / 2
    cx = (v[:-1, 1:, 0] + v[:-1, :-1, 0]) / 2
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    y = ii.astype(np.float32) - cy
    x = jj.astype(np.float32) - cx
    y = np.maximum(0, np.minimum(N, y))
    x = np.maximum(0, np.minimum(N, x))
    oob = (y > N - 1) | (x > N - 1)                         # interpn(bounds_error=False, fill_value=0)
    i0 = np.clip(np.floor(y).astype(np.int64), 0, N - 2)
    j0 = np.clip(np.floor(x).astype(np.int64), 0, N - 2)
    ty, tx = y - i0, x - j0
    sy, sx = 1 - ty, 1 - tx
    out = []
    for f in fields:
        val = np.array([0.])
        for (ia, wa) in ((i0, sy), (i0 + 1, ty)):
            for (ib, wb) in ((j0, sx), (j0 + 1, tx)):
                w = np.array([1.]) * wa * wb
                val = val + f[ia, ib] * w
        val[oob] = 0
        out.append(val.astype(f.dtype))
    return out


# ---------------------------------------------------------------------------------------------- the rollout

def domain():
    fluid = fluid_mask()
    return {"fluid": fluid, "vmask": velocity_mask(fluid), "stencil": pressure_stencil(fluid)}


def _pad128(d):
    a = np.zeros((S, S), dtype=float)
    a[:-1, :-1] = d
    return a


def solver(init_velocity, init_density, c1, c2, per_timelength=256, dom=None, dot=None, return_iters=False):
    """evaluate_solver.solver (dt = 1): init_velocity (128,128,2), init_density (nx,nx), c1 / c2 (nt,nx,nx) ->
    the reference's seven arrays, sub-sampled in time and space the same way."""
    dom = dom or domain()
    nt, nx = c1.shape[0], c1.shape[1]
    num_t = per_timelength
    ti, si = int(num_t / nt), int(128 / nx)
    dens0 = np.tile(init_density.reshape(nx, 1, nx, 1), (1, si, 1, si)).reshape(128, 128)
    c1 = np.tile(c1.reshape(nt, 1, nx, 1, nx, 1), (1, ti, 1, si, 1, si)).reshape(num_t, 128, 128)
    c2 = np.tile(c2.reshape(nt, 1, nx, 1, nx, 1), (1, ti, 1, si, 1, si)).reshape(num_t, 128, 128)
    dens = dens0[:-1, :-1].copy()
    dz, dzs = dens.copy(), dens.copy()
    vel = np.asarray(init_velocity).reshape(128, 128, 2)
    each, concat, keep = bucket_masks()
    each_s, concat_s, keep_s = bucket_masks_safe()
    outs, outs_s = np.zeros(len(each)), np.zeros(len(each_s))
    densitys, zero_densitys, velocitys, rec, rec_s, iters = [], [], [], [], [], []

    def book(dz, dzs):
        az, azs = _pad128(dz), _pad128(dzs)
        if np.sum(az * concat) > 0:
            for i in range(len(each)):
                outs[i] += np.sum(az * each[i])
            dz = (dz * keep[:-1, :-1]).astype(dz.dtype)
        if np.sum(azs * concat_s) > 0:
            for i in range(len(each_s)):
                outs_s[i] += np.sum(azs * each_s[i])
            dzs = (dzs * keep_s[:-1, :-1]).astype(dzs.dtype)
        az, azs = _pad128(dz), _pad128(dzs)
        with np.errstate(all="ignore"):
            rec.append(outs[1] / (np.sum(outs) + np.sum(az)))
            rec_s.append(outs_s[0] / (np.sum(outs_s) + np.sum(azs)))
        return dz, dzs, az

    velocitys.append(np.array(vel, dtype=float))
    densitys.append(_pad128(dens))
    dz, dzs, az = book(dz, dzs)
    zero_densitys.append(az)
    for frame in range(num_t - 1):
        vel, it = evolve(dom, vel, c1[frame], c2[frame], dot=dot)
        iters.append(it)
        dens, dz, dzs = advect(vel, [dens, dz, dzs])
        dz, dzs, az = book(dz, dzs)
        densitys.append(_pad128(dens))
        zero_densitys.append(az)
        velocitys.append(vel.copy())
    rec = np.tile(np.stack(rec)[:, None, None], (1, 128, 128))
    rec_s = np.tile(np.stack(rec_s)[:, None, None], (1, 128, 128))
    res = (np.stack(densitys)[::ti, ::si, ::si], np.stack(zero_densitys)[::ti, ::si, ::si],
           np.stack(velocitys)[::ti, ::si, ::si], c1[::ti, ::si, ::si], c2[::ti, ::si, ::si],
           rec[::ti, ::si, ::si], rec_s[::ti, ::si, ::si])
    return (res, iters) if return_iters else res


def init_velocity():
    """init_velocity_ (evaluate_solver.py:68-79): vx = 0, vy = 0.8 in float32."""
    v = np.empty((128, 128, 2), np.float32)
    v[..., 0] = 0
    v[..., 1] = 0.8
    return v


def multi_evaluate_fields(pred, data, per_timelength=256):
    """The solver part of InferencePipeline.multi_evaluate (2d/inference_2d.py:407-456): pred, data (B,32,7,64,64)
    numpy, un-rescaled -> solver_out (B,32,7,64,64) float64."""
    pred = np.array(pred, copy=True)
    pred[:, 0, 0] = data[:, 0, 0]
    pred[:, :, 3:5, 8:56, 8:56] = 0
    out = np.zeros_like(pred, dtype=float)
    dom = domain()
    for i in range(pred.shape[0]):
        r = solver(init_velocity(), data[i, 0, 0], pred[i, :, 3], pred[i, :, 4], per_timelength, dom=dom)
        out[i, :, 0] = r[0]
        out[i, :, 1] = r[2][..., 0]
        out[i, :, 2] = r[2][..., 1]
        out[i, :, 3] = r[3]
        out[i, :, 4] = r[4]
        out[i, :, 5] = r[5]
        out[i, :, 6] = r[6]
    return out
