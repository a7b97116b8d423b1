"""GPU probe: sdc_smoke_rollout against the reference fixtures (errors) and its time per batch size."""
import _libsel  # noqa: F401,E402  (SDC_LIB_PATH -> safediffcon_amd._lib.use_library, tools only)
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safediffcon_amd import smoke_solver as ss

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = torch.device("cuda:0")
sim = ss.init_sim_128()
for name in ["short_a", "short_nan", "short_128", "full_a", "full_b"]:
    f = np.load(os.path.join(G, f"smoke_solver_{name}.npz"))
    c1 = torch.from_numpy(f["c1"]).to(dev)[None]; c2 = torch.from_numpy(f["c2"]).to(dev)[None]
    d0 = torch.from_numpy(f["init_density"]).to(dev)[None]
    torch.cuda.synchronize(); t0 = time.time()
    out = ss.solver(sim, ss.init_velocity_(), d0, c1, c2, int(f["per_timelength"]))
    torch.cuda.synchronize(); dt = time.time() - t0
    dens, zd, vel, oc1, oc2, rec, recs = [o[0].cpu().numpy() for o in out]
    if "velocitys" in f:
        ev = np.abs(vel - f["velocitys"]).max()
    else:
        ev = np.abs(vel[list(f["f64_frames"])] - f["velocitys_f64"]).max()
    print(name, "time %.3fs" % dt, "dens", np.abs(dens - f["densitys"]).max(), "zdens", np.abs(zd - f["zero_densitys"]).max(),
          "vel", ev, "velscale", np.abs(vel).max(),
          "rec", np.nanmax(np.abs(rec[:, 0, 0] - f["smoke_out_record"])) if not np.isnan(f["smoke_out_record"]).all() else "nan-eq %s" % np.isnan(rec).all(),
          "recs", np.nanmax(np.abs(recs[:, 0, 0] - f["smoke_out_safe_record"])) if not np.isnan(f["smoke_out_safe_record"]).all() else "nan",
          "c1 eq", np.array_equal(oc1, f["out_c1"]), flush=True)
f = np.load(os.path.join(G, "smoke_solver_full_b.npz"))
for B in (1, 16, 64, 128, 256):
    c1 = torch.from_numpy(f["c1"]).to(dev)[None].repeat(B, 1, 1, 1); c2 = torch.from_numpy(f["c2"]).to(dev)[None].repeat(B, 1, 1, 1)
    d0 = torch.from_numpy(f["init_density"]).to(dev)[None].repeat(B, 1, 1)
    torch.cuda.synchronize(); t0 = time.time()
    out = ss.solver(sim, ss.init_velocity_(), d0, c1, c2, 256)
    torch.cuda.synchronize(); dt = time.time() - t0
    same = all(torch.equal(o[0], o[-1]) or (torch.isnan(o[0]) == torch.isnan(o[-1])).all() for o in out)
    print("B", B, "time %.3f s" % dt, "per CG iteration %.2f us" % (dt / 255 / 500 * 1e6), "first == last sample:", same, flush=True)
