"""safediffcon_amd -- MI355X-native sampler for SafeDiffCon's PDE-control hot path.

Drop-in classes (same names / signatures as the reference):
  Unet2D, Unet1D, Unet3D_with_Conv3D                      (safediffcon_amd.unet)
  GaussianDiffusionBurgers / Tokamak / Smoke              (safediffcon_amd.diffusion)
  BurgersGuidance / TokamakGuidance / SmokeGuidance       (closed-form guidance specs)
  ConformalCalculator, conformal helpers                  (safediffcon_amd.conformal)
  control_trajectories (Burgers rollout)                  (safediffcon_amd.solvers)
  KSTARSolver, control_trajectories, evaluate_samples     (safediffcon_amd.kstar: the tokamak score check)
  GraphedLossStep                                         (safediffcon_amd.train_graph: a fine-tuning step as one hipGraph)

All compute goes through libsdc_hip.so (include/sdc.h); importing this package
does not load it, the first kernel call does -- and raises if it is missing.
"""
from .unet import Unet2D, Unet1D, Unet3D_with_Conv3D                                   # noqa: F401
from .diffusion import (GaussianDiffusion, GaussianDiffusionBurgers, GaussianDiffusionTokamak,   # noqa: F401
                        GaussianDiffusionSmoke, GuidanceSpec, BurgersGuidance, TokamakGuidance, SmokeGuidance,
                        schedule_tables)
from .train_graph import GraphedLossStep                                               # noqa: F401

__version__ = "0.1.0"
