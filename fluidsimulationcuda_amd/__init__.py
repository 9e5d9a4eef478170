"""MI355X-native Stable-Fluids step (drop-in for the vel_step + dens_step path
of ArbiterMob/FluidSimulationCuda's project/sequential).  HIP only: importing
works anywhere, computing needs libfluid_amd.so and a gfx950 GPU."""
from . import capi  # noqa: F401
from .solver import DIFF, DT, ITERS, VIS, FluidSolver, coefficients, step, step_src  # noqa: F401
