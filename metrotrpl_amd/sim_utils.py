"""Grid value object -- same fields and errors as the reference's
sim_utils.Grid (sim_utils.py:246-277)."""
from sys import float_info

import numpy as np

# sim_utils.py:17-23
DEFAULT_HMAX = 4
DEFAULT_TEMPER_FREQ = 10
NEGATIVE_FRAC_TOL = 0.2


class Grid:
    """Collection of values describing the grid dimensions for the simulation."""

    def __init__(self, thickness, nx, tSteps, hmax):
        self.thickness = thickness
        self.nx = nx
        self.dx = self.thickness / self.nx
        self.xSteps = np.linspace(self.dx / 2, self.thickness - self.dx / 2, self.nx)
        if tSteps[0] != 0:
            raise ValueError("Grid error - times must start at t=0")
        self.tSteps = tSteps
        self.start_time = 0
        self.nt = len(tSteps) - 1
        self.hmax = hmax
        self.final_time = self.tSteps[-1]
        self.min_y = float_info.min
