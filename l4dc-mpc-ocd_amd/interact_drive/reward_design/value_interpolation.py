"""`ValueFeature`: trilinear interpolation of a value table, usable as NaivePlanner's `leaf_evaluation`.
Mirrors interact_drive/reward_design/value_interpolation.py:5-61.

The reference takes an arbitrary Python `proj(world_state)`; the compiled planner knows the two coarse
states of the reference's own value iteration (coarse_value_iteration.py:117-124): `proj_xyv` =
(x, y, v) of the planning car and `proj_xy_vertical_speed` = (x, y, v * sin(heading)).
"""
import numpy as np

from ..tensor import Tensor


def proj_xyv(world_state):
    s = np.asarray(world_state[0], dtype=np.float32)
    return s[:3]


def proj_xy_vertical_speed(world_state):
    s = np.asarray(world_state[0], dtype=np.float32)
    return np.array([s[0], s[1], np.float32(s[2] * np.float32(np.sin(np.float32(s[3]))))], dtype=np.float32)


_PROJ_KINDS = {proj_xyv: 0, proj_xy_vertical_speed: 1}


class LeafEvaluation:
    """What ValueFeature.interpolate_value(t) returns: the table slice the planner kernel reads for the
    last horizon step (naive_planner.py:69-70).  Calling it evaluates the interpolation on the GPU."""

    def __init__(self, disc_grid, values, proj_kind, world_car=None):
        self.disc_grid = [np.ascontiguousarray(g, dtype=np.float32) for g in disc_grid]
        self.values = np.ascontiguousarray(values, dtype=np.float32)
        self.proj_kind = int(proj_kind)
        if self.values.shape != tuple(len(g) for g in self.disc_grid):
            raise ValueError("v_grids[t] must have one axis per disc_grid row")

    def key(self):
        return (self.proj_kind, tuple(g.tobytes() for g in self.disc_grid), self.values.tobytes())


class ValueFeature:
    def __init__(self, proj, value_data, dim=3, scale=1):
        assert dim == 3
        if proj not in _PROJ_KINDS:
            raise NotImplementedError("only proj_xyv / proj_xy_vertical_speed are compiled into the planner kernel")
        self.proj = proj
        self.dims = dim
        self.scale = scale                      # stored and unused, as in the reference
        self.disc_grid = [np.asarray(row, dtype=np.float32) for row in value_data['disc_grid']]
        self.v_grids = np.asarray(value_data['v_grids'], dtype=np.float32)

    def interpolate_value(self, t=0) -> LeafEvaluation:
        return LeafEvaluation(self.disc_grid, self.v_grids[t], _PROJ_KINDS[self.proj])
