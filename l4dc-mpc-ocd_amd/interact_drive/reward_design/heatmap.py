"""Reward heat map on the GPU.  Mirrors the grid evaluation of interact_drive/visualizer.py:211-238,
264-271 (128 x 128 serial reward_fn calls there; one reward_batch launch here).  No GL."""
import numpy as np

from .._describe import describe, engine_for


def reward_heatmap(car, world, weights=None, min_coord=(-0.15, -1.0), max_coord=(0.15, 1.0), size=(128, 128)):
    """vals[j, i] = reward of the world with `car` moved to (x_i, y_j), keeping its speed and heading
    (visualizer.py:221-229), x_i / y_j on np.linspace(min + 1e-6, max - 1e-6, size) (visualizer.py:264-266).
    Returns a float32 array of shape (size[1], size[0])."""
    xs = np.linspace(min_coord[0] + 1e-6, max_coord[0] - 1e-6, size[0])
    ys = np.linspace(min_coord[1] + 1e-6, max_coord[1] - 1e-6, size[1])
    gx, gy = np.meshgrid(xs, ys)                                   # [size1, size0]
    states = np.stack([np.asarray(c.state, dtype=np.float32) for c in world.cars])
    ws = np.broadcast_to(states, (gx.size,) + states.shape).copy()
    ws[:, car.index, 0] = gx.reshape(-1)
    ws[:, car.index, 1] = gy.reshape(-1)
    w = car.weights if weights is None else np.asarray(weights, dtype=np.float32)
    eng = engine_for(describe(world, car, getattr(car, "horizon", 1)))
    _, rew = eng.reward_batch(ws, w)
    return rew.reshape(gx.shape)
