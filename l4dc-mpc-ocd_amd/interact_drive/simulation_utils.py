"""Car dynamics on the GPU.  Mirrors interact_drive/simulation_utils.py:9-123,321-326."""
from typing import Iterable, Union

import numpy as np

from .tensor import Tensor
from ..engine import default_ops


def car_dynamics_step(x, y, v, angle, acc, ang_vel, dt, friction):
    """simulation_utils.py:9-21 (clip acc to [-8,4], ang_vel to [-4,4], quadratic drag)."""
    st = np.stack(np.broadcast_arrays(*[np.asarray(a, dtype=np.float32) for a in (x, y, v, angle)]), axis=-1)
    shape = st.shape[:-1]
    u = np.stack(np.broadcast_arrays(np.asarray(acc, dtype=np.float32), np.asarray(ang_vel, dtype=np.float32)), axis=-1)
    u = np.broadcast_to(u, shape + (2,))
    out = default_ops().dynamics_batch(st.reshape(-1, 4), u.reshape(-1, 2), float(dt), float(friction)).reshape(shape + (4,))
    return tuple(Tensor(out[..., k]) for k in range(4))


def batched_next_car_state(state, control, dt: float, friction: float = 0.) -> Tensor:
    """simulation_utils.py:24-70."""
    state = np.asarray(state, dtype=np.float32)
    control = np.asarray(control, dtype=np.float32)
    if len(state.shape) != 2 or state.shape[1] != 4:
        raise ValueError("Input state has incorrect length {}".format(control.shape))
    if len(control.shape) != 2 or control.shape[1] != 2:
        raise ValueError("Input control has incorrect shape".format(control.shape))
    return Tensor(default_ops().dynamics_batch(state, control, float(dt), float(friction)))


def next_car_state(state: Union[np.ndarray, Iterable], control: Union[np.ndarray, Iterable],
                   dt: float, friction: float = 0.) -> Tensor:
    """simulation_utils.py:73-123: [x, y, v, angle] x [acc, angle_vel] -> next state."""
    state = np.asarray(state, dtype=np.float32)
    control = np.asarray(control, dtype=np.float32)
    if state.shape[0] != 4:
        raise ValueError("Input state has incorrect length {}".format(len(state)))
    if control.shape[0] != 2:
        raise ValueError("Input control has incorrect length {}".format(len(control)))
    return Tensor(default_ops().dynamics_batch(state[None], control[None], float(dt), float(friction))[0])


def get_dynamics_fn(friction):
    """simulation_utils.py:321-326."""
    friction = float(np.asarray(friction, dtype=np.float32))

    def next_state(state, control, dt):
        return next_car_state(state, control, dt, friction)

    return next_state
