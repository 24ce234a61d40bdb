"""Host-side mirror of the reference's ``interact_drive`` API for the planner path.

Same class names, argument meaning and error behaviour as the reference
(interact_drive/{world,car/*,planner/*,simulation_utils,reward_design/mpc_ord}.py
and the scenario factories under experiments/), so code written against the
reference runs unchanged; every computation is a call into the HIP C-ABI
library (no TensorFlow, no CPU fallback).  Values come back as ``Tensor``
objects: float32 ``numpy.ndarray`` views with the ``.numpy()`` method the
reference's callers use on ``tf.Tensor``.
"""
from .tensor import Tensor, constant  # noqa: F401
from . import simulation_utils, world  # noqa: F401
