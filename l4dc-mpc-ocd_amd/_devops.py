"""Scenario-independent device calls used by the host-side API (lazy: no GPU work at import time)."""


def dynamics_batch(states, controls, dt, friction):
    from .engine import default_ops
    return default_ops().dynamics_batch(states, controls, float(dt), float(friction))
