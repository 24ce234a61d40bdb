"""MI355X-native batched MPC planner for interact_drive's receding-horizon path.

The directory is named ``l4dc-mpc-ocd_amd`` (not an importable identifier); the
top-level shim ``l4dc_mpc_ocd_amd.py`` loads it under the importable name
``l4dc_mpc_ocd_amd``.

Only the hot path of avikj/L4DC-MPC-OCD lives here:
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/ocd.h)
  abi.py           ctypes mirror of the C ABI, loader (no CPU fallback)
  scenarios.py     the reference's scenario constants as descriptors
  engine.py        device buffers / streams around the C ABI (torch used for allocation only)
  interact_drive/  host-side mirror of the reference's planner / car / world / reward_design API
  sharding.py      episode sharding across ranks + the one gather per generation
"""
from . import abi, scenarios  # noqa: F401

__all__ = ["abi", "scenarios"]
