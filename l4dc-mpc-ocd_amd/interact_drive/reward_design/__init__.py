from .mpc_ord import MPC_ORD, finite_horizon_env  # noqa: F401
from .utils import evaluate_weights  # noqa: F401
from .value_interpolation import ValueFeature, proj_xyv, proj_xy_vertical_speed  # noqa: F401
