"""CMA-ES generation wall-clock through the reference-shaped API (MPC_ORD.optimize_cmaes) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from l4dc_mpc_ocd_amd.interact_drive.reward_design import MPC_ORD, finite_horizon_env
car, world, inits = finite_horizon_env(horizon=10, env_seeds=list(range(1, 9)))
m = MPC_ORD(world, car, inits, 15)
t0 = time.perf_counter()
best = m.optimize_cmaes(seed=1, sigma0=0.05, popsize=16, maxiter=12)
dt = time.perf_counter() - t0
gs = np.array(m.generation_seconds) * 1e3
print(f"CMA-ES H=10 pop 16 x 8 inits: {len(gs)} generations, generation wall-clock median {np.median(gs):.2f} ms (min {gs.min():.2f}, first {gs[0]:.2f}); "
      f"designer-weights return {m.history[0][1]:.5f} -> best {max(h[1] for h in m.history):.5f}; total {dt:.2f} s")
