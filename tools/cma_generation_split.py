import sys, numpy as np
sys.path.insert(0,'/root/repo')
from l4dc_mpc_ocd_amd import scenarios
from l4dc_mpc_ocd_amd.interact_drive.experiments import run_mpc_ord as rmo
cfg = scenarios.BASELINE_CONFIGS[3]
m = rmo.make_mpc_ord(cfg["scenario"], horizon=cfg["horizon"], n_inits=cfg["n_inits"], seed=1)
m.optimize_cmaes(seed=1, sigma0=0.05, popsize=64, maxiter=14)
hs = m.host_split
names = list(hs)
print("gen_total_us  fitness_us |", " ".join(names))
for g in range(len(m.generation_seconds)):
    print(f"{m.generation_seconds[g]*1e6:9.1f} {m.fitness_seconds[g]*1e6:9.1f} |", " ".join(f"{hs[k][g]*1e6:7.1f}" for k in names), "| sum", f"{sum(hs[k][g] for k in names)*1e6:8.1f}")
