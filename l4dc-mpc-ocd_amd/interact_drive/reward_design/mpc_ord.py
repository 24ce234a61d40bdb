"""MPC_ORD: the reward-design fitness loop.  Mirrors interact_drive/reward_design/mpc_ord.py:10-207.

``eval_weights`` keeps the reference's scalar signature; ``eval_population`` is the batched entry
point ([P, D] -> [P]) that one CMA-ES generation needs and the GPU is built for.  With
torch.distributed initialised (backend "nccl" = RCCL), candidates are sharded over the ranks and the
per-episode returns are all-gathered once per call (sharding.py).
"""
import pickle
import time

import numpy as np

from .cmaes import CMAES
from .. import experiments  # noqa: F401
from ..experiments._sampling import make_get_init_state
from .._describe import describe, engine_for
from ... import sharding
from ...scenarios import planner_weights_fp32, planner_weights_fp32_batch, row_dots


class list2(list):  # mutable list that can carry a .seed attribute (mpc_ord.py:10-12)
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)


def _pickle_as_reference_list2():
    """History pickles must load in the reference's own scripts (bar_plot.py:105-130, generalization_plot.py:
    167-177 unpickle `interact_drive.reward_design.mpc_ord.list2`): pickle the class under the reference's
    module path, and make that path resolve to this mirror in this process unless the real package is there."""
    import importlib
    import sys
    ref = "interact_drive.reward_design.mpc_ord"
    here = sys.modules[__name__]
    pkg_rd = sys.modules[__name__.rsplit(".", 1)[0]]
    pkg_id = sys.modules[__name__.rsplit(".", 2)[0]]
    try:
        mod = importlib.import_module(ref)          # the real reference is installed: its class wins
        if getattr(mod, "list2", None) is not list2 and mod is not here:
            return
    except Exception:
        sys.modules.setdefault("interact_drive", pkg_id)
        sys.modules.setdefault("interact_drive.reward_design", pkg_rd)
        sys.modules.setdefault(ref, here)
    list2.__module__ = ref


class MPC_ORD:
    def __init__(self, world, car, init_car_states, designer_horizon, save_path=None, num_samples=1):
        self.world = world
        self.car = car
        self.designer_horizon = designer_horizon
        self.init_car_states = init_car_states
        self.save_path = save_path
        self.designer_weights = car.weights / np.linalg.norm(car.weights)
        self.weight_dim = len(car.weights)
        self.history = list2()
        self.iter = 0
        self.should_save_history = False
        self.done = False
        self.num_samples = num_samples
        self.verbose = False
        self.last_returns = None

    # ------------------------------------------------------------------ GPU plumbing
    def _engine(self):
        pa = self.car.planner_args
        desc = describe(self.world, self.car, self.car.horizon, pa.get("learning_rate", 0.1),
                        pa.get("n_iter", 100), pa.get("extra_inits", False),
                        episode_len=self.designer_horizon, n_samples=self.num_samples,
                        designer_weights=self.designer_weights)
        return engine_for(desc)

    def _init_states_dev(self, eng, init):
        """Device copy of the init states, re-uploaded only when they change (they are the same every generation)."""
        import torch
        key = (str(eng.device), init.tobytes())
        if getattr(self, "_init_dev_key", None) != key:
            self._init_dev = torch.as_tensor(init).to(eng.device)
            self._init_dev_key = key
        return self._init_dev

    def _returns(self, inits, weights_2d, while_running=None):
        """fp32 sample rewards [P, N, S] of every (candidate, init, sample) episode; sharded when distributed.
        `while_running()` (host bookkeeping that does not need the returns) runs between launch and readback."""
        import torch
        import torch.distributed as dist
        eng = self._engine()
        if isinstance(weights_2d, np.ndarray) and weights_2d.ndim == 2:
            W = weights_2d
        else:
            W = np.asarray([np.asarray(w, dtype=np.float64).reshape(-1) for w in weights_2d])
        w32 = planner_weights_fp32_batch(W)
        init = np.ascontiguousarray(np.asarray(inits, dtype=np.float32).reshape(-1, 4))
        P, N, S = w32.shape[0], init.shape[0], self.num_samples
        init_dev = self._init_states_dev(eng, init)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            e0, e1 = sharding.episode_range(P, N, S, dist.get_world_size(), dist.get_rank())
            local = eng.rollout(init_dev, w32, ep_begin=e0, ep_end=e1, to_numpy=False)["returns"]
            if while_running is not None:
                while_running()
            full = sharding.gather_returns(local, P, N, S)
            ret = full.cpu().numpy()
        else:
            dev = eng.rollout(init_dev, w32, to_numpy=False)["returns"]
            if while_running is not None:
                while_running()
            ret = dev.cpu().numpy()                                # synchronises with the launch stream
        # keep world.reset() side effects in step with the reference (ReplanningCarWorld toggles per reset)
        if hasattr(self.world, "unlucky_car_idx") and (P * N * S) % 2:
            self.world.unlucky_car_idx = 2 if self.world.unlucky_car_idx == 1 else 1
        return ret.reshape(P, N, S)

    # ------------------------------------------------------------------ reference API
    def eval_weights_for_init(self, init, weights, render, heatmap_show=False):
        """Designer return of one init (summed over samples), mpc_ord.py:67-106."""
        if render:
            raise NotImplementedError("rendering is outside the accelerated planner path")
        weights = np.asarray(weights)
        if weights.ndim == 2:
            weights = weights[0]
        r = self._returns([init], [weights])[0, 0]
        designer_reward = np.float32(0)
        for s in range(self.num_samples):
            designer_reward = np.float32(designer_reward + r[s])
        self.car.weights = weights / np.linalg.norm(weights)
        self.car.init_state = type(self.car.state)(init)
        return designer_reward

    def eval_weights_for_inits(self, weights_2d, inits):
        """[W, D] weight vectors x [N, 4] init states -> [W, N] designer returns (fp32, summed over
        samples): the held-out generalisation sweep of experiments/generalization_data.py:64-107
        (eval_weights_for_init for every (test_init, chosen_weights) pair) as one launch."""
        W = np.asarray(weights_2d, dtype=np.float64).reshape(-1, self.weight_dim)
        ret = self._returns(inits, W)                                  # [W, N, S]
        out = np.zeros(ret.shape[:2], dtype=np.float32)
        for s in range(ret.shape[2]):
            out = (out + ret[:, :, s]).astype(np.float32)
        return out

    def eval_population(self, weights_2d):
        """[P, D] candidate weights -> [P] costs (-expected designer return), one launch per rank."""
        W = np.asarray(weights_2d, dtype=np.float64).reshape(-1, self.weight_dim)
        hist = {}

        def normalised_for_history():                           # runs while the GPU works
            hist["Wn"] = W / np.sqrt(row_dots(W))[:, None]      # np.linalg.norm(row) == sqrt(row.dot(row))

        ret = self._returns(self.init_car_states, W, while_running=normalised_for_history)
        self.last_returns = ret
        P, N, S = ret.shape
        cost = sharding.fitness_from_returns(ret.reshape(-1), P, N, S)
        self.history.extend(zip(hist["Wn"], -cost))
        self.iter += P
        if self.should_save_history and self.save_path is not None:
            self.save_history()
        return cost

    def eval_weights(self, weights, gif=None, heatmap_show=False):
        """The CMA-ES fitness callable of the reference (mpc_ord.py:109-151): one candidate -> cost."""
        if gif:
            raise NotImplementedError("gif rendering is outside the accelerated planner path")
        if isinstance(weights, list):
            weights = np.array(weights)
        if weights.ndim == 2:
            weights = weights[0]
        if self.verbose:
            print('ITERATION', self.iter)
            print('eval', weights / np.linalg.norm(weights))
        return float(self.eval_population(weights[None])[0])

    def optimize_cmaes(self, seed=1, sigma0=0.1, popsize=None, maxiter=None, maxfevals=None):
        """mpc_ord.py:33-45, with whole generations evaluated per launch."""
        self.history.seed = seed
        assert seed != 0
        assert not self.done
        self.should_save_history = True
        self.eval_weights(self.designer_weights)                       # "Iteration 0" baseline
        es = CMAES(list(self.designer_weights), sigma0, popsize=popsize, seed=seed)
        self.generation_seconds = []
        self.fitness_seconds = []
        while True:
            t0 = time.perf_counter()                               # a generation: ask, fitness of the population, tell
            X = es.ask()
            t1 = time.perf_counter()
            f = self.eval_population(X)
            self.fitness_seconds.append(time.perf_counter() - t1)
            es.tell(X, f)
            self.generation_seconds.append(time.perf_counter() - t0)
            if es.stop(maxiter=maxiter, last_fitness=f) or (maxfevals and es.counteval >= maxfevals):
                break
        self.should_save_history = False
        self.done = True
        self.es = es
        return es.best_x

    def optimize_random_search(self, n_iter=1000, seed=1):
        """mpc_ord.py:47-65: same candidate stream (np.random.rand under np.random.seed), one launch."""
        self.history.seed = seed
        assert not self.done
        self.should_save_history = True
        self.iter = 0
        self.eval_weights(self.designer_weights)
        np.random.seed(seed)
        W = np.stack([np.random.rand(*self.designer_weights.shape) * 2 - 1 for _ in range(n_iter)])
        self.eval_population(W)
        self.should_save_history = False
        self.done = True
        return max(self.history, key=lambda a: a[1])

    def save_history(self):
        assert self.save_path is not None
        with open(self.save_path, 'wb') as file:
            pickle.dump(self.history, file)


_pickle_as_reference_list2()


def finite_horizon_env(horizon=5, env_seeds=[1], debug=True, extra_inits=False):
    """The finite_horizon scenario factory of the reference's mpc_ord.py, built from scenarios.finite_horizon."""
    from ..experiments._build import world_from_scenario
    from ... import scenarios
    scn = scenarios.finite_horizon(horizon=horizon, extra_inits=extra_inits)
    dist = scn.init_dist
    init_states = [make_get_init_state(dist.x, dist.y, dist.v)(s) for s in env_seeds]
    our_car, _, world = world_from_scenario(scn, init_states[0], debug=debug,
                                            visualizer_args=dict(name="Switch Lanes"))
    return our_car, world, init_states
