"""MPC_ORD: the reward-design fitness loop.  Mirrors interact_drive/reward_design/mpc_ord.py:10-207.

``eval_weights`` keeps the reference's scalar signature; ``eval_population`` is the batched entry
point ([P, D] -> [P]) that one CMA-ES generation needs and the GPU is built for.  With
torch.distributed initialised (backend "nccl" = RCCL), candidates are sharded over the ranks and the
per-episode returns are all-gathered once per call (sharding.py).
"""
import copyreg
import os
import pickle
import time

import numpy as np

from .cmaes import CMAES, NativeCMAES, fitness_from_returns_native  # noqa: F401
from .. import experiments  # noqa: F401
from ..experiments._sampling import make_get_init_state
from .._describe import describe, engine_for
from ... import abi, sharding
from ...scenarios import planner_weights_fp32, planner_weights_fp32_batch, row_dots


_REF_MODULE = "interact_drive.reward_design.mpc_ord"


class _ModuleByName:
    """Pickles as importlib.import_module(name): a module reference that needs nothing of this package to load."""

    def __init__(self, name):
        self.name = name

    def __reduce__(self):
        import importlib
        return importlib.import_module, (self.name,)


class _PickledUnderReferencePath(type):
    """Metaclass of list2: the CLASS pickles as getattr(import_module('interact_drive.reward_design.mpc_ord'),
    'list2') (registered with copyreg below; pickle consults the dispatch table for classes with a custom metaclass
    before it falls back to a by-name global).  History pickles therefore name the reference's module path -- its
    own scripts load them (bar_plot.py:105-130, generalization_plot.py:167-177) -- and dumping imports nothing and
    registers nothing in sys.modules."""


def _reduce_list2_class(cls):
    return getattr, (_ModuleByName(_REF_MODULE), cls.__name__)


copyreg.pickle(_PickledUnderReferencePath, _reduce_list2_class)


class list2(list, metaclass=_PickledUnderReferencePath):  # mutable list that can carry a .seed attribute (mpc_ord.py:10-12)
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)


class _ReferencePathFinder:
    """LOADING such a pickle in a process that has this package but not the reference: a finder at the END of
    sys.meta_path resolves `interact_drive`, `interact_drive.reward_design` and `...mpc_ord` -- the two parents as
    empty stub packages, the leaf as this module -- only when no other finder did, i.e. only when the real
    `interact_drive` is not importable.  Where it is (side-by-side comparison, the reference's plot scripts) the
    real package wins and nothing here is registered; until somebody imports one of the three names nothing is
    registered either.  The stubs hold no other name: `import interact_drive.car` still raises
    ModuleNotFoundError instead of loading a second copy of the mirror under another name."""

    _names = ("interact_drive", "interact_drive.reward_design", _REF_MODULE)

    @classmethod
    def find_spec(cls, fullname, path=None, target=None):
        if fullname not in cls._names:
            return None
        import importlib.machinery
        return importlib.machinery.ModuleSpec(fullname, cls, is_package=fullname != _REF_MODULE)

    @staticmethod
    def create_module(spec):
        import sys
        import types
        if spec.name == _REF_MODULE:
            return sys.modules[__name__]
        stub = types.ModuleType(spec.name, "stub package: only names the reference path of MPC_ORD history pickles")
        stub.__path__ = []
        stub._ocd_pickle_stub = True
        return stub

    @staticmethod
    def exec_module(module):
        pass


def _install_reference_path_finder():
    import sys
    if not any(f is _ReferencePathFinder for f in sys.meta_path):
        sys.meta_path.append(_ReferencePathFinder)


class MPC_ORD:
    def __init__(self, world, car, init_car_states, designer_horizon, save_path=None, num_samples=1):
        self.world = world
        self.car = car
        self.designer_horizon = designer_horizon
        self.init_car_states = init_car_states
        self.save_path = save_path
        self.designer_weights = car.weights / np.linalg.norm(car.weights)
        self.weight_dim = len(car.weights)
        self._history = list2()
        self._history_chunks = []        # (weights [G, P, D], costs [G, P]) blocks of the native loops, not yet in the list
        self.iter = 0
        self.should_save_history = False
        self.done = False
        self.num_samples = num_samples
        self.verbose = False
        self.last_returns = None
        self.host_split = {}
        self._pending_history = None
        self._defer_history = False
        self._overlap_hooks = []
        self.n_nonfinite = []            # per eval_population call: costs that came back NaN or +-inf
        self.max_nan_resamples = 10      # rounds of pycma-style rejection sampling per generation

    # the (weights, reward) list of the reference (mpc_ord.py:27,146).  The native generation loops hand their rows over
    # as arrays, many generations at a time; they become list entries when somebody looks (len, iteration, pickling) --
    # 64 tuples per generation built inside the loop cost 35 us of a 1.5 ms generation
    @property
    def history(self):
        if self._history_chunks:
            chunks, self._history_chunks = self._history_chunks, []
            for W, cost in chunks:
                for g in range(W.shape[0]):
                    self._history.extend(zip(W[g], -cost[g]))
        return self._history

    @history.setter
    def history(self, value):
        self._history_chunks = []
        self._history = value

    # ------------------------------------------------------------------ GPU plumbing
    def _engine(self):
        pa = self.car.planner_args
        desc = describe(self.world, self.car, self.car.horizon, pa.get("learning_rate", 0.1),
                        pa.get("n_iter", 100), pa.get("extra_inits", False),
                        episode_len=self.designer_horizon, n_samples=self.num_samples,
                        designer_weights=self.designer_weights)
        return engine_for(desc)

    def _init_key_array(self, inits):
        """[N, 4] fp32 contiguous init states.  Inside optimize_cmaes (the loop owns world, car and init states for its
        duration: _eng_fixed) the conversion of the same list object is done once; every other caller gets a fresh
        conversion, so init states edited between two eval_weights calls are always seen."""
        c = getattr(self, "_init_np_cache", None)
        if getattr(self, "_eng_fixed", None) is not None and c is not None and c[0] is inits:
            return c[1]
        arr = np.ascontiguousarray(np.asarray(inits, dtype=np.float32).reshape(-1, 4))
        self._init_np_cache = (inits, arr)
        return arr

    def _init_states_dev(self, eng, init):
        """Device copy of the init states, re-uploaded only when they change (they are the same every generation)."""
        import torch
        key = (str(eng.device), init.tobytes())
        if getattr(self, "_init_dev_key", None) != key:
            self._init_dev = torch.as_tensor(init).to(eng.device)
            self._init_dev_key = key
        return self._init_dev

    def _staging(self, eng, P, N, S, D, n_local):
        """Buffers of one (device, shape), allocated once.  The candidate weights and the returns live in PINNED
        host memory that the GPU addresses directly (HIP maps pinned allocations into the device's address space):
        the kernel reads its 7 weights per trajectory and writes its one return per episode over the link itself,
        so a single-process generation is one launch and one event wait -- no copy calls.  Sharded runs keep a
        device buffer for the returns, which the RCCL all-gather needs."""
        import torch
        key = (str(eng.device), P, N, S, D, n_local)
        cache = self.__dict__.setdefault("_stages", {})            # a resampled sub-population has its own shape
        st = cache.get(key)
        if st is None:
            if len(cache) >= 8:
                cache.clear()
            st = dict(key=key,
                      w_host=torch.empty((P, D), dtype=torch.float32).pin_memory(),
                      ret_dev=torch.empty((n_local,), dtype=torch.float32, device=eng.device),
                      ret_host=torch.empty((P * N * S,), dtype=torch.float32).pin_memory(),
                      done=torch.cuda.Event())
            st["w_np"] = st["w_host"].numpy()
            st["ret_np"] = st["ret_host"].numpy()
            st["w_ptr"] = st["w_host"].data_ptr()
            st["ret_dev_ptr"] = st["ret_dev"].data_ptr()
            st["ret_host_ptr"] = st["ret_host"].data_ptr()
            cache[key] = st
        return st

    def _flush_history(self):
        """Append the last generation's (normalised weights, reward) entries (mpc_ord.py:146).  Inside
        optimize_cmaes this runs while the NEXT generation's kernel does; everywhere else immediately."""
        if self._pending_history is not None:
            Wn, cost = self._pending_history
            self.history.extend(zip(Wn, -cost))                    # (the property first lists what the native loops left)
            self._pending_history = None

    def _tick(self, name, t0):
        """Accumulate host-side wall time of one segment of a generation (bench.py reports the medians)."""
        t1 = time.perf_counter()
        self.host_split.setdefault(name, []).append(t1 - t0)
        return t1

    def host_split_ms(self):
        """Median milliseconds per generation of each host-side segment over the last 32 generations of the run (the
        first ones pay allocations and run on a GPU whose clocks are still rising)."""
        return {k: float(np.median(v[-32:]) * 1e3) for k, v in self.host_split.items()}

    def _returns(self, inits, weights_2d, while_running=None):
        """fp32 sample rewards [P, N, S] of every (candidate, init, sample) episode; sharded when distributed.
        `while_running()` (host bookkeeping that does not need the returns) runs between launch and readback."""
        import torch
        import torch.distributed as dist
        t = time.perf_counter()
        eng = getattr(self, "_eng_fixed", None) or self._engine()
        if isinstance(weights_2d, np.ndarray) and weights_2d.ndim == 2:
            W = weights_2d
        else:
            W = np.asarray([np.asarray(w, dtype=np.float64).reshape(-1) for w in weights_2d])
        if torch.cuda.current_device() != eng.device.index:        # rare: run the same body under the engine's device
            with torch.cuda.device(eng.device):
                return self._returns(inits, weights_2d, while_running)
        init = self._init_key_array(inits)
        P, N, S, D = W.shape[0], init.shape[0], self.num_samples, W.shape[1]
        init_dev = self._init_states_dev(eng, init)
        sharded = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                   and not getattr(self, "_local_only", False))      # (lockstep over RUNS shards the runs, not a population)
        e0, e1 = sharding.episode_range(P, N, S, dist.get_world_size(), dist.get_rank()) if sharded else (0, P * N * S)
        st = self._staging(eng, P, N, S, D, e1 - e0)
        # three float64 normalisations + the fp32 cast, written straight into the pinned rows the kernel reads
        planner_weights_fp32_batch(W, out=st["w_np"])
        t = self._tick("normalise", t)
        stream = torch.cuda.current_stream()
        abi.check(eng.lib, eng.lib.ocd_rollout_episodes(
            eng._h, init_dev.data_ptr(), st["w_ptr"], P, N, e0, e1,
            st["ret_dev_ptr"] if sharded else st["ret_host_ptr"], None, None, stream.cuda_stream))
        t = self._tick("launch", t)
        if while_running is not None:
            while_running()
        t = self._tick("overlapped_bookkeeping", t)
        if sharded:
            full = sharding.gather_returns(st["ret_dev"], P, N, S)
            st["ret_host"].copy_(full, non_blocking=full.is_cuda)   # (gloo rehearsal: gathered on the host already)
        st["done"].record(stream)
        st["done"].synchronize()                                   # the launch (and the gather and its copy) are done
        ret = st["ret_np"].copy()
        t = self._tick("kernel_gather_readback", t)
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

        def bookkeeping():                                      # runs while the GPU works
            self._flush_history()                               # the previous generation's entries (deferred)
            for hook in self._overlap_hooks:
                hook()
            hist["Wn"] = W / np.sqrt(row_dots(W))[:, None]      # np.linalg.norm(row) == sqrt(row.dot(row))

        ret = self._returns(self.init_car_states, W, while_running=bookkeeping)
        t = time.perf_counter()
        self.last_returns = ret
        P, N, S = ret.shape
        cost = fitness_from_returns_native(ret.reshape(-1), P, N, S)    # == sharding.fitness_from_returns, one C call
        # a runaway trajectory (the dynamics have no speed floor) scores inf or NaN: counted, never hidden
        self.n_nonfinite.append(P - int(np.isfinite(cost).sum()))
        self._pending_history = (hist["Wn"], cost)
        if not self._defer_history:
            self._flush_history()
        self.iter += P
        self._tick("reduce", t)
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

    # pycma's CMA_active option (default True: negative recombination weights in the rank-mu update, tutorial eqs. 46-53)
    cma_active = True

    def optimize_cmaes(self, seed=1, sigma0=0.1, popsize=None, maxiter=None, maxfevals=None, termination=None):
        """mpc_ord.py:33-45, with whole generations evaluated per launch.

        Around the sampler this follows what pycma's fmin2 does with the reference's fitness callable: a candidate
        whose cost is NaN is rejected and redrawn (ask_and_eval; at most `max_nan_resamples` rounds per generation,
        each one small launch -- what is still NaN then ranks last), every evaluation lands in the history, and the
        run ends on pycma's default termination rules (cmaes._Termination: maxiter = 100 + 150 (N+3)^2 / sqrt(popsize)
        unless `maxiter` is given, tolfun, tolfunhist, tolx, tolstagnation, condition of C, ...; `termination`: a dict
        of option overrides, e.g. {"tolfacupx": inf} -- the cost is invariant to the scale of the weights
        (mpc_ord.py:120 normalises them), so the step size drifts upwards and `tolfacupx` is what usually ends a long
        run).  `self.stop_reason` keeps the satisfied conditions, `self.n_nonfinite` the non-finite costs per
        evaluation call."""
        self.history.seed = seed
        assert seed != 0
        assert not self.done
        self.should_save_history = True
        self.eval_weights(self.designer_weights)                       # "Iteration 0" baseline
        es = NativeCMAES(list(self.designer_weights), sigma0, popsize=popsize, seed=seed, active=self.cma_active)   # csrc/ocd_cma.c
        self.generation_seconds = []                               # per generation: ask ... termination test
        self.generation_wall_seconds = []                          # the same plus the interpreter's bookkeeping (history rows,
        self.fitness_seconds = []                                  #   counters): wall-clock of the loop / generations
        self.host_split = {}
        self.n_resampled = 0
        self.stop_reason = {}
        self._eng_fixed = self._engine()                           # world and car do not change inside the loop
        self._defer_history = self.save_path is None               # (a saved history must be complete at every dump)
        self._overlap_hooks = [es.prepare]                          # the next population's deviates, while the GPU works
        overrides = dict(maxiter=maxiter, maxfevals=maxfevals, **(termination or {}))
        try:
            if self._native_loop_possible(es):
                self._optimize_cmaes_native(es, overrides)
            else:
                while True:
                    t0 = time.perf_counter()                       # a generation: ask, fitness of the population, tell
                    X = es.ask()
                    t1 = self._tick("ask", t0)
                    f = self.eval_population(X)
                    if np.isnan(f).any():
                        f = self._resample_nan(es, X, f)
                    t2 = time.perf_counter()
                    self.fitness_seconds.append(t2 - t1)
                    es.tell(X, f)
                    t3 = self._tick("tell", t2)
                    why = es.stop(**overrides)
                    self._tick("stop", t3)
                    self.generation_seconds.append(time.perf_counter() - t0)     # ask ... termination test
                    self.generation_wall_seconds.append(self.generation_seconds[-1])
                    if why:
                        self.stop_reason = why
                        break
        finally:                                                   # also on an exception or Ctrl-C inside a long run
            self._eng_fixed = None
            self._init_np_cache = None
            self._defer_history = False
            self._overlap_hooks = []
            self._flush_history()
        self.should_save_history = False
        self.done = True
        self.es = es
        return es.best_x

    # ------------------------------------------------------------------ generations in native code
    def _native_loop_possible(self, es):
        """One process, no history file to keep complete after every evaluation, and a native normalisation that
        reproduces numpy's on this machine: then whole generations run inside csrc/ocd_cma.c (ocd_cma_run)."""
        import torch.distributed as dist
        from ...scenarios import _native_normalise_variant
        if self.save_path is not None or getattr(self, "force_python_loop", False):
            return False
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not getattr(self, "_local_only", False):
            return False
        return hasattr(es, "run") and _native_normalise_variant(self.weight_dim) is not None

    native_chunk = 64                                                # generations per ocd_cma_run call

    def _optimize_cmaes_native(self, es, overrides, chunk=None):
        """The loop of optimize_cmaes through ocd_cma_run: ask, normalise into the pinned rows, launch, (next deviates
        and history rows while the GPU works,) wait, reduce, tell, termination test -- generation after generation
        without returning to the interpreter; Python only books the results every `chunk` generations and handles the
        rare generation with a NaN cost (pycma's rejection sampling) itself."""
        import ctypes as C
        import torch
        from .cmaes import RunArgs
        from ...scenarios import _native_normalise_variant
        eng = self._eng_fixed
        chunk = int(chunk or self.native_chunk)
        with torch.cuda.device(eng.device):
            init = self._init_key_array(self.init_car_states)
            P, N, S, D = es.lam, init.shape[0], self.num_samples, self.weight_dim
            E = P * N * S
            init_dev = self._init_states_dev(eng, init)
            st = self._staging(eng, P, N, S, D, E)
            hist_w = np.empty((chunk, P, D), dtype=np.float64)
            hist_c = np.empty((chunk, P), dtype=np.float64)
            secs = np.zeros((chunk, 8), dtype=np.float64)
            nonf = np.zeros(chunk, dtype=np.int32)
            a = RunArgs()
            a.scn, a.init_dev, a.N, a.S = eng._h.value, init_dev.data_ptr(), N, S
            a.w_pinned, a.ret_pinned = st["w_ptr"], st["ret_host_ptr"]
            a.stream = torch.cuda.current_stream().cuda_stream
            a.rollout = C.cast(eng.lib.ocd_rollout_episodes, C.c_void_p).value
            a.sync = C.cast(eng.lib.ocd_stream_synchronize, C.c_void_p).value
            a.normalise_variant = _native_normalise_variant(D)
            a.max_generations = chunk
            a.hist_w, a.hist_cost = hist_w.ctypes.data, hist_c.ctypes.data
            a.seconds, a.nonfinite = secs.ctypes.data, nonf.ctypes.data
            names = ("ask", "normalise", "launch", "overlapped_bookkeeping", "kernel_gather_readback", "reduce", "tell")
            toggles = hasattr(self.world, "unlucky_car_idx") and (E % 2)

            def book(g0, g1):                                      # generations [g0, g1) of this call's buffers
                if g1 > g0:
                    self._history_chunks.append((hist_w[g0:g1].copy(), hist_c[g0:g1].copy()))   # (mpc_ord.py:146, lazily)
                    self.iter += P * (g1 - g0)
                    if toggles and (g1 - g0) % 2:                  # world.reset() side effects, as _returns keeps them
                        self.world.unlucky_car_idx = 2 if self.world.unlucky_car_idx == 1 else 1

            while True:
                t_chunk = time.perf_counter()
                done, why, pending = es.run(a, overrides)
                book(0, done)
                if done:                                           # (whole columns at a time: no interpreter work per generation)
                    self.generation_seconds.extend(secs[:done, 0].tolist())
                    self.fitness_seconds.extend(secs[:done, 2:7].sum(axis=1).tolist())
                    self.n_nonfinite.extend(nonf[:done].tolist())
                    for k, name in enumerate(names):
                        self.host_split.setdefault(name, []).extend(secs[:done, 1 + k].tolist())
                if done:        # what the C timers do not see: this call's share of the interpreter (history, counters)
                    self.generation_wall_seconds.extend([(time.perf_counter() - t_chunk) / done] * done)
                if pending:                                        # generation `done` is evaluated, not told: redraw its NaN
                    t0 = time.perf_counter()
                    book(done, done + 1)
                    f = es._f.copy()
                    self.n_nonfinite.append(P - int(np.isfinite(f).sum()))
                    f = self._resample_nan(es, es._X, f)
                    self._flush_history()                          # (the redrawn candidates' entries, in evaluation order)
                    es.tell(es._X, f)
                    why = es.stop(**overrides)
                    self.generation_seconds.append(float(secs[done, 0]) + time.perf_counter() - t0)
                    self.generation_wall_seconds.append(self.generation_seconds[-1])
                    self.fitness_seconds.append(self.generation_seconds[-1])
                if why:
                    self.stop_reason = why
                    break
            self.last_returns = st["ret_np"].copy().reshape(P, N, S)

    def _resample_nan(self, es, X, f):
        """pycma's rejection sampling (ask_and_eval behind mpc_ord.py:41): redraw and re-evaluate the candidates
        whose cost is NaN; +-inf costs are kept (pycma keeps them too) and rank where they fall."""
        f = np.array(f, dtype=np.float64)
        for _ in range(self.max_nan_resamples):
            rows = np.nonzero(np.isnan(f))[0]
            if rows.size == 0:
                break
            Xr = es.resample(rows)
            self.n_resampled += int(rows.size)
            es.add_evals(rows.size)                                # pycma counts the rejected evaluations too
            f[rows] = self.eval_population(Xr)
        return f

    def optimize_cmaes_many(self, runs, popsize=None, maxiter=None, maxfevals=None, termination=None, save_paths=None,
                            groups=None, host_threads=None, chunk=None):
        """R independent optimize_cmaes runs over this world and car -- `runs` = [(init_states, seed, sigma0), ...], what
        the reference hands to a multiprocessing.Pool, one process per init group (run_mpc_ord.py:83-90) -- advanced in
        LOCKSTEP with one episode launch per generation (optimize_cmaes_lockstep).  Returns a LockstepResult; its `.runs`
        are R MPC_ORD objects, each with the history, stop_reason, es and counters the run would have alone."""
        ords = []
        for k, (init_states, _seed, _sigma0) in enumerate(runs):
            ords.append(MPC_ORD(self.world, self.car, init_states, self.designer_horizon,
                                save_path=None if save_paths is None else save_paths[k], num_samples=self.num_samples))
        return optimize_cmaes_lockstep(ords, [r[1] for r in runs], [r[2] for r in runs], popsize=popsize, maxiter=maxiter,
                                       maxfevals=maxfevals, termination=termination, groups=groups, host_threads=host_threads,
                                       chunk=chunk)

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
        # uniform weights in [-1, 1] can reward leaving the road: such an episode scores NaN / -inf.  The
        # reference's max() over a history holding NaN depends on where the NaN sits (every comparison with it is
        # False); here the best FINITE entry is returned and the others are counted in self.n_nonfinite.
        finite = [h for h in self.history if np.isfinite(h[1])]
        return max(finite or self.history, key=lambda a: a[1])

    def save_history(self):
        assert self.save_path is not None
        self._flush_history()
        with open(self.save_path, 'wb') as file:
            pickle.dump(self.history, file)


_install_reference_path_finder()


class LockstepResult:
    """What optimize_cmaes_lockstep returns: `runs` (the MPC_ORD objects, in order), `best` (their best candidates),
    `lockstep` (True: one launch per generation for all runs; False: the runs were made one after another),
    `generation_seconds` / `generation_wall_seconds` (per lockstep generation: native timers / including the
    interpreter's bookkeeping), `episodes_per_generation` (episodes of each generation's launch), `host_split`
    (seconds per segment and generation, as MPC_ORD.host_split), `launch` (what the last launch chose); under
    torch.distributed `ranks` and `per_rank` (the timing fields above are then the slowest rank's)."""

    def __init__(self, runs):
        self.runs, self.best, self.lockstep = runs, [], True
        self.generation_seconds, self.generation_wall_seconds, self.episodes_per_generation = [], [], []
        self.host_split, self.launch = {}, None
        self.groups = 1                          # launches per generation (csrc/ocd_cma.c: groups of runs on their own streams)
        self.host_threads = 1                    # threads that shared the tells of a generation (results do not depend on it)
        self.ranks, self.per_rank = 1, None      # torch.distributed: ranks the runs were dealt over, each rank's timings

    def host_split_ms(self):
        return {k: float(np.median(v[-32:]) * 1e3) for k, v in self.host_split.items()}


def optimize_cmaes_lockstep(ords, seeds, sigma0s, popsize=None, maxiter=None, maxfevals=None, termination=None, chunk=None,
                            groups=None, host_threads=None):
    """See _lockstep_local (one process) and _lockstep_over_ranks (torch.distributed): the public entry point.
    groups: launches per generation (None: _lockstep_groups -- four, else two, while every group's launch is a latency build within
    its share of the chip, else one).
    host_threads: threads that share the runs' tells inside a native call (None: _lockstep_host_threads; 1: the caller's only).
    chunk: generations per native call (None: up to 128 while the call's history block stays under 4 MB; the interpreter
    books the results between calls -- ~1 ms for 28 runs, i.e. 0.03 ms per generation at 32, 0.01 at 128)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and \
            not any(getattr(o, "_local_only", False) for o in ords):
        return _lockstep_over_ranks(ords, seeds, sigma0s, popsize, maxiter, maxfevals, termination, chunk, groups, host_threads)
    return _lockstep_local(ords, seeds, sigma0s, popsize, maxiter, maxfevals, termination, chunk, groups, host_threads)


def _lockstep_groups(eng, run_episodes, cus=None):
    """Launches per generation: the runs go out as G groups on G streams while every group's launch -- planned for its 1 / G
    of the compute units ("concurrent_launches") -- is a latency build that fits that share one wavefront per SIMD: the
    groups then sit side by side on the chip, each cycles wait -> tell -> ask -> launch by itself, and one group's host work
    hides under the others' kernels.  Four groups where that holds (HIP's four hardware queues: a fifth stream shares one and
    its launches queue behind another group's, profiles/r06_lockstep_streams.txt), else two, else one launch."""
    R = len(run_episodes)
    if cus is None:
        import torch
        cus = torch.cuda.get_device_properties(eng.device).multi_processor_count
    for G in (4, 2):
        if R < G:
            continue
        share = cus // G
        ok = share >= 1
        for k in range(G):
            e = int(sum(run_episodes[R * k // G:R * (k + 1) // G]))
            plan = eng.plan_launch(e, share) if ok else None
            per_wg = plan["wavefronts_per_workgroup"] if plan else 1
            room = 4 * share if per_wg == 1 else share * (4 // per_wg)
            if not plan or plan["build_wavefronts_per_simd"] != 1 or plan["workgroups"] > room:
                ok = False                                             # (not a latency build: nothing is claimed; or no room)
                break
        if ok:
            return G
    return 1


def _lockstep_host_threads(n_runs):
    """Threads of one ocd_cma_run_many call (csrc/ocd_cma.c, ABI 8): the tells of a generation are independent and, at the
    reference's shape, the largest host item between a kernel's end and the next launch (28 x 3.9 us after a 1.15 ms
    kernel).  At most four, at most one per four runs, never more than this process's share of the cores less one
    (torchrun: LOCAL_WORLD_SIZE ranks share the node); OCD_CMA_THREADS overrides.  Results do not depend on it."""
    def _int(name, default):
        try:
            return int(os.environ.get(name, "").strip() or default)
        except ValueError:
            return default
    forced = _int("OCD_CMA_THREADS", 0)
    if forced > 0:
        return forced
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    share = max(1, cores // max(1, _int("LOCAL_WORLD_SIZE", 1)))
    return max(1, min(4, share - 1, n_runs // 4))


def _lockstep_local(ords, seeds, sigma0s, popsize=None, maxiter=None, maxfevals=None, termination=None, chunk=None, groups=None,
                    host_threads=None):
    """MPC_ORD.optimize_cmaes (mpc_ord.py:33-45) for R MPC_ORD objects over the SAME world, car and planner -- their init
    states, seeds and step sizes differ -- with the generation's episodes of ALL runs in one launch.

    The reference's only parallel axis is a process pool over such runs (run_mpc_ord.py:83-90 --one_by_one; 28 chosen
    weights in generalization_data.py:78-84).  One run of the reference's shape (popsize 9 x 3 inits = 27 episodes) keeps
    81 of the chip's 1 024 SIMDs busy for ~1.1 ms per generation; R runs in lockstep take the same wall time per
    generation until the launch outgrows one wavefront per SIMD.  Each run makes exactly the calls it makes alone
    (csrc/ocd_cma.c: ocd_cma_run_many vs ocd_cma_run -- ask, normalise, prepare, tell, stop on its own state and random
    stream; NaN costs redrawn by the same Python code), so histories, pickles, stop reasons and counters are bit for
    bit those of the runs made one after another; a run that stops drops out of the launch.

    Under torch.distributed with G ranks the RUNS are dealt over the ranks (run r on rank r mod G: the reference's
    Pool of processes as one process per GPU) -- each rank advances its runs in lockstep on its own GPU, nothing is
    exchanged while they run, and ONE all_gather_object at the end hands every rank the histories, stop reasons and
    best candidates of all runs (`res.runs[k].es` is None for a run another rank made).

    Falls back to making the runs one after another (same results) when lockstep is not possible: with
    force_python_loop, or where the native weight normalisation does not reproduce numpy's on this machine."""
    import ctypes as C
    import torch
    from .cmaes import MAX_RUNS, RunManyArgs, STOP_NAMES, N_STOP, load_cma_library
    from ...scenarios import _native_normalise_variant
    R = len(ords)
    if not (R == len(seeds) == len(sigma0s)) or R < 1:
        raise ValueError("one seed and one sigma0 per run")
    res = LockstepResult(list(ords))
    first = ords[0]
    D = first.weight_dim
    engines = [o._engine() for o in ords]
    overrides = dict(maxiter=maxiter, maxfevals=maxfevals, **(termination or {}))
    odd = any(hasattr(o.world, "unlucky_car_idx") and (len(o.init_car_states) * o.num_samples) % 2 for o in ords)
    if (any(e is not engines[0] for e in engines) or any(o.weight_dim != D or o.num_samples != first.num_samples for o in ords)):
        raise ValueError("optimize_cmaes_lockstep: the runs must share world, car, planner arguments and num_samples")
    if (R > MAX_RUNS or odd or _native_normalise_variant(D) is None
            or any(getattr(o, "force_python_loop", False) for o in ords)):
        res.lockstep = False
        for o, seed, sigma0 in zip(ords, seeds, sigma0s):
            res.best.append(o.optimize_cmaes(seed=seed, sigma0=sigma0, popsize=popsize, maxiter=maxiter, maxfevals=maxfevals,
                                             termination=termination))
            res.generation_seconds.extend(o.generation_seconds)
            res.generation_wall_seconds.extend(o.generation_wall_seconds)
        return res
    eng = engines[0]
    lib = load_cma_library()
    # (the reference dumps the history after EVERY evaluation (mpc_ord.py:146-148): with a save path a native call is one
    #  generation -- chunk = 1 below -- so a crash loses at most the generation in flight (the single-run path refuses the
    #  native loop for the same reason).  Histories and pickles are the same either way.)
    ess = []
    for o, seed, sigma0 in zip(ords, seeds, sigma0s):                  # the head of optimize_cmaes, per run
        o.history.seed = seed
        assert seed != 0
        assert not o.done
        o.should_save_history = True
        o.eval_weights(o.designer_weights)                             # "Iteration 0" baseline
        ess.append(NativeCMAES(list(o.designer_weights), sigma0, popsize=popsize, seed=seed, active=o.cma_active))
        o.generation_seconds, o.generation_wall_seconds, o.fitness_seconds = [], [], []
        o.host_split, o.n_resampled, o.stop_reason = {}, 0, {}
        o._eng_fixed = eng
    S = first.num_samples
    lams = np.array([es.lam for es in ess], dtype=np.int64)
    run_N = np.array([len(o.init_car_states) for o in ords], dtype=np.int64)
    run_n0 = np.concatenate([[0], np.cumsum(run_N)[:-1]]).astype(np.int64)
    run_p0 = np.concatenate([[0], np.cumsum(lams)[:-1]]).astype(np.int64)
    P_rows, N_rows = int(lams.sum()), int(run_N.sum())
    E_max = int((lams * run_N).sum() * S)
    if not chunk or int(chunk) < 1:                                    # (0 generations per call would never finish)
        chunk = int(min(128, max(8, (4 << 20) // (P_rows * D * 8))))
    chunk = int(chunk)
    if any(o.save_path is not None for o in ords):
        chunk = 1
    concurrent = False
    try:
        with torch.cuda.device(eng.device):
            inits = np.concatenate([np.asarray(o.init_car_states, dtype=np.float32).reshape(-1, 4) for o in ords])
            init_dev = torch.as_tensor(np.ascontiguousarray(inits)).to(eng.device)
            w_host = torch.empty((P_rows, D), dtype=torch.float32).pin_memory()
            idx_host = torch.empty((E_max, 3), dtype=torch.int32).pin_memory()
            ret_host = torch.empty((E_max,), dtype=torch.float32).pin_memory()
            hist_w = np.empty((chunk, P_rows, D), dtype=np.float64)
            hist_c = np.empty((chunk, P_rows), dtype=np.float64)
            evaluated = np.zeros((chunk, R), dtype=np.uint8)
            secs = np.zeros((chunk, 8), dtype=np.float64)
            nonf = np.zeros((chunk, R), dtype=np.int32)
            launched = np.zeros(chunk, dtype=np.int64)
            active = np.ones(R, dtype=np.uint8)
            pending = np.zeros(R, dtype=np.uint8)
            flags = np.zeros((R, N_STOP), dtype=np.int32)
            opts = ess[0]._stop_opts(overrides)
            stop_opts = np.array([opts.get(k, 0.0) for k in STOP_NAMES], dtype=np.float64)
            X_ptrs = (C.c_void_p * R)(*[es._X_ptr for es in ess])
            f_ptrs = (C.c_void_p * R)(*[es._f_ptr for es in ess])
            es_ptrs = (C.c_void_p * R)(*[es._h.value for es in ess])
            a = RunManyArgs()
            a.scn, a.init_dev, a.N_rows, a.P_rows, a.S, a.R = eng._h.value, init_dev.data_ptr(), N_rows, P_rows, S, R
            a.normalise_variant = _native_normalise_variant(D)
            a.run_n0, a.run_N, a.run_p0, a.run_reset_phase = run_n0.ctypes.data, run_N.ctypes.data, run_p0.ctypes.data, None
            a.w_pinned, a.index_pinned, a.ret_pinned = w_host.data_ptr(), idx_host.data_ptr(), ret_host.data_ptr()
            a.stream = torch.cuda.current_stream().cuda_stream
            # Groups (round 6): while the launch fits one wavefront per SIMD (the reference's shapes: 28 runs x 27 episodes =
            # 756 wavefronts), the runs go out as FOUR (else two) launches on as many streams, side by side on the chip; a
            # group's tells and asks (0.10 ms per generation for all 28 runs on one thread) run under the other groups'
            # kernels (csrc/ocd_cma.c: ocd_cma_run_many).  Every run's own call sequence, hence its history, is unchanged.
            n_groups = int(groups) if groups else _lockstep_groups(eng, (lams * run_N * S).tolist())
            streams = []
            if n_groups > 1:
                # streams on DIFFERENT hardware queues, found by measurement once per engine (Engine.side_by_side_streams);
                # each group's launch is planned for its share of the compute units
                eng.set_option("concurrent_launches", n_groups)
                concurrent = True
                streams = list(eng.side_by_side_streams(n_groups, inits[0], int((lams * run_N * S).sum()) // n_groups))
                if len(streams) < n_groups and not groups:             # (fewer queues than groups: as many groups as streams)
                    n_groups = 2 if len(streams) >= 2 else 1
                    streams = streams[:n_groups] if n_groups > 1 else []
                    eng.set_option("concurrent_launches", n_groups)
                while len(streams) < n_groups:                         # (a forced group count keeps its count)
                    streams.append(torch.cuda.Stream(device=eng.device))
            for st_ in streams:
                st_.wait_stream(torch.cuda.current_stream())           # (after the baseline evaluations and the uploads above)
            stream_ptrs = (C.c_void_p * max(n_groups, 1))(*[st_.cuda_stream for st_ in streams])
            a.n_groups, a.streams = (n_groups if n_groups > 1 else 0), (C.cast(stream_ptrs, C.c_void_p).value if n_groups > 1 else None)
            res.groups = max(n_groups, 1)
            a.host_threads = res.host_threads = int(host_threads) if host_threads else _lockstep_host_threads(R)
            a.rollout = C.cast(eng.lib.ocd_rollout_indexed, C.c_void_p).value
            a.sync = C.cast(eng.lib.ocd_stream_synchronize, C.c_void_p).value
            a.max_generations = chunk
            a.stop_opts, a.active = stop_opts.ctypes.data, active.ctypes.data
            a.X, a.cost = C.cast(X_ptrs, C.c_void_p).value, C.cast(f_ptrs, C.c_void_p).value
            a.hist_w, a.hist_cost, a.evaluated = hist_w.ctypes.data, hist_c.ctypes.data, evaluated.ctypes.data
            a.seconds, a.nonfinite, a.episodes_launched = secs.ctypes.data, nonf.ctypes.data, launched.ctypes.data
            a.stop_flags, a.pending_nan = flags.ctypes.data, pending.ctypes.data
            names = ("ask", "normalise", "launch", "overlapped_bookkeeping", "kernel_gather_readback", "reduce", "tell")
            done = C.c_int64(0)
            while active.any():
                t_chunk = time.perf_counter()
                st = lib.ocd_cma_run_many(es_ptrs, C.byref(a), C.byref(done))
                if st != 0:
                    raise RuntimeError(f"ocd_cma_run_many -> {st}: {eng.lib.ocd_last_error().decode()}")
                G = int(done.value)
                for r in range(R):                                     # one block per run and native call (mpc_ord.py:146, lazily)
                    took = np.nonzero(evaluated[:G, r])[0]
                    if took.size == 0:
                        continue
                    o, p0, lam = ords[r], int(run_p0[r]), int(lams[r])
                    o._history_chunks.append((hist_w[took, p0:p0 + lam], hist_c[took, p0:p0 + lam]))    # (fancy index: copies)
                    o.iter += lam * int(took.size)
                    o.generation_seconds.extend(secs[took, 0].tolist())
                    told = took[:-1] if pending[r] else took            # (a pending generation is counted after its redraw)
                    o.n_nonfinite.extend(int(v) for v in nonf[told, r])
                res.generation_seconds.extend(secs[:G, 0].tolist())
                res.episodes_per_generation.extend(int(v) for v in launched[:G])
                for k, name in enumerate(names):
                    res.host_split.setdefault(name, []).extend(secs[:G, 1 + k].tolist())
                for r in np.nonzero(pending)[0]:                       # pycma's rejection sampling, as the run alone does it
                    o, es = ords[r], ess[r]
                    f = es._f.copy()
                    o.n_nonfinite.append(int(lams[r]) - int(np.isfinite(f).sum()))
                    f = o._resample_nan(es, es._X, f)
                    o._flush_history()
                    es.tell(es._X, f)
                    why = es.stop(**overrides)
                    if why:
                        active[r] = 0
                        o.stop_reason = why
                for r in range(R):
                    if flags[r].any():
                        ords[r].stop_reason = {k: opts.get(k) for i, k in enumerate(STOP_NAMES) if flags[r, i]}
                if G:
                    res.generation_wall_seconds.extend([(time.perf_counter() - t_chunk) / G] * G)
                for o in ords:                                         # a saved history is complete after every native call
                    if o.save_path is not None:
                        o.save_history()
            res.launch = eng.last_launch()
    finally:
        if concurrent:
            eng.set_option("concurrent_launches", 1)
        for o, es in zip(ords, ess):
            o._eng_fixed = None
            o._init_np_cache = None
            o._flush_history()
            o.es = es
    for o, es in zip(ords, ess):
        o.should_save_history = False
        o.done = True
        es.lib.ocd_cma_stop_state(es._h, es._ss_ptr)
        es.last_nonfinite, es.nonfinite_total = int(es._ss[8]), int(es._ss[9])
        res.best.append(es.best_x)
    return res


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


def _lockstep_over_ranks(ords, seeds, sigma0s, popsize, maxiter, maxfevals, termination, chunk, groups=None, host_threads=None):
    """optimize_cmaes_lockstep under torch.distributed: rank g makes runs g, g + G, g + 2G, ... in lockstep on its own GPU
    (no collective while they run -- the runs are independent, run_mpc_ord.py:83-90), then one all_gather_object."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = list(range(rank, len(ords), world))
    for o in ords:
        o._local_only = True
    try:
        # A rank whose runs fail must still reach the collective: the others would wait in all_gather_object until the
        # backend's timeout.  The exception travels as text and every rank raises after the gather.
        local, payload, stats, failure = None, {}, None, None
        try:
            if mine:
                # (the distributed flag is read through dist.get_world_size(): the local call sees itself as unsharded
                #  because every MPC_ORD it touches is marked _local_only)
                local = _lockstep_local([ords[k] for k in mine], [seeds[k] for k in mine], [sigma0s[k] for k in mine],
                                        popsize, maxiter, maxfevals, termination, chunk, groups, host_threads)
            for j, k in enumerate(mine):
                o = ords[k]
                payload[k] = dict(history=[(np.asarray(w), float(r)) for w, r in o.history], seed=o.history.seed, iter=o.iter,
                                  stop_reason=o.stop_reason, n_nonfinite=list(o.n_nonfinite), n_resampled=o.n_resampled,
                                  generation_seconds=list(o.generation_seconds), best=np.asarray(local.best[j]),
                                  best_f=float(o.es.best_f))
            stats = None if local is None else dict(generation_seconds=local.generation_seconds,
                                                    generation_wall_seconds=local.generation_wall_seconds,
                                                    episodes_per_generation=local.episodes_per_generation,
                                                    lockstep=local.lockstep, launch=local.launch)
        except Exception as exc:                                       # noqa: BLE001 -- re-raised on every rank below
            import traceback
            failure = f"rank {rank}: {type(exc).__name__}: {exc}\n{traceback.format_exc()}"
            payload, stats = {}, None
        gathered = [None] * world
        dist.all_gather_object(gathered, (payload, stats, failure))
        failures = [f for _, _, f in gathered if f]
        if failures:
            raise RuntimeError("optimize_cmaes_lockstep failed on " + "; ".join(f.splitlines()[0] for f in failures)
                               + "\n" + failures[0])
        everything = [(pl, st) for pl, st, _ in gathered]
    finally:
        for o in ords:
            o._local_only = False
    res = LockstepResult(list(ords))
    res.ranks = world
    res.per_rank = [st for _, st in everything]
    res.lockstep = all(st is None or st["lockstep"] for _, st in everything)
    slowest = max((st for _, st in everything if st is not None), key=lambda st: sum(st["generation_wall_seconds"]), default=None)
    if slowest is not None:                                            # the job lasts as long as its slowest rank
        res.generation_seconds, res.generation_wall_seconds = slowest["generation_seconds"], slowest["generation_wall_seconds"]
        res.episodes_per_generation, res.launch = slowest["episodes_per_generation"], slowest["launch"]
    best = {}
    for pl, _ in everything:
        for k, rec in pl.items():
            o = ords[k]
            if k not in mine:                                          # a run another rank made: its results, no strategy object
                o.history = list2(rec["history"])
                o.history.seed = rec["seed"]
                o.iter, o.stop_reason = rec["iter"], rec["stop_reason"]
                o.n_nonfinite, o.n_resampled = rec["n_nonfinite"], rec["n_resampled"]
                o.generation_seconds = rec["generation_seconds"]
                o.es, o.done, o.should_save_history = None, True, False
                o.best_f = rec["best_f"]
            best[k] = rec["best"]
    res.best = [best[k] for k in range(len(ords))]
    return res
