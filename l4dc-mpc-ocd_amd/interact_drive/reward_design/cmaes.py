"""A compact (mu/mu_w, lambda)-CMA-ES with an ask/tell interface.

The reference calls pycma's ``cma.evolution_strategy.fmin2`` (mpc_ord.py:41; pycma is an unpinned,
un-vendored dependency, setup.py:6, and is not installed here).  This is the textbook algorithm
(N. Hansen, "The CMA Evolution Strategy: A Tutorial", 2016, default strategy parameters) so that a
whole population can be handed to the batched GPU fitness at once.  The sampling sequence is NOT
pycma's: optimisation traces are "parity unpinned" (SURVEY.md 8c); the fitness values it is fed
are bit-exact.

Two implementations of the same algorithm:
  NativeCMAES  csrc/ocd_cma.c through ctypes (include/ocd_cma.h): what MPC_ORD.optimize_cmaes runs -- around a
               1.6 ms kernel the numpy calls of ask + tell cost ~45 us per generation, the native ones ~3 us;
  CMAES        numpy; the twin the tests compare the native one with (same normal deviates bit for bit, same
               candidates to rounding: both sample m + sigma * C^(1/2) z with the SYMMETRIC root, which does not
               depend on the order or sign of the eigenvectors LAPACK / the native Jacobi sweep return).
"""
import ctypes as C
import os

import numpy as np


class CMAES:
    def __init__(self, x0, sigma0, popsize=None, seed=1):
        self.n = n = len(x0)
        self.mean = np.asarray(x0, dtype=np.float64).copy()
        self.sigma = float(sigma0)
        self.lam = int(popsize) if popsize else 4 + int(3 * np.log(n))
        self.mu = self.lam // 2
        w = np.log(self.mu + 0.5) - np.log(np.arange(1, self.mu + 1))
        self.weights = w / w.sum()
        self.mueff = 1.0 / np.sum(self.weights ** 2)
        self.cc = (4 + self.mueff / n) / (n + 4 + 2 * self.mueff / n)
        self.cs = (self.mueff + 2) / (n + self.mueff + 5)
        self.c1 = 2 / ((n + 1.3) ** 2 + self.mueff)
        self.cmu = min(1 - self.c1, 2 * (self.mueff - 2 + 1 / self.mueff) / ((n + 2) ** 2 + self.mueff))
        self.damps = 1 + 2 * max(0.0, np.sqrt((self.mueff - 1) / (n + 1)) - 1) + self.cs
        self.pc = np.zeros(n)
        self.ps = np.zeros(n)
        self.B = np.eye(n)
        self.Dg = np.ones(n)
        self.C = np.eye(n)
        self.sqrtC = np.eye(n)
        self.invsqrtC = np.eye(n)
        self.chiN = np.sqrt(n) * (1 - 1 / (4 * n) + 1 / (21 * n * n))
        self.rng = np.random.RandomState(seed)
        self.gen = 0
        self.counteval = 0
        self._z = None
        self._best_x, self._best_f = None, np.inf
        self._told = None

    def ask(self):
        """lambda candidate vectors [lam, n]."""
        self._z = self.rng.standard_normal((self.lam, self.n))
        self._y = self._z @ self.sqrtC                              # C^(1/2) z, symmetric root
        return self.mean + self.sigma * self._y

    def tell(self, X, fitness):
        fitness = np.asarray(fitness, dtype=np.float64)
        order = np.argsort(fitness, kind="stable")
        self._update_best()                                        # (a previous generation finish_tell() did not see)
        self._told = (X, fitness, order[0])                        # best-so-far bookkeeping: finish_tell()
        self.counteval += len(fitness)
        n = self.n
        if self.invsqrtC is None:                                   # finish_tell() was not called in between
            self.invsqrtC = (self.B / self.Dg) @ self.B.T
        ysel = self._y[order[: self.mu]]
        yw = self.weights @ ysel
        self.mean = self.mean + self.sigma * yw
        self.ps = (1 - self.cs) * self.ps + np.sqrt(self.cs * (2 - self.cs) * self.mueff) * (self.invsqrtC @ yw)
        ps_norm = np.sqrt(self.ps.dot(self.ps))
        hsig = (ps_norm / np.sqrt(1 - (1 - self.cs) ** (2 * self.counteval / self.lam)) / self.chiN
                < 1.4 + 2 / (n + 1))
        self.pc = (1 - self.cc) * self.pc + hsig * np.sqrt(self.cc * (2 - self.cc) * self.mueff) * yw
        rank_mu = (ysel.T * self.weights) @ ysel
        C = ((1 - self.c1 - self.cmu) * self.C
             + self.c1 * (np.outer(self.pc, self.pc) + (1 - hsig) * self.cc * (2 - self.cc) * self.C)
             + self.cmu * rank_mu)
        # rank_mu is symmetric only up to rounding: keep C bit-symmetric, so that what eigh reads (the lower
        # triangle) and what any other reader of C sees are the same matrix
        self.C = (C + C.T) * 0.5
        self.sigma *= np.exp((self.cs / self.damps) * (ps_norm / self.chiN - 1))
        ev, self.B = np.linalg.eigh(self.C)
        self.Dg = np.sqrt(np.maximum(ev, 1e-20))
        self.sqrtC = (self.B * self.Dg) @ self.B.T
        self.invsqrtC = None                                       # needed by the NEXT tell only: finish_tell()
        self.gen += 1

    def finish_tell(self):
        """The part of tell() the next ask() does not need (C^-1/2 for the next path update, best-so-far): callers
        with something to wait for -- a running GPU generation -- call it meanwhile; tell() / best_x catch up
        otherwise."""
        if self.invsqrtC is None:
            self.invsqrtC = (self.B / self.Dg) @ self.B.T
        self._update_best()

    def _update_best(self):
        if self._told is not None:
            X, fitness, i = self._told
            if fitness[i] < self._best_f:
                self._best_f, self._best_x = float(fitness[i]), np.array(X[i])
            self._told = None

    @property
    def best_x(self):
        self.finish_tell()
        return self._best_x

    @property
    def best_f(self):
        self.finish_tell()
        return self._best_f

    def stop(self, tolfun=1e-11, tolx=1e-11, maxiter=None, last_fitness=None):
        if maxiter is not None and self.gen >= maxiter:
            return True
        if self.sigma * np.max(self.Dg) < tolx:
            return True
        if last_fitness is not None and self.gen > 10 and np.ptp(last_fitness) < tolfun:
            return True
        return False


# ---------------------------------------------------------------- native implementation (csrc/ocd_cma.c)
_CMA_LIB = None
_D = C.POINTER(C.c_double)


def load_cma_library():
    """dlopen csrc/libocd_cma.so (built by `make -C csrc`, plain C) and bind include/ocd_cma.h."""
    global _CMA_LIB
    if _CMA_LIB is not None:
        return _CMA_LIB
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(os.path.dirname(here)), "csrc", "libocd_cma.so")
    if not os.path.exists(path):
        import subprocess
        r = subprocess.run(["make", "-C", os.path.dirname(path), "libocd_cma.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise FileNotFoundError(f"{path} is missing and could not be built:\n{r.stderr[-1500:]}")
    lib = C.CDLL(path)
    lib.ocd_cma_create.restype = C.c_int32
    lib.ocd_cma_create.argtypes = [C.c_int32, _D, C.c_double, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.ocd_cma_destroy.restype = None
    lib.ocd_cma_destroy.argtypes = [C.c_void_p]
    lib.ocd_cma_popsize.restype = C.c_int32
    lib.ocd_cma_popsize.argtypes = [C.c_void_p]
    lib.ocd_cma_ask.restype = C.c_int32
    lib.ocd_cma_ask.argtypes = [C.c_void_p, C.c_void_p]
    lib.ocd_cma_prepare.restype = C.c_int32
    lib.ocd_cma_prepare.argtypes = [C.c_void_p]
    lib.ocd_cma_tell.restype = C.c_int32
    lib.ocd_cma_tell.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ocd_cma_state.restype = C.c_int32
    lib.ocd_cma_state.argtypes = [C.c_void_p, _D, _D, _D, _D, _D, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _D]
    lib.ocd_fitness_from_returns.restype = C.c_int32
    lib.ocd_fitness_from_returns.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    _CMA_LIB = lib
    return lib


def fitness_from_returns_native(returns: np.ndarray, P: int, N: int, S: int, out: np.ndarray = None) -> np.ndarray:
    """sharding.fitness_from_returns (mpc_ord.py:102,126-151) in one native call; `returns` fp32 contiguous [P*N*S]."""
    lib = load_cma_library()
    r = np.ascontiguousarray(returns, dtype=np.float32)
    cost = np.empty(P, dtype=np.float64) if out is None else out
    if lib.ocd_fitness_from_returns(r.ctypes.data, P, N, S, cost.ctypes.data) != 0:
        raise ValueError("ocd_fitness_from_returns: bad arguments")
    return cost


class NativeCMAES:
    """The same ask / tell interface as CMAES over csrc/ocd_cma.c."""

    def __init__(self, x0, sigma0, popsize=None, seed=1):
        self.lib = load_cma_library()
        self.n = len(x0)
        x = np.ascontiguousarray(x0, dtype=np.float64)
        h = C.c_void_p()
        if self.lib.ocd_cma_create(self.n, x.ctypes.data_as(_D), float(sigma0), int(popsize or 0), int(seed) & 0xffffffff,
                                   C.byref(h)) != 0:
            raise ValueError("ocd_cma_create: bad arguments")
        self._h = h
        self.lam = int(self.lib.ocd_cma_popsize(h))
        self.mu = self.lam // 2
        self._X = np.empty((self.lam, self.n), dtype=np.float64)
        self._X_ptr = self._X.ctypes.data
        self._f = np.empty(self.lam, dtype=np.float64)
        self._f_ptr = self._f.ctypes.data

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.ocd_cma_destroy(h)
            self._h = None

    def ask(self):
        """lambda candidate vectors [lam, n] (a view of an internal buffer, valid until the next ask)."""
        if self.lib.ocd_cma_ask(self._h, self._X_ptr) != 0:
            raise RuntimeError("ocd_cma_ask failed")
        return self._X

    def tell(self, X, fitness):
        if X is not self._X:
            self._X[...] = X
        self._f[...] = fitness
        if self.lib.ocd_cma_tell(self._h, self._X_ptr, self._f_ptr) != 0:
            raise RuntimeError("ocd_cma_tell failed")

    def finish_tell(self):                                         # (the numpy twin defers work; nothing to do here)
        pass

    def prepare(self):
        """Draw the next population's normal deviates now (while the GPU runs this generation): same stream."""
        self.lib.ocd_cma_prepare(self._h)

    def _state(self):
        n = self.n
        mean, Cm, bx = np.empty(n), np.empty((n, n)), np.empty(n)
        sigma, bf, md = C.c_double(), C.c_double(), C.c_double()
        gen, ce = C.c_int64(), C.c_int64()
        self.lib.ocd_cma_state(self._h, mean.ctypes.data_as(_D), C.byref(sigma), Cm.ctypes.data_as(_D), bx.ctypes.data_as(_D),
                               C.byref(bf), C.byref(gen), C.byref(ce), C.byref(md))
        return dict(mean=mean, sigma=sigma.value, C=Cm, best_x=bx, best_f=bf.value, gen=gen.value, counteval=ce.value,
                    max_axis=md.value)

    mean = property(lambda self: self._state()["mean"])
    sigma = property(lambda self: self._state()["sigma"])
    C = property(lambda self: self._state()["C"])
    best_x = property(lambda self: self._state()["best_x"])
    best_f = property(lambda self: self._state()["best_f"])
    gen = property(lambda self: self._state()["gen"])
    counteval = property(lambda self: self._state()["counteval"])

    def stop(self, tolfun=1e-11, tolx=1e-11, maxiter=None, last_fitness=None):
        st = self._state()
        if maxiter is not None and st["gen"] >= maxiter:
            return True
        if st["sigma"] * st["max_axis"] < tolx:
            return True
        if last_fitness is not None and st["gen"] > 10 and np.ptp(last_fitness) < tolfun:
            return True
        return False
