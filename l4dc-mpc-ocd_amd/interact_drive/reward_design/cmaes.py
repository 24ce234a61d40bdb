"""A compact (mu/mu_w, lambda)-CMA-ES with an ask/tell interface.

The reference calls pycma's ``cma.evolution_strategy.fmin2`` (mpc_ord.py:41; pycma is an unpinned,
un-vendored dependency, setup.py:6, and is not installed here).  This is the textbook algorithm
(N. Hansen, "The CMA Evolution Strategy: A Tutorial", 2016, default strategy parameters) so that a
whole population can be handed to the batched GPU fitness at once.  The sampling sequence is NOT
pycma's: optimisation traces are "parity unpinned" (SURVEY.md 8c); the fitness values it is fed
are bit-exact.
"""
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
        self._y = (self._z * self.Dg) @ self.B.T
        return self.mean + self.sigma * self._y

    def tell(self, X, fitness):
        fitness = np.asarray(fitness, dtype=np.float64)
        order = np.argsort(fitness, kind="stable")
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
        self.invsqrtC = None                                       # needed by the NEXT tell only: finish_tell()
        self.gen += 1

    def finish_tell(self):
        """The part of tell() the next ask() does not need (C^-1/2 for the next path update, best-so-far): callers
        with something to wait for -- a running GPU generation -- call it meanwhile; tell() / best_x catch up
        otherwise."""
        if self.invsqrtC is None:
            self.invsqrtC = (self.B / self.Dg) @ self.B.T
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
