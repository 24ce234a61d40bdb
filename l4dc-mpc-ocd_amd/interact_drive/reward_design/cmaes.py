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
        self.best_x, self.best_f = None, np.inf

    def ask(self):
        """lambda candidate vectors [lam, n]."""
        self._z = self.rng.standard_normal((self.lam, self.n))
        self._y = (self._z * self.Dg) @ self.B.T
        return self.mean + self.sigma * self._y

    def tell(self, X, fitness):
        fitness = np.asarray(fitness, dtype=np.float64)
        order = np.argsort(fitness, kind="stable")
        if fitness[order[0]] < self.best_f:
            self.best_f, self.best_x = float(fitness[order[0]]), np.array(X[order[0]])
        self.counteval += len(fitness)
        n = self.n
        ysel = self._y[order[: self.mu]]
        yw = self.weights @ ysel
        self.mean = self.mean + self.sigma * yw
        self.ps = (1 - self.cs) * self.ps + np.sqrt(self.cs * (2 - self.cs) * self.mueff) * (self.invsqrtC @ yw)
        hsig = (np.linalg.norm(self.ps) / np.sqrt(1 - (1 - self.cs) ** (2 * self.counteval / self.lam)) / self.chiN
                < 1.4 + 2 / (n + 1))
        self.pc = (1 - self.cc) * self.pc + hsig * np.sqrt(self.cc * (2 - self.cc) * self.mueff) * yw
        rank_mu = (ysel.T * self.weights) @ ysel
        self.C = ((1 - self.c1 - self.cmu) * self.C
                  + self.c1 * (np.outer(self.pc, self.pc) + (1 - hsig) * self.cc * (2 - self.cc) * self.C)
                  + self.cmu * rank_mu)
        self.sigma *= np.exp((self.cs / self.damps) * (np.linalg.norm(self.ps) / self.chiN - 1))
        ev, self.B = np.linalg.eigh(self.C)                         # reads the lower triangle only
        self.Dg = np.sqrt(np.maximum(ev, 1e-20))
        self.invsqrtC = (self.B / self.Dg) @ self.B.T
        self.gen += 1

    def stop(self, tolfun=1e-11, tolx=1e-11, maxiter=None, last_fitness=None):
        if maxiter is not None and self.gen >= maxiter:
            return True
        if self.sigma * np.max(self.Dg) < tolx:
            return True
        if last_fitness is not None and self.gen > 10 and np.ptp(last_fitness) < tolfun:
            return True
        return False
