"""A compact (mu/mu_w, lambda)-CMA-ES with an ask/tell interface.

The reference calls pycma's ``cma.evolution_strategy.fmin2`` (mpc_ord.py:41; pycma is an unpinned,
un-vendored dependency, setup.py:6, and is not installed here).  This is the published algorithm
(N. Hansen, "The CMA Evolution Strategy: A Tutorial", arXiv 1604.00772, default strategy parameters of
its Table 1: recombination weights ln((lambda+1)/2) - ln i over all lambda ranks, eq. 49; the negative
ones scaled per eqs. 50-53 and used in the rank-mu update only -- "active" CMA, eqs. 46-47, pycma's
CMA_active=True default; ``active=False`` drops them) so that a whole population can be handed to the
batched GPU fitness at once.  tests/test_cma_paper_constants.py types the constants of n = 7,
lambda = 9 (the reference's shape) from the equations, not from either twin.  The sampling sequence is NOT
pycma's: optimisation traces are "parity unpinned" (SURVEY.md 8c); the fitness values it is fed
are bit-exact.

Two implementations of the same algorithm:
  NativeCMAES  csrc/ocd_cma.c through ctypes (include/ocd_cma.h): what MPC_ORD.optimize_cmaes runs -- around a
               1.6 ms kernel the numpy calls of ask + tell cost ~45 us per generation, the native ones ~3 us;
  CMAES        numpy; the twin the tests compare the native one with (same normal deviates bit for bit, same
               candidates to rounding: both sample m + sigma * C^(1/2) z with the SYMMETRIC root, which does not
               depend on the order or sign of the eigenvectors LAPACK / the native Jacobi sweep return).

Non-finite costs (both twins, identically): a population is ranked with NaN LAST (numpy's argsort order, +inf just
before it) and tell() returns how many costs were not finite; resample(k) draws a replacement for slot k the way
pycma's ask_and_eval does for a candidate whose cost came back NaN (rejection sampling); MPC_ORD.optimize_cmaes
uses both.  Termination (`stop()`, one implementation shared by the twins) follows pycma's documented default
options -- maxiter = 100 + 150 (N + 3)^2 / sqrt(popsize), tolfun 1e-11, tolfunhist 1e-12, tolx 1e-11,
tolstagnation = 100 + 100 N^1.5 / popsize, tolconditioncov 1e14, tolfacupx 1e3, tolupsigma 1e20, flat fitness --
restated from pycma's option table (pycma itself is absent: "parity unpinned").
"""
import ctypes as C
import math
import os

import numpy as np


class _Termination:
    """pycma's termination rules on the numbers both twins expose through _stop_state().

    Options and defaults as pycma's CMAOptions documents them (cma.evolution_strategy, the dependency behind
    mpc_ord.py:41; not installed here, restated):
        maxiter          100 + 150 * (N + 3) ** 2 // popsize ** 0.5
        maxfevals        inf
        tolfun           1e-11   range of the current population's costs AND of the recent best costs below it
        tolfunhist       1e-12   range of the recent best costs (at least 10 generations of them)
        tolx             1e-11   sigma * sqrt(C_ii) and sigma * |pc_i| below it in every coordinate
        tolfacupx        1e3     sigma * sqrt(C_ii) grew beyond tolfacupx * sigma0 (initial step far too small)
        tolconditioncov  1e14    condition number of C
        tolupsigma       1e20    sigma / sigma0 above tolupsigma * the longest axis of C ("creeping")
        tolstagnation    100 + 100 * N ** 1.5 / popsize   generations without progress of the median / best costs
        tolflatfitness   1       generations in a row whose best and median cost coincide
        noeffectaxis     -       0.1 sigma along principal axis (generation mod N, axes by ascending length) no longer
                                 changes the mean in any coordinate (no option value in pycma: always on)
        noeffectcoord    -       0.2 sigma sqrt(C_jj) no longer changes coordinate j of the mean (always on)
    The "recent best costs" are those of the last 10 + 30 N / popsize generations, as in pycma.  stop() returns
    pycma's dict of the satisfied conditions ({} = go on), e.g. {'tolx': 1e-11}."""

    def _init_termination(self, sigma0):
        n, lam = self.n, self.lam
        self.sigma0 = float(sigma0)
        self.opts = dict(maxiter=100 + 150 * (n + 3) ** 2 // lam ** 0.5, maxfevals=np.inf, tolfun=1e-11,
                         tolfunhist=1e-12, tolx=1e-11, tolfacupx=1e3, tolconditioncov=1e14, tolupsigma=1e20,
                         tolstagnation=int(100 + 100 * n ** 1.5 / lam), tolflatfitness=1)
        self._hist = []                 # best cost of the last 10 + 30 N / popsize generations, newest first
        self._hist_len = 10 + 30 * n / lam
        self._histbest, self._histmedian = [], []
        self._flat = 0
        self._last_pop_range = np.inf
        self.nonfinite_total = 0
        self.last_nonfinite = 0

    def _record(self, best, median, worst, nonfinite):
        """After every tell: this population's best / median / worst cost (NaN kept out of best and worst)."""
        self.last_nonfinite = int(nonfinite)
        self.nonfinite_total += int(nonfinite)
        self._hist.insert(0, best)
        if len(self._hist) > self._hist_len:
            self._hist.pop()
        self._histbest.append(best)                               # oldest first (pycma keeps them newest first)
        self._histmedian.append(median)
        if len(self._histbest) > 20000:
            del self._histbest[0], self._histmedian[0]
        self._last_pop_range = worst - best if math.isfinite(worst) and math.isfinite(best) else np.inf
        self._flat = self._flat + 1 if best == median else 0

    def stop(self, tolfun=None, tolx=None, maxiter=None, last_fitness=None, **overrides):
        """{} while the search should go on, else the satisfied conditions.  Keyword arguments override the
        defaults for this call (maxiter=None keeps pycma's default cap; last_fitness is accepted for older
        callers and ignored: the population's costs are recorded by tell())."""
        o = dict(self.opts)
        for k, v in dict(tolfun=tolfun, tolx=tolx, maxiter=maxiter, **overrides).items():
            if v is not None:
                if k not in o:
                    raise TypeError(f"unknown termination option {k!r}")
                o[k] = v
        st = self._stop_state()
        out = {}
        if st["gen"] >= o["maxiter"]:
            out["maxiter"] = o["maxiter"]
        if st["counteval"] >= o["maxfevals"]:
            out["maxfevals"] = o["maxfevals"]
        if st["gen"] == 0:
            return out
        h = [v for v in self._hist if math.isfinite(v)]
        hist_range = (max(h) - min(h)) if h else np.inf
        if self._last_pop_range < o["tolfun"] and hist_range < o["tolfun"]:
            out["tolfun"] = o["tolfun"]
        if len(self._hist) > 9 and hist_range < o["tolfunhist"]:
            out["tolfunhist"] = o["tolfunhist"]
        if st["sigma_max_std"] < o["tolx"] and st["sigma_max_pc"] < o["tolx"]:
            out["tolx"] = o["tolx"]
        if st["sigma_max_std"] > self.sigma0 * o["tolfacupx"]:
            out["tolfacupx"] = o["tolfacupx"]
        if st["max_axis"] > o["tolconditioncov"] ** 0.5 * st["min_axis"]:
            out["tolconditioncov"] = o["tolconditioncov"]
        if st["sigma"] / self.sigma0 > o["tolupsigma"] * st["max_axis"]:
            out["tolupsigma"] = o["tolupsigma"]
        if self._flat > o["tolflatfitness"]:
            out["tolflatfitness"] = o["tolflatfitness"]
        nb = len(self._histbest)
        if st["gen"] > self.n * (5 + 100 / self.lam) and nb > 100:
            # the newest ell generations against the ell just before them: pycma keeps its lists newest first and
            # compares histbest[:l] with histbest[l:2l] (not with the start of the run, where the costs are worst)
            ell = int(max(o["tolstagnation"] / 5. / 2, nb / 10))
            if 2 * ell < nb and (np.median(self._histmedian[-ell:]) >= np.median(self._histmedian[-2 * ell:-ell])
                                 and np.median(self._histbest[-ell:]) >= np.median(self._histbest[-2 * ell:-ell])):
                out["tolstagnation"] = o["tolstagnation"]
        if st.get("noeffectaxis"):
            out["noeffectaxis"] = None
        if st.get("noeffectcoord"):
            out["noeffectcoord"] = None
        return out


def _population_stats(fitness, order):
    """(best, median, worst, non-finite count) of one population as csrc/ocd_cma.c computes them: NaN ranks last and
    stays out of best / worst; the median is numpy's (NaN if any cost is NaN)."""
    n_nan = int(np.count_nonzero(np.isnan(fitness)))
    m = len(fitness) - n_nan
    best = float(fitness[order[0]]) if m else np.nan
    worst = float(fitness[order[m - 1]]) if m else np.nan
    return best, float(np.median(fitness)), worst, int(np.count_nonzero(~np.isfinite(fitness)))


class CMAES(_Termination):
    def __init__(self, x0, sigma0, popsize=None, seed=1, active=True):
        self.n = n = len(x0)
        self.mean = np.asarray(x0, dtype=np.float64).copy()
        self.sigma = float(sigma0)
        self.lam = lam = int(popsize) if popsize else 4 + int(3 * np.log(n))
        self.mu = mu = self.lam // 2
        self.active = bool(active)
        # eq. 49: w'_i = ln((lambda + 1) / 2) - ln i for i = 1..lambda; positive exactly for i <= mu
        wraw = np.log((lam + 1) / 2.0) - np.log(np.arange(1, lam + 1))
        self.mueff = wraw[:mu].sum() ** 2 / np.sum(wraw[:mu] ** 2)                       # Table 1
        self.cc = (4 + self.mueff / n) / (n + 4 + 2 * self.mueff / n)                    # eq. 56
        self.cs = (self.mueff + 2) / (n + self.mueff + 5)                                # eq. 55
        self.c1 = 2 / ((n + 1.3) ** 2 + self.mueff)                                      # eq. 57 (alpha_cov = 2)
        self.cmu = min(1 - self.c1, 2 * (self.mueff - 2 + 1 / self.mueff) / ((n + 2) ** 2 + self.mueff))   # eq. 58
        # eqs. 50-53: positive weights sum to 1; negative ones sum to -min(alpha_mu, alpha_mueff, alpha_posdef)
        wneg = wraw[mu:]
        mueff_neg = wneg.sum() ** 2 / np.sum(wneg ** 2) if np.any(wneg != 0) else 0.0
        alpha = min(1 + self.c1 / self.cmu, 1 + 2 * mueff_neg / (self.mueff + 2),
                    (1 - self.c1 - self.cmu) / (n * self.cmu))
        neg_sum = np.abs(wneg).sum()
        self.weights_all = np.concatenate([wraw[:mu] / wraw[:mu].sum(),
                                           alpha * wneg / neg_sum if (self.active and neg_sum > 0) else 0.0 * wneg])
        self.weights = self.weights_all[:mu]
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
        self._z_next = None
        self._init_termination(sigma0)

    def prepare(self):
        """Draw the next population's normal deviates now (the native twin does this while the GPU works): the
        stream is the same, only resample() calls in between see it advanced."""
        if self._z_next is None:
            self._z_next = self.rng.standard_normal((self.lam, self.n))

    def ask(self):
        """lambda candidate vectors [lam, n]."""
        self.prepare()
        self._z, self._z_next = self._z_next, None
        self._y = self._z @ self.sqrtC                              # C^(1/2) z, symmetric root
        self._X = self.mean + self.sigma * self._y
        return self._X

    def resample(self, rows):
        """Replace the candidates in `rows` of the population last asked for by fresh draws (pycma's rejection
        sampling of NaN costs); returns the new rows [len(rows), n] (also written into the array ask() returned)."""
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        for k in rows:
            z = self.rng.standard_normal(self.n)
            self._y[k] = self.sqrtC @ z
            self._X[k] = self.mean + self.sigma * self._y[k]
        return self._X[rows]

    def add_evals(self, n):
        """Evaluations made outside tell(): candidates redrawn after a NaN cost (pycma's ask_and_eval counts them)."""
        self.counteval += int(n)

    def tell(self, X, fitness):
        """Returns the number of non-finite costs (they rank last: np.argsort puts NaN after +inf)."""
        fitness = np.asarray(fitness, dtype=np.float64)
        order = np.argsort(fitness, kind="stable")
        self._record(*_population_stats(fitness, order))
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
        # eq. 47: all lambda ranks; a negative weight is rescaled by n / ||C^(-1/2) y_i||^2 (eq. 46)
        if self.active:
            yall = self._y[order]
            wo = self.weights_all.copy()
            neg = wo < 0
            m2 = np.sum((yall[neg] @ self.invsqrtC) ** 2, axis=1)
            wo[neg] = np.where(m2 > 0, wo[neg] * n / np.where(m2 > 0, m2, 1.0), 0.0)
            rank_mu = (yall.T * wo) @ yall
        else:
            rank_mu = (ysel.T * self.weights) @ ysel
        C = ((1 - self.c1 - self.cmu * self.weights_all.sum()) * self.C
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
        return self.last_nonfinite

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

    def _stop_state(self):
        d = np.sqrt(np.diag(self.C))
        i = self.gen % self.n                                       # eigh: axes in ascending order of their length
        axis = bool(np.all(self.mean == self.mean + 0.1 * self.sigma * self.Dg[i] * self.B[:, i]))
        coord = bool(np.any(self.mean == self.mean + 0.2 * self.sigma * d))
        return dict(noeffectaxis=axis, noeffectcoord=coord, sigma=self.sigma, max_axis=float(np.max(self.Dg)), min_axis=float(np.min(self.Dg)), gen=self.gen,
                    counteval=self.counteval, sigma_max_std=float(self.sigma * d.max()),
                    sigma_max_pc=float(self.sigma * np.abs(self.pc).max()))


# ---------------------------------------------------------------- native implementation (csrc/ocd_cma.c)
_CMA_LIB = None
_D = C.POINTER(C.c_double)

# the order of opts / flags of ocd_cma_stop (include/ocd_cma.h)
STOP_NAMES = ("maxiter", "maxfevals", "tolfun", "tolfunhist", "tolx", "tolfacupx", "tolconditioncov", "tolupsigma",
              "tolstagnation", "tolflatfitness", "noeffectaxis", "noeffectcoord")
N_STOP = len(STOP_NAMES)                                            # OCD_CMA_N_STOP


class RunArgs(C.Structure):
    """struct ocd_cma_run_args (include/ocd_cma.h)."""
    _fields_ = [("scn", C.c_void_p), ("init_dev", C.c_void_p), ("N", C.c_int64), ("S", C.c_int64),
                ("w_pinned", C.c_void_p), ("ret_pinned", C.c_void_p), ("stream", C.c_void_p),
                ("rollout", C.c_void_p), ("sync", C.c_void_p), ("normalise_variant", C.c_int32), ("reserved", C.c_int32),
                ("max_generations", C.c_int64), ("stop_opts", C.c_double * 12), ("X", C.c_void_p), ("cost", C.c_void_p),
                ("hist_w", C.c_void_p), ("hist_cost", C.c_void_p), ("seconds", C.c_void_p), ("nonfinite", C.c_void_p)]


class RunManyArgs(C.Structure):
    """struct ocd_cma_many_args (include/ocd_cma.h): R runs in lockstep, one indexed launch per generation."""
    _fields_ = [("scn", C.c_void_p), ("init_dev", C.c_void_p), ("N_rows", C.c_int64), ("P_rows", C.c_int64), ("S", C.c_int64),
                ("R", C.c_int32), ("normalise_variant", C.c_int32),
                ("run_n0", C.c_void_p), ("run_N", C.c_void_p), ("run_p0", C.c_void_p), ("run_reset_phase", C.c_void_p),
                ("w_pinned", C.c_void_p), ("index_pinned", C.c_void_p), ("ret_pinned", C.c_void_p), ("stream", C.c_void_p),
                ("rollout", C.c_void_p), ("sync", C.c_void_p), ("max_generations", C.c_int64), ("stop_opts", C.c_void_p),
                ("active", C.c_void_p), ("X", C.c_void_p), ("cost", C.c_void_p), ("hist_w", C.c_void_p),
                ("hist_cost", C.c_void_p), ("evaluated", C.c_void_p), ("seconds", C.c_void_p), ("nonfinite", C.c_void_p),
                ("episodes_launched", C.c_void_p), ("stop_flags", C.c_void_p), ("pending_nan", C.c_void_p),
                ("n_groups", C.c_int32), ("host_threads", C.c_int32), ("streams", C.c_void_p)]


MAX_RUNS = 256                                                       # OCD_CMA_MAX_RUNS
MAX_GROUPS = 8                                                       # OCD_CMA_MAX_GROUPS


def load_cma_library():
    """dlopen csrc/libocd_cma.so (built by `make -C csrc`, plain C) and bind include/ocd_cma.h."""
    global _CMA_LIB
    if _CMA_LIB is not None:
        return _CMA_LIB
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(os.path.dirname(os.path.dirname(here)), "csrc", "libocd_cma.so")
    root = os.path.dirname(os.path.dirname(os.path.dirname(here)))
    srcs = [os.path.join(os.path.dirname(path), "ocd_cma.c"), os.path.join(root, "include", "ocd_cma.h")]
    stale = (not os.path.exists(path)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(path) for f in srcs)
    make_error = ""
    if stale:                                                       # plain gcc, no GPU involved; a no-op when current
        # every rank of a torchrun job imports this: ONE of them builds (exclusive lock on a file beside the target),
        # the others wait for the lock and find the target current; the Makefile writes a temporary and renames it
        import fcntl
        import subprocess
        try:
            lock = open(os.path.join(os.path.dirname(path), ".libocd_cma.lock"), "w")
        except OSError:
            lock = None                                             # read-only tree: nothing to build into either
        try:
            if lock is not None:
                fcntl.flock(lock, fcntl.LOCK_EX)
            r = subprocess.run(["make", "-C", os.path.dirname(path), "libocd_cma.so"], capture_output=True, text=True)
            if r.returncode != 0:
                make_error = r.stderr[-1500:]
                if not os.path.exists(path):
                    raise FileNotFoundError(f"{path} is missing and could not be built:\n{make_error}")
        finally:
            if lock is not None:
                fcntl.flock(lock, fcntl.LOCK_UN)
                lock.close()
    lib = C.CDLL(path)
    want = _header_abi_version(srcs[1])
    have = lib.ocd_cma_abi_version() if hasattr(lib, "ocd_cma_abi_version") else 1
    if want is not None and have != want:
        raise RuntimeError(f"{path} was built for ocd_cma.h ABI {have}, the header says {want}: run `make -C "
                           f"{os.path.dirname(path)}` (a stale library would silently mismatch the ctypes signatures)"
                           + (f"; the rebuild attempted just now failed:\n{make_error}" if make_error else ""))
    lib.ocd_cma_create.restype = C.c_int32
    lib.ocd_cma_create.argtypes = [C.c_int32, _D, C.c_double, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.ocd_cma_destroy.restype = None
    lib.ocd_cma_destroy.argtypes = [C.c_void_p]
    lib.ocd_cma_popsize.restype = C.c_int32
    lib.ocd_cma_popsize.argtypes = [C.c_void_p]
    lib.ocd_cma_set_active.restype = C.c_int32
    lib.ocd_cma_set_active.argtypes = [C.c_void_p, C.c_int32]
    lib.ocd_cma_weights.restype = C.c_int32
    lib.ocd_cma_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ocd_cma_ask.restype = C.c_int32
    lib.ocd_cma_ask.argtypes = [C.c_void_p, C.c_void_p]
    lib.ocd_cma_prepare.restype = C.c_int32
    lib.ocd_cma_prepare.argtypes = [C.c_void_p]
    lib.ocd_cma_tell.restype = C.c_int32
    lib.ocd_cma_tell.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ocd_cma_resample.restype = C.c_int32
    lib.ocd_cma_resample.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.ocd_cma_add_evals.restype = C.c_int32
    lib.ocd_cma_add_evals.argtypes = [C.c_void_p, C.c_int64]
    lib.ocd_cma_stop_state.restype = C.c_int32
    lib.ocd_cma_stop_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.ocd_cma_abi_version.restype = C.c_int32
    lib.ocd_cma_abi_version.argtypes = []
    lib.ocd_normalise_weights.restype = C.c_int32
    lib.ocd_normalise_weights.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    lib.ocd_cma_stop.restype = C.c_int32
    lib.ocd_cma_stop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ocd_eval_generations.restype = C.c_int32
    lib.ocd_eval_generations.argtypes = [C.POINTER(RunArgs), C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_double)]
    lib.ocd_cma_run.restype = C.c_int32
    lib.ocd_cma_run.argtypes = [C.c_void_p, C.POINTER(RunArgs), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int32)]
    lib.ocd_cma_run_many.restype = C.c_int32
    lib.ocd_cma_run_many.argtypes = [C.c_void_p, C.POINTER(RunManyArgs), C.POINTER(C.c_int64)]
    lib.ocd_cma_state.restype = C.c_int32
    lib.ocd_cma_state.argtypes = [C.c_void_p, _D, _D, _D, _D, _D, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _D]
    lib.ocd_fitness_from_returns.restype = C.c_int32
    lib.ocd_fitness_from_returns.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    _CMA_LIB = lib
    return lib


def _header_abi_version(header):
    import re
    try:
        m = re.search(r"#define\s+OCD_CMA_ABI_VERSION\s+(\d+)", open(header).read())
    except OSError:
        return None
    return int(m.group(1)) if m else None


def fitness_from_returns_native(returns: np.ndarray, P: int, N: int, S: int, out: np.ndarray = None) -> np.ndarray:
    """sharding.fitness_from_returns (mpc_ord.py:102,126-151) in one native call; `returns` fp32 contiguous [P*N*S]."""
    lib = load_cma_library()
    r = np.ascontiguousarray(returns, dtype=np.float32)
    cost = np.empty(P, dtype=np.float64) if out is None else out
    if lib.ocd_fitness_from_returns(r.ctypes.data, P, N, S, cost.ctypes.data) != 0:
        raise ValueError("ocd_fitness_from_returns: bad arguments")
    return cost


class NativeCMAES(_Termination):
    """The same ask / tell interface as CMAES over csrc/ocd_cma.c."""

    def __init__(self, x0, sigma0, popsize=None, seed=1, active=True):
        self.lib = load_cma_library()
        self.n = len(x0)
        x = np.ascontiguousarray(x0, dtype=np.float64)
        h = C.c_void_p()
        if self.lib.ocd_cma_create(self.n, x.ctypes.data_as(_D), float(sigma0), int(popsize or 0), int(seed) & 0xffffffff,
                                   C.byref(h)) != 0:
            raise ValueError("ocd_cma_create: bad arguments")
        self._h = h
        self.active = bool(active)
        if not active and self.lib.ocd_cma_set_active(h, 0) != 0:
            raise RuntimeError("ocd_cma_set_active failed")
        self.lam = int(self.lib.ocd_cma_popsize(h))
        self.mu = self.lam // 2
        self._X = np.empty((self.lam, self.n), dtype=np.float64)
        self._X_ptr = self._X.ctypes.data
        self._f = np.empty(self.lam, dtype=np.float64)
        self._f_ptr = self._f.ctypes.data
        self._ss = np.empty(13, dtype=np.float64)
        self._ss_ptr = self._ss.ctypes.data
        self._opts_c = np.zeros(N_STOP, dtype=np.float64)
        self._flags_c = np.zeros(N_STOP, dtype=np.int32)
        self._init_termination(sigma0)

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

    def resample(self, rows):
        """Replace the candidates in `rows` of the population last asked for by fresh draws (pycma's rejection
        sampling of NaN costs); returns the new rows [len(rows), n] (also written into ask()'s array)."""
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        for k in rows:
            if self.lib.ocd_cma_resample(self._h, int(k), self._X_ptr) != 0:
                raise IndexError(f"ocd_cma_resample: no slot {int(k)} in a population of {self.lam}")
        return self._X[rows]

    def add_evals(self, n):
        """Evaluations made outside tell(): candidates redrawn after a NaN cost (pycma's ask_and_eval counts them)."""
        if self.lib.ocd_cma_add_evals(self._h, int(n)) != 0:
            raise ValueError("ocd_cma_add_evals: bad arguments")

    def tell(self, X, fitness):
        """The update from the population last asked for.  X must BE that population (the array ask() returned, rows
        replaced by resample() included): the native update uses the steps it drew, not the numbers in X -- a
        modified X is refused.  Returns the number of non-finite costs (ranked last)."""
        if X is not self._X and not np.array_equal(X, self._X):
            raise ValueError("tell(X, ...) must get the population ask() returned, unmodified")
        self._f[...] = fitness
        nonfinite = self.lib.ocd_cma_tell(self._h, self._X_ptr, self._f_ptr)
        if nonfinite < 0:
            raise RuntimeError("ocd_cma_tell failed")
        self.last_nonfinite = nonfinite                            # (the termination history lives in the C state)
        self.nonfinite_total += nonfinite
        return nonfinite

    def _stop_opts(self, overrides):
        o = dict(self.opts)
        for k, v in overrides.items():
            if v is not None:
                if k not in o:
                    raise TypeError(f"unknown termination option {k!r}")
                o[k] = v
        return o

    def stop(self, tolfun=None, tolx=None, maxiter=None, last_fitness=None, **overrides):
        """pycma's termination rules, evaluated by csrc/ocd_cma.c:ocd_cma_stop on the history ocd_cma_tell keeps (the
        numpy twin runs the same rules in Python: _Termination.stop).  {} = go on."""
        o = self._stop_opts(dict(tolfun=tolfun, tolx=tolx, maxiter=maxiter, **overrides))
        for i, k in enumerate(STOP_NAMES):
            self._opts_c[i] = o.get(k, 0.0)
        n = self.lib.ocd_cma_stop(self._h, self._opts_c.ctypes.data, self._flags_c.ctypes.data)
        if n < 0:
            raise RuntimeError("ocd_cma_stop failed")
        return {k: o.get(k) for i, k in enumerate(STOP_NAMES) if self._flags_c[i]} if n else {}

    def run(self, args: "RunArgs", overrides):
        """ocd_cma_run: generations in native code until a termination rule holds, `args.max_generations` are done, or
        a generation is left evaluated-but-not-told because a cost is NaN.  Returns (generations told, stop dict,
        pending_nan)."""
        o = self._stop_opts(overrides)
        for i, k in enumerate(STOP_NAMES):
            args.stop_opts[i] = o.get(k, 0.0)
        args.X, args.cost = self._X_ptr, self._f_ptr
        done, pending = C.c_int64(0), C.c_int32(0)
        st = self.lib.ocd_cma_run(self._h, C.byref(args), C.byref(done), self._flags_c.ctypes.data, C.byref(pending))
        if st != 0:
            raise RuntimeError(f"ocd_cma_run -> {st}")
        self.lib.ocd_cma_stop_state(self._h, self._ss_ptr)
        self.last_nonfinite, self.nonfinite_total = int(self._ss[8]), int(self._ss[9])
        why = {k: o.get(k) for i, k in enumerate(STOP_NAMES) if self._flags_c[i]}
        return int(done.value), why, bool(pending.value)

    def strategy_parameters(self):
        """(weights [lam], dict of mueff, cc, cs, c1, cmu, damps, chiN, weight_sum) as the native state holds them."""
        w, c = np.empty(self.lam), np.empty(8)
        if self.lib.ocd_cma_weights(self._h, w.ctypes.data, c.ctypes.data) != 0:
            raise RuntimeError("ocd_cma_weights failed")
        return w, dict(zip(("mueff", "cc", "cs", "c1", "cmu", "damps", "chiN", "weight_sum"), c.tolist()))

    def finish_tell(self):                                         # (the numpy twin defers work; nothing to do here)
        pass

    def prepare(self):
        """Draw the next population's normal deviates now (while the GPU runs this generation): same stream."""
        if self.lib.ocd_cma_prepare(self._h) != 0:
            raise RuntimeError("ocd_cma_prepare failed")

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

    def _stop_state(self):
        self.lib.ocd_cma_stop_state(self._h, self._ss_ptr)
        ss = self._ss
        return dict(sigma=ss[0], max_axis=ss[1], min_axis=ss[2], gen=int(ss[3]), counteval=int(ss[4]),
                    sigma_max_std=ss[10], sigma_max_pc=ss[11])
