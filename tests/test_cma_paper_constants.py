"""The own CMA-ES (both twins) against the PUBLISHED algorithm, not against each other.

pycma -- what the reference hands MPC_ORD.eval_weights to (mpc_ord.py:33-45) -- is absent here, so the strategy parameters are
pinned to N. Hansen, "The CMA Evolution Strategy: A Tutorial" (arXiv 1604.00772), Table 1 and eqs. 46-58, for the reference's
own shape: n = 7 weights, pycma's default population lambda = 4 + floor(3 ln 7) = 9, mu = 4.

The numbers below were worked out from the equations with a 40-digit decimal calculator (not with either twin, not with numpy):

  eq. 49   w'_i = ln((lambda+1)/2) - ln i = ln 5 - ln i:
           1.609438, 0.916291, 0.510826, 0.223144, 0, -0.182322, -0.336472, -0.470004, -0.587787
           (round 5's twins used ln(mu + 0.5) = ln 4.5 here: the same thing for even lambda, not for lambda = 9)
  Table 1  mueff  = (sum_{i<=4} w'_i)^2 / sum_{i<=4} w'_i^2 = 3.259699^2 / 3.740604 = 2.840610
           mueff- = (sum_{i>4} w'_i)^2 / sum_{i>4} w'_i^2                          = 3.486867
  eq. 55   c_sigma = (mueff + 2) / (n + mueff + 5)                                 = 0.326173
           d_sigma = 1 + 2 max(0, sqrt((mueff - 1) / (n + 1)) - 1) + c_sigma       = 1.326173   (the max is 0)
  eq. 56   c_c  = (4 + mueff / n) / (n + 4 + 2 mueff / n)                          = 0.373006
  eq. 57   c_1  = 2 / ((n + 1.3)^2 + mueff)                                        = 0.027882
  eq. 58   c_mu = min(1 - c_1, 2 (mueff - 2 + 1 / mueff) / ((n + 2)^2 + mueff))    = 0.028450
  eq. 50   alpha_mu-     = 1 + c_1 / c_mu                                          = 1.980027   <- the minimum
  eq. 51   alpha_mueff-  = 1 + 2 mueff- / (mueff + 2)                              = 2.440672
  eq. 52   alpha_posdef- = (1 - c_1 - c_mu) / (n c_mu)                             = 4.738418
  eq. 53   w_i = w'_i / 3.259699 (i <= 4);  w_i = 1.980027 w'_i / 1.576585 (i > 4)
"""
import numpy as np
import pytest

from l4dc_mpc_ocd_amd.interact_drive.reward_design.cmaes import CMAES, NativeCMAES

W = [0.493738377484340919, 0.281096832480643307, 0.156709502558070078, 0.068455287476945696, 0.0,
     -0.228977013800854956, -0.422574320465514076, -0.590275935510337549, -0.738199243402432935]
CONST = dict(mueff=2.840610429717054077, cc=0.373006229336514121, cs=0.326173269801904198, c1=0.027882099260254255,
             cmu=0.028450351990791164, damps=1.326173269801904198, chiN=2.553831379703503017)


def _params(es):
    if isinstance(es, NativeCMAES):
        return es.strategy_parameters()
    return es.weights_all, dict(mueff=es.mueff, cc=es.cc, cs=es.cs, c1=es.c1, cmu=es.cmu, damps=es.damps, chiN=es.chiN,
                                weight_sum=es.weights_all.sum())


@pytest.mark.parametrize("cls", [CMAES, NativeCMAES])
def test_strategy_parameters_of_the_reference_shape_are_the_tutorials(cls):
    es = cls([0.0] * 7, 0.05)                                   # popsize None: pycma's default
    assert es.lam == 9 and es.mu == 4
    w, c = _params(es)
    np.testing.assert_allclose(w, W, rtol=0, atol=2e-15)
    for k, v in CONST.items():
        assert abs(c[k] - v) <= 4e-15 * max(1.0, abs(v)), (k, c[k], v)
    # the positive weights sum to one; the negative ones to -alpha_mu- = -(1 + c1 / cmu): the decay factor of eq. 47,
    # 1 - c1 - cmu sum_j w_j, is then exactly 1 (the tutorial's remark under eq. 53)
    assert abs(sum(w[:4]) - 1.0) < 1e-15 and abs(c["weight_sum"] - (1.0 - 1.980026513179139516)) < 4e-15
    assert abs((1 - c["c1"] - c["cmu"] * c["weight_sum"]) - 1.0) < 1e-15


@pytest.mark.parametrize("cls", [CMAES, NativeCMAES])
def test_without_the_active_update_the_negative_weights_are_zero(cls):
    es = cls([0.0] * 7, 0.05, active=False)
    w, c = _params(es)
    np.testing.assert_allclose(w[:4], W[:4], rtol=0, atol=2e-15)
    assert np.all(np.asarray(w[4:]) == 0.0) and abs(c["weight_sum"] - 1.0) < 1e-15
    for k, v in CONST.items():                                  # the learning rates do not depend on the switch
        assert abs(c[k] - v) <= 4e-15 * max(1.0, abs(v)), k


def test_even_populations_keep_round_fives_positive_weights():
    """ln((lambda+1)/2) = ln(mu + 0.5) when lambda is even: BASELINE's populations (16, 64, 128, 256) draw the same mean
    updates as before; only the rank-mu term gained the negative ranks."""
    for lam in (16, 64, 128, 256):
        es = CMAES([0.0] * 7, 0.05, popsize=lam)
        mu = lam // 2
        old = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
        np.testing.assert_allclose(es.weights, old / old.sum(), rtol=0, atol=1e-15)
        assert np.all(es.weights_all[mu:] < 0)


@pytest.mark.parametrize("active", [True, False])
def test_one_covariance_update_is_eq_47_term_by_term(active):
    """tell() against eq. 47 written out with plain loops over the ranks (from the paper, not from the twins' matrix code),
    in the first generation where C = I, so C^(-1/2) y = y."""
    n, lam = 7, 9
    a, b = CMAES([0.3] * n, 0.2, seed=4, active=active), NativeCMAES([0.3] * n, 0.2, seed=4, active=active)
    X = a.ask().copy()
    Xb = b.ask()
    np.testing.assert_allclose(Xb, X, rtol=1e-13, atol=1e-15)
    f = np.sum((X - 1.0) ** 2 * np.arange(1, n + 1), axis=1)
    y = (X - 0.3) / 0.2
    order = np.argsort(f, kind="stable")
    w = np.array(W) if active else np.array(W[:4] + [0.0] * 5)
    c1, cmu, cc, cs, mueff, chiN = (CONST[k] for k in ("c1", "cmu", "cc", "cs", "mueff", "chiN"))
    yw = sum(w[k] * y[order[k]] for k in range(4))
    ps = np.sqrt(cs * (2 - cs) * mueff) * yw                     # C^(-1/2) = I, ps was 0
    hsig = np.linalg.norm(ps) / np.sqrt(1 - (1 - cs) ** 2) / chiN < 1.4 + 2 / (n + 1)
    pc = hsig * np.sqrt(cc * (2 - cc) * mueff) * yw
    Cn = (1 + c1 * (1 - hsig) * cc * (2 - cc) - c1 - cmu * w.sum()) * np.eye(n) + c1 * np.outer(pc, pc)
    for k in range(lam):
        yk = y[order[k]]
        wk = w[k] if w[k] >= 0 else w[k] * n / float(yk @ yk)    # eq. 46
        Cn = Cn + cmu * wk * np.outer(yk, yk)
    a.tell(X, f)
    b.tell(Xb, f)
    np.testing.assert_allclose(a.C, Cn, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(b.C, Cn, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(a.mean, 0.3 + 0.2 * yw, rtol=1e-13)
    assert np.all(np.linalg.eigvalsh(a.C) > 0)


def test_active_update_converges_faster_on_an_ill_conditioned_quadratic():
    """What the negative weights are for (tutorial section 3.3 / Jastrebski & Arnold 2006): the same seed reaches the same
    cost in fewer generations."""
    def gens(active):
        es = NativeCMAES([1.0] * 7, 0.3, seed=3, active=active)
        for g in range(2000):
            X = es.ask()
            f = np.sum((X ** 2) * 10.0 ** (np.arange(7) * 4 / 6), axis=1)
            es.tell(X, f)
            if es.best_f < 1e-10:
                return g
        return 2000
    assert gens(True) < gens(False) < 2000
