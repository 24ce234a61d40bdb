/* ocd_cma.c -- host side of a CMA-ES generation in native code: ask, tell and the float64 fitness reduction.
 *
 * What it replaces (reference file:line): the reference hands MPC_ORD.eval_weights to pycma's
 * cma.evolution_strategy.fmin2 (interact_drive/reward_design/mpc_ord.py:33-45; pycma is an unpinned, un-vendored
 * dependency, setup.py:6) and reduces the episode returns in Python (mpc_ord.py:126-151).  Around a 1.6 ms
 * episode kernel the numpy version of these steps costs ~70 us of call overhead per generation for ~2 us of
 * arithmetic (7 weights, 64 candidates); here they are three C calls.
 *
 * Algorithm: the (mu/mu_w, lambda)-CMA-ES of N. Hansen, "The CMA Evolution Strategy: A Tutorial" (arXiv 1604.00772),
 * with the default strategy parameters of its Table 1 -- recombination weights w'_i = ln((lambda+1)/2) - ln i over ALL
 * lambda ranks (eq. 49), the positive ones normalised to sum 1, the negative ones scaled by min(alpha_mu^-,
 * alpha_mueff^-, alpha_posdef^-) (eqs. 50-53) and used in the rank-mu update only ("active" CMA, eqs. 46-47: what pycma
 * runs by default, CMA_active=True); ocd_cma_set_active(es, 0) drops the negative weights -- the same formulas as
 * interact_drive/reward_design/cmaes.py (the numpy twin the tests compare it with; tests/test_cma_paper_constants.py
 * types the constants of n = 7, lambda = 9 from the equations).  Candidates are m + sigma * C^(1/2) z with the SYMMETRIC square root, so the sample path
 * does not depend on the order or sign of the eigenvectors (numpy's LAPACK and the Jacobi sweep below give the same
 * candidates to rounding).  z comes from MT19937 + the polar method exactly as numpy.random.RandomState(seed)
 * .standard_normal does (bit-identical stream; tests/test_host_mirror.py).  The sampling sequence is not pycma's:
 * optimisation traces are "parity unpinned" (SURVEY.md 8c); the fitness values are bit-exact.
 *
 * Plain C11, no HIP: libocd_cma.so.  Declared in include/ocd_cma.h. */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime, pthreads under -std=c11 */
#include "../../include/ocd_cma.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MT_N 624
#define MT_M 397

struct ocd_cma {
    int n, lam, mu, active;
    double sigma, mueff, cc, cs, c1, cmu, damps, chiN;
    double wsum_all;          /* sum of all lambda weights as the C update sees them (1 without the negative ones) */
    double *mean, *weights, *wraw, *wo, *pc, *ps, *C, *B, *D, *sqrtC, *invsqrtC, *y, *best_x, *tmp, *work;
    double best_f, max_d, min_d;
    double fit_best, fit_median, fit_worst;   /* of the population last told (non-finite costs excluded from best / worst) */
    int64_t gen, counteval, nonfinite_total;
    int last_nonfinite;
    /* termination history (pycma's rules, see ocd_cma_stop): best costs of the last hist_cap generations (newest
     * first), every generation's best / median cost (oldest first, capped), flat-fitness run, last population range */
    double sigma0, last_pop_range;
    double *hist, *histbest, *histmedian, *medtmp;
    int hist_cap, hist_len, flat;
    int64_t nhb, hb_cap;
    int *order;
    double *z;                /* [lam, n] the normal deviates of the NEXT population, drawn ahead by ocd_cma_prepare */
    int z_ready;
    /* numpy.random.RandomState(seed): MT19937 + cached second value of the polar method */
    uint32_t mt[MT_N];
    int mti, has_gauss;
    double gauss;
};

/* ---- MT19937 as numpy's legacy RandomState seeds and draws it ---- */
static void mt_seed(ocd_cma *es, uint32_t seed)
{
    for (int pos = 0; pos < MT_N; ++pos) {
        es->mt[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
    es->mti = MT_N;
    es->has_gauss = 0;
    es->gauss = 0.0;
}

static uint32_t mt_next(ocd_cma *es)
{
    if (es->mti >= MT_N) {
        uint32_t *mt = es->mt, y;
        int kk;
        for (kk = 0; kk < MT_N - MT_M; ++kk) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < MT_N - 1; ++kk) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        es->mti = 0;
    }
    uint32_t y = es->mt[es->mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

static double mt_double(ocd_cma *es)
{
    const int32_t a = (int32_t)(mt_next(es) >> 5), b = (int32_t)(mt_next(es) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

static double gauss(ocd_cma *es)          /* numpy's legacy_gauss */
{
    if (es->has_gauss) {
        const double t = es->gauss;
        es->has_gauss = 0;
        es->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(es) - 1.0;
        x2 = 2.0 * mt_double(es) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    es->gauss = f * x1;
    es->has_gauss = 1;
    return f * x2;
}

/* ---- symmetric eigendecomposition, cyclic Jacobi: A (n x n, symmetric, destroyed) -> eigenvalues d, vectors V (columns) ---- */
static void jacobi_eigh(int n, double *A, double *d, double *V)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-60 + 1e-34 * diag) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {              /* A <- A J */
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {              /* A <- J^T A */
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) d[i] = A[i * n + i];
}

/* sqrtC = B diag(D) B^T, invsqrtC = B diag(1/D) B^T from C (D = sqrt(max(eigenvalue, 1e-20))) */
static void decompose(ocd_cma *es)
{
    const int n = es->n;
    memcpy(es->work, es->C, sizeof(double) * n * n);
    jacobi_eigh(n, es->work, es->D, es->B);
    es->max_d = 0.0;
    es->min_d = INFINITY;
    for (int i = 0; i < n; ++i) {
        es->D[i] = sqrt(es->D[i] > 1e-20 ? es->D[i] : 1e-20);
        if (es->D[i] > es->max_d) es->max_d = es->D[i];
        if (es->D[i] < es->min_d) es->min_d = es->D[i];
    }
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < n; ++k) {
                const double bb = es->B[i * n + k] * es->B[j * n + k];
                a += bb * es->D[k];
                b += bb / es->D[k];
            }
            es->sqrtC[i * n + j] = es->sqrtC[j * n + i] = a;
            es->invsqrtC[i * n + j] = es->invsqrtC[j * n + i] = b;
        }
}

void ocd_cma_destroy(ocd_cma *es);

/* eqs. 50-53: the final weights from w' -- positive ones / their sum; negative ones * min(alpha_mu^-, alpha_mueff^-,
 * alpha_posdef^-) / the sum of their magnitudes (zero without the active update). */
static void set_weights(ocd_cma *es)
{
    const int lam = es->lam, mu = es->mu;
    double pos = 0.0, neg = 0.0, nsum = 0.0, nsq = 0.0;
    for (int i = 0; i < mu; ++i) pos += es->wraw[i];
    for (int i = mu; i < lam; ++i) { neg += fabs(es->wraw[i]); nsum += es->wraw[i]; nsq += es->wraw[i] * es->wraw[i]; }
    const double mueff_neg = nsq > 0.0 ? nsum * nsum / nsq : 0.0;
    const double a_mu = 1.0 + es->c1 / es->cmu;                                   /* eq. 50 */
    const double a_mueff = 1.0 + 2.0 * mueff_neg / (es->mueff + 2.0);             /* eq. 51 */
    const double a_posdef = (1.0 - es->c1 - es->cmu) / ((double)es->n * es->cmu); /* eq. 52 */
    double a = a_mu < a_mueff ? a_mu : a_mueff;
    if (a_posdef < a) a = a_posdef;
    es->wsum_all = 0.0;
    for (int i = 0; i < lam; ++i) {
        if (i < mu) es->weights[i] = es->wraw[i] / pos;
        else es->weights[i] = (es->active && neg > 0.0) ? a * es->wraw[i] / neg : 0.0;
        es->wsum_all += es->weights[i];
    }
}

/* The active update (negative recombination weights in the rank-mu term) on / off; on by default, as in pycma.  Only
 * before the first tell. */
int32_t ocd_cma_set_active(ocd_cma *es, int32_t on)
{
    if (!es || es->gen != 0) return -1;
    es->active = on ? 1 : 0;
    set_weights(es);
    return 0;
}

int32_t ocd_cma_weights(const ocd_cma *es, double *w, double *consts)
{
    if (!es) return -1;
    if (w) memcpy(w, es->weights, sizeof(double) * (size_t)es->lam);
    if (consts) { consts[0] = es->mueff; consts[1] = es->cc; consts[2] = es->cs; consts[3] = es->c1; consts[4] = es->cmu;
                  consts[5] = es->damps; consts[6] = es->chiN; consts[7] = es->wsum_all; }
    return 0;
}

int32_t ocd_cma_create(int32_t n, const double *x0, double sigma0, int32_t popsize, uint32_t seed, ocd_cma **out)
{
    if (!out) return -1;
    *out = NULL;
    if (n < 1 || n > OCD_CMA_MAX_DIM || !x0 || !(sigma0 > 0.0) || popsize < 0) return -1;
    ocd_cma *es = (ocd_cma *)calloc(1, sizeof(ocd_cma));
    if (!es) return -1;
    es->n = n;
    es->lam = popsize > 0 ? popsize : 4 + (int)(3.0 * log((double)n));
    if (es->lam < 2) { free(es); return -1; }
    es->mu = es->lam / 2;
    const size_t nn = (size_t)n * n;
    const size_t total = (size_t)n * 6 + (size_t)es->lam * 3 + nn * 6 + (size_t)es->lam * n * 2;
    double *mem = (double *)calloc(total, sizeof(double));
    es->order = (int *)calloc((size_t)es->lam, sizeof(int));
    if (!mem || !es->order) { free(mem); free(es->order); free(es); return -1; }
    double *p = mem;
    es->mean = p; p += n; es->pc = p; p += n; es->ps = p; p += n; es->D = p; p += n; es->best_x = p; p += n; es->tmp = p; p += n;
    es->weights = p; p += es->lam;            /* [lam]: positive (first mu), then <= 0 */
    es->wraw = p; p += es->lam;               /* w'_i of eq. 49 */
    es->wo = p; p += es->lam;                 /* scratch of tell: the weights of this update (eq. 46) */
    es->C = p; p += nn; es->B = p; p += nn; es->sqrtC = p; p += nn; es->invsqrtC = p; p += nn; es->work = p; p += 2 * nn;
    es->y = p; p += (size_t)es->lam * n;
    es->z = p;
    memcpy(es->mean, x0, sizeof(double) * n);
    es->sigma = sigma0;
    /* eq. 49: w'_i = ln((lambda + 1) / 2) - ln i, i = 1..lambda; positive exactly for i <= mu = floor(lambda / 2)
     * (w'_(mu+1) = 0 when lambda is odd) */
    double wsum = 0.0, w2 = 0.0;
    for (int i = 0; i < es->lam; ++i) es->wraw[i] = log(((double)es->lam + 1.0) / 2.0) - log((double)(i + 1));
    for (int i = 0; i < es->mu; ++i) { wsum += es->wraw[i]; w2 += es->wraw[i] * es->wraw[i]; }
    es->mueff = wsum * wsum / w2;             /* Table 1: (sum_{i<=mu} w'_i)^2 / sum_{i<=mu} w'_i^2 */
    const double me = es->mueff, dn = (double)n;
    es->cc = (4 + me / dn) / (dn + 4 + 2 * me / dn);                 /* eq. 56 */
    es->cs = (me + 2) / (dn + me + 5);                               /* eq. 55 */
    es->c1 = 2 / ((dn + 1.3) * (dn + 1.3) + me);                     /* eq. 57, alpha_cov = 2 */
    es->cmu = 2 * (me - 2 + 1 / me) / ((dn + 2) * (dn + 2) + me);    /* eq. 58, alpha_cov = 2 */
    if (es->cmu > 1 - es->c1) es->cmu = 1 - es->c1;
    es->active = 1;
    set_weights(es);
    const double dm = sqrt((me - 1) / (dn + 1)) - 1;
    es->damps = 1 + 2 * (dm > 0.0 ? dm : 0.0) + es->cs;
    es->chiN = sqrt(dn) * (1 - 1 / (4 * dn) + 1 / (21 * dn * dn));
    for (int i = 0; i < n; ++i) {
        es->C[i * n + i] = es->B[i * n + i] = es->sqrtC[i * n + i] = es->invsqrtC[i * n + i] = 1.0;
        es->D[i] = 1.0;
    }
    es->max_d = es->min_d = 1.0;
    es->best_f = INFINITY;
    es->fit_best = es->fit_median = es->fit_worst = NAN;
    es->sigma0 = sigma0;
    es->last_pop_range = INFINITY;
    es->hist_cap = (int)floor(10.0 + 30.0 * (double)n / (double)es->lam);
    es->hb_cap = 20000;
    es->hist = (double *)calloc((size_t)es->hist_cap + 1, sizeof(double));
    es->histbest = (double *)calloc((size_t)es->hb_cap, sizeof(double));
    es->histmedian = (double *)calloc((size_t)es->hb_cap, sizeof(double));
    es->medtmp = (double *)calloc((size_t)es->hb_cap, sizeof(double));
    if (!es->hist || !es->histbest || !es->histmedian || !es->medtmp) { ocd_cma_destroy(es); return -1; }
    mt_seed(es, seed);
    *out = es;
    return 0;
}

void ocd_cma_destroy(ocd_cma *es)
{
    if (!es) return;
    free(es->mean);           /* the one block all double arrays live in */
    free(es->order);
    free(es->hist); free(es->histbest); free(es->histmedian); free(es->medtmp);
    free(es);
}

int32_t ocd_cma_popsize(const ocd_cma *es) { return es ? es->lam : -1; }
int32_t ocd_cma_abi_version(void) { return OCD_CMA_ABI_VERSION; }

/* The deviates do not depend on the state of the search: a caller with something to wait for (the running episode
 * kernel) draws the next population's while it waits; ask() draws them itself otherwise.  Same stream either way. */
int32_t ocd_cma_prepare(ocd_cma *es)
{
    if (!es) return -1;
    if (!es->z_ready) {
        const size_t m = (size_t)es->lam * es->n;
        for (size_t i = 0; i < m; ++i) es->z[i] = gauss(es);
        es->z_ready = 1;
    }
    return 0;
}

int32_t ocd_cma_ask(ocd_cma *es, double *X)
{
    if (!es || !X) return -1;
    const int n = es->n, lam = es->lam;
    ocd_cma_prepare(es);
    es->z_ready = 0;
    for (int k = 0; k < lam; ++k) {
        const double *z = es->z + (size_t)k * n;
        double *y = es->y + (size_t)k * n;
        for (int i = 0; i < n; ++i) {               /* y = C^(1/2) z (symmetric root) */
            double a = 0.0;
            for (int j = 0; j < n; ++j) a += es->sqrtC[i * n + j] * z[j];
            y[i] = a;
            X[(size_t)k * n + i] = es->mean[i] + es->sigma * a;
        }
    }
    return 0;
}

/* One more candidate for row k of the population last asked for: pycma's ask_and_eval draws a replacement for a
 * candidate whose cost is NaN and evaluates that instead (rejection sampling, cma.evolution_strategy.fmin2 as called
 * from mpc_ord.py:41).  n fresh deviates from the stream (after any block ocd_cma_prepare has drawn ahead). */
int32_t ocd_cma_resample(ocd_cma *es, int32_t k, double *X)
{
    if (!es || !X || k < 0 || k >= es->lam) return -1;
    const int n = es->n;
    double *z = es->work;                        /* scratch: decompose() is not running */
    for (int i = 0; i < n; ++i) z[i] = gauss(es);
    double *y = es->y + (size_t)k * n;
    for (int i = 0; i < n; ++i) {
        double a = 0.0;
        for (int j = 0; j < n; ++j) a += es->sqrtC[i * n + j] * z[j];
        y[i] = a;
        X[(size_t)k * n + i] = es->mean[i] + es->sigma * a;
    }
    return 0;
}

/* Sort key of a cost: NaN ranks after everything else (numpy's argsort order, which the numpy twin uses; a raw `>`
 * comparison would leave a NaN wherever it was inserted and stop smaller costs from passing it). */
static inline int cost_after(double a, double b)       /* a strictly after b */
{
    if (isnan(a)) return !isnan(b);
    if (isnan(b)) return 0;
    return a > b;
}

int32_t ocd_cma_tell(ocd_cma *es, const double *X, const double *fitness)
{
    if (!es || !X || !fitness) return -1;
    const int n = es->n, lam = es->lam, mu = es->mu;
    /* stable argsort of the fitness, NaN last (insertion sort: lam is small) */
    int nonfinite = 0, n_nan = 0;
    for (int k = 0; k < lam; ++k) {
        int j = k;
        while (j > 0 && cost_after(fitness[es->order[j - 1]], fitness[k])) { es->order[j] = es->order[j - 1]; --j; }
        es->order[j] = k;
        if (!isfinite(fitness[k])) ++nonfinite;
        if (isnan(fitness[k])) ++n_nan;
    }
    es->last_nonfinite = nonfinite;
    es->nonfinite_total += nonfinite;
    {   /* statistics of this population for the termination rules: best, median (numpy's: NaN if any), worst non-NaN */
        const int m = lam - n_nan;
        es->fit_best = m > 0 ? fitness[es->order[0]] : NAN;
        es->fit_worst = m > 0 ? fitness[es->order[m - 1]] : NAN;
        es->fit_median = n_nan ? NAN : ((lam & 1) ? fitness[es->order[lam / 2]]
                                                  : 0.5 * (fitness[es->order[lam / 2 - 1]] + fitness[es->order[lam / 2]]));
    }
    const int b = es->order[0];
    if (fitness[b] < es->best_f) {
        es->best_f = fitness[b];
        memcpy(es->best_x, X + (size_t)b * n, sizeof(double) * n);
    }
    es->counteval += lam;
    double *yw = es->tmp;
    for (int i = 0; i < n; ++i) yw[i] = 0.0;
    for (int k = 0; k < mu; ++k) {
        const double *y = es->y + (size_t)es->order[k] * n;
        for (int i = 0; i < n; ++i) yw[i] += es->weights[k] * y[i];
    }
    for (int i = 0; i < n; ++i) es->mean[i] += es->sigma * yw[i];
    const double cps = sqrt(es->cs * (2 - es->cs) * es->mueff);
    double ps2 = 0.0;
    for (int i = 0; i < n; ++i) {
        double a = 0.0;
        for (int j = 0; j < n; ++j) a += es->invsqrtC[i * n + j] * yw[j];
        es->ps[i] = (1 - es->cs) * es->ps[i] + cps * a;
        ps2 += es->ps[i] * es->ps[i];
    }
    const double ps_norm = sqrt(ps2);
    const int hsig = ps_norm / sqrt(1 - pow(1 - es->cs, 2.0 * (double)es->counteval / lam)) / es->chiN < 1.4 + 2.0 / (n + 1);
    const double cpc = hsig ? sqrt(es->cc * (2 - es->cc) * es->mueff) : 0.0;
    for (int i = 0; i < n; ++i) es->pc[i] = (1 - es->cc) * es->pc[i] + cpc * yw[i];
    /* eq. 47: C <- (1 + c1 delta(hsig) - c1 - cmu sum_j w_j) C + c1 pc pc^T + cmu sum_i w_i^o y_i y_i^T over all lambda
     * ranks, w_i^o = w_i for w_i >= 0 and w_i * n / ||C^(-1/2) y_i||^2 for the negative ones (eq. 46) */
    const double keep = (1 - es->c1 - es->cmu * es->wsum_all) + es->c1 * (hsig ? 0.0 : es->cc * (2 - es->cc));
    const int n_rank = es->active ? lam : mu;
    double *wo = es->wo;
    for (int k = 0; k < n_rank; ++k) {
        wo[k] = es->weights[k];
        if (wo[k] < 0.0) {
            const double *y = es->y + (size_t)es->order[k] * n;
            double m2 = 0.0;
            for (int i = 0; i < n; ++i) {
                double a = 0.0;
                for (int j = 0; j < n; ++j) a += es->invsqrtC[i * n + j] * y[j];
                m2 += a * a;
            }
            wo[k] = m2 > 0.0 ? wo[k] * (double)n / m2 : 0.0;
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double rank_mu = 0.0;
            for (int k = 0; k < n_rank; ++k) {
                const double *y = es->y + (size_t)es->order[k] * n;
                rank_mu += wo[k] * y[i] * y[j];
            }
            const double c = keep * es->C[i * n + j] + es->c1 * es->pc[i] * es->pc[j] + es->cmu * rank_mu;
            es->C[i * n + j] = es->C[j * n + i] = c;        /* symmetric by construction */
        }
    es->sigma *= exp((es->cs / es->damps) * (ps_norm / es->chiN - 1));
    decompose(es);
    es->gen += 1;
    {   /* termination history (cmaes._Termination._record) */
        const double best = es->fit_best, worst = es->fit_worst, median = es->fit_median;
        const int keep = es->hist_len < es->hist_cap ? es->hist_len : es->hist_cap - 1;
        memmove(es->hist + 1, es->hist, sizeof(double) * (size_t)(keep > 0 ? keep : 0));
        es->hist[0] = best;
        es->hist_len = keep + 1;
        if (es->nhb == es->hb_cap) {
            memmove(es->histbest, es->histbest + 1, sizeof(double) * (size_t)(es->hb_cap - 1));
            memmove(es->histmedian, es->histmedian + 1, sizeof(double) * (size_t)(es->hb_cap - 1));
            es->nhb -= 1;
        }
        es->histbest[es->nhb] = best;
        es->histmedian[es->nhb] = median;
        es->nhb += 1;
        es->last_pop_range = (isfinite(worst) && isfinite(best)) ? worst - best : INFINITY;
        es->flat = (best == median) ? es->flat + 1 : 0;
    }
    return nonfinite;
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* numpy.median of v[0..n): NaN if any element is NaN */
static double median_of(ocd_cma *es, const double *v, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { if (isnan(v[i])) return NAN; es->medtmp[i] = v[i]; }
    qsort(es->medtmp, (size_t)n, sizeof(double), cmp_double);
    return (n & 1) ? es->medtmp[n / 2] : 0.5 * (es->medtmp[n / 2 - 1] + es->medtmp[n / 2]);
}

/* pycma's termination rules (reward_design/cmaes.py: _Termination.stop is the same logic in Python, used by the numpy
 * twin; tests compare the two).  opts / flags in the order: maxiter, maxfevals, tolfun, tolfunhist, tolx, tolfacupx,
 * tolconditioncov, tolupsigma, tolstagnation, tolflatfitness, noeffectaxis, noeffectcoord (the last two have no option
 * value).  flags[i] = 1 where the condition holds; returns how many. */
int32_t ocd_cma_stop(ocd_cma *es, const double opts[OCD_CMA_N_STOP], int32_t flags[OCD_CMA_N_STOP])
{
    if (!es || !opts || !flags) return -1;
    const int n = es->n;
    int count = 0;
    for (int i = 0; i < OCD_CMA_N_STOP; ++i) flags[i] = 0;
    if ((double)es->gen >= opts[0]) flags[0] = 1;
    if ((double)es->counteval >= opts[1]) flags[1] = 1;
    if (es->gen > 0) {
        double hmax = -INFINITY, hmin = INFINITY;
        int nfin = 0;
        for (int i = 0; i < es->hist_len; ++i)
            if (isfinite(es->hist[i])) { ++nfin; if (es->hist[i] > hmax) hmax = es->hist[i]; if (es->hist[i] < hmin) hmin = es->hist[i]; }
        const double hist_range = nfin ? hmax - hmin : INFINITY;
        double dmax = 0.0, pmax = 0.0;
        for (int i = 0; i < n; ++i) {
            const double d = sqrt(es->C[i * n + i]), pc = fabs(es->pc[i]);
            if (d > dmax) dmax = d;
            if (pc > pmax) pmax = pc;
        }
        const double smax_std = es->sigma * dmax, smax_pc = es->sigma * pmax;
        if (es->last_pop_range < opts[2] && hist_range < opts[2]) flags[2] = 1;
        if (es->hist_len > 9 && hist_range < opts[3]) flags[3] = 1;
        if (smax_std < opts[4] && smax_pc < opts[4]) flags[4] = 1;
        if (smax_std > es->sigma0 * opts[5]) flags[5] = 1;
        if (es->max_d > sqrt(opts[6]) * es->min_d) flags[6] = 1;
        if (es->sigma / es->sigma0 > opts[7] * es->max_d) flags[7] = 1;
        if ((double)es->flat > opts[9]) flags[9] = 1;
        const int64_t nb = es->nhb;
        if ((double)es->gen > (double)n * (5.0 + 100.0 / (double)es->lam) && nb > 100) {
            const double a = opts[8] / 5.0 / 2.0, b = (double)nb / 10.0;
            const int64_t ell = (int64_t)(a > b ? a : b);
            if (2 * ell < nb && ell > 0) {
                /* the newest ell generations against the ell JUST BEFORE them (pycma keeps its lists newest first and
                 * compares histbest[:l] with histbest[l:2l]) -- not against the start of the run, where costs are worst */
                const double m_new = median_of(es, es->histmedian + (nb - ell), ell);
                const double m_old = median_of(es, es->histmedian + (nb - 2 * ell), ell);
                const double b_new = median_of(es, es->histbest + (nb - ell), ell);
                const double b_old = median_of(es, es->histbest + (nb - 2 * ell), ell);
                if (m_new >= m_old && b_new >= b_old) flags[8] = 1;
            }
        }
        /* noeffectaxis: a step of 0.1 sigma along principal axis (generation mod n) -- axes in ascending order of
         * their length, as numpy's eigh returns them to the numpy twin and to pycma -- no longer changes the mean in
         * any coordinate; noeffectcoord: a step of 0.2 sigma sqrt(C_jj) no longer changes coordinate j.  pycma has
         * no option value for either (always on). */
        {
            const int want = (int)(es->gen % n);
            int k = 0;
            for (int c = 0; c < n; ++c) {                 /* the axis with exactly `want` shorter (or equal, earlier) axes */
                int rank = 0;
                for (int j = 0; j < n; ++j) rank += (es->D[j] < es->D[c]) || (es->D[j] == es->D[c] && j < c);
                if (rank == want) { k = c; break; }
            }
            int same = 0, coord = 0;
            for (int j = 0; j < n; ++j) {
                volatile double moved = es->mean[j] + 0.1 * es->sigma * es->D[k] * es->B[j * n + k];
                same += (es->mean[j] == moved);
                volatile double moved_c = es->mean[j] + 0.2 * es->sigma * sqrt(es->C[j * n + j]);
                coord |= (es->mean[j] == moved_c);
            }
            if (same == n) flags[10] = 1;
            if (coord) flags[11] = 1;
        }
    }
    for (int i = 0; i < OCD_CMA_N_STOP; ++i) count += flags[i];
    return count;
}

/* Evaluations made outside tell(): the candidates redrawn after a NaN cost (pycma's ask_and_eval counts every
 * evaluation, the rejected ones included, so maxfevals and the hsig correction see them). */
int32_t ocd_cma_add_evals(ocd_cma *es, int64_t n)
{
    if (!es || n < 0) return -1;
    es->counteval += n;
    return 0;
}

int32_t ocd_cma_state(const ocd_cma *es, double *mean, double *sigma, double *C, double *best_x, double *best_f,
                      int64_t *gen, int64_t *counteval, double *max_axis)
{
    if (!es) return -1;
    const int n = es->n;
    if (mean) memcpy(mean, es->mean, sizeof(double) * n);
    if (sigma) *sigma = es->sigma;
    if (C) memcpy(C, es->C, sizeof(double) * n * n);
    if (best_x) memcpy(best_x, es->best_x, sizeof(double) * n);
    if (best_f) *best_f = es->best_f;
    if (gen) *gen = es->gen;
    if (counteval) *counteval = es->counteval;
    if (max_axis) *max_axis = es->max_d;
    return 0;
}

/* The numbers the termination rules of reward_design/cmaes.py read after a tell, in one call:
 * out[0..4] = sigma, largest and smallest sqrt-eigenvalue of C, generations, evaluations;
 * out[5..7] = best / median / worst (NaN excluded from best and worst) cost of the population last told;
 * out[8..9] = non-finite costs in that population / in all populations told so far;
 * out[10] = max_i sigma * sqrt(C_ii), out[11] = max_i sigma * |pc_i| (pycma's tolx reads both), out[12] = min_i sqrt(C_ii). */
int32_t ocd_cma_stop_state(const ocd_cma *es, double out[13])
{
    if (!es || !out) return -1;
    const int n = es->n;
    double dmax = 0.0, dmin = INFINITY, pmax = 0.0;
    for (int i = 0; i < n; ++i) {
        const double d = sqrt(es->C[i * n + i]), p = fabs(es->pc[i]);
        if (d > dmax) dmax = d;
        if (d < dmin) dmin = d;
        if (p > pmax) pmax = p;
    }
    out[0] = es->sigma; out[1] = es->max_d; out[2] = es->min_d; out[3] = (double)es->gen; out[4] = (double)es->counteval;
    out[5] = es->fit_best; out[6] = es->fit_median; out[7] = es->fit_worst;
    out[8] = (double)es->last_nonfinite; out[9] = (double)es->nonfinite_total;
    out[10] = es->sigma * dmax; out[11] = es->sigma * pmax; out[12] = dmin;
    return 0;
}

/* [P*N*S] fp32 sample rewards -> [P] costs with the reference's accumulation types (mpc_ord.py:102,126-151): samples
 * summed sequentially in fp32, inits sequentially in float64, / num_samples, negated. */
int32_t ocd_fitness_from_returns(const float *returns, int64_t P, int64_t N, int64_t S, double *cost_out)
{
    if (!returns || !cost_out || P < 0 || N < 1 || S < 1) return -1;
    for (int64_t p = 0; p < P; ++p) {
        double total = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            const float *r = returns + (p * N + i) * S;
            float per_init = r[0];
            for (int64_t s = 1; s < S; ++s) per_init = per_init + r[s];
            total = (i == 0) ? (double)per_init : total + (double)per_init;
        }
        total = total / (double)S;
        cost_out[p] = -total;
    }
    return 0;
}

/* The three float64 normalisations a candidate goes through on its way to the planning car (mpc_ord.py:120, :71,
 * linear_reward_car.py:45-47: weights / np.linalg.norm(weights), then the fp32 assign), for P rows of D weights.
 * np.linalg.norm of a 1-D float64 vector is sqrt(dot(x, x)) and the dot is the BLAS numpy links: its summation order
 * is not ours to define, so the caller picks the variant that reproduces it on THIS machine (self-check in
 * scenarios.py) -- 0: dot = dot + x*x left to right; 1: dot = fma(x, x, dot) left to right (OpenBLAS' scalar tail
 * loop, compiled with contraction: what numpy 2.2 does for D < 16 here) -- or keeps the numpy path. */
int32_t ocd_normalise_weights(const double *W, int64_t P, int64_t D, int32_t variant, float *out)
{
    if (!W || !out || P < 0 || D < 1 || D > OCD_CMA_MAX_DIM || variant < 0 || variant > 1) return -1;
    double row[OCD_CMA_MAX_DIM];
    for (int64_t p = 0; p < P; ++p) {
        for (int64_t i = 0; i < D; ++i) row[i] = W[p * D + i];
        for (int pass = 0; pass < 3; ++pass) {
            double dot = 0.0;
            if (variant == 1) for (int64_t i = 0; i < D; ++i) dot = __builtin_fma(row[i], row[i], dot);
            else for (int64_t i = 0; i < D; ++i) dot = dot + row[i] * row[i];
            const double nrm = sqrt(dot);
            for (int64_t i = 0; i < D; ++i) row[i] = row[i] / nrm;
        }
        for (int64_t i = 0; i < D; ++i) out[p * D + i] = (float)row[i];
    }
    return 0;
}

/* ---- whole generations in native code ------------------------------------------------------------------------------
 * ask -> normalise into the pinned rows the kernel reads -> launch (through the caller's function pointer: this
 * library stays free of HIP) -> draw the next population's deviates and write the history rows while the GPU works ->
 * wait -> float64 reduction -> tell -> termination test, generation after generation, without returning to Python.
 * Around a 1.6 ms kernel the Python loop of MPC_ORD.optimize_cmaes costs ~45 us per generation in interpreter and
 * ctypes overhead; this one costs what the arithmetic and the launch cost.  Same functions, same order, same bits. */
static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void row_normalise_once(const double *x, int n, int variant, double *out)
{
    double dot = 0.0;
    if (variant == 1) for (int i = 0; i < n; ++i) dot = __builtin_fma(x[i], x[i], dot);
    else for (int i = 0; i < n; ++i) dot = dot + x[i] * x[i];
    const double nrm = sqrt(dot);
    for (int i = 0; i < n; ++i) out[i] = x[i] / nrm;
}

int32_t ocd_cma_run(ocd_cma *es, const ocd_cma_run_args *a, int64_t *generations_done, int32_t stop_flags[OCD_CMA_N_STOP],
                    int32_t *pending_nan)
{
    if (!es || !a || !generations_done || !stop_flags || !pending_nan) return -1;
    if (!a->rollout || !a->sync || !a->w_pinned || !a->ret_pinned || !a->X || !a->cost || a->N < 1 || a->S < 1 ||
        a->max_generations < 0 || a->normalise_variant < 0 || a->normalise_variant > 1) return -1;
    const int n = es->n, lam = es->lam;
    const int64_t E = (int64_t)lam * a->N * a->S;
    *generations_done = 0;
    *pending_nan = 0;
    for (int i = 0; i < OCD_CMA_N_STOP; ++i) stop_flags[i] = 0;
    for (int64_t g = 0; g < a->max_generations; ++g) {
        double *sg = a->seconds ? a->seconds + 8 * g : NULL;
        const double t0 = now_s();
        if (ocd_cma_ask(es, a->X) != 0) return -1;
        const double t1 = now_s();
        if (ocd_normalise_weights(a->X, lam, n, a->normalise_variant, a->w_pinned) != 0) return -1;
        const double t2 = now_s();
        const int32_t st = a->rollout(a->scn, a->init_dev, a->w_pinned, lam, a->N, 0, E, a->ret_pinned, NULL, NULL, a->stream);
        if (st != 0) return st < 0 ? st : -1;
        const double t3 = now_s();
        /* while the GPU works: the next population's deviates, this generation's history rows (mpc_ord.py:120,146) */
        ocd_cma_prepare(es);
        if (a->hist_w)
            for (int k = 0; k < lam; ++k)
                row_normalise_once(a->X + (size_t)k * n, n, a->normalise_variant, a->hist_w + ((size_t)g * lam + k) * n);
        const double t4 = now_s();
        if (a->sync(a->stream) != 0) return -1;
        const double t5 = now_s();
        if (ocd_fitness_from_returns(a->ret_pinned, lam, a->N, a->S, a->cost) != 0) return -1;
        int any_nan = 0;
        for (int k = 0; k < lam; ++k) any_nan |= isnan(a->cost[k]);
        if (a->hist_cost) memcpy(a->hist_cost + (size_t)g * lam, a->cost, sizeof(double) * (size_t)lam);
        const double t6 = now_s();
        if (sg) { sg[1] = t1 - t0; sg[2] = t2 - t1; sg[3] = t3 - t2; sg[4] = t4 - t3; sg[5] = t5 - t4; sg[6] = t6 - t5; sg[7] = 0.0; sg[0] = t6 - t0; }
        if (any_nan) {                       /* pycma redraws NaN candidates: the caller does that, then tells */
            *pending_nan = 1;
            *generations_done = g;           /* generation g is evaluated (X, cost, history row written) but not told */
            return 0;
        }
        const int32_t nonfinite = ocd_cma_tell(es, a->X, a->cost);
        if (nonfinite < 0) return -1;
        if (a->nonfinite) a->nonfinite[g] = nonfinite;
        const int32_t stops = ocd_cma_stop(es, a->stop_opts, stop_flags);
        const double t7 = now_s();
        if (sg) { sg[7] = t7 - t6; sg[0] = t7 - t0; }
        *generations_done = g + 1;
        if (stops > 0) return 0;
    }
    return 0;
}

/* ---- host threads for the per-run work of ocd_cma_run_many (ABI 8) ---------------------------------------------------
 * The tells of a lockstep generation are independent (each run has its own strategy state and random stream) and, at the
 * reference's shape, the largest item between one kernel's end and the next launch: 28 runs x 3.9 us of 7 x 7
 * eigendecomposition after a 1.15 ms kernel.  A pool of a->host_threads - 1 workers lives for the duration of one call and
 * takes runs off a shared counter together with the calling thread; every run is told by exactly one thread, with the
 * same function on the same state, so nothing about the results depends on the number of threads.
 * Workers spin (a few milliseconds: longer than an episode kernel of the latency builds) and then sleep on a condition
 * variable; a job is published by bumping `epoch` to odd (closed: no worker may enter), waiting for the workers still
 * inside the previous job's claim loop, writing the job, and bumping it to even (open). */
#define OCD_CMA_MAX_THREADS 16
#define OCD_CMA_POOL_SPIN_S 4e-3
typedef void (*cma_run_fn)(void *ctx, int r);
typedef struct {
    int n_workers;
    pthread_t th[OCD_CMA_MAX_THREADS];
    atomic_uint epoch;
    atomic_int busy, quit, sleepers;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    cma_run_fn fn;            /* the job: written while epoch is odd and busy == 0 */
    void *ctx;
    int r1;
    atomic_int next, pending; /* next run to claim; runs not finished yet */
} cma_pool;

static inline void cpu_relax(void)
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    sched_yield();
#endif
}

static void pool_claim(cma_pool *p)
{
    const cma_run_fn fn = p->fn;
    void *ctx = p->ctx;
    const int r1 = p->r1;
    for (;;) {
        const int r = atomic_fetch_add(&p->next, 1);
        if (r >= r1) break;
        fn(ctx, r);
        atomic_fetch_sub_explicit(&p->pending, 1, memory_order_release);
    }
}

static void *pool_worker(void *arg)
{
    cma_pool *p = (cma_pool *)arg;
    unsigned seen = 0;
    double idle_since = now_s();
    for (;;) {
        unsigned e;
        unsigned spins = 0;
        for (;;) {                                           /* wait for an open job this worker has not seen */
            if (atomic_load(&p->quit)) return NULL;
            e = atomic_load(&p->epoch);
            if (e != seen && !(e & 1u)) break;
            cpu_relax();
            if ((++spins & 1023u) == 0 && now_s() - idle_since > OCD_CMA_POOL_SPIN_S) {
                pthread_mutex_lock(&p->mu);
                atomic_fetch_add(&p->sleepers, 1);
                for (;;) {
                    e = atomic_load(&p->epoch);
                    if (atomic_load(&p->quit) || (e != seen && !(e & 1u))) break;
                    pthread_cond_wait(&p->cv, &p->mu);
                }
                atomic_fetch_sub(&p->sleepers, 1);
                pthread_mutex_unlock(&p->mu);
                idle_since = now_s();
            }
        }
        seen = e;
        atomic_fetch_add(&p->busy, 1);
        if (atomic_load(&p->epoch) == e) pool_claim(p);      /* (else: closed again under us -- the next job will do) */
        atomic_fetch_sub(&p->busy, 1);
        idle_since = now_s();
    }
}

static void pool_wake(cma_pool *p)
{
    if (atomic_load(&p->sleepers) > 0) {
        pthread_mutex_lock(&p->mu);
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
    }
}

/* NULL when threads <= 1 or no worker could be started: the caller then runs everything itself */
static cma_pool *pool_create(int threads)
{
    if (threads > OCD_CMA_MAX_THREADS) threads = OCD_CMA_MAX_THREADS;
    if (threads <= 1) return NULL;
    cma_pool *p = (cma_pool *)calloc(1, sizeof(cma_pool));
    if (!p) return NULL;
    atomic_init(&p->epoch, 0u); atomic_init(&p->busy, 0); atomic_init(&p->quit, 0); atomic_init(&p->sleepers, 0);
    atomic_init(&p->next, 0); atomic_init(&p->pending, 0);
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv, NULL);
    for (int i = 0; i < threads - 1; ++i) {
        if (pthread_create(&p->th[p->n_workers], NULL, pool_worker, p) != 0) break;
        p->n_workers += 1;
    }
    if (p->n_workers == 0) { pthread_cond_destroy(&p->cv); pthread_mutex_destroy(&p->mu); free(p); return NULL; }
    return p;
}

static void pool_destroy(cma_pool *p)
{
    if (!p) return;
    atomic_store(&p->quit, 1);
    pthread_mutex_lock(&p->mu);
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
    for (int i = 0; i < p->n_workers; ++i) pthread_join(p->th[i], NULL);
    pthread_cond_destroy(&p->cv);
    pthread_mutex_destroy(&p->mu);
    free(p);
}

/* fn(ctx, r) for r in [r0, r1), each r exactly once, on the calling thread and the pool's workers; returns when all are done */
static void pool_for(cma_pool *p, cma_run_fn fn, void *ctx, int r0, int r1)
{
    if (!p || r1 - r0 < 2) {
        for (int r = r0; r < r1; ++r) fn(ctx, r);
        return;
    }
    atomic_fetch_add(&p->epoch, 1u);                          /* odd: closed */
    while (atomic_load(&p->busy) != 0) cpu_relax();
    p->fn = fn; p->ctx = ctx; p->r1 = r1;
    atomic_store(&p->next, r0);
    atomic_store(&p->pending, r1 - r0);
    atomic_fetch_add(&p->epoch, 1u);                          /* even: open */
    pool_wake(p);
    pool_claim(p);
    while (atomic_load_explicit(&p->pending, memory_order_acquire) != 0) cpu_relax();
}

/* ---- R independent runs in lockstep, one launch per generation -------------------------------------------------------
 * The reference runs one optimisation per init group in a multiprocessing.Pool (run_mpc_ord.py:83-90): R processes, each
 * looping ask -> fitness -> tell on its own population and its own init states.  Here the R strategies advance together
 * and a generation's episodes -- sum over the active runs of popsize_r x N_r x S -- are ONE indexed launch
 * (ocd_rollout_indexed): a single run of the reference's shape fills 81 of the chip's 1 024 SIMDs, 28 of them most of it,
 * in the same wall time.  Every run goes through exactly the calls ocd_cma_run makes for it alone (ask, normalise,
 * prepare, tell, stop, in that order, on its own state and random stream), so its history is bit for bit the history of
 * the run alone; a run that stops drops out of the index, a run with a NaN cost is handed back untold (pending_nan[r])
 * for the caller's rejection sampling while the others are told. */
/* The runs are dealt to G = a->n_groups groups of neighbouring runs (1 when the field is 0), each with its own stream and
 * its own region of the index / returns buffers.  A group cycles  wait -> reduce -> tell -> stop -> ask -> normalise ->
 * launch -> (next deviates, history rows)  by itself, and the groups take turns: while one group's host work runs (28 tells
 * + asks of the reference's shape: ~0.1 ms per generation after a 1.15 ms kernel), the other groups' kernels are still
 * on the GPU -- launches of at most one wavefront per SIMD, side by side on different SIMDs (round 6).  With one group the
 * call sequence is exactly round 5's.  A run's own sequence of calls does not depend on the grouping. */
#define OCD_CMA_MAX_GROUPS 8
typedef struct {
    int r0, r1;               /* runs [r0, r1) */
    void *stream;
    int64_t e_cap0;           /* first row of the group's region in index_pinned / ret_pinned */
    int64_t E;                /* episodes of its current index */
    int dirty;                /* the index must be rebuilt (a run dropped out) */
    int in_flight;            /* a launch of generation `gen` is on the stream, not yet reduced / told */
    int64_t gen;              /* the generation of that launch */
} cma_group;

static void build_group_index(const ocd_cma_many_args *a, ocd_cma *const *es, cma_group *gr, int64_t *ret_off)
{
    int64_t e = gr->e_cap0;
    for (int r = gr->r0; r < gr->r1; ++r) {
        ret_off[r] = -1;
        if (!a->active[r]) continue;
        ret_off[r] = e;
        const int64_t lam = es[r]->lam, N = a->run_N[r], phase = a->run_reset_phase ? a->run_reset_phase[r] : 0;
        int64_t flat = 0;
        for (int64_t p = 0; p < lam; ++p)
            for (int64_t i = 0; i < N; ++i)
                for (int64_t sidx = 0; sidx < a->S; ++sidx, ++flat, ++e) {
                    int32_t *row = a->index_pinned + 3 * e;
                    row[0] = (int32_t)(a->run_p0[r] + p);
                    row[1] = (int32_t)(a->run_n0[r] + i);
                    row[2] = (int32_t)(phase + flat);
                }
    }
    gr->E = e - gr->e_cap0;
}

/* ask -> normalise -> (index) -> launch -> the host work that overlaps the kernel, for the active runs of one group and
 * generation g; t[0..3] += ask, normalise, launch, overlapped seconds */
/* per-run pieces of a generation that the pool's threads share out (each touches run r's strategy, its rows of the
 * history / flag arrays and nothing else) */
typedef struct {
    ocd_cma *const *es;
    const ocd_cma_many_args *a;
    int64_t g;
    int n;
    int32_t *status;          /* [R] run_tell: -1 where a call failed */
    uint8_t *stopped;         /* [R] run_tell: 1 where a termination rule ended run r */
} run_ctx;

static void run_overlapped(void *ctx, int r)                    /* the next deviates, this generation's history rows */
{
    const run_ctx *c = (const run_ctx *)ctx;
    const ocd_cma_many_args *a = c->a;
    if (!a->active[r]) return;
    const int n = c->n;
    if (a->evaluated) a->evaluated[(size_t)c->g * a->R + r] = 1;
    ocd_cma_prepare(c->es[r]);
    if (a->hist_w)
        for (int k = 0; k < c->es[r]->lam; ++k)
            row_normalise_once(a->X[r] + (size_t)k * n, n, a->normalise_variant,
                               a->hist_w + ((size_t)c->g * a->P_rows + a->run_p0[r] + k) * n);
}

static void run_tell(void *ctx, int r)                          /* tell -> stop */
{
    const run_ctx *c = (const run_ctx *)ctx;
    const ocd_cma_many_args *a = c->a;
    c->status[r] = 0;
    c->stopped[r] = 0;
    if (!a->active[r] || a->pending_nan[r]) return;
    const int32_t nonfinite = ocd_cma_tell(c->es[r], a->X[r], a->cost[r]);
    if (nonfinite < 0) { c->status[r] = -1; return; }
    if (a->nonfinite) a->nonfinite[(size_t)c->g * a->R + r] = nonfinite;
    if (ocd_cma_stop(c->es[r], a->stop_opts, a->stop_flags + (size_t)r * OCD_CMA_N_STOP) > 0) c->stopped[r] = 1;
}

static int32_t group_launch(ocd_cma *const *es, const ocd_cma_many_args *a, cma_group *gr, int64_t g, int n, int64_t *ret_off,
                            double *t, cma_pool *pool)
{
    int n_active = 0;
    for (int r = gr->r0; r < gr->r1; ++r) n_active += a->active[r] != 0;
    if (!n_active) return 0;
    const double t0 = now_s();
    for (int r = gr->r0; r < gr->r1; ++r)
        if (a->active[r] && ocd_cma_ask(es[r], a->X[r]) != 0) return -1;
    const double t1 = now_s();
    for (int r = gr->r0; r < gr->r1; ++r)
        if (a->active[r] && ocd_normalise_weights(a->X[r], es[r]->lam, n, a->normalise_variant,
                                                  a->w_pinned + (size_t)a->run_p0[r] * n) != 0) return -1;
    if (gr->dirty) { build_group_index(a, es, gr, ret_off); gr->dirty = 0; }
    const double t2 = now_s();
    const int32_t st = a->rollout(a->scn, a->init_dev, a->N_rows, a->w_pinned, a->P_rows, a->index_pinned + 3 * gr->e_cap0, gr->E,
                                  a->ret_pinned + gr->e_cap0, NULL, NULL, gr->stream);
    if (st != 0) return st < 0 ? st : -1;
    gr->in_flight = 1;
    gr->gen = g;
    if (a->episodes_launched) a->episodes_launched[g] += gr->E;
    const double t3 = now_s();
    run_ctx c = {es, a, g, n, NULL, NULL};
    pool_for(pool, run_overlapped, &c, gr->r0, gr->r1);         /* while the GPU works */
    const double t4 = now_s();
    t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2; t[3] += t4 - t3;
    return 0;
}

/* wait -> reduce -> tell -> stop for the group's launch in flight; t[0..2] += wait, reduce, tell seconds;
 * *any_pending |= some run of the group got a NaN cost (left evaluated, not told) */
static int32_t group_finish(ocd_cma *const *es, const ocd_cma_many_args *a, cma_group *gr, const int64_t *ret_off, double *t,
                            int *any_pending, cma_pool *pool)
{
    const int64_t g = gr->gen;
    const double t0 = now_s();
    if (a->sync(gr->stream) != 0) return -1;
    gr->in_flight = 0;
    const double t1 = now_s();
    for (int r = gr->r0; r < gr->r1; ++r) {
        if (!a->active[r]) continue;
        const int lam = es[r]->lam;
        if (ocd_fitness_from_returns(a->ret_pinned + ret_off[r], lam, a->run_N[r], a->S, a->cost[r]) != 0) return -1;
        if (a->hist_cost) memcpy(a->hist_cost + (size_t)g * a->P_rows + a->run_p0[r], a->cost[r], sizeof(double) * (size_t)lam);
        int any_nan = 0;
        for (int k = 0; k < lam; ++k) any_nan |= isnan(a->cost[r][k]);
        if (any_nan) { a->pending_nan[r] = 1; *any_pending = 1; }
    }
    const double t2 = now_s();
    int32_t status[OCD_CMA_MAX_RUNS];
    uint8_t stopped[OCD_CMA_MAX_RUNS];
    run_ctx c = {es, a, g, es[gr->r0]->n, status, stopped};
    pool_for(pool, run_tell, &c, gr->r0, gr->r1);
    for (int r = gr->r0; r < gr->r1; ++r) {
        if (status[r] != 0) return -1;
        if (stopped[r]) { a->active[r] = 0; gr->dirty = 1; }
    }
    const double t3 = now_s();
    t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2;
    return 0;
}

int32_t ocd_cma_run_many(ocd_cma *const *es, const ocd_cma_many_args *a, int64_t *generations_done)
{
    if (!es || !a || !generations_done) return -1;
    if (a->R < 1 || a->R > OCD_CMA_MAX_RUNS || !a->rollout || !a->sync || !a->w_pinned || !a->ret_pinned || !a->index_pinned ||
        !a->run_n0 || !a->run_N || !a->run_p0 || !a->active || !a->X || !a->cost || !a->stop_flags || !a->pending_nan ||
        !a->stop_opts || a->S < 1 || a->N_rows < 1 || a->P_rows < 1 || a->max_generations < 0 ||
        a->normalise_variant < 0 || a->normalise_variant > 1 || a->n_groups < 0 || a->n_groups > OCD_CMA_MAX_GROUPS ||
        (a->n_groups > 1 && !a->streams) || a->host_threads < 0) return -1;
    const int R = a->R;
    int n = 0;
    for (int r = 0; r < R; ++r) {
        if (!es[r] || !a->X[r] || !a->cost[r]) return -1;
        if (r == 0) n = es[r]->n;
        if (es[r]->n != n || a->run_N[r] < 1 || a->run_n0[r] < 0 || a->run_n0[r] + a->run_N[r] > a->N_rows ||
            a->run_p0[r] < 0 || a->run_p0[r] + es[r]->lam > a->P_rows) return -1;
        a->pending_nan[r] = 0;
        for (int i = 0; i < OCD_CMA_N_STOP; ++i) a->stop_flags[(size_t)r * OCD_CMA_N_STOP + i] = 0;
    }
    *generations_done = 0;
    /* the groups: neighbouring runs, each group's region of the index / returns buffers sized for all of its runs */
    int G = a->n_groups > 1 ? a->n_groups : 1;
    if (G > R) G = R;
    cma_group grp[OCD_CMA_MAX_GROUPS];
    int64_t ret_off[OCD_CMA_MAX_RUNS];
    {
        int64_t e = 0;
        for (int k = 0; k < G; ++k) {
            cma_group *gr = &grp[k];
            gr->r0 = (int)((int64_t)R * k / G);
            gr->r1 = (int)((int64_t)R * (k + 1) / G);
            gr->stream = G > 1 ? a->streams[k] : a->stream;
            gr->e_cap0 = e;
            gr->E = 0; gr->dirty = 1; gr->in_flight = 0; gr->gen = 0;
            for (int r = gr->r0; r < gr->r1; ++r) e += (int64_t)es[r]->lam * a->run_N[r] * a->S;
        }
    }
    if (a->max_generations == 0) return 0;
    {
        int n_active = 0;
        for (int r = 0; r < R; ++r) n_active += a->active[r] != 0;
        if (!n_active) return 0;
    }
    int32_t err = 0;
    int drain = 0;                             /* a run is pending (NaN cost): finish what is in flight, launch nothing more */
    cma_pool *pool = pool_create(a->host_threads);   /* NULL: this thread does everything (host_threads 0 / 1) */
    double t_prev = now_s();
    double tl[4] = {0, 0, 0, 0};               /* the launches made for the generation that follows */
    if (a->episodes_launched) a->episodes_launched[0] = 0;
    if (a->evaluated) memset(a->evaluated, 0, (size_t)R);
    for (int k = 0; k < G; ++k) {
        err = group_launch(es, a, &grp[k], 0, n, ret_off, tl, pool);
        if (err != 0) goto fail;
    }
    for (int64_t g = 0; g < a->max_generations; ++g) {
        double *sg = a->seconds ? a->seconds + 8 * g : NULL;
        double tf[3] = {0, 0, 0};
        double tn[4] = {0, 0, 0, 0};
        int any_in_flight = 0;
        if (g + 1 < a->max_generations) {                   /* (group_launch marks who takes part in g + 1) */
            if (a->episodes_launched) a->episodes_launched[g + 1] = 0;
            if (a->evaluated) memset(a->evaluated + (size_t)(g + 1) * R, 0, (size_t)R);
        }
        for (int k = 0; k < G; ++k) {
            cma_group *gr = &grp[k];
            if (!gr->in_flight || gr->gen != g) continue;   /* (a group without active runs has nothing in flight) */
            any_in_flight = 1;
            int pend = 0;
            err = group_finish(es, a, gr, ret_off, tf, &pend, pool);
            if (err != 0) goto fail;
            if (pend) drain = 1;
            /* this group's next generation goes out before the next group is waited for: its kernel runs while the others'
             * results are reduced and told */
            if (!drain && g + 1 < a->max_generations) {
                err = group_launch(es, a, gr, g + 1, n, ret_off, tn, pool);
                if (err != 0) goto fail;
            }
        }
        if (!any_in_flight) break;                          /* no run was active in generation g */
        const double t_now = now_s();
        if (sg) { sg[1] = tl[0]; sg[2] = tl[1]; sg[3] = tl[2]; sg[4] = tl[3]; sg[5] = tf[0]; sg[6] = tf[1]; sg[7] = tf[2]; sg[0] = t_now - t_prev; }
        t_prev = t_now;
        for (int i = 0; i < 4; ++i) tl[i] = tn[i];
        *generations_done = g + 1;
        int still = 0;
        for (int k = 0; k < G; ++k) still |= grp[k].in_flight;
        if (!still) break;                                  /* nothing launched for g + 1: pending, the cap, or no active run */
    }
    pool_destroy(pool);
    return 0;
fail:                                          /* nothing stays in flight behind an error: the caller frees the buffers */
    for (int k = 0; k < G; ++k)
        if (grp[k].in_flight) { (void)a->sync(grp[k].stream); grp[k].in_flight = 0; }
    pool_destroy(pool);
    return err;
}

/* K fitness evaluations of one fixed population, back to back: launch, wait, float64 reduction -- the part of a
 * generation that bench.py times as its "step", through the same function pointers and without the interpreter between
 * steps.  The rows of a->w_pinned are evaluated as they are (already normalised fp32); cost_out [P] holds the last
 * step's costs; seconds_out (or NULL) the wall time of the K steps as this loop saw it. */
int32_t ocd_eval_generations(const ocd_cma_run_args *a, int64_t P, int64_t K, double *cost_out, double *seconds_out)
{
    if (!a || !a->rollout || !a->sync || !a->w_pinned || !a->ret_pinned || !cost_out || P < 1 || K < 0 || a->N < 1 || a->S < 1)
        return -1;
    const int64_t E = P * a->N * a->S;
    const double t0 = now_s();
    for (int64_t k = 0; k < K; ++k) {
        const int32_t st = a->rollout(a->scn, a->init_dev, a->w_pinned, P, a->N, 0, E, a->ret_pinned, NULL, NULL, a->stream);
        if (st != 0) return st < 0 ? st : -1;
        if (a->sync(a->stream) != 0) return -1;
        if (ocd_fitness_from_returns(a->ret_pinned, P, a->N, a->S, cost_out) != 0) return -1;
    }
    if (seconds_out) *seconds_out = now_s() - t0;
    return 0;
}

