/* CPU restatement of the reference's PSMC forward recursion and its gradient, float64.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py: parity pinning; who may link this).
 * It is the checker for the HIP kernels at sizes numpy loops cannot reach, and the
 * "cpu_baseline" (kind "port") that bench.py times beside the GPU.  It is NOT product code:
 * phlash_amd never loads it.
 *
 * Follows (paths relative to jthlab/phlash v1.0.6):
 *   src/phlash/hmm.py:52-65   matvec_smc  -- O(K) product with the SMC' matrix
 *   src/phlash/hmm.py:68-82   psmc_ll     -- transition, then emission (ob=-1 -> 1), c = sum,
 *                                            alpha /= c, ll += log(c)
 *   src/phlash/gpu.py:504-527 CUDA twins of the two (same arithmetic, same order)
 *   src/phlash/gpu.py:108-110 observations clipped to [-1, 1]
 *   src/phlash/model.py:52-57 warm-up prefix: alpha_hat after the first W sites replaces pi and
 *                             only the remaining sites are scored
 * The gradient is reverse mode (scaled forward-backward), not the reference's forward-mode
 * tangent propagation (gpu.py:575-692); tests/test_oracle_pins.py checks it against autograd of
 * the plain recursion and against central finite differences, which is what the reference's own
 * tests do with its kernel (tests/test_gpu.py:27-31, 58-64).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ROW_B 0
#define ROW_D 1
#define ROW_U 2
#define ROW_V 3
#define ROW_E0 4
#define ROW_E1 5
#define ROW_PI 6

/* out = x A, A[i][j] = b[j] (i>j), d[j] (i==j), u[i] v[j] (i<j).  hmm.py:52-65 */
static void matvec(const double *p, int K, const double *x, double *out) {
    const double *b = p + ROW_B * K, *d = p + ROW_D * K, *u = p + ROW_U * K, *v = p + ROW_V * K;
    double s = 0.0;
    for (int j = 0; j < K; ++j) {
        out[j] = x[j] * d[j] + s * v[j];
        s += u[j] * x[j];
    }
    s = 0.0;
    for (int j = K - 1; j >= 0; --j) {
        out[j] += s * b[j];
        s += x[j];
    }
}

/* one forward site: alpha <- normalise((alpha A) .* e_ob); returns c.  hmm.py:74-79 */
static double fwd_site(const double *p, int K, int ob, double *alpha, double *tmp) {
    matvec(p, K, alpha, tmp);
    double c = 0.0;
    if (ob >= 0) {
        const double *e = p + (ob >= 1 ? ROW_E1 : ROW_E0) * K;
        for (int j = 0; j < K; ++j) {
            tmp[j] *= e[j];
            c += tmp[j];
        }
    } else {
        for (int j = 0; j < K; ++j) c += tmp[j];
    }
    for (int j = 0; j < K; ++j) alpha[j] = tmp[j] / c;
    return c;
}

/* ll of sites W..L-1 given the whole row of L sites; alpha_out (optional) = alpha_hat_L. */
int oracle_psmc_ll(const double *params, int K, const int8_t *data, int64_t L, int64_t W,
                   double *ll_out, double *alpha_out) {
    double *alpha = (double *)malloc(sizeof(double) * 2 * K);
    if (!alpha) return 1;
    double *tmp = alpha + K;
    memcpy(alpha, params + ROW_PI * K, sizeof(double) * K);
    double ll = 0.0;
    for (int64_t t = 0; t < L; ++t) {
        double c = fwd_site(params, K, data[t], alpha, tmp);
        if (t >= W) ll += log(c);
    }
    *ll_out = ll;
    if (alpha_out) memcpy(alpha_out, alpha, sizeof(double) * K);
    free(alpha);
    return 0;
}

/* ll and d ll / d theta, rows b,d,u,v,emis0,emis1,pi (plain derivative).  Stores every
 * alpha_hat (L*K doubles) -- a CPU oracle can afford it. */
int oracle_psmc_ll_grad(const double *params, int K, const int8_t *data, int64_t L, int64_t W,
                        double *ll_out, double *grad) {
    const double *b = params + ROW_B * K, *d = params + ROW_D * K, *u = params + ROW_U * K,
                 *v = params + ROW_V * K;
    double *alphas = (double *)malloc(sizeof(double) * ((size_t)(L + 1) * K + (size_t)(L + 1) + 6 * K));
    if (!alphas) return 1;
    double *cs = alphas + (size_t)(L + 1) * K;
    double *tmp = cs + (L + 1);
    double *beta = tmp + K, *w = beta + K, *sufa = w + K, *preua = sufa + K, *nb = preua + K;
    memcpy(alphas, params + ROW_PI * K, sizeof(double) * K);
    double ll = 0.0;
    for (int64_t t = 1; t <= L; ++t) {
        double *a = alphas + (size_t)t * K;
        memcpy(a, a - K, sizeof(double) * K);
        cs[t] = fwd_site(params, K, data[t - 1], a, tmp);
        if (t > W) ll += log(cs[t]);
    }
    memset(grad, 0, sizeof(double) * 7 * K);
    for (int j = 0; j < K; ++j) beta[j] = 1.0;
    for (int64_t t = L; t >= 1; --t) {
        if (t == W)
            for (int j = 0; j < K; ++j) beta[j] -= 1.0;
        const double *ap = alphas + (size_t)(t - 1) * K;
        const int ob = data[t - 1];
        const double *e = ob < 0 ? NULL : params + (ob >= 1 ? ROW_E1 : ROW_E0) * K;
        const double rc = 1.0 / cs[t];
        for (int j = 0; j < K; ++j) w[j] = (e ? e[j] : 1.0) * beta[j] * rc;
        double s = 0.0;
        for (int j = K - 1; j >= 0; --j) { sufa[j] = s; s += ap[j]; }
        s = 0.0;
        for (int j = 0; j < K; ++j) { preua[j] = s; s += u[j] * ap[j]; }
        for (int j = 0; j < K; ++j) {
            grad[ROW_B * K + j] += w[j] * sufa[j];
            grad[ROW_D * K + j] += w[j] * ap[j];
            grad[ROW_V * K + j] += w[j] * preua[j];
        }
        if (e) {
            double *ge = grad + (ob >= 1 ? ROW_E1 : ROW_E0) * K;
            for (int j = 0; j < K; ++j)
                ge[j] += (d[j] * ap[j] + v[j] * preua[j] + b[j] * sufa[j]) * beta[j] * rc;
        }
        s = 0.0; /* suffix of v.*w -> du and the upper part of the new beta */
        for (int j = K - 1; j >= 0; --j) {
            nb[j] = d[j] * w[j] + u[j] * s;
            grad[ROW_U * K + j] += ap[j] * s;
            s += v[j] * w[j];
        }
        s = 0.0; /* prefix of b.*w -> lower part of the new beta */
        for (int j = 0; j < K; ++j) {
            nb[j] += s;
            s += b[j] * w[j];
        }
        memcpy(beta, nb, sizeof(double) * K);
    }
    memcpy(grad + ROW_PI * K, beta, sizeof(double) * K);
    *ll_out = ll;
    free(alphas);
    return 0;
}

/* Batched over (particle b, chunk s), OpenMP over sequences.
 * params: element strides pstride_b / pstride_s (pstride_s = 0 broadcasts one block over s).
 * data [N, Ltot] int8 row-major, inds[S] row indices, ll [B,S], grad [B,S,7,K] or NULL. */
int oracle_batch(const double *params, int64_t pstride_b, int64_t pstride_s, int K,
                 const int8_t *data, int64_t N, int64_t Ltot, const int64_t *inds, int64_t B,
                 int64_t S, int64_t W, double *ll, double *grad, int nthreads) {
    int err = 0;
    (void)N;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
#endif
    for (int64_t q = 0; q < B * S; ++q) {
        const int64_t bb = q / S, ss = q % S;
        const double *p = params + bb * pstride_b + ss * pstride_s;
        const int8_t *row = data + inds[ss] * Ltot;
        if (grad)
            err |= oracle_psmc_ll_grad(p, K, row, Ltot, W, ll + q, grad + (size_t)q * 7 * K);
        else
            err |= oracle_psmc_ll(p, K, row, Ltot, W, ll + q, NULL);
    }
    return err;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
