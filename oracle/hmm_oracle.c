/* hmm_oracle.c — plain-C twin of oracle/textbook.py and oracle/viterbi.py for the large sizes.
 *
 * TEST INFRASTRUCTURE: built into oracle/_build/liboracle.so by oracle/build.py and loaded only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Never linked into the product.
 *
 *  oracle_posterior   float64 scaled forward-backward with the reference cell's eps clamps
 *                     (hmm_layer/MsaHmmCell.py:73-106; forward :102-103, reverse :96-100):
 *                     gamma (b,L,q), loglik (b)
 *  oracle_viterbi     Q16 fixed-point max-plus Viterbi, lowest-index tie-break (oracle/viterbi.py;
 *                     the reference has no Viterbi: parity unpinned)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_posterior(const float *A, const float *pi, const float *E, int b, int L, int q, double eps,
                      double *gamma, double *loglik) {
    double *ah = (double *)malloc((size_t)L * q * sizeof(double));
    double *x = (double *)malloc((size_t)q * sizeof(double));
    double *y = (double *)malloc((size_t)q * sizeof(double));
    for (int n = 0; n < b; ++n) {
        const float *En = E + (size_t)n * L * q;
        double ll = 0.0;
        for (int t = 0; t < L; ++t) {
            double S = 0.0;
            for (int j = 0; j < q; ++j) {
                double R;
                if (t == 0) R = pi[j];
                else { R = 0.0; for (int i = 0; i < q; ++i) R += x[i] * (double)A[i * q + j]; }
                double e = En[(size_t)t * q + j];
                R = R > eps ? R : eps;
                e = e > eps ? e : eps;
                y[j] = e * R;
                S += y[j];
            }
            ll += log(S);
            for (int j = 0; j < q; ++j) { x[j] = y[j] / S; ah[(size_t)t * q + j] = x[j]; }
        }
        loglik[n] = ll;
        for (int j = 0; j < q; ++j) x[j] = 1.0;
        for (int t = L - 1; t >= 0; --t) {
            double S = 0.0, G = 0.0;
            double *g = gamma + ((size_t)n * L + t) * q;
            for (int i = 0; i < q; ++i) {
                double R;
                if (t == L - 1) R = 1.0;
                else { R = 0.0; for (int j = 0; j < q; ++j) R += (double)A[i * q + j] * x[j]; }
                R = R > eps ? R : eps;
                y[i] = R;
                g[i] = ah[(size_t)t * q + i] * R;
                G += g[i];
            }
            for (int i = 0; i < q; ++i) g[i] /= G;
            for (int i = 0; i < q; ++i) {
                double e = En[(size_t)t * q + i];
                e = e > eps ? e : eps;
                y[i] *= e;
                S += y[i];
            }
            for (int i = 0; i < q; ++i) x[i] = y[i] / S;
        }
    }
    free(ah); free(x); free(y);
}

static int64_t quant(float v) {
    if (!(v > -1024.0f)) v = -1024.0f;      /* also catches -inf and NaN */
    if (v > 1024.0f) v = 1024.0f;
    return (int64_t)rintf(v * 65536.0f);
}

void oracle_viterbi(const float *logA, const float *logpi, const float *logE, int b, int L, int q,
                    int32_t *path, double *score) {
    int64_t *a = (int64_t *)malloc((size_t)q * q * sizeof(int64_t));
    int64_t *d = (int64_t *)malloc((size_t)q * sizeof(int64_t));
    int64_t *dn = (int64_t *)malloc((size_t)q * sizeof(int64_t));
    int8_t *bp = (int8_t *)malloc((size_t)L * q);
    for (int i = 0; i < q * q; ++i) a[i] = quant(logA[i]);
    for (int n = 0; n < b; ++n) {
        const float *En = logE + (size_t)n * L * q;
        for (int j = 0; j < q; ++j) d[j] = quant(logpi[j]) + quant(En[j]);
        for (int t = 1; t < L; ++t) {
            for (int j = 0; j < q; ++j) {
                int64_t best = d[0] + a[j];
                int arg = 0;
                for (int i = 1; i < q; ++i) {
                    int64_t c = d[i] + a[i * q + j];
                    if (c > best) { best = c; arg = i; }
                }
                dn[j] = best + quant(En[(size_t)t * q + j]);
                bp[(size_t)t * q + j] = (int8_t)arg;
            }
            memcpy(d, dn, (size_t)q * sizeof(int64_t));
        }
        int s = 0;
        for (int j = 1; j < q; ++j) if (d[j] > d[s]) s = j;
        score[n] = (double)d[s] / 65536.0;
        for (int t = L - 1; t >= 0; --t) {
            path[(size_t)n * L + t] = s;
            if (t > 0) s = bp[(size_t)t * q + s];
        }
    }
    free(a); free(d); free(dn); free(bp);
}
