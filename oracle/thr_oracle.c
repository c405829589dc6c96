/* CPU oracle, C half  --  TEST INFRASTRUCTURE ONLY (see oracle/thr_oracle.py).
 *
 * Same contracts as the numpy restatement, in plain C so that full-size
 * corpora (1M x 768) can be checked exactly in seconds and so that bench.py's
 * cpu_baseline leg has a threaded scalar port to time.  Built by
 * oracle/Makefile into oracle/_build/libthr_oracle.so; only tests/, smoke()
 * and bench.py's cpu_baseline leg load it.
 *
 * No -ffast-math, no FMA contraction: every operation below is one IEEE
 * float64 rounding, in the written order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* sequential left-to-right float64 accumulation of float32 products
 * (contract: oracle/thr_oracle.py seq_dot_f64) */
static double seq_dot(const float *a, const float *b, int d)
{
    double s = 0.0;
    for (int i = 0; i < d; ++i)
        s += (double)a[i] * (double)b[i];
    return s;
}

void oracle_doc_norms_f64(const float *docs, int64_t n, int d, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        out[r] = sqrt(seq_dot(docs + r * (int64_t)d, docs + r * (int64_t)d, d));
}

/* cosine_scores_f64: sim = dot / (qn * dn); dn == 0 -> -inf; qn == 0 -> 0 */
void oracle_cosine_scores_f64(const float *docs, int64_t n, int d, const float *q,
                              const double *dnorm, double *out)
{
    const double qn = sqrt(seq_dot(q, q, d));
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        if (!(dnorm[r] > 0.0)) { out[r] = -INFINITY; continue; }
        if (!(qn > 0.0)) { out[r] = 0.0; continue; }
        out[r] = seq_dot(docs + r * (int64_t)d, q, d) / (qn * dnorm[r]);
    }
}

/* total order (score desc, id asc) */
static int better(double sa, int64_t ia, double sb, int64_t ib)
{
    return sa > sb || (sa == sb && ia < ib);
}

/* exact top-k of one score vector by insertion into a k-sized sorted list */
void oracle_topk_f64(const double *scores, int64_t n, int k, int64_t id_base,
                     double *out_s, int64_t *out_i, int *out_n)
{
    int m = 0;
    for (int64_t r = 0; r < n; ++r) {
        double s = scores[r];
        if (!isfinite(s)) continue;
        int64_t id = r + id_base;
        if (m == k && !better(s, id, out_s[m - 1], out_i[m - 1])) continue;
        int p = (m < k) ? m++ : m - 1;
        while (p > 0 && better(s, id, out_s[p - 1], out_i[p - 1])) {
            out_s[p] = out_s[p - 1]; out_i[p] = out_i[p - 1]; --p;
        }
        out_s[p] = s; out_i[p] = id;
    }
    *out_n = m;
}

/* dense channel, exact, Q queries (threads across docs inside each query) */
void oracle_dense_topk_exact(const float *docs, int64_t n, int d, const double *dnorm,
                             const float *queries, int nq, int k, int64_t id_base,
                             double *out_s, int64_t *out_i, int *out_n)
{
    double *sc = (double *)malloc(sizeof(double) * (size_t)n);
    for (int q = 0; q < nq; ++q) {
        oracle_cosine_scores_f64(docs, n, d, queries + (int64_t)q * d, dnorm, sc);
        oracle_topk_f64(sc, n, k, id_base, out_s + (int64_t)q * k, out_i + (int64_t)q * k, out_n + q);
    }
    free(sc);
}

/* BM25, one query -> dense score vector (contract: thr_oracle.py bm25_scores) */
void oracle_bm25_scores(const int64_t *rowptr, const int32_t *post_doc, const int32_t *post_tf,
                        const float *doclen, const double *idf, double avgdl,
                        const int32_t *terms, int nterms, int64_t n_docs,
                        double k1, double b, double *score)
{
    unsigned char *hit = (unsigned char *)calloc((size_t)n_docs, 1);
    for (int64_t i = 0; i < n_docs; ++i) score[i] = 0.0;
    for (int t = 0; t < nterms; ++t) {
        int32_t term = terms[t];
        if (term < 0) continue;
        for (int64_t p = rowptr[term]; p < rowptr[term + 1]; ++p) {
            int32_t dd = post_doc[p];
            double tf = (double)post_tf[p];
            double dl = (double)doclen[dd];
            double nrm = k1 * ((1.0 - b) + b * (dl / avgdl));
            double contrib = idf[term] * ((tf * (k1 + 1.0)) / (tf + nrm));
            score[dd] = score[dd] + contrib;
            hit[dd] = 1;
        }
    }
    for (int64_t i = 0; i < n_docs; ++i)
        if (!hit[i]) score[i] = -INFINITY;
    free(hit);
}

/* MaxSim in float64 from float16 bit patterns (contract: maxsim_scores) */
static double h2d(uint16_t h)
{
    uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    double v;
    if (e == 0) v = ldexp((double)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexp((double)(m | 1024), (int)e - 25);
    return s ? -v : v;
}

void oracle_maxsim(const uint16_t *qtok, int nq, int qt, int dim, const uint16_t *dtok, int dt,
                   const int32_t *cand, int ncand, double *out)
{
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int q = 0; q < nq; ++q)
        for (int c = 0; c < ncand; ++c) {
            int32_t doc = cand[(int64_t)q * ncand + c];
            if (doc < 0) { out[(int64_t)q * ncand + c] = -INFINITY; continue; }
            const uint16_t *Q = qtok + (int64_t)q * qt * dim;
            const uint16_t *D = dtok + (int64_t)doc * dt * dim;
            double total = 0.0;
            for (int i = 0; i < qt; ++i) {
                double best = -INFINITY;
                for (int j = 0; j < dt; ++j) {
                    double s = 0.0;
                    for (int x = 0; x < dim; ++x)
                        s += h2d(Q[i * dim + x]) * h2d(D[j * dim + x]);
                    if (s > best) best = s;
                }
                total += best;
            }
            out[(int64_t)q * ncand + c] = total;
        }
}
