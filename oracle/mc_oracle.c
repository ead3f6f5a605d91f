/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of /root/reference/MaxCorrelation.c ("MC"): the step after PW_ReAligner -- for every
 * variation (column, symbol) of the realigned MSA the largest significance of its co-occurrence with a variation at least
 * 20 columns away (MC:745-837), written one "%f" per line (MC:516-532).
 *
 * PARITY UNPINNED: the reference needs GSL (gsl_cdf_hypergeometric_Q, MC:415), which this image does not have, so it cannot
 * be compiled here and no fixture made with it exists.  hyper_Q below restates the algorithm of GSL 2.x
 * cdf/hypergeometric.c + randist/hyperg.c (sum of pdf terms by ratio recurrences away from k, pdf = exp of three
 * lnchoose, each lgamma-based) from its published description, not from its source; results should agree with GSL to
 * about 1e-12 relative, the output has six decimals.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double lnchoose(unsigned n, unsigned m)
{
    if (m == n || m == 0) return 0.0;
    if (2 * m > n) m = n - m;
    return lgamma(n + 1.0) - lgamma(m + 1.0) - lgamma(n - m + 1.0);
}

/* P(X = k), X = successes among t draws without replacement from n1 successes and n2 failures */
static double hyper_pdf(unsigned k, unsigned n1, unsigned n2, unsigned t)
{
    if (t > n1 + n2) t = n1 + n2;
    if (k > n1 || k > t) return 0.0;
    if (t > n2 && k + n2 < t) return 0.0;
    return exp(lnchoose(n1, k) + lnchoose(n2, t - k) - lnchoose(n1 + n2, t));
}

static double hyper_lower(unsigned k, unsigned n1, unsigned n2, unsigned t)     /* P(X <= k) */
{
    unsigned i = k;
    double s = hyper_pdf(i, n1, n2, t), P = s;
    while (i > 0) {
        const double factor = (i / (n1 - i + 1.0)) * ((n2 + i - t) / (t - i + 1.0));
        s *= factor; P += s;
        if (s / P < 2.2204460492503131e-16) break;
        i--;
    }
    return P;
}

static double hyper_upper(unsigned k, unsigned n1, unsigned n2, unsigned t)     /* P(X > k) */
{
    unsigned i = k + 1;
    double s = hyper_pdf(i, n1, n2, t), Q = s;
    while (i < t) {
        const double factor = ((n1 - i) / (i + 1.0)) * ((t - i) / (n2 + i + 1.0 - t));
        s *= factor; Q += s;
        if (s / Q < 2.2204460492503131e-16) break;
        i++;
    }
    return Q;
}

double mco_hyper_Q(unsigned k, unsigned n1, unsigned n2, unsigned t)            /* gsl_cdf_hypergeometric_Q */
{
    if (k >= n1 || k >= t) return 0.0;
    const double midpoint = ((double)t * n1) / ((double)n1 + n2);
    if (k < midpoint) return 1.0 - hyper_lower(k, n1, n2, t);
    return hyper_upper(k, n1, n2, t);
}

/* MC:413-434 PositiveSignificance from the four counts and the two group sizes */
double mco_significance(int schnitt, int cov, int gr1, int gr2, int size1, int size2)
{
    if (gr1 == 0 || gr2 == 0) return 0.0;
    if (schnitt < 1) return 0.0;
    double Z = -1.0 * log10(mco_hyper_Q((unsigned)(schnitt - 1), (unsigned)gr2, (unsigned)(cov - gr2), (unsigned)gr1));   /* MC:415-416 */
    if (isinf(Z) || Z > 99) Z = 99.0;                                                                                      /* MC:417 */
    if (isinf(Z) || Z > 98.0) {                                                                                            /* MC:432 */
        double F = 2.0 * schnitt;                                                                                          /* F_beta(.,.,1), MC:396-410 */
        F = F < 0.0001 ? 0.0 : F / (2.0 * schnitt + (size1 - schnitt) + (size2 - schnitt));
        Z = 98.0 + F;
    }
    return Z;
}

static int popc(uint64_t x) { return __builtin_popcountll(x); }

/* MC:270-393 Einlesen + MC:745-837 / 839-905 for text = T rows of W characters; out[W * 5] */
int mco_maxcorrs(int T, int W, const unsigned char *text, int mincov, double *out)
{
    const int sc = T / 64 + 1;                                                   /* MC:338 */
    uint64_t *G = calloc((size_t)W * 5 * sc, 8), *LC = calloc((size_t)W * sc, 8);
    int *gsize = calloc((size_t)W * 5, sizeof(int)), *cover = calloc(W, sizeof(int));
    if (!G || !LC || !gsize || !cover) return -1;
    for (int j = 0; j < T; j++)
        for (int i = 0; i < W; i++) {
            int c;
            switch (text[(size_t)j * W + i]) {                                   /* MC:303-330 */
            case 'a': case 'A': c = 0; break;
            case 'c': case 'C': c = 1; break;
            case 'g': case 'G': c = 2; break;
            case 't': case 'T': c = 3; break;
            case '-': case '_': c = 4; break;
            default: c = 5;
            }
            if (c < 5) {
                G[((size_t)i * 5 + c) * sc + j / 64] |= 1ull << (j % 64);
                LC[(size_t)i * sc + j / 64] |= 1ull << (j % 64);
                gsize[i * 5 + c]++; cover[i]++;
            }
        }
    for (int i = 0; i < W * 5; i++) out[i] = 0.0;
    const int maxgroup = T;                                                      /* MC:1007 */
    for (int ii = 0; ii < W; ii++) {
        const int baseno = gsize[ii * 5] + gsize[ii * 5 + 1] + gsize[ii * 5 + 2] + gsize[ii * 5 + 3];
        for (int k = 0; k < 5; k++) {
            const int i = ii * 5 + k;
            if (!(gsize[i] > mincov / 4 && gsize[i] < maxgroup && baseno > cover[ii] / 2)) continue;   /* MC:796 */
            const uint64_t *Gi = G + (size_t)i * sc, *Li = LC + (size_t)ii * sc;
            for (int jj = ii + 20; jj < W; jj++) {                                /* MC:798 */
                const uint64_t *Lj = LC + (size_t)jj * sc;
                int cov = 0;
                for (int w = 0; w < sc; w++) cov += popc(Li[w] & Lj[w]);
                if (cov < mincov) break;                                          /* MC:801-804 */
                for (int kk = 0; kk < 5; kk++) {
                    const int j = jj * 5 + kk;
                    if (!(gsize[j] > mincov / 4 && gsize[j] < maxgroup)) continue; /* MC:811 */
                    const uint64_t *Gj = G + (size_t)j * sc;
                    int s = 0, g1 = 0, g2 = 0;
                    for (int w = 0; w < sc; w++) { s += popc(Gi[w] & Gj[w]); g1 += popc(Gi[w] & Lj[w]); g2 += popc(Gj[w] & Li[w]); }
                    const double Z = mco_significance(s, cov, g1, g2, gsize[i], gsize[j]);
                    if (Z > out[i]) out[i] = Z;                                   /* MC:816-817 */
                    if (Z > out[j]) out[j] = Z;
                }
            }
        }
    }
    free(G); free(LC); free(gsize); free(cover);
    return 0;
}

/* lines as the reference takes them (MC:286-336): the first line sets the width, lines of another width are skipped */
int mco_run(const char *msa_path, const char *out_path, int mincov)
{
    FILE *f = fopen(msa_path, "r");
    if (!f) { printf("MA is missing.\n"); return 1; }
    size_t cap = 1 << 20, used = 0;
    unsigned char *text = malloc(cap);
    char *buf = malloc(150000);
    int W = -1, T = 0;
    while (fgets(buf, 150000 - 2, f)) {
        const int len = (int)strlen(buf) - 1;
        if (W < 0) W = len;
        if (len != W) continue;
        if (used + W > cap) { while (used + W > cap) cap *= 2; text = realloc(text, cap); }
        memcpy(text + used, buf, W); used += W; T++;
    }
    fclose(f);
    if (W <= 0) return 1;
    double *mc = malloc(sizeof(double) * (size_t)W * 5);
    if (mco_maxcorrs(T, W, text, mincov, mc)) return 1;
    FILE *o = fopen(out_path, "w");
    if (!o) return 1;
    for (int i = 0; i < W * 5; i++) fprintf(o, "%f\n", mc[i]);                   /* MC:526-529 */
    fclose(o);
    free(text); free(buf); free(mc);
    return 0;
}

#ifdef MCO_MAIN
int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: mc_oracle MSA out [mincov]\n"); return 2; }
    return mco_run(argv[1], argv[2], argc > 3 ? atoi(argv[3]) : 30);
}
#endif
