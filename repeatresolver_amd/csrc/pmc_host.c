/* pmc_host.c -- host side of the MaxCorrelation drop-in, plain C: file in, file out (MC:270-336, MC:516-532, MC:916-1020). */
#define _POSIX_C_SOURCE 200809L
#include "pmc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int pmc_read_msa(const char *path, int *rows, int *width, unsigned char **text)
{
    FILE *f = fopen(path, "r");
    if (!f) return PWR_ERR_INPUT;                                                  /* MC:283 "MA is missing." */
    size_t cap = (size_t)1 << 22, used = 0;
    unsigned char *t = malloc(cap);
    char *buf = malloc(PMC_MAX_COLUMNS + 3);
    int W = -1, T = 0, rc = PWR_OK;
    if (!t || !buf) { fclose(f); free(t); free(buf); return PWR_ERR_NOMEM; }
    while (fgets(buf, PMC_MAX_COLUMNS + 3 - 2, f)) {                               /* MC:286 */
        const int len = (int)strlen(buf) - 1;
        if (W < 0) W = len;                                                        /* MC:290 */
        if (len != W) continue;                                                    /* MC:299 */
        if (T >= PMC_MAX_ROWS) { rc = PWR_ERR_RANGE; break; }                      /* MC:278 Signatures[Max_Sig_Anzahl] */
        if (used + (size_t)W > cap) {
            while (used + (size_t)W > cap) cap *= 2;
            unsigned char *nt = realloc(t, cap);
            if (!nt) { rc = PWR_ERR_NOMEM; break; }
            t = nt;
        }
        memcpy(t + used, buf, (size_t)W); used += (size_t)W; T++;
    }
    fclose(f);
    free(buf);
    if (rc == PWR_OK && W <= 0) rc = PWR_ERR_INPUT;
    if (rc) { free(t); return rc; }
    *rows = T; *width = W; *text = t;
    return PWR_OK;
}

int pmc_write(const char *path, int nvars, const double *maxcorrs)
{
    FILE *f = fopen(path, "w");
    if (!f) return PWR_ERR_IO;                                                     /* MC:521 */
    for (int i = 0; i < nvars; i++) fprintf(f, "%f\n", maxcorrs[i]);               /* MC:526-529 */
    return fclose(f) == 0 ? PWR_OK : PWR_ERR_IO;
}

int pmc_run_file(const char *msa_path, int mincov, int device, FILE *log)
{
    int T = 0, W = 0;
    unsigned char *text = NULL;
    const time_t start = time(NULL);
    int rc = pmc_read_msa(msa_path, &T, &W, &text);
    if (rc == PWR_ERR_INPUT) { fprintf(log, "MA is missing.\n"); return 1; }       /* MC:283 */
    if (rc) { fprintf(log, "MaxCorrelation: %s\n", pwr_strerror(rc)); return 1; }
    fprintf(log, "There are %d sequences.\n", T);                                  /* MC:333-334 */
    fprintf(log, "Siglength is %d.\n", W);
    fprintf(log, "From %d to %d\n", 0, W);                                         /* MC:981-986 */
    char name[4096];
    snprintf(name, sizeof name, "MaxCorrsOf_%s", msa_path);                        /* MC:991-993 */
    fprintf(log, "%s\n", name);
    fprintf(log, "Cutoff %f\n", -1.0 * log10(1.0 / ((double)W * 5.0)));            /* MC:998-1000 */
    fprintf(log, "AllMaxCorrs\n");
    double *mc = malloc(sizeof(double) * (size_t)W * 5);
    rc = mc ? pmc_maxcorrs(T, W, text, mincov, device, mc) : PWR_ERR_NOMEM;
    if (rc == PWR_OK) rc = pmc_write(name, W * 5, mc);
    free(mc); free(text);
    if (rc) { fprintf(log, "MaxCorrelation: %s\n", pwr_strerror(rc)); return 1; }
    fprintf(log, "Runtime: %lu sec.\n", (unsigned long)(time(NULL) - start));      /* MC:1017 */
    return 0;
}
