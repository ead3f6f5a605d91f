/* pwr_host.c -- host side of the PW_ReAligner drop-in, plain C (the reference's host code is C).
 *
 * File format, stdout lines, exit codes and the round loop follow PW_ReAligner.c ("PW:") main(),
 * PW:1610-1759; all MSA work goes through the C ABI of include/pwr.h into the HIP kernels.
 */
#define _POSIX_C_SOURCE 200809L
#include "pwr.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

int pwr_read_msa_file(const char *path, int *rows, int *width, unsigned char **text, char *err, size_t errcap)
{
    FILE *f = fopen(path, "r");
    if (!f) { if (err) snprintf(err, errcap, "MA is missing."); return PWR_ERR_INPUT; }        /* PW:121 */
    const size_t bufsz = PWR_MAX_LINE + 3;
    char *buf = malloc(bufsz);
    unsigned char *data = NULL;
    size_t cap = 0, used = 0;
    int T = 0, W = -1, rc = PWR_OK;
    if (!buf) { fclose(f); return PWR_ERR_NOMEM; }
    while (fgets(buf, (int)(PWR_MAX_LINE + 1), f)) {                                           /* PW:119-122 */
        size_t n = strlen(buf);
        if (n == 0 || buf[n - 1] != '\n') {                                                     /* PW:134 */
            if (err) snprintf(err, errcap, "line %d is not terminated by a newline (or is longer than %d)", T + 1, PWR_MAX_LINE);
            rc = PWR_ERR_INPUT; break;
        }
        n--;
        if (W < 0) W = (int)n;
        if ((int)n != W) {
            /* the reference does not notice and silently loses bases; we refuse */
            if (err) snprintf(err, errcap, "line %d has %zu characters, the first line has %d", T + 1, n, W);
            rc = PWR_ERR_INPUT; break;
        }
        if (used + n > cap) {
            size_t ncap = cap ? cap * 2 : ((size_t)1 << 20);
            while (ncap < used + n) ncap *= 2;
            unsigned char *nd = realloc(data, ncap);
            if (!nd) { rc = PWR_ERR_NOMEM; break; }
            data = nd; cap = ncap;
        }
        memcpy(data + used, buf, n);
        used += n;
        T++;
    }
    fclose(f);
    free(buf);
    if (rc == PWR_OK && (T == 0 || W <= 0)) { if (err) snprintf(err, errcap, "empty MSA"); rc = PWR_ERR_INPUT; }
    if (rc != PWR_OK) { free(data); return rc; }
    *rows = T; *width = W; *text = data;
    return PWR_OK;
}

int pwr_write_msa_file(const char *path, int rows, int width, const unsigned char *text)
{
    FILE *f = fopen(path, "w");                                                                /* PW:1566 */
    if (!f) return PWR_ERR_IO;
    for (int r = 0; r < rows; r++) {
        if (width > 0 && fwrite(text + (size_t)r * width, 1, (size_t)width, f) != (size_t)width) { fclose(f); return PWR_ERR_IO; }
        if (fputc('\n', f) == EOF) { fclose(f); return PWR_ERR_IO; }
    }
    return fclose(f) == 0 ? PWR_OK : PWR_ERR_IO;
}

/* PW:945-961: (millions, units) with units in (0, 1e6], printed as %lu%06lu */
static void print_score(FILE *log, uint64_t total)
{
    uint64_t m = 0, u = 0;
    if (total > 0) { m = (total - 1) / 1000000u; u = (total - 1) % 1000000u + 1; }
    fprintf(log, "OverallScore: %lu%06lu\n", (unsigned long)m, (unsigned long)u);
}

static int write_current(pwr_ctx *ctx, const char *out_path)
{
    int T = 0, W = 0;
    int rc = pwr_dims(ctx, &T, &W);
    if (rc) return rc;
    unsigned char *buf = malloc((size_t)T * (size_t)(W > 0 ? W : 1));
    if (!buf) return PWR_ERR_NOMEM;
    rc = pwr_export_rows(ctx, buf, (size_t)T * W);
    if (rc == PWR_OK) rc = pwr_write_msa_file(out_path, T, W, buf);
    free(buf);
    return rc;
}

int pwr_run_file(const char *in_path, const char *out_path, int bandwidth, int device, int max_rounds, FILE *log)
{
    char err[256];
    int T = 0, W = 0, rc;
    unsigned char *text = NULL;
    pwr_ctx *ctx = NULL;
    fprintf(log, "output file: %s\n", out_path);                                               /* PW:1649 */
    fprintf(log, "bandwidth %d\n", bandwidth);                                                 /* PW:1650 */
    rc = pwr_read_msa_file(in_path, &T, &W, &text, err, sizeof err);
    if (rc) { fprintf(log, "%s\n", err); return 1; }
    rc = pwr_create(&ctx, T, W, text, bandwidth, device);
    free(text);
    if (rc) { fprintf(log, "PW_ReAligner: %s\n", pwr_strerror(rc)); return 1; }
    rc = pwr_trim_ends(ctx);                                                                   /* PW:1655 */
    if (rc == PWR_OK) rc = pwr_dims(ctx, &T, &W);
    if (rc) goto fail;
    fprintf(log, "Rows %d, Columns %d.\n", T, W);                                              /* PW:1657 */
    uint64_t best = 0, tot = 0;
    if ((rc = pwr_total_score(ctx, &best))) goto fail;                                         /* PW:1664-1665 */
    print_score(log, best);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int round = 0; round < 10000; round++) {                                              /* PW:1681 */
        if (max_rounds >= 0 && round >= max_rounds) break;
        if ((rc = pwr_realign_round(ctx))) goto fail;                                          /* PW:1695-1737 */
        if ((rc = pwr_total_score(ctx, &tot))) goto fail;
        print_score(log, tot);
        if (tot < best) {                                                                      /* PW:1741 */
            best = tot;
            if ((rc = write_current(ctx, out_path))) goto fail;
        } else break;                                                                          /* PW:1742 */
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(log, "Total time: %f min.\n", ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / 60.0);
    if ((rc = pwr_trim_ends(ctx))) goto fail;                                                  /* PW:1753 */
    if ((rc = pwr_total_score(ctx, &tot))) goto fail;
    print_score(log, tot);
    if (tot < best && (rc = write_current(ctx, out_path))) goto fail;                          /* PW:1754 */
    pwr_destroy(ctx);
    return 0;
fail:
    if (rc == PWR_ERR_IO) fprintf(log, "DateiVerbratei!\n");                                   /* PW:1570 */
    else if (rc == PWR_ERR_INTERNAL) fprintf(log, "\nStuff gone wrong\n");                     /* PW:1414 */
    else fprintf(log, "PW_ReAligner: %s\n", pwr_strerror(rc));
    pwr_destroy(ctx);
    return 1;
}
