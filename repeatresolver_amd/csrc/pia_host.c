/* pia_host.c -- host side of the InitialAligner drop-in, plain C (the reference's host code is C).
 *
 * File formats, stdout lines and exit codes follow InitialAligner.c ("IA:") main(), IA:667-770; the alignments themselves
 * (IntoAligner, IA:282-453) go through pia_align into the HIP kernels.
 */
#define _POSIX_C_SOURCE 200809L
#include "pia.h"

#include <stdlib.h>
#include <string.h>

static char base_of(char c)
{
    switch (c) {                                                                  /* IA:190-195, IA:240-245 */
    case 'A': case 'a': return 'a';
    case 'C': case 'c': return 'c';
    case 'G': case 'g': return 'g';
    case 'T': case 't': return 't';
    default: return 0;                                                            /* everything else is skipped */
    }
}

int pia_read_template(const char *path, char **templ, int *len)
{
    FILE *f = fopen(path, "r");
    if (!f) return PWR_ERR_INPUT;                                                 /* IA:226: exit(1) */
    char *buf = malloc(PIA_MAX_LINE), *t = malloc(PIA_MAX_LINE + 1);
    int n = 0, rc = PWR_OK;
    if (!buf || !t) { fclose(f); free(buf); free(t); return PWR_ERR_NOMEM; }
    while (fgets(buf, PIA_MAX_LINE, f)) {                                         /* IA:230-252 */
        if (buf[0] == '>') continue;
        for (int i = 0; buf[i] != '\n' && buf[i] != '\0'; i++) {
            const char c = base_of(buf[i]);
            if (!c) continue;
            if (n >= PIA_MAX_LINE) { rc = PWR_ERR_RANGE; break; }                 /* IA:214 Template[70000] */
            t[n++] = c;
        }
        if (rc) break;
    }
    fclose(f);
    free(buf);
    if (rc) { free(t); return rc; }
    *templ = t; *len = n;
    return PWR_OK;
}

int pia_read_fasta(const char *path, int *nreads, char **bases, long long **off)
{
    FILE *f = fopen(path, "r");
    if (!f) return PWR_ERR_INPUT;                                                 /* IA:79 */
    char *buf = malloc(PIA_MAX_LINE);
    size_t cap = (size_t)1 << 20, used = 0;
    char *b = malloc(cap);
    int ocap = 1024, n = 0, rc = PWR_OK;
    long long *o = malloc(sizeof(long long) * (ocap + 1));
    if (!buf || !b || !o) { fclose(f); free(buf); free(b); free(o); return PWR_ERR_NOMEM; }
    while (fgets(buf, PIA_MAX_LINE, f)) {
        if (buf[0] == '>') {                                                      /* IA:80-83: a record starts */
            if (n == ocap) { ocap *= 2; long long *no = realloc(o, sizeof(long long) * (ocap + 1)); if (!no) { rc = PWR_ERR_NOMEM; break; } o = no; }
            o[n++] = (long long)used;
        } else if (n > 0) {                                                       /* IA:184-201 */
            for (int i = 0; buf[i] != '\n' && buf[i] != '\0'; i++) {
                const char c = base_of(buf[i]);
                if (!c) continue;
                if (used == cap) { cap *= 2; char *nb = realloc(b, cap); if (!nb) { rc = PWR_ERR_NOMEM; break; } b = nb; }
                b[used++] = c;
            }
            if (rc) break;
        }
    }
    fclose(f);
    free(buf);
    if (rc) { free(b); free(o); return rc; }
    o[n] = (long long)used;
    *nreads = n; *bases = b; *off = o;
    return PWR_OK;
}

int pia_build_msa(const char *msa_path, const char *class_path, int nreads, const char *bases, const long long *off,
                  const int *align, const int *dist, double cutoff, int templ_len)
{
    FILE *fm = fopen(msa_path, "w");                                              /* IA:555-559 */
    FILE *fc = fopen(class_path, "w");
    if (!fm || !fc) { if (fm) fclose(fm); if (fc) fclose(fc); return PWR_ERR_IO; }
    int *gapcount = calloc((size_t)templ_len + 1, sizeof(int));
    if (!gapcount) { fclose(fm); fclose(fc); return PWR_ERR_NOMEM; }
    /* the widest run of bases any read places before template base i (slot templ_len: behind the last), IA:571-597;
     * every read counts, also those the cut-off turns away */
    for (int j = 0; j < nreads; j++) {
        const int rl = (int)(off[j + 1] - off[j]);
        const int *al = align + off[j];
        int i = 0, count = 0;
        while (i < rl && al[i] == -1) i++;
        if (i >= rl) continue;                       /* no base aligned at all (the reference reads past the array here, IA:581) */
        int gap = al[i];
        for (i = 0; i < rl; i++) {
            if (al[i] == -1) { count++; if (count > gapcount[gap]) gapcount[gap] = count; }
            else { gap = al[i] + 1; count = 0; }
        }
    }
    size_t width = 0;
    for (int i = 0; i < templ_len + 1; i++) width += (size_t)gapcount[i] + 1;
    char *line = malloc(width + 2);
    if (!line) { free(gapcount); fclose(fm); fclose(fc); return PWR_ERR_NOMEM; }
    for (int j = 0; j < nreads; j++) {                                            /* IA:602-655 */
        const int rl = (int)(off[j + 1] - off[j]);
        const int *al = align + off[j];
        const char *rd = bases + off[j];
        if ((double)dist[j] / (double)rl < cutoff) {                              /* IA:352, IA:606 (0/0 for an empty read: NaN, not smaller) */
            fputs("r\n", fc);
            size_t w = 0;
            int k = 0;
            for (int i = 0; i < templ_len + 1; i++) {
                int count = 0;
                while (k < rl && al[k] == -1) { line[w++] = rd[k]; k++; count++; }
                for (int l = count; l < gapcount[i]; l++) line[w++] = '-';
                if (k < rl && al[k] == i) { line[w++] = rd[k]; k++; }
                else line[w++] = '-';
            }
            line[w++] = '\n';
            fwrite(line, 1, w, fm);
        } else fputs("l\n", fc);
    }
    free(line);
    free(gapcount);
    if (fclose(fm) != 0) { fclose(fc); return PWR_ERR_IO; }
    return fclose(fc) == 0 ? PWR_OK : PWR_ERR_IO;
}

int pia_run_files(const char *templ_path, const char *reads_path, const char *msa_path, const char *class_path,
                  double cutoff, int cutoff_given, int device, FILE *log)
{
    char *templ = NULL, *bases = NULL;
    long long *off = NULL;
    int *align = NULL, *dist = NULL;
    int L2 = 0, n = 0, rc;
    pia_ctx *ctx = NULL;
    if (cutoff_given) fprintf(log, "errorcutoff %f.\n", cutoff);                   /* IA:722 */
    if (pia_read_template(templ_path, &templ, &L2)) return 1;                      /* IA:226 */
    fprintf(log, "template length %d\n", L2);                                      /* IA:739 */
    fprintf(log, "output file: %s\n", msa_path);                                   /* IA:745-746 */
    fprintf(log, "seqclass file: %s\n", class_path);
    if (pia_read_fasta(reads_path, &n, &bases, &off)) { free(templ); return 1; }   /* IA:79 */
    fprintf(log, "read count %d\n", n);                                            /* IA:750 */
    align = malloc(sizeof(int) * (size_t)(off[n] > 0 ? off[n] : 1));
    dist = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    rc = (align && dist) ? pia_create(&ctx, templ, L2, device) : PWR_ERR_NOMEM;
    if (rc == PWR_OK) rc = pia_align(ctx, n, bases, off, align, dist);             /* IA:754 Parallel_Aligning */
    if (rc == PWR_OK) {
        fprintf(log, "Writing the msa.\n");                                        /* IA:756 */
        rc = pia_build_msa(msa_path, class_path, n, bases, off, align, dist, cutoff, L2);
        if (rc == PWR_OK) fprintf(log, "\nFiles written.\n");                      /* IA:658 */
    }
    if (rc != PWR_OK) fprintf(log, "InitialAligner: %s\n", pwr_strerror(rc));
    pia_destroy(ctx);
    free(templ); free(bases); free(off); free(align); free(dist);
    return rc == PWR_OK ? 0 : 1;
}
