/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement of /root/reference/InitialAligner.c ("IA"): the step before PW_ReAligner in the pipeline.
 * Used only as the checker for the HIP InitialAligner (tests/, never repeatresolver_amd/); pinned against the compiled
 * reference (oracle/_ref/initial_aligner) through tests/golden/ia_*.  Every function cites the IA lines it restates.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define IA_MAX_LINE 70000            /* IA:84, IA:214: fgets buffers */
#define IA_MAX_READ 40000            /* IA:742 maxlength1 */

/* IA:282-453 IntoAligner: semi-global edit distance of `read` (rows) into `templ` (columns): free start and end along the
 * template, every read base is aligned or dropped.  align[x] = template position of read base x, or -1 (IA:420-446).
 * Returns the distance; *entry = the template column the alignment ends in (IA:333-345). */
long iao_align(const char *read, int L1, const char *templ, int L2, int *align, int *entry_out, unsigned char *codes /* L1*L2 */)
{
    long *rowsh = malloc(sizeof(long) * ((size_t)L2 + 1)), *row = rowsh + 1;
    for (int y = -1; y < L2; y++) row[y] = 0;                                   /* IA:296 */
    for (int x = 0; x < L1; x++) {                                              /* IA:300-328 */
        long upper = x;
        row[-1] = x + 1;
        for (int y = 0; y < L2; y++) {
            const int m = read[x] == templ[y] ? 0 : 1;
            long e = upper + m;
            unsigned char c = m ? 0 : 3;
            if (row[y - 1] + 1 < e) { e = row[y - 1] + 1; c = 1; }
            if (row[y] + 1 < e) { e = row[y] + 1; c = 2; }
            upper = row[y];
            row[y] = e;
            codes[(size_t)x * L2 + y] = c;
        }
    }
    int y = L2 - 1, entry = y;                                                  /* IA:333-345: minimum of the last row, ties -> largest y; column 0 is not looked at */
    long mn = row[entry];
    for (int i = L2 - 1; i > 0; i--) if (row[i] < mn) { mn = row[i]; entry = i; }
    char *script = malloc((size_t)L1 + L2 + 2);
    int count = 0, x = L1 - 1;
    while (y > entry) { script[count++] = 'i'; y--; }                           /* IA:359-364 */
    while (x > -1 && y > -1) {                                                  /* IA:366-383 */
        const unsigned char c = codes[(size_t)x * L2 + y];
        if (c == 0) { script[count++] = 's'; x--; y--; }
        else if (c == 3) { script[count++] = 'm'; x--; y--; }
        else if (c == 1) { script[count++] = 'i'; y--; }
        else { script[count++] = 'd'; x--; }
    }
    while (x > -1) { script[count++] = 'd'; x--; }                              /* IA:389-400 */
    while (y > -1) { script[count++] = 'i'; y--; }
    x = 0; y = 0;                                                               /* IA:402-446 (script read back to front = inverted) */
    for (int i = count - 1; i >= 0; i--) {
        if (script[i] == 's' || script[i] == 'm') { align[x++] = y++; }
        else if (script[i] == 'i') y++;
        else align[x++] = -1;
    }
    if (entry_out) *entry_out = entry;
    free(script);
    free(rowsh);
    return mn;
}

/* ---- FASTA as the reference reads it ---- */
typedef struct { char **seq; int *len; int n; } iao_reads;

/* IA:156-213 ReadingFasta / IA:66-153: records start at lines beginning with '>', bases are aAcCgGtT (lower-cased), everything
 * else is skipped; a read ends at the next '>' or at the end of the file. */
static int load_fasta(const char *path, iao_reads *out)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char *buf = malloc(IA_MAX_LINE);
    int cap = 0, n = 0;
    char **seq = NULL; int *len = NULL, *scap = NULL;
    while (fgets(buf, IA_MAX_LINE, f)) {
        if (buf[0] == '>') {
            if (n == cap) { cap = cap ? 2 * cap : 1024; seq = realloc(seq, sizeof(char *) * cap); len = realloc(len, sizeof(int) * cap); scap = realloc(scap, sizeof(int) * cap); }
            seq[n] = NULL; len[n] = 0; scap[n] = 0; n++;
        } else if (n > 0) {
            for (int i = 0; buf[i] != '\n' && buf[i] != '\0'; i++) {
                char c = 0;
                switch (buf[i]) { case 'A': case 'a': c = 'a'; break; case 'C': case 'c': c = 'c'; break; case 'G': case 'g': c = 'g'; break; case 'T': case 't': c = 't'; break; default: break; }
                if (!c) continue;
                if (len[n - 1] == scap[n - 1]) { scap[n - 1] = scap[n - 1] ? 2 * scap[n - 1] : 4096; seq[n - 1] = realloc(seq[n - 1], scap[n - 1] + 1); }
                seq[n - 1][len[n - 1]++] = c;
            }
        }
    }
    fclose(f);
    free(buf); free(scap);
    out->seq = seq; out->len = len; out->n = n;
    return 0;
}

/* IA:218-262 ReadingTemplate: every non-'>' line of the file contributes its bases */
static int load_template(const char *path, char **out, int *len_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char *buf = malloc(IA_MAX_LINE), *t = malloc(IA_MAX_LINE + 1);
    int n = 0;
    while (fgets(buf, IA_MAX_LINE, f)) {
        if (buf[0] == '>') continue;
        for (int i = 0; buf[i] != '\n' && buf[i] != '\0'; i++) {
            char c = 0;
            switch (buf[i]) { case 'A': case 'a': c = 'a'; break; case 'C': case 'c': c = 'c'; break; case 'G': case 'g': c = 'g'; break; case 'T': case 't': c = 't'; break; default: break; }
            if (c && n < IA_MAX_LINE) t[n++] = c;
        }
    }
    fclose(f);
    free(buf);
    *out = t; *len_out = n;
    return 0;
}

/* IA:553-663 Building_MSA on in-memory alignments.  gap slot i = insertions before template base i (slot L2: after the
 * last); EVERY read widens the slots, also the ones the cut-off rejects (IA:576-597); rows only for reads with
 * error < cutoff (IA:606), class file 'r' / 'l' per read. */
int iao_build_msa(const char *msa_path, const char *class_path, const iao_reads *rd, int **align, const double *err, double cutoff, int L2)
{
    FILE *fm = fopen(msa_path, "w"), *fc = fopen(class_path, "w");
    if (!fm || !fc) return -1;
    int *gapcount = calloc((size_t)L2 + 1, sizeof(int));
    for (int j = 0; j < rd->n; j++) {
        int i = 0, count = 0;
        while (i < rd->len[j] && align[j][i] == -1) i++;
        if (i >= rd->len[j]) continue;                                          /* (no aligned base: the reference reads past the array here, IA:581) */
        int gap = align[j][i];
        for (i = 0; i < rd->len[j]; i++) {
            if (align[j][i] == -1) { count++; if (count > gapcount[gap]) gapcount[gap] = count; }
            else { gap = align[j][i] + 1; count = 0; }
        }
    }
    for (int j = 0; j < rd->n; j++) {
        if (err[j] < cutoff) {
            fputs("r\n", fc);
            const int rl = rd->len[j];
            if (rl > 0) {
                int k = 0;
                for (int i = 0; i < L2 + 1; i++) {
                    int count = 0;
                    while (k < rl && align[j][k] == -1) { fputc(rd->seq[j][k], fm); k++; count++; }
                    for (int l = count; l < gapcount[i]; l++) fputc('-', fm);
                    if (k < rl && align[j][k] == i) { fputc(rd->seq[j][k], fm); k++; }
                    else fputc('-', fm);
                }
            } else {
                for (int i = 0; i < L2 + 1; i++) { fputc('-', fm); for (int k = 0; k < gapcount[i]; k++) fputc('-', fm); }
            }
            fputc('\n', fm);
        } else fputs("l\n", fc);
    }
    fclose(fm); fclose(fc);
    free(gapcount);
    return 0;
}

/* the whole program, IA:667-770: returns 0, or -1 when a file is missing */
int iao_run(const char *templ_path, const char *reads_path, const char *msa_path, const char *class_path, double cutoff)
{
    char *templ; int L2;
    iao_reads rd;
    if (load_template(templ_path, &templ, &L2)) return -1;
    if (load_fasta(reads_path, &rd)) return -1;
    int **align = malloc(sizeof(int *) * (rd.n ? rd.n : 1));
    double *err = malloc(sizeof(double) * (rd.n ? rd.n : 1));
    int maxl = 1;
    for (int j = 0; j < rd.n; j++) if (rd.len[j] > maxl) maxl = rd.len[j];
    unsigned char *codes = malloc((size_t)maxl * (size_t)(L2 > 0 ? L2 : 1));
    for (int j = 0; j < rd.n; j++) {
        align[j] = malloc(sizeof(int) * (rd.len[j] ? rd.len[j] : 1));
        const long d = iao_align(rd.seq[j], rd.len[j], templ, L2, align[j], NULL, codes);
        err[j] = (double)d / (double)rd.len[j];                                /* IA:352 */
    }
    const int rc = iao_build_msa(msa_path, class_path, &rd, align, err, cutoff, L2);
    for (int j = 0; j < rd.n; j++) { free(align[j]); free(rd.seq[j]); }
    free(align); free(err); free(codes); free(rd.seq); free(rd.len); free(templ);
    return rc;
}

#ifdef IAO_MAIN
int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: ia_oracle template.fasta Seq.fasta out_msa out_seqclass [cutoff]\n"); return 2; }
    return iao_run(argv[1], argv[2], argv[3], argv[4], argc > 5 ? atof(argv[5]) : 0.30) ? 1 : 0;
}
#endif
