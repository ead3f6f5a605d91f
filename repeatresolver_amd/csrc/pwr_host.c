/* pwr_host.c -- host side of the PW_ReAligner drop-in, plain C (the reference's host code is C).
 *
 * File format, stdout lines, exit codes and the round loop follow PW_ReAligner.c ("PW:") main(),
 * PW:1610-1759; all MSA work goes through the C ABI of include/pwr.h into the HIP kernels.
 */
#define _POSIX_C_SOURCE 200809L
#define _FILE_OFFSET_BITS 64
#include "pwr.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

static int read_msa_by_line(const char *path, int *rows, int *width, unsigned char **text, char *err, size_t errcap)
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


/* The regular case in one read: a file of T lines of W characters, each ended by '\n', is T * (W + 1) bytes whose every
 * (W + 1)-th byte is the only newline of its line.  The check and the copy without the newlines are row-parallel (1.8 GB
 * at benchmark scale).  Anything else -- a short last line, unequal lengths, a line beyond PWR_MAX_LINE -- goes to the
 * line-by-line reader above, which words the refusal as PW:118-136 would. */
typedef struct { const unsigned char *raw; unsigned char *dst; size_t W; int r0, r1; int bad; } read_share;
static void *read_share_run(void *arg)
{
    read_share *q = arg;
    for (int r = q->r0; r < q->r1; r++) {
        const unsigned char *line = q->raw + (size_t)r * (q->W + 1);
        if (line[q->W] != '\n' || memchr(line, '\n', q->W) || memchr(line, 0, q->W)) { q->bad = 1; return NULL; }
        memcpy(q->dst + (size_t)r * q->W, line, q->W);
    }
    return NULL;
}

int pwr_read_msa_file(const char *path, int *rows, int *width, unsigned char **text, char *err, size_t errcap)
{
    FILE *f = fopen(path, "r");
    if (!f) { if (err) snprintf(err, errcap, "MA is missing."); return PWR_ERR_INPUT; }        /* PW:121 */
    unsigned char *raw = NULL, *data = NULL;
    long long size = -1;
    if (fseeko(f, 0, SEEK_END) == 0) { size = (long long)ftello(f); rewind(f); }
    int fast = 0;
    size_t W = 0, T = 0;
    if (size > 0 && (raw = malloc((size_t)size)) != NULL && fread(raw, 1, (size_t)size, f) == (size_t)size) {
        const unsigned char *nl = memchr(raw, '\n', (size_t)size);
        if (nl && nl > raw && (size_t)(nl - raw) <= PWR_MAX_LINE - 1 && (size_t)size % (size_t)(nl - raw + 1) == 0) {
            W = (size_t)(nl - raw); T = (size_t)size / (W + 1);
            fast = T > 0 && T < 0x7fffffff;
        }
    }
    fclose(f);
    if (fast && (data = malloc(T * W)) != NULL) {
        enum { MAXT = 16 };
        long nc = sysconf(_SC_NPROCESSORS_ONLN);
        int nt = (int)(nc < 1 ? 1 : (nc > MAXT ? MAXT : nc));
        if ((size_t)nt > T / 64 + 1) nt = (int)(T / 64 + 1);
        pthread_t th[MAXT];
        read_share sh[MAXT];
        int started[MAXT];
        for (int t = 0; t < nt; t++) {
            sh[t] = (read_share){raw, data, W, (int)((long long)T * t / nt), (int)((long long)T * (t + 1) / nt), 0};
            started[t] = pthread_create(&th[t], NULL, read_share_run, &sh[t]) == 0;
            if (!started[t]) read_share_run(&sh[t]);
        }
        int bad = 0;
        for (int t = 0; t < nt; t++) { if (started[t]) pthread_join(th[t], NULL); bad |= sh[t].bad; }
        free(raw);
        if (!bad) { *rows = (int)T; *width = (int)W; *text = data; return PWR_OK; }
        free(data);
    } else free(raw);
    return read_msa_by_line(path, rows, width, text, err, errcap);
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

/* The file the reference rewrites after every improving round (PW:1741, MMA_Auslesen PW:1556-1598) is 1.8 GB at benchmark
 * scale.  Its image is taken in stream order on the device (pwr_snapshot_begin) and the NEXT round starts at once; a writer
 * thread waits for the image's copy to the host and writes the file.  The file is opened here, before the round goes on
 * (PW:1566-1572: a path that cannot be written ends the run there and then), writes never overlap (the one before is
 * joined first), and the last one is joined before the run returns: whenever the reference's file is complete, so is ours,
 * byte for byte. */
typedef struct { pwr_snapshot *snap; FILE *f; int rc; pthread_t th; int running; } out_writer;

static void *out_writer_run(void *arg)
{
    out_writer *w = arg;
    const unsigned char *img = NULL;
    size_t bytes = 0;
    w->rc = pwr_snapshot_wait(w->snap, &img, &bytes, NULL, NULL);
    if (w->rc == PWR_OK) {
        const size_t chunk = (size_t)64 << 20;
        for (size_t o = 0; o < bytes && w->rc == PWR_OK; o += chunk) {
            const size_t n = bytes - o < chunk ? bytes - o : chunk;
            if (fwrite(img + o, 1, n, w->f) != n) w->rc = PWR_ERR_IO;
        }
    }
    if (fclose(w->f) != 0 && w->rc == PWR_OK) w->rc = PWR_ERR_IO;
    w->f = NULL;
    pwr_snapshot_free(w->snap);
    w->snap = NULL;
    return NULL;
}

static int out_writer_join(out_writer *w)
{
    if (!w->running) return PWR_OK;
    pthread_join(w->th, NULL);
    w->running = 0;
    return w->rc;
}

static int write_current(pwr_ctx *ctx, const char *out_path, out_writer *w)
{
    int rc = out_writer_join(w);                                                               /* the write before this one */
    if (rc) return rc;
    w->f = fopen(out_path, "w");                                                               /* PW:1566 */
    if (!w->f) return PWR_ERR_IO;
    setvbuf(w->f, NULL, _IONBF, 0);
    if ((rc = pwr_snapshot_begin(ctx, &w->snap))) { fclose(w->f); w->f = NULL; return rc; }
    w->rc = PWR_OK;
    if (pthread_create(&w->th, NULL, out_writer_run, w) != 0) { out_writer_run(w); return w->rc; }   /* (no thread: write it here) */
    w->running = 1;
    return PWR_OK;
}

int pwr_run_file(const char *in_path, const char *out_path, int bandwidth, int device, int max_rounds, FILE *log)
{
    char err[256];
    int T = 0, W = 0, rc;
    unsigned char *text = NULL;
    pwr_ctx *ctx = NULL;
    out_writer wr = {NULL, NULL, PWR_OK, 0, 0};
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
            if ((rc = write_current(ctx, out_path, &wr))) goto fail;
        } else break;                                                                          /* PW:1742 */
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(log, "Total time: %f min.\n", ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec)) / 60.0);
    if ((rc = pwr_trim_ends(ctx))) goto fail;                                                  /* PW:1753 */
    if ((rc = pwr_total_score(ctx, &tot))) goto fail;
    print_score(log, tot);
    if (tot < best && (rc = write_current(ctx, out_path, &wr))) goto fail;                     /* PW:1754 */
    if ((rc = out_writer_join(&wr))) goto fail;
    pwr_destroy(ctx);
    return 0;
fail:
    (void)out_writer_join(&wr);
    if (rc == PWR_ERR_IO) fprintf(log, "DateiVerbratei!\n");                                   /* PW:1570 */
    else if (rc == PWR_ERR_INTERNAL) fprintf(log, "\nStuff gone wrong\n");                     /* PW:1414 */
    else fprintf(log, "PW_ReAligner: %s\n", pwr_strerror(rc));
    pwr_destroy(ctx);
    return 1;
}
