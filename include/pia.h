/*
 * pia.h -- C ABI of the MI355X-native InitialAligner (part of libpwr.so), the step before PW_ReAligner in the
 * RepeatResolver pipeline.
 *
 * Reference: PhilippBongartz/RepeatResolver, InitialAligner.c ("IA:").  Like PW_ReAligner it has no library interface: its
 * boundary is the process (`./InitialAligner template.fasta Seq.fasta [-o msa] [-s seqclass] [-e cutoff] [-p threads]`,
 * IA:270-279, IA:667-735) and, inside, IntoAligner (IA:282-453, one semi-global edit-distance alignment per read, a full
 * len x template direction matrix each) and Building_MSA (IA:553-663).  The alignments are independent, so this is the
 * embarrassingly parallel part of the pipeline: pia_align runs them on the GPU, the rest is host C.
 * Error codes are those of pwr.h.
 */
#ifndef PIA_H
#define PIA_H

#include <stdio.h>

#include "pwr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PIA_MAX_LINE 70000       /* IA:84, IA:214: fgets buffers, and the template array Template[70000] */
#define PIA_MAX_READ 40000       /* IA:742: maxlength1 */

typedef struct pia_ctx pia_ctx;

/* The template (acgt in either case, what ReadingTemplate IA:214-262 leaves) is uploaded once.  Any other byte -- here and in
 * the reads of pia_align -- is refused with PWR_ERR_INPUT (the reference's reader reduces its input to the four bases before
 * it compares anything, IA:190-209; the kernels would silently take another byte for one of them). */
int pia_create(pia_ctx **out, const char *templ, int templ_len, int device);
void pia_destroy(pia_ctx *ctx);

/* IntoAligner (IA:282-453) for nreads reads: bases = the reads concatenated (lower-case acgt), off[nreads + 1] their
 * offsets.  Out: align[off[j] + x] = template position of base x of read j, or -1 when the base is placed between two
 * template bases (IA:420-446); dist[j] = the edit distance Row[entry] (IA:352: AlignmentError = dist / length). */
int pia_align(pia_ctx *ctx, int nreads, const char *bases, const long long *off, int *align, int *dist);
/* DP cells filled so far (read length x template length per read) and the summed duration of the fill kernel in ms. */
int pia_get_stats(pia_ctx *ctx, unsigned long long *cells, double *fill_ms);
/* "mem_budget": bytes of device memory for the stored direction bits of one batch of reads (0 = the default: 22 GB, or
 * less when the card has less to spare -- three quarters of the free memory over the four buffers in flight; an allocation
 * that fails all the same halves it and the batches are planned again); a read whose band alone is larger still gets a
 * batch of its own. */
int pia_set_option(pia_ctx *ctx, const char *key, long long value);
/* Where the last pia_align spent its time, ms: [0] all of it, [1] set-up and upload, [2] pass 1, [3] pass 2 with the
 * tracebacks, [4] the number of batches pass 2 took (not a time), [5] download. */
int pia_get_timing(pia_ctx *ctx, double *ms6);

/* ---- host side, plain C (pia_host.c) ---- */
/* ReadingTemplate (IA:214-262): every line that does not start with '>' contributes its aAcCgGtT, lower-cased. */
int pia_read_template(const char *path, char **templ, int *len);
/* ReadCounter / Offsetter / ReadingFasta (IA:66-213): records start at '>' lines; *bases and *off are malloc'ed. */
int pia_read_fasta(const char *path, int *nreads, char **bases, long long **off);
/* Building_MSA (IA:553-663): writes the MSA (lower-case acgt and '-', equal-width rows, only reads with
 * dist / length < cutoff) and the class file ('r' / 'l' per read). */
int pia_build_msa(const char *msa_path, const char *class_path, int nreads, const char *bases, const long long *off,
                  const int *align, const int *dist, double cutoff, int templ_len);
/* main() of the reference (IA:667-770): same stdout lines (without the progress percentages), same files; returns
 * the process exit code. */
int pia_run_files(const char *templ_path, const char *reads_path, const char *msa_path, const char *class_path,
                  double cutoff, int cutoff_given, int device, FILE *log);

#ifdef __cplusplus
}
#endif
#endif /* PIA_H */
