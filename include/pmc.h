/*
 * pmc.h -- C ABI of the MI355X-native MaxCorrelation (part of libpwr.so), the step after PW_ReAligner in the RepeatResolver
 * pipeline.
 *
 * Reference: PhilippBongartz/RepeatResolver, MaxCorrelation.c ("MC:").  Its boundary is the process
 * (`./MaxCorrelation MSApath [-c coverage] [-p threads]`, MC:916-1020): it reads the realigned MSA, and for every variation
 * (column, symbol in a c g t -) writes the largest significance of its co-occurrence with a variation at least 20 columns
 * away into `MaxCorrsOf_<MSApath>`, one "%f" per line (MC:516-532).  The hot loop (HilfsMaxCorrsRechner, MC:745-837) is
 * all pairs of variations x four intersections of row bit sets x one hypergeometric tail (GSL, MC:415): here a tiled
 * bit-set product with the tail evaluated in its epilogue.  Floating point: results agree with the reference's to the
 * accuracy of the tail sum (about 1e-12 relative), not bit for bit.  Error codes are those of pwr.h.
 */
#ifndef PMC_H
#define PMC_H

#include <stdio.h>

#include "pwr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PMC_MAX_COLUMNS 149997   /* MC:17, MC:274, MC:286: fgets(buffer, Max_Var_Anzahl - 2) */
#define PMC_MAX_ROWS 30000       /* MC:18 Max_Sig_Anzahl */

/* Parallel_AllMaxCorrsRechner(threads, 0, siglength, mincov, signumber, cutoff) (MC:839-905) on text = rows x width
 * characters (a c g t A C G T, '-' or '_' = gap, anything else = not covered, MC:303-330): maxcorrs[width * 5]. */
int pmc_maxcorrs(int rows, int width, const unsigned char *text, int mincov, int device, double *maxcorrs);
/* Duration of the last pmc_maxcorrs' device work, ms: [0] all, [1] bit sets, [2] ranges (MC:801), [3] pairs; [4] pairs evaluated.
 * (Kept per process, not per call: meaningful when pmc_maxcorrs is not called from several threads at once.) */
int pmc_last_timing(double *ms5);

/* ---- host side, plain C (pmc_host.c) ---- */
/* Einlesen (MC:270-336): the first line sets the width, lines of another width are skipped; *text is malloc'ed. */
int pmc_read_msa(const char *path, int *rows, int *width, unsigned char **text);
/* MaxCorrsRausschreiben (MC:516-532) */
int pmc_write(const char *path, int nvars, const double *maxcorrs);
/* main() (MC:916-1020): the reference's structural stdout lines, output file MaxCorrsOf_<path>; returns the exit code */
int pmc_run_file(const char *msa_path, int mincov, int device, FILE *log);

#ifdef __cplusplus
}
#endif
#endif /* PMC_H */
