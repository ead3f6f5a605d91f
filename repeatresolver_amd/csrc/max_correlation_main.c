/* Drop-in for the reference's `MaxCorrelation` (MaxCorrelation.c main(), MC:916-1020): same argv, same output file
 * MaxCorrsOf_<MSApath>; the pair loop runs on the GPU behind include/pmc.h.  Extra flag: -g <device>. */
#include <stdio.h>
#include <stdlib.h>

#include "pmc.h"

int main(int argc, char **argv)
{
    if (argc < 2) { printf("Usage: ./MaxCorrelation MSApath <options>\n"); return 0; }     /* MC:922 */
    int cov = 30, device = 0;                                                               /* MC:925 */
    for (int i = 2; i < argc; i++) {
        if (argv[i][0] != '-') continue;
        if (argv[i][1] == 'p' && i + 1 < argc) printf("NTHREADS: %ld\n", strtol(argv[i + 1], NULL, 10));     /* MC:939-940: accepted, the GPU does the pairs */
        if (argv[i][1] == 'c' && i + 1 < argc) { cov = (int)strtol(argv[i + 1], NULL, 10); printf("Coverage %d\n", cov); }   /* MC:952-954 */
        if (argv[i][1] == 'f' && i + 2 < argc) printf("Full coverage from column %ld until %ld.\n", strtol(argv[i + 1], NULL, 10), strtol(argv[i + 2], NULL, 10));
        if (argv[i][1] == 'g' && i + 1 < argc) device = atoi(argv[i + 1]);
    }
    return pmc_run_file(argv[1], cov, device, stdout);
}
