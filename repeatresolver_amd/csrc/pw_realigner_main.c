/* Drop-in for the reference's ./PW_ReAligner (PW:1610-1647): same argv scan (two-character prefix
 * match, from argv[1] on), same defaults (output "MSAreal" PW:1619, bandwidth 1000 PW:1625).
 * Ours only: -g <device> selects the GPU, -r <n> stops after n rounds. */
#include "pwr.h"

#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc < 2) { printf("Usage: ./PW_ReAligner MApath\n"); return 0; }                      /* PW:1615 */
    const char *out = "MSAreal";
    int bandwidth = 1000, device = 0, max_rounds = -1;
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-' && argv[i][1] == 'o') { printf("%s\n", argv[i]); if (i + 1 < argc) out = argv[i + 1]; }  /* PW:1631-1635 */
        if (argv[i][0] == '-' && argv[i][1] == 'b' && i + 1 < argc) bandwidth = atoi(argv[i + 1]);                   /* PW:1637-1641 */
        if (argv[i][0] == '-' && argv[i][1] == 'g' && i + 1 < argc) device = atoi(argv[i + 1]);
        if (argv[i][0] == '-' && argv[i][1] == 'r' && i + 1 < argc) max_rounds = atoi(argv[i + 1]);
        if (argv[i][0] == '-' && argv[i][1] == 'h') {                                                                /* PW:1600-1607 */
            printf("Usage: ./PW_ReAligner MApath\n");
            printf("Flags:\n");
            printf("-o msa_path    Path of the refined multiple sequence alignment. Default: MSAreal.\n");
            printf("-b <1000>      The width of the band that is calculated in the alignment matrix.\n");
            printf("-g <0>         GPU to use.\n");
            return 0;
        }
    }
    return pwr_run_file(argv[1], out, bandwidth, device, max_rounds, stdout);
}
