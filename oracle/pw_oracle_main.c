/* TEST INFRASTRUCTURE -- CLI around the CPU restatement, same argv as the reference (PW:1610-1647). */
#include "pw_oracle.h"
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc < 2) { printf("Usage: ./PW_ReAligner MApath\n"); return 0; }          /* PW:1615 */
    const char *out = "MSAreal";                                                    /* PW:1619 */
    int bandwidth = 1000, max_rounds = -1;                                          /* PW:1625 */
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] == '-' && argv[i][1] == 'o') { printf("%s\n", argv[i]); if (i + 1 < argc) out = argv[i + 1]; }
        if (argv[i][0] == '-' && argv[i][1] == 'b' && i + 1 < argc) bandwidth = atoi(argv[i + 1]);
        if (argv[i][0] == '-' && argv[i][1] == 'r' && i + 1 < argc) max_rounds = atoi(argv[i + 1]); /* ours */
        if (argv[i][0] == '-' && argv[i][1] == 'h') {
            printf("Usage: ./PW_ReAligner MApath\nFlags:\n-o msa_path\n-b <1000>\n");
            return 0;
        }
    }
    return pwo_run(argv[1], out, bandwidth, stdout, max_rounds);
}
