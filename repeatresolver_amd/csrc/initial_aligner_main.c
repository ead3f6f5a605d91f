/* Drop-in for the reference's `InitialAligner` (InitialAligner.c main(), IA:667-770): same argv, same files, same stdout
 * lines (minus the progress percentages); the alignments run on the GPU behind include/pia.h.  Extra flag: -g <device>. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pia.h"

static void help(void)
{
    printf("Usage: ./InitialAligner_parallel template.fasta Seq.fasta\n");                                              /* IA:272-278 */
    printf("Flags:\n");
    printf("-o msa_path    Path of the resulting multiple sequence alignment. Default: SimulatedMSA.\n");
    printf("-s <150>       Path of the seq class information: Which seqs are instances of the repeat. Default: SimulatedSeqClass\n");
    printf("-e <0.30>      The mapping error cutoff being used to detect instances of the template.\n");
    exit(0);
}

int main(int argc, char **argv)
{
    if (argc < 3) help();                                                                                              /* IA:669 */
    const char *templ_path = argv[1], *reads_path = argv[2];
    /* default outputs: the template path's prefix before "Template.fasta" + "MSA" / "SeqClass" (IA:676-700) */
    char prefix[300], out_msa[320], out_cls[320];
    size_t i = 0;
    const size_t tl = strlen(templ_path);
    while (i < tl && i < sizeof prefix - 1 && strcmp(templ_path + i, "Template.fasta") != 0) { prefix[i] = templ_path[i]; i++; }
    if (strcmp(templ_path + i, "Template.fasta") != 0) i = 0;
    prefix[i] = '\0';
    snprintf(out_msa, sizeof out_msa, "%sMSA", prefix);
    snprintf(out_cls, sizeof out_cls, "%sSeqClass", prefix);
    const char *msa = out_msa, *cls = out_cls;
    double cutoff = 0.30;
    int cutoff_given = 0, device = 0;
    for (int a = 1; a < argc; a++) {                                                                                   /* IA:705-735 */
        if (argv[a][0] != '-') continue;
        if (argv[a][1] == 'o' && a + 1 < argc) msa = argv[a + 1];
        if (argv[a][1] == 's' && a + 1 < argc) cls = argv[a + 1];
        if (argv[a][1] == 'e' && a + 1 < argc) { cutoff = atof(argv[a + 1]); cutoff_given = 1; }
        if (argv[a][1] == 'g' && a + 1 < argc) device = atoi(argv[a + 1]);
        if (argv[a][1] == 'h') help();
        /* -p <threads> is accepted and ignored: the reads are spread over the GPU instead */
    }
    return pia_run_files(templ_path, reads_path, msa, cls, cutoff, cutoff_given, device, stdout);
}
