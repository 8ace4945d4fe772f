/*
 * correct_bam -- command-line counterpart of the reference's correct_bam (/root/reference/programs/src/correct_bam.c): applies a
 * relabel list (`secphase`'s <prefix>.out.log) to a BAM by swapping the primary / secondary flags, plus the reference's filters.
 * Same options (getopt table of correct_bam.c:222-238, option string :257); the work is spx_correct_bam (spx_correct.cpp).
 * One addition: --samText writes SAM text instead of BAM.
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spx.h"

static struct option long_options[] = {{"inputBam", required_argument, NULL, 'i'},
                                       {"outputBam", required_argument, NULL, 'o'},
                                       {"phasingLog", required_argument, NULL, 'P'},
                                       {"mapqTable", required_argument, NULL, 'M'},
                                       {"minReadLen", required_argument, NULL, 'm'},
                                       {"minAlignmentLen", required_argument, NULL, 'a'},
                                       {"primaryOnly", no_argument, NULL, 'p'},
                                       {"exclude", required_argument, NULL, 'e'},
                                       {"threads", required_argument, NULL, 'n'},
                                       {"noTag", no_argument, NULL, 't'},
                                       {"maxMapq", required_argument, NULL, 'x'},
                                       {"maxDiv", required_argument, NULL, 'd'},
                                       {"samText", no_argument, NULL, 1000},
                                       {NULL, 0, NULL, 0}};

int main(int argc, char *argv[])
{
    const char *in = NULL, *out = NULL;
    spx_correct_options o;
    spx_correct_default_options(&o);
    const char *program = strrchr(argv[0], '/');
    program = program ? program + 1 : argv[0];
    int c;
    while (~(c = getopt_long(argc, argv, "i:o:x:e:P:M:tpm:a:n:d:h", long_options, NULL))) {
        switch (c) {
        case 'i': in = optarg; break;
        case 'o': out = optarg; break;
        case 'x': o.max_mapq = atoi(optarg); break;
        case 'e': o.exclude = optarg; break;
        case 'P': o.phasing_log = optarg; break;
        case 'M': o.mapq_table = optarg; break;
        case 't': o.no_tag = 1; break;
        case 'p': o.primary_only = 1; break;
        case 'm': o.min_read_length = atoi(optarg); break;
        case 'a': o.min_alignment_length = atoi(optarg); break;
        case 'n': o.threads = atoi(optarg); break;
        case 'd': o.max_divergence = atof(optarg); break;
        case 1000: o.sam_text = 1; break;
        default:
            if (c != 'h') fprintf(stderr, "[E::%s] undefined option %c\n", __func__, c);
            fprintf(stderr, "\nUsage: %s -i <INPUT_BAM> -o <OUTPUT_BAM> [-P <PHASING_LOG>] [-M <MAPQ_TABLE>]\n"
                            "\tapplies the relabel list of secphase (primary / secondary flags swapped where the list says so), sets MAPQs from a table,\n"
                            "\tfilters secondary alignments, short reads, short alignments, high MAPQ, divergent alignments; can drop the optional fields\n"
                            "Options:\n"
                            "         --inputBam, -i         input bam file (BAM only: SAM and CRAM input, which the reference reads through htslib, are not read here)\n"
                            "         --outputBam, -o        output bam file\n"
                            "         --maxMapq, -x          maximum mapq [default:100]\n"
                            "         --phasingLog, -P       the phasing log path (output of secphase) [optional]\n"
                            "         --mapqTable, -M        tab-delimited, no header: read_name, contig_name, 1_based_contig_start, new_mapq [optional]\n"
                            "         --exclude, -e          file with the read names to exclude [optional]\n"
                            "         --noTag, -t            output no optional fields\n"
                            "         --primaryOnly, -p      output only primary alignments\n"
                            "         --minReadLen, -m       min read length [default: 5k]\n"
                            "         --minAlignmentLen, -a  min alignment length [default: 5k]\n"
                            "         --maxDiv, -d           max gap-compressed divergence (\"de\" tag) [default: 0.12]\n"
                            "         --threads, -n          number of threads (for bam I/O) [default: 2]\n"
                            "         --samText              write SAM text instead of BAM\n",
                    program);
            return 1;
        }
    }
    if (!in || !out) { fprintf(stderr, "%s: -i and -o are required\n", program); return 1; }
    spx_correct_stats st;
    const int rc = spx_correct_bam(in, out, &o, &st);
    if (rc != SPX_OK) { fprintf(stderr, "%s: %s\n", program, spx_last_error()); return 1; }
    fprintf(stderr, "%s: %lld records read, %lld written; %lld reads in the phasing table; %lld records made primary, %lld made secondary\n", program,
            (long long)st.records_in, (long long)st.records_out, (long long)st.table_reads, (long long)st.made_primary, (long long)st.made_secondary);
    return 0;
}
