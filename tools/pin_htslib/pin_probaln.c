/*
 * pin_probaln.c -- dumps what the REAL htslib probaln_glocal returns for a list of problems, so that the oracle
 * (oracle/probaln_oracle.c) and the MI355X kernels can be pinned against it.  Not built in this repository's
 * container (htslib is absent there): run tools/pin_htslib/run.sh on any machine that has htslib 1.17.
 *
 *   pin_probaln < problems.txt > answers.txt
 * problems.txt, one problem per line:   l_ref l_query bw d e set_q <ref codes 0-4> <query codes 0-4>
 * answers.txt, one line per problem:    Pr state[0..l_query) q[0..l_query)
 *
 * The call is the one secphase makes (/root/reference/programs/submodules/ptMarker/ptMarker.c:747-757): constant
 * base qualities set_q, conf = {d, e, bw}.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <htslib/hts.h>

int main(void)
{
    static char ref_s[1 << 20], qry_s[1 << 20];
    int l_ref, l_query, bw, set_q;
    double d, e;
    while (scanf("%d %d %d %lf %lf %d %1048575s %1048575s", &l_ref, &l_query, &bw, &d, &e, &set_q, ref_s, qry_s) == 8) {
        uint8_t *ref = malloc(l_ref), *qry = malloc(l_query), *iq = malloc(l_query), *q = malloc(l_query);
        int *state = malloc(sizeof(int) * l_query);
        for (int k = 0; k < l_ref; ++k) ref[k] = (uint8_t)(ref_s[k] - '0');
        for (int k = 0; k < l_query; ++k) { qry[k] = (uint8_t)(qry_s[k] - '0'); iq[k] = (uint8_t)set_q; }
        probaln_par_t par = {(float)d, (float)e, bw};
        int pr = probaln_glocal(ref, l_ref, qry, l_query, iq, &par, state, q);
        printf("%d", pr);
        for (int k = 0; k < l_query; ++k) printf(" %d", state[k]);
        for (int k = 0; k < l_query; ++k) printf(" %d", (int)q[k]);
        printf("\n");
        free(ref); free(qry); free(iq); free(q); free(state);
    }
    return 0;
}
