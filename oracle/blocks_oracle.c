/*
 * blocks_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the ptBlock sort / merge routines behind secphase's BED side outputs
 * (/root/reference/programs/submodules/ptBlock/ptBlock.c:166-170 comparator, :238-272 merge,
 * :274-428 merge-with-count "v2", called from src/secphase.c:59-72).  PINNED: these are the only
 * routines of the reference with known-answer tests (src/secphase_test.c:30-231); the vectors are
 * committed as tests/golden/ptblock_kats.json and checked in tests/test_blocks.py.
 */
#include <stdlib.h>
#include <string.h>

#include "secphase_oracle.h"

typedef struct { int s, e, c; } seg;
typedef struct { seg *v; int n, cap; } segv;
static void spush(segv *a, int s, int e, int c)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 32; a->v = realloc(a->v, sizeof(seg) * a->cap); }
    a->v[a->n].s = s; a->v[a->n].e = e; a->v[a->n].c = c; a->n++;
}
static int cmp_s(const void *a, const void *b) { return ((const seg *)a)->s - ((const seg *)b)->s; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return b < a ? a : b; }

/* ptBlock_sort_stHash_by_rfs on one contig: qsort by start (ptBlock_cmp_rfs) */
void orc_blocks_sort(int n, int *s, int *e, int *c)
{
    seg *v = malloc(sizeof(seg) * (n ? n : 1));
    int i;
    for (i = 0; i < n; ++i) { v[i].s = s[i]; v[i].e = e[i]; v[i].c = c ? c[i] : 0; }
    qsort(v, n, sizeof(seg), cmp_s);
    for (i = 0; i < n; ++i) { s[i] = v[i].s; e[i] = v[i].e; if (c) c[i] = v[i].c; }
    free(v);
}

/* ptBlock_merge_blocks (ptBlock.c:238-272) on blocks sorted by start; counts add up when has_count */
int orc_blocks_merge(int n, const int *s, const int *e, const int *c, int *os, int *oe, int *oc)
{
    int i, m = 0, ms = 0, me = 0, mc = 0;
    for (i = 0; i < n; ++i) {
        if (i == 0) { ms = s[0]; me = e[0]; mc = c ? c[0] : 0; continue; }
        if (me < s[i]) {
            os[m] = ms; oe[m] = me; if (oc) oc[m] = mc; m++;
            ms = s[i]; me = e[i]; mc = c ? c[i] : 0;
        } else {
            me = imax(me, e[i]);
            if (c) mc += c[i];
        }
    }
    if (n > 0) { os[m] = ms; oe[m] = me; if (oc) oc[m] = mc; m++; }
    return m;
}

/* ptBlock_merge_blocks_v2 (ptBlock.c:274-428) on blocks sorted by start: disjoint segments, each
 * carrying the sum of the counts of the blocks that cover it (no count: c == NULL).
 * Output arrays must hold at least 2*n+1 entries.  Returns the number of segments. */
int orc_blocks_merge_v2(int n, const int *s, const int *e, const int *c, int *os, int *oe, int *oc)
{
    segv fin = {0}, on = {0}, tmp;
    int i, j, m;
    for (i = 0; i < n; ++i) {
        const int s2 = s[i], e2 = e[i], c2 = c ? c[i] : 0;
        int e1 = 0;
        if (on.n == 0) { spush(&on, s2, e2, c2); continue; }
        tmp = on;
        memset(&on, 0, sizeof on);
        for (j = 0; j < tmp.n; ++j) {
            const int s1 = tmp.v[j].s, c1 = tmp.v[j].c;
            e1 = tmp.v[j].e;
            if (e1 < s2) spush(&fin, s1, e1, c1);
            else if (s1 <= s2) {
                if (s1 < s2) spush(&fin, s1, s2 - 1, c1);
                spush(&on, s2, imin(e1, e2), c1 + c2);
                if (e2 < e1) spush(&on, e2 + 1, e1, c1);
            } else if (e1 <= e2) spush(&on, s1, e1, c1 + c2);
            else {
                if (s1 <= e2) spush(&on, s1, e2, c1 + c2);
                spush(&on, imax(e2 + 1, s1), e1, c1);
            }
        }
        if (imax(e1 + 1, s2) <= e2) spush(&on, imax(e1 + 1, s2), e2, c2);
        free(tmp.v);
    }
    for (j = 0; j < on.n; ++j) spush(&fin, on.v[j].s, on.v[j].e, on.v[j].c);
    for (m = 0; m < fin.n; ++m) { os[m] = fin.v[m].s; oe[m] = fin.v[m].e; if (oc) oc[m] = fin.v[m].c; }
    m = fin.n;
    free(fin.v); free(on.v);
    return m;
}
