/*
 * spx_probaln_general.hip -- probaln_glocal with PER-BASE query qualities (htslib's `iqual[i]`, which samtools' BAQ passes;
 * secphase itself always passes one constant: /root/reference/programs/submodules/ptMarker/ptMarker.c:747-757).
 *
 * The scoring kernels of spx_kernels.hip keep the two emission values of a problem's constant quality in registers for the
 * whole problem; a per-base quality would put a load into every DP row of the three hottest kernels.  The drop-in symbol
 * spx_probaln_glocal (include/spx.h) still has to honour htslib's contract, so problems with a non-constant `iqual` take
 * THIS kernel: one lane per problem runs the banded forward / backward / MAP recursions in the reference's own order (columns
 * ascending in the forward pass and the row sums, descending in the backward pass), full rows in HBM scratch like the CPU
 * implementation.  It is the contract that counts here, not the rate: the operations, their order and their roundings
 * are those of the other kernels and of the CPU restatement (no contraction: the Makefile's -ffp-contract=off; IEEE division;
 * the phred through the host-libm thresholds), so state[], q[] and the scaling factors equal theirs bit for bit.
 * Band index as in kprobaln: cell (i, k) of a band of half width bw lives at u = (k - i + bw) * 3 + 3 (+0 M, +1 I, +2 D)... in
 * the row's slice of 3 * (2 * bw + 1) + 6 doubles; the htslib releases that store only min(2 * bw + 1, l_ref) cells per row use
 * x = max(i - bw, 0), u = (k - x + 1) * 3 -- the same cell set, another address, identical arithmetic.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr double kEI = 0.25, kEM = 0.33333333333;

/* thr[k] = largest x with (int)(-4.343 * log(x) + .499) >= k, from the host libm (spx_host_tables) */
__device__ uint32_t phred_from_x(double x, const double *__restrict__ thr)
{
    if (!(x > 0.0)) return 0; /* x == 0 (log = -inf) or NaN: x86's conversion gives INT_MIN -> 0 */
    int lo = 0, hi = 101;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (x <= thr[mid]) lo = mid; else hi = mid - 1;
    }
    return lo > 100 ? 99u : (uint32_t)lo;
}

struct Args {
    const uint8_t *ref;   /* 0..3 = ACGT, > 3 ambiguous */
    const uint8_t *query;
    const float *qual;    /* per query base: (float)pow(10, -iqual / 10.), computed on the host */
    int32_t l_ref, l_query, bw;
    double m[9], bM, bI, sM, sI;
    double *f, *b, *s;    /* (l_query + 1) * i_dim zeroed doubles each (+ tail), l_query + 2 */
    int64_t i_dim;
    int32_t *state;
    uint8_t *q;
    const double *thr;
    int32_t drop_last; /* terminal-guard reading "row" and l_query <= bw, 2*bw+1 > l_ref: column l_ref is left out of the termination */
};

__device__ __forceinline__ int64_t slot(int bw, int i, int k) { return ((int64_t)(k - i + bw) * 3 + 3); }

__global__ void probaln_general_kernel(Args A)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int L = A.l_query, R = A.l_ref, bw = A.bw;
    const int64_t D = A.i_dim;
    const int bw2 = bw * 2 + 1;
    double *f = A.f, *b = A.b, *s = A.s;
    const double *m = A.m;
    f[slot(bw, 0, 0)] = 1.;
    s[0] = 1.;
    { /* row 1: no D state, division (not a reciprocal) by the row sum */
        double *fi = f + D, sum = 0.;
        const int end = R < bw + 1 ? R : bw + 1;
        for (int k = 1; k <= end; ++k) {
            const int64_t u = slot(bw, 1, k);
            const double ql = (double)A.qual[0];
            const double e = (A.ref[k - 1] > 3 || A.query[0] > 3) ? 1. : A.ref[k - 1] == A.query[0] ? 1. - ql : ql * kEM;
            fi[u + 0] = e * A.bM;
            fi[u + 1] = kEI * A.bI;
            sum += fi[u] + fi[u + 1];
        }
        s[1] = sum;
        for (int64_t k = slot(bw, 1, 1); k <= slot(bw, 1, end) + 2; ++k) fi[k] /= sum;
    }
    for (int i = 2; i <= L; ++i) {
        double *fi = f + (int64_t)i * D, *fp = f + (int64_t)(i - 1) * D, sum = 0.;
        const double qli = (double)A.qual[i - 1];
        const uint8_t qyi = A.query[i - 1];
        int beg = 1, end = R;
        if (beg < i - bw) beg = i - bw;
        if (end > i + bw) end = i + bw;
        for (int k = beg; k <= end; ++k) {
            const int64_t u = slot(bw, i, k), v11 = slot(bw, i - 1, k - 1), v10 = slot(bw, i - 1, k), v01 = slot(bw, i, k - 1);
            const double e = (A.ref[k - 1] > 3 || qyi > 3) ? 1. : A.ref[k - 1] == qyi ? 1. - qli : qli * kEM;
            fi[u + 0] = e * (m[0] * fp[v11 + 0] + m[3] * fp[v11 + 1] + m[6] * fp[v11 + 2]);
            fi[u + 1] = kEI * (m[1] * fp[v10 + 0] + m[4] * fp[v10 + 1]);
            fi[u + 2] = m[2] * fi[v01 + 0] + m[8] * fi[v01 + 2];
            sum += fi[u] + fi[u + 1] + fi[u + 2];
        }
        s[i] = sum;
        sum = 1. / sum;
        for (int64_t k = slot(bw, i, beg); k <= slot(bw, i, end) + 2; ++k) fi[k] *= sum;
    }
    /* (in this kernel's kprobaln addressing `u >= bw2*3+3` is plain band membership; the other reading of htslib's guard, u >= i_dim-3
     * in ITS addressing, drops column l_ref in the regime spx_logic.h terminal_drop() names: the host passes that as drop_last) */
    const int Rt = A.drop_last ? R - 1 : R;
    { /* termination */
        double sum = 0.;
        for (int k = 1; k <= Rt; ++k) {
            const int64_t u = slot(bw, L, k);
            if (u < 3 || u >= (int64_t)bw2 * 3 + 3) continue;
            sum += f[(int64_t)L * D + u + 0] * A.sM + f[(int64_t)L * D + u + 1] * A.sI;
        }
        s[L + 1] = sum;
    }
    /* backward */
    for (int k = 1; k <= Rt; ++k) {
        const int64_t u = slot(bw, L, k);
        double *bi = b + (int64_t)L * D;
        if (u < 3 || u >= (int64_t)bw2 * 3 + 3) continue;
        bi[u + 0] = A.sM / s[L] / s[L + 1];
        bi[u + 1] = A.sI / s[L] / s[L + 1];
    }
    for (int i = L - 1; i >= 1; --i) {
        double *bi = b + (int64_t)i * D, *bn = b + (int64_t)(i + 1) * D;
        double y = i > 1 ? 1. : 0.;
        const double qli1 = (double)A.qual[i];
        const uint8_t qyi1 = A.query[i];
        int beg = 1, end = R;
        if (beg < i - bw) beg = i - bw;
        if (end > i + bw) end = i + bw;
        for (int k = end; k >= beg; --k) {
            const int64_t u = slot(bw, i, k), v11 = slot(bw, i + 1, k + 1), v10 = slot(bw, i + 1, k), v01 = slot(bw, i, k + 1);
            const double e = (k >= R ? 0. : (A.ref[k] > 3 || qyi1 > 3) ? 1. : A.ref[k] == qyi1 ? 1. - qli1 : qli1 * kEM) * bn[v11];
            bi[u + 0] = e * m[0] + kEI * m[1] * bn[v10 + 1] + m[2] * bi[v01 + 2];
            bi[u + 1] = e * m[3] + kEI * m[4] * bn[v10 + 1];
            bi[u + 2] = (e * m[6] + m[8] * bi[v01 + 2]) * y;
        }
        y = 1. / s[i];
        for (int64_t k = slot(bw, i, beg); k <= slot(bw, i, end) + 2; ++k) bi[k] *= y;
    }
    /* MAP */
    for (int i = 1; i <= L; ++i) {
        const double *fi = f + (int64_t)i * D, *bi = b + (int64_t)i * D;
        double sum = 0., mx = 0.;
        int beg = 1, end = R, max_k = -1;
        if (beg < i - bw) beg = i - bw;
        if (end > i + bw) end = i + bw;
        for (int k = beg; k <= end; ++k) {
            const int64_t u = slot(bw, i, k);
            double z = fi[u + 0] * bi[u + 0];
            if (z > mx) { mx = z; max_k = (k - 1) << 2 | 0; }
            sum += z;
            z = fi[u + 1] * bi[u + 1];
            if (z > mx) { mx = z; max_k = (k - 1) << 2 | 1; }
            sum += z;
        }
        mx /= sum;
        A.state[i - 1] = max_k;
        A.q[i - 1] = (uint8_t)phred_from_x(1. - mx, A.thr);
    }
}

} // namespace

/* one problem; f / b: zeroed device scratch of (l_query + 1) * i_dim + 8 doubles each, i_dim = 3 * (2 * bw + 1) + 6 */
extern "C" hipError_t spx_launch_probaln_general(const uint8_t *d_ref, int32_t l_ref, const uint8_t *d_query, int32_t l_query, const float *d_qual, int32_t bw,
                                                 const double *hmm9_bM_bI_sM_sI /* host: m[0..8], bM, bI, sM, sI */, double *d_f, double *d_b, double *d_s,
                                                 int64_t i_dim, int32_t *d_state, uint8_t *d_q, const double *d_thr, int32_t drop_last_column, hipStream_t st)
{
    Args A;
    A.ref = d_ref; A.query = d_query; A.qual = d_qual;
    A.l_ref = l_ref; A.l_query = l_query; A.bw = bw;
    for (int k = 0; k < 9; ++k) A.m[k] = hmm9_bM_bI_sM_sI[k];
    A.bM = hmm9_bM_bI_sM_sI[9]; A.bI = hmm9_bM_bI_sM_sI[10]; A.sM = hmm9_bM_bI_sM_sI[11]; A.sI = hmm9_bM_bI_sM_sI[12];
    A.f = d_f; A.b = d_b; A.s = d_s; A.i_dim = i_dim;
    A.state = d_state; A.q = d_q; A.thr = d_thr; A.drop_last = drop_last_column;
    hipLaunchKernelGGL(probaln_general_kernel, dim3(1), dim3(64), 0, st, A);
    return hipGetLastError();
}
