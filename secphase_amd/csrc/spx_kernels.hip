/*
 * spx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the secphase hot path.
 *
 *  baq_kernel<G,C>  banded profile-HMM forward/backward + MAP at the wanted
 *                   rows (= htslib-1.17 probaln_glocal as called from
 *                   /root/reference/programs/submodules/ptMarker/ptMarker.c:755-757)
 *                   fused with secphase's write-back rule (ptMarker.c:778-779,786).
 *  score_kernel     filter_lowq_markers + calc_alignment_score + the
 *                   deterministic part of get_best_record_index
 *                   (ptMarker.c:110-153,307-325; ptAlignment.c:137-177).
 *
 * Mapping: FP64-VALU bound, not HBM bound.  One DP problem
 * is owned by G adjacent lanes of a wavefront (64/G problems per wave); the
 * band is stored on DIAGONALS (slot j <-> column k = i - bw + j) so that the
 * M recurrence is lane-local, and each lane keeps C consecutive slots of the
 * current row entirely in VGPRs (3*C doubles).  Bit-exactness with the CPU
 * order of operations is kept by construction: no FMA contraction
 * (-ffp-contract=off), IEEE division, the D-state recurrence and the row sum
 * are evaluated in the reference's sequential column order (G short masked
 * passes with a wave-shuffle carry), and the per-row scale factor is applied
 * exactly where the reference applies it.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spx_device.h"

#define SPX_EI 0.25

/* ---- neighbour exchange inside a group of G adjacent lanes -------------- */
/* G <= 16: DPP row shifts (VALU, no LDS round trip); wider groups: ds_bpermute */
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int G>
__device__ __forceinline__ double shfl_up1(double v) /* lane l <- lane l-1 */
{
    if constexpr (G == 1) return v;
    else if constexpr (G <= 16) return dpp_f64<0x111>(v); /* row_shr:1 */
    else return __shfl_up(v, 1, G);
}
template <int G>
__device__ __forceinline__ double shfl_down1(double v) /* lane l <- lane l+1 */
{
    if constexpr (G == 1) return v;
    else if constexpr (G <= 16) return dpp_f64<0x101>(v); /* row_shl:1 */
    else return __shfl_down(v, 1, G);
}

__device__ __forceinline__ uint32_t fetch_code(const uint8_t *__restrict__ pool, int64_t nib0, int idx, int n)
{
    if ((unsigned)idx >= (unsigned)n) return SPX_CODE_OUT;
    int64_t a = nib0 + idx;
    uint32_t b = pool[a >> 1];
    return (a & 1) ? (b >> 4) : (b & 0xfu);
}

/* byte-packed window of C codes */
template <int C>
struct CodeWin {
    static constexpr int NW = (C + 3) / 4;
    uint32_t w[NW];
    __device__ __forceinline__ uint32_t get(int c) const { return (w[c >> 2] >> (8 * (c & 3))) & 0xffu; }
    __device__ __forceinline__ void set(int c, uint32_t v)
    {
        w[c >> 2] = (w[c >> 2] & ~(0xffu << (8 * (c & 3)))) | (v << (8 * (c & 3)));
    }
    /* slot c <- slot c+1, slot C-1 <- v */
    __device__ __forceinline__ void shift_down(uint32_t v)
    {
#pragma unroll
        for (int k = 0; k < NW - 1; ++k) w[k] = (w[k] >> 8) | (w[k + 1] << 24);
        w[NW - 1] = (w[NW - 1] >> 8);
        set(C - 1, v);
    }
    /* slot c <- slot c-1, slot 0 <- v */
    __device__ __forceinline__ void shift_up(uint32_t v)
    {
#pragma unroll
        for (int k = NW - 1; k > 0; --k) w[k] = (w[k] << 8) | (w[k - 1] >> 24);
        w[0] = (w[0] << 8) | v;
    }
};

/* FAST: interior rows of problems without ambiguous bases -- every band slot below W is a real
 * cell, so the emission is a two-way select and no validity masks are needed.  (Slots whose column
 * is < 1 hold exact zeros by induction and stay zero whatever finite emission they are given.) */
template <bool FAST>
__device__ __forceinline__ double emission(uint32_t code, uint32_t qy, double e_match, double e_mis)
{
    double e = (code == qy) ? e_match : e_mis;
    if constexpr (!FAST) {
        if ((code | qy) & SPX_CODE_N) e = 1.0;
        if (code & SPX_CODE_OUT) e = 0.0;
    }
    return e;
}

/* phred of the posterior: (int)(-4.343*log(x)+.499) with x = 1 - max/sum,
 * evaluated through thresholds computed on the host with the host libm so
 * that the result is identical to the CPU path bit for bit. */
__device__ __forceinline__ uint32_t phred_from_x(double x, const double *__restrict__ thr)
{
    if (!(x > 0.0)) return 0; /* x == 0 (log = -inf) or NaN: x86 (int) conversion gives INT_MIN -> 0 */
    /* thr[k] = largest x with f(x) >= k, k = 1..101, decreasing in k */
    int lo = 0, hi = 101;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (x <= thr[mid]) lo = mid; else hi = mid - 1;
    }
    return lo > 100 ? 99u : (uint32_t)lo;
}

struct HmmC {
    double m0, m1, m2, m3, m4, m6, m8, e_match, e_mis;
};

/* one forward row (i >= 2), in place: on entry fM,fI,fD = scaled row i-1; on exit scaled row i.
 * Returns the row sum s[i]. */
template <int G, int C, bool FAST>
__device__ __forceinline__ double fwd_row(double (&fM)[C], double (&fI)[C], double (&fD)[C], const CodeWin<C> &ew,
                                          uint32_t qy, const HmmC &h, int g, int Wu, int tlast, double &inv_out)
{
    double nM = shfl_down1<G>(fM[0]), nI = shfl_down1<G>(fI[0]);
    if (g == G - 1) { nM = 0.0; nI = 0.0; }
    /* parallel phase: fM<-M(i,.), fI<-I(i,.), fD<-m2*M(i,k-1) */
    double prevM;
    {
        const double S0 = (h.m0 * fM[0] + h.m3 * fI[0]) + h.m6 * fD[0];
        const double e0 = emission<FAST>(ew.get(0), qy, h.e_match, h.e_mis);
        const double pMn = C > 1 ? fM[1] : nM, pIn = C > 1 ? fI[1] : nI;
        prevM = e0 * S0;
        fI[0] = SPX_EI * (h.m1 * pMn + h.m4 * pIn);
        fM[0] = prevM;
    }
#pragma unroll
    for (int c = 1; c < C; ++c) {
        const double S = (h.m0 * fM[c] + h.m3 * fI[c]) + h.m6 * fD[c];
        const double e = emission<FAST>(ew.get(c), qy, h.e_match, h.e_mis);
        const double pMn = (c + 1 < C) ? fM[c + 1] : nM, pIn = (c + 1 < C) ? fI[c + 1] : nI;
        const double newM = e * S;
        fI[c] = SPX_EI * (h.m1 * pMn + h.m4 * pIn);
        fD[c] = h.m2 * prevM;
        fM[c] = newM;
        prevM = newM;
    }
    {
        double pl = shfl_up1<G>(prevM);
        if (g == 0 || g > tlast) pl = 0.0; /* lanes beyond the band keep exact zeros */
        fD[0] = h.m2 * pl;
    }
    /* serial phase: D recurrence and row sum in column order, one lane of the group at a time.
     * Straight-line code only (uniform selects, no branches inside the unrolled register arrays). */
    double carryD = 0.0, carryS = 0.0, mysum = 0.0;
    for (int t = 0; t <= tlast; ++t) {
        if (g == t) {
            double d = carryD, s = carryS;
            if (FAST && t < tlast) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    d = fD[c] + h.m8 * d;
                    fD[c] = d;
                    s = s + ((fM[c] + fI[c]) + d);
                }
            } else {
                const int nc = min(C, Wu - t * C);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    /* slots beyond the band (c >= nc) keep D = 0: it feeds M of that slot in the next row */
                    const bool valid = (c < nc) && (FAST || !(ew.get(c) & SPX_CODE_OUT));
                    const double dn = fD[c] + h.m8 * d;
                    d = valid ? dn : 0.0;
                    fD[c] = d;
                    const double tt = (fM[c] + fI[c]) + d;
                    s = (FAST || valid) ? s + tt : s; /* FAST: pad slots hold exact zeros */
                }
            }
            carryD = d; carryS = s; mysum = s;
        }
        if (t < tlast) {
            carryD = shfl_up1<G>(carryD);
            carryS = shfl_up1<G>(carryS);
        }
    }
    const double tot = __shfl(mysum, tlast, G);
    const double inv = 1.0 / tot;
#pragma unroll
    for (int c = 0; c < C; ++c) { fM[c] *= inv; fI[c] *= inv; fD[c] *= inv; }
    inv_out = inv;
    return tot;
}

/* one backward row (1 <= i <= L-1), in place: on entry bM,bI = scaled row i+1; on exit scaled row i.
 * ew holds the code of column k+1 (ref index i - bw + j) per slot. */
template <int G, int C, bool FAST>
__device__ __forceinline__ void bwd_row(double (&bM)[C], double (&bI)[C], double (&bD)[C], const CodeWin<C> &ew,
                                        uint32_t qy, const HmmC &h, int g, int Wu, int tlast, double inv, bool first_row)
{
    double lI = shfl_up1<G>(bI[C - 1]);
    if (g == 0) lI = 0.0;
    const double em1 = SPX_EI * h.m1, em4 = SPX_EI * h.m4;
    /* parallel phase A (descending, in place): bM<-e*m0+EI*m1*bI', bI<-e*m3+EI*m4*bI', bD<-e*m6 */
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        const double e = emission<FAST>(ew.get(c), qy, h.e_match, h.e_mis) * bM[c];
        const double bin = c > 0 ? bI[c - 1] : lI;
        const double u = e * h.m0 + em1 * bin;
        const double v = e * h.m3 + em4 * bin;
        bD[c] = e * h.m6;
        bM[c] = u;
        bI[c] = v;
    }
    /* serial phase: D recurrence over descending columns; row 1 has y = 0 */
    if (!first_row) {
        double carryD = 0.0;
        for (int t = tlast; t >= 0; --t) {
            if (g == t) {
                double d = carryD;
                if (t < tlast) {
#pragma unroll
                    for (int c = C - 1; c >= 0; --c) {
                        d = bD[c] + h.m8 * d;
                        bD[c] = d;
                    }
                } else {
                    const int nc = min(C, Wu - t * C);
#pragma unroll
                    for (int c = C - 1; c >= 0; --c) {
                        const double x = (c < nc) ? bD[c] : 0.0; /* D of a column that does not exist stays 0 */
                        d = x + h.m8 * d;
                        bD[c] = d;
                    }
                }
                carryD = d;
            }
            if (t > 0) carryD = shfl_down1<G>(carryD);
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) bD[c] = 0.0;
    }
    /* parallel phase B: M += m2*D(i,k+1); scale */
    double hD = shfl_down1<G>(bD[0]);
    if (g == G - 1 || g >= tlast) hD = 0.0; /* the slot above the band has D = 0 */
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double dn = (c + 1 < C) ? bD[c + 1] : hD; /* slots beyond the band hold D = 0 (see below) */
        bM[c] = (bM[c] + h.m2 * dn) * inv;
        bI[c] = bI[c] * inv;
    }
}

#ifndef SPX_WAVES
#define SPX_WAVES 2
#endif
template <int G, int C>
__global__ __launch_bounds__(64, SPX_WAVES) void baq_kernel(spx_dev_batch B)
{
    constexpr int PPW = 64 / G; /* problems per wave */
    constexpr int SLOTS = G * C;
    const int lane = threadIdx.x & 63;
    const int g = lane % G;
    const int grp = lane / G;
    const int oslot = blockIdx.x * PPW + grp;
    const int pid = oslot < B.n_order ? B.order[oslot] : -1;
    const bool act = pid >= 0;

    int L = 0, R = 0, bw = 0;
    int64_t ref0 = 0, qry0 = 0;
    HmmC h = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int hasN = 0;
    if (act) {
        L = B.L[pid]; R = B.R[pid]; bw = B.bw[pid];
        ref0 = B.ref_nib[pid]; qry0 = B.qry_nib[pid];
        const double *hp = B.hmm + (int64_t)pid * SPX_H_N;
        h.m0 = hp[SPX_H_M0]; h.m1 = hp[SPX_H_M1]; h.m2 = hp[SPX_H_M2]; h.m3 = hp[SPX_H_M3]; h.m4 = hp[SPX_H_M4];
        h.m6 = hp[SPX_H_M6]; h.m8 = hp[SPX_H_M8]; h.e_match = hp[SPX_H_EMATCH]; h.e_mis = hp[SPX_H_EMIS];
        hasN = hp[SPX_H_PAD0] != 0.0; /* host flag: window or query holds an ambiguous base */
    }
    /* wave-uniform quantities: band width (host guarantees one W per wave), the longest query, the last
     * row every problem of the wave treats as interior, and whether any problem holds an N */
    int Wu = 0, Lw = 0, fwd_fast_end, bwd_fast_end;
    {
        int w = act ? 2 * bw + 1 : 0, l = L;
        int ff = act ? R - bw : 0x7fffffff;      /* forward row i is interior iff i + bw <= R       */
        int bf = act ? R - bw - 1 : 0x7fffffff;  /* backward row i is interior iff i + bw <  R       */
        int anyN = hasN;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            w = max(w, __shfl_xor(w, o));
            l = max(l, __shfl_xor(l, o));
            ff = min(ff, __shfl_xor(ff, o));
            bf = min(bf, __shfl_xor(bf, o));
            anyN |= __shfl_xor(anyN, o);
        }
        Wu = __builtin_amdgcn_readfirstlane(w);
        Lw = __builtin_amdgcn_readfirstlane(l);
        fwd_fast_end = __builtin_amdgcn_readfirstlane(anyN ? 0 : ff);
        bwd_fast_end = __builtin_amdgcn_readfirstlane(anyN ? 0 : bf);
    }
    if (Lw == 0) return;
    const int tlast = (Wu - 1) / C; /* last lane of a group that owns band slots */
    const int jbase = g * C;

    double fM[C], fI[C], fD[C];
    CodeWin<C> cw, padw; /* padw: SPX_CODE_OUT in the slots beyond the band (j >= W), fixed per problem */
    double *sinv = B.sinv + (act ? B.s_off[pid] : 0);
    const int nrows = act ? B.n_rows[pid] : 0;
    const int row0 = act ? B.row_off[pid] : 0;
    double *fsave = B.fsave + (act ? B.fsave_off[pid] : 0);
    const int64_t fstride = B.fsave_stride;

    /* ------------------------------------------------------------------ */
    /* forward row 1: f(1,k) = e*bM, EI*bI for k in [1, min(R, bw+1)], divided by the row sum */
    double s_cur = 1.0;
    int wnext = 0;
    int next_row = nrows > 0 ? B.rows[row0] : 0x7fffffff;
    {
        const double bM = act ? B.hmm[(int64_t)pid * SPX_H_N + SPX_H_BM] : 0.0;
        const double bI = act ? B.hmm[(int64_t)pid * SPX_H_N + SPX_H_BI] : 0.0;
        const uint32_t qy = act ? fetch_code(B.qry4, qry0, 0, L) : 0;
        /* window for row 1: slot j <-> ref idx r = 1 - bw + j - 1 */
#pragma unroll
        for (int k = 0; k < CodeWin<C>::NW; ++k) { cw.w[k] = 0; padw.w[k] = 0; }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            int j = jbase + c;
            uint32_t code = act ? fetch_code(B.ref4, ref0, j - bw, R) : SPX_CODE_OUT;
            cw.set(c, code);
            padw.set(c, j < Wu ? 0u : (uint32_t)SPX_CODE_OUT);
        }
        double carry = 0.0, mysum = 0.0;
        CodeWin<C> ew;
#pragma unroll
        for (int k = 0; k < CodeWin<C>::NW; ++k) ew.w[k] = cw.w[k] | padw.w[k];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint32_t code = ew.get(c);
            bool valid = !(code & SPX_CODE_OUT);
            double e = emission<false>(code, qy, h.e_match, h.e_mis);
            fM[c] = valid ? e * bM : 0.0;
            fI[c] = valid ? SPX_EI * bI : 0.0;
            fD[c] = 0.0;
        }
        for (int t = 0; t <= tlast; ++t) {
            if (g == t) {
                double s = carry;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    bool valid = !(ew.get(c) & SPX_CODE_OUT);
                    double tt = fM[c] + fI[c];
                    s = valid ? s + tt : s;
                }
                mysum = s;
                carry = s;
            }
            carry = shfl_up1<G>(carry);
        }
        double tot = __shfl(mysum, tlast, G);
        s_cur = tot;
        if (act) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                fM[c] = fM[c] / tot;
                fI[c] = fI[c] / tot;
            }
            if (g == 0) sinv[1] = 1.0 / tot;
        }
        if (act && next_row == 1) {
            double *dst = fsave + (int64_t)wnext * fstride + jbase;
#pragma unroll
            for (int c = 0; c < C; ++c) { dst[c] = fM[c]; dst[SLOTS + c] = fI[c]; }
            wnext++;
            next_row = wnext < nrows ? B.rows[row0 + wnext] : 0x7fffffff;
        }
    }
    /* ------------------------------------------------------------------ */
    /* forward rows 2..L */
    uint32_t qy_n = (act && L >= 2) ? fetch_code(B.qry4, qry0, 1, L) : 0;
    uint32_t rc_n = (act && L >= 2) ? fetch_code(B.ref4, ref0, 2 - bw + (jbase + C - 1) - 1, R) : SPX_CODE_OUT;
    for (int i = 2; i <= Lw; ++i) {
        const bool on = act && i <= L;
        if (on) {
            const uint32_t qy = qy_n;
            cw.shift_down(rc_n);
            CodeWin<C> ew;
#pragma unroll
            for (int k = 0; k < CodeWin<C>::NW; ++k) ew.w[k] = cw.w[k] | padw.w[k];
            /* prefetch next row's query base and incoming ref code */
            qy_n = fetch_code(B.qry4, qry0, i, L);
            rc_n = fetch_code(B.ref4, ref0, (i + 1) - bw + (jbase + C - 1) - 1, R);
            double inv;
            if (i <= fwd_fast_end) s_cur = fwd_row<G, C, true>(fM, fI, fD, ew, qy, h, g, Wu, tlast, inv);
            else s_cur = fwd_row<G, C, false>(fM, fI, fD, ew, qy, h, g, Wu, tlast, inv);
            if (g == 0) sinv[i] = inv;
            if (i == next_row) {
                double *dst = fsave + (int64_t)wnext * fstride + jbase;
#pragma unroll
                for (int c = 0; c < C; ++c) { dst[c] = fM[c]; dst[SLOTS + c] = fI[c]; }
                wnext++;
                next_row = wnext < nrows ? B.rows[row0 + wnext] : 0x7fffffff;
            }
        }
    }
    /* ------------------------------------------------------------------ */
    /* terminal: s[L+1] = sum_k f(L,k).M*sM + f(L,k).I*sI in column order */
    {
        const double sM = act ? B.hmm[(int64_t)pid * SPX_H_N + SPX_H_SM] : 0.0;
        const double sI = act ? B.hmm[(int64_t)pid * SPX_H_N + SPX_H_SI] : 0.0;
        double carry = 0.0, mysum = 0.0;
        for (int t = 0; t <= tlast; ++t) {
            const int nc = min(C, Wu - t * C);
            if (g == t) {
                double s = carry;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    int k = L - bw + jbase + c;
                    bool valid = c < nc && k >= 1 && k <= R;
                    double tt = fM[c] * sM + fI[c] * sI;
                    s = valid ? s + tt : s;
                }
                carry = s; mysum = s;
            }
            carry = shfl_up1<G>(carry);
        }
        const double sL1 = __shfl(mysum, tlast, G);
        /* backward row L */
        const double vM = (sM / s_cur) / sL1, vI = (sI / s_cur) / sL1;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            int j = jbase + c, k = L - bw + j;
            bool valid = act && j < Wu && k >= 1 && k <= R;
            fM[c] = valid ? vM : 0.0;
            fI[c] = valid ? vI : 0.0;
            fD[c] = 0.0;
        }
    }
    __threadfence_block(); /* sinv[] written by lane g==0 is read by the whole group below */
    /* MAP of one row: f from fsave, b in registers.  "First strictly greater" in column order:
     * within a lane the scan is in column order, across lanes the lower lane wins ties. */
    auto do_map = [&](int i, int w) {
        const double *src = fsave + (int64_t)w * fstride + jbase;
        double best = 0.0;
        int best_k = -1;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            int j = jbase + c, k = i - bw + j;
            bool valid = j < Wu && k >= 1 && k <= R;
            double a = src[c] * fM[c], b = src[SLOTS + c] * fI[c];
            if (valid && a > best) { best = a; best_k = ((k - 1) << 2) | 0; }
            if (valid && b > best) { best = b; best_k = ((k - 1) << 2) | 1; }
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {
            double ob = __shfl_up(best, o, G);
            int ok = __shfl_up(best_k, o, G);
            if (g >= o && ob >= best && ok >= 0) { best = ob; best_k = ok; }
        }
        best = __shfl(best, G - 1, G);
        best_k = __shfl(best_k, G - 1, G);
        /* sequential sum in column order (products recomputed: this runs on a few rows only) */
        double carry = 0.0, mysum = 0.0;
        for (int t = 0; t <= tlast; ++t) {
            if (g == t) {
                double s = carry;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    int j = jbase + c, k = i - bw + j;
                    if (j < Wu && k >= 1 && k <= R) {
                        s += src[c] * fM[c];
                        s += src[SLOTS + c] * fI[c];
                    }
                }
                carry = s; mysum = s;
            }
            carry = __shfl_up(carry, 1, G);
        }
        const double sum = __shfl(mysum, tlast, G);
        if (g == 0) {
            double mx = best / sum;
            uint32_t q = phred_from_x(1.0 - mx, B.qthr);
            int ridx = row0 + w;
            if (B.out_state) B.out_state[ridx] = best_k;
            if (B.out_q) B.out_q[ridx] = (uint8_t)q;
            if (B.out_bq) {
                int expect = B.row_expect[ridx];
                uint32_t raw = B.row_rawq[ridx];
                uint32_t bq = ((best_k & 3) != 0 || (best_k >> 2) != expect) ? 0u : (raw < q ? raw : q);
                B.out_bq[ridx] = (uint8_t)(bq < 94 ? bq : 93);
            }
        }
    };

    int wprev = nrows - 1;
    int prev_row = wprev >= 0 ? B.rows[row0 + wprev] : -1;
    if (act && prev_row == L) {
        do_map(L, wprev);
        wprev--;
        prev_row = wprev >= 0 ? B.rows[row0 + wprev] : -1;
    }
    /* ------------------------------------------------------------------ */
    /* backward rows L-1..1.  Window for row i holds the code of ref idx i - bw + j (= column k+1). */
#pragma unroll
    for (int c = 0; c < C; ++c) {
        int j = jbase + c;
        uint32_t code = (act && L >= 2) ? fetch_code(B.ref4, ref0, (L - 1) - bw + j, R) : SPX_CODE_OUT;
        cw.set(c, code);
    }
    uint32_t qy_p = (act && L >= 2) ? fetch_code(B.qry4, qry0, L - 1, L) : 0;
    uint32_t rc_p = SPX_CODE_OUT;
    double inv_p = (act && L >= 2) ? sinv[L - 1] : 0.0;
    for (int i = Lw - 1; i >= 1; --i) {
        const bool on = act && i <= L - 1;
        if (on) {
            const uint32_t qy = qy_p;
            const double inv = inv_p;
            if (i != L - 1) cw.shift_up(rc_p);
            CodeWin<C> ew;
#pragma unroll
            for (int k = 0; k < CodeWin<C>::NW; ++k) ew.w[k] = cw.w[k] | padw.w[k];
            /* prefetch for row i-1 */
            if (i >= 2) {
                qy_p = fetch_code(B.qry4, qry0, i - 1, L);
                rc_p = fetch_code(B.ref4, ref0, (i - 1) - bw + jbase, R);
                inv_p = sinv[i - 1];
            }
            /* pad slots (j >= W): in the general path their emission is 0, so their D stays 0; the fast path
             * ignores the pad code, hence D of pad slots is forced to 0 by never running the recurrence there
             * and by zeroing e*m6 below */
            if (i <= bwd_fast_end) {
                bwd_row<G, C, true>(fM, fI, fD, ew, qy, h, g, Wu, tlast, inv, i == 1);
            } else {
                bwd_row<G, C, false>(fM, fI, fD, ew, qy, h, g, Wu, tlast, inv, i == 1);
            }
            if (i == prev_row) {
                do_map(i, wprev);
                wprev--;
                prev_row = wprev >= 0 ? B.rows[row0 + wprev] : -1;
            }
        }
    }
}

/* ---------------------------------------------------------------------- */
/* marker filter + score + deterministic part of the decision, one thread per group */
__global__ __launch_bounds__(256) void score_kernel(spx_dev_groups Gd)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= Gd.n_groups) return;
    const int m0 = Gd.mk_first[gi], m1 = Gd.mk_first[gi + 1];
    const int n = Gd.n_aln[gi];
    const uint32_t sec = Gd.sec_mask[gi];
    double *out = Gd.score + (int64_t)gi * 10;
    double max_score = -1.7976931348623157e308, prim_score = -1.7976931348623157e308;
    int max_idx = -1, prim_idx = -1;
    for (int a = 0; a < n; ++a) {
        double sc = 0.0;
        int p = m0;
        while (p < m1) {
            /* one read position: min quality over its markers (filter_lowq_markers) */
            int e = p, mine = -1, is_match = 0;
            int minq = 100;
            do {
                const spx_dev_marker mk = Gd.markers[e];
                int q = mk.row >= 0 ? Gd.out_bq[mk.row] : mk.qfix;
                if (q < minq) minq = q;
                if (mk.aln == a) { mine = e; is_match = mk.is_match; }
                ++e;
            } while (e < m1 && !Gd.markers[e].first_of_pos);
            if (minq > Gd.min_q && mine >= 0) sc += is_match ? Gd.match_tbl[minq] : Gd.mis_tbl[minq];
            p = e;
        }
        out[a] = sc;
        if (!((sec >> a) & 1)) { prim_idx = a; prim_score = sc; }
        else if (max_score < sc) { max_idx = a; max_score = sc; }
    }
    uint32_t tie = 0;
    for (int a = 0; a < n; ++a)
        if (((sec >> a) & 1) && max_score <= out[a]) tie |= 1u << a;
    Gd.prim_idx[gi] = (uint8_t)prim_idx;
    Gd.max_idx[gi] = (uint8_t)max_idx;
    Gd.tie_mask[gi] = (uint16_t)tie;
    Gd.pass[gi] = !(prim_idx == -1 || max_score <= (prim_score + Gd.prim_margin) || max_score < Gd.min_score);
}

/* fixed-size decision records for the cross-rank gather (one RCCL collective):
 * {group index u32 | prim i8 | max_idx i8 | tie_mask u16 low byte.. } packed in 8 bytes */
__global__ __launch_bounds__(256) void pack_kernel(spx_dev_groups Gd, const int32_t *__restrict__ grp_index,
                                                   int32_t group_base, unsigned long long *__restrict__ out)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= Gd.n_groups) return;
    unsigned long long r = (unsigned long long)(uint32_t)(grp_index[gi] + group_base);
    r |= (unsigned long long)Gd.prim_idx[gi] << 32;
    r |= (unsigned long long)Gd.max_idx[gi] << 40;
    r |= (unsigned long long)Gd.tie_mask[gi] << 48;
    r |= (unsigned long long)(Gd.pass[gi] ? 1 : 0) << 63;
    out[gi] = r;
}

extern "C" hipError_t spx_launch_pack(const spx_dev_groups *Gd, const int32_t *grp_index, int32_t group_base,
                                      unsigned long long *out, hipStream_t st)
{
    if (Gd->n_groups <= 0) return hipSuccess;
    int blocks = (Gd->n_groups + 255) / 256;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, *Gd, grp_index, group_base, out);
    return hipGetLastError();
}

/* ---------------------------------------------------------------------- */
extern "C" hipError_t spx_launch_baq(int cls, const spx_dev_batch *B, hipStream_t st)
{
    if (B->n_order <= 0) return hipSuccess;
#define SPX_LAUNCH(G_, C_)                                                                    \
    {                                                                                         \
        int ppw = 64 / G_, blocks = (B->n_order + ppw - 1) / ppw;                             \
        hipLaunchKernelGGL((baq_kernel<G_, C_>), dim3(blocks), dim3(64), 0, st, *B);          \
    }                                                                                         \
    break;
    switch (cls) {
    case 0: SPX_LAUNCH(4, 12)
    case 1: SPX_LAUNCH(4, 16)
    case 2: SPX_LAUNCH(8, 16)
    case 3: SPX_LAUNCH(16, 16)
    case 4: SPX_LAUNCH(32, 16)
    case 5: SPX_LAUNCH(64, 16)
    case 6: SPX_LAUNCH(64, 32)
    default: return hipErrorInvalidValue;
    }
#undef SPX_LAUNCH
    return hipGetLastError();
}

extern "C" hipError_t spx_launch_score(const spx_dev_groups *Gd, hipStream_t st)
{
    if (Gd->n_groups <= 0) return hipSuccess;
    int blocks = (Gd->n_groups + 255) / 256;
    hipLaunchKernelGGL(score_kernel, dim3(blocks), dim3(256), 0, st, *Gd);
    return hipGetLastError();
}
