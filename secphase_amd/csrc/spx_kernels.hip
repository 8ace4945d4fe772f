/*
 * spx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the secphase hot path.
 *
 *  baq_fwd1_kernel<W>        banded profile-HMM forward pass, one lane per DP problem (the HiFi band widths)
 *  baq_fwd_kernel<G,C,..>    the same with G lanes x C slots per problem (every other band width)
 *  baq_bwd_kernel<G,C,..>    backward pass; z = f*b at the wanted rows
 *  map_kernel                arg-max, ordered sum, phred and secphase's write-back rule per wanted row
 *                            (= htslib-1.17 probaln_glocal as called from
 *                            /root/reference/programs/submodules/ptMarker/ptMarker.c:755-757, with
 *                            ptMarker.c:778-779,786)
 *  posmin/score/decide       filter_lowq_markers + calc_alignment_score + the deterministic part of
 *                            get_best_record_index (ptMarker.c:110-153,307-325; ptAlignment.c:137-177)
 *  pack_kernel               16-byte decision records for the multi-GPU gather
 *
 * Mapping: FP64 vector-ALU work, not HBM- or MFMA-bound.  The band is stored on DIAGONALS
 * (slot j <-> column k = i - bw + j) so that the M recurrence is slot-local; a problem is owned by G adjacent lanes
 * of a wavefront, each holding C consecutive slots of the current row in VGPRs (the one-lane kernel parks its D row
 * in LDS).  Bit-exactness with the CPU order of operations is kept by construction: no FMA contraction
 * (-ffp-contract=off), IEEE division, the D-state recurrence and the row sum evaluated in the reference's
 * sequential column order (G short masked passes with a lane-to-lane carry), and the per-row scale factor applied
 * exactly where the reference applies it.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/spx.h"
#include "spx_device.h"
#include "spx_dp_dev.h"
#include "spx_prep_dev.h"

#define SPX_EI 0.25 /* (the compiler turns x * 0.25 into v_ldexp_f64; forcing v_mul_f64 instead changes nothing: 954.7 k vs 955.6 k groups/s) */


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


/* where the D row of a problem lives: in VGPRs (scaled together with M and I), or in LDS ([slot][lane],
 * bank-conflict free, left UNSCALED: the factor is applied when the value is read back in the next row,
 * which is the very multiplication the reference performs when it rescales the row).  LDS storage frees
 * 2*C VGPRs per lane, which is what lets wide (ONT) bands use 4 lanes per problem instead of 8. */
template <int C>
struct DRegs {
    static constexpr bool lds = false;
    double v[C];
    __device__ __forceinline__ double get(int c) const { return v[c]; }
    __device__ __forceinline__ void set(int c, double x) { v[c] = x; }
};
template <int C>
struct DLds {
    static constexpr bool lds = true;
    double *base; /* &sD[0][lane] */
    __device__ __forceinline__ double get(int c) const { return base[c * 64]; }
    __device__ __forceinline__ void set(int c, double x) { base[c * 64] = x; }
};


/* one forward row (i >= 2), in place: on entry fM,fI,fD = scaled row i-1; on exit scaled row i.
 * Returns the row sum s[i]. */
template <int G, int C, bool FAST, int W0, class DS>
__device__ __forceinline__ double fwd_row(double (&fM)[C], double (&fI)[C], DS &D, double dinv, const CodeWin<C> &ew,
                                          uint32_t qy, const HmmC &h, int g, int Wu, int tlast, int t_first, int t_stop,
                                          double &inv_out)
{
    double nM = shfl_down1<G>(fM[0]), nI = shfl_down1<G>(fI[0]);
    if (g == G - 1) { nM = 0.0; nI = 0.0; }
    /* masked rows, narrow lanes: the validity of a slot is folded into the two inputs of its D recurrence in the
     * parallel phase (m2*M(i,k-1) -> 0, m8 -> 0), so the serial passes -- executed by the whole wave for one lane's
     * benefit -- need no selects at all */
    constexpr bool MASKC = !FAST && !DS::lds && C <= 16;
    double m8v[MASKC ? C : 1];
    const int nc_own = g < tlast ? C : (g == tlast ? (W0 ? W0 - ((W0 - 1) / C) * C : Wu - tlast * C) : 0);
    /* parallel phase: fM<-M(i,.), fI<-I(i,.), fD<-m2*M(i,k-1) */
    double prevM;
    {
        const double pD0 = DS::lds ? D.get(0) * dinv : D.get(0);
        const double S0 = (h.m0 * fM[0] + h.m3 * fI[0]) + h.m6 * pD0;
        const double e0 = emission<FAST>(ew.get(0), qy, h.e_match, h.e_mis);
        const double pMn = C > 1 ? fM[1] : nM, pIn = C > 1 ? fI[1] : nI;
        prevM = e0 * S0;
        fI[0] = SPX_EI * (h.m1 * pMn + h.m4 * pIn);
        fM[0] = prevM;
    }
#pragma unroll
    for (int c = 1; c < C; ++c) {
        const double pD = DS::lds ? D.get(c) * dinv : D.get(c);
        const double S = (h.m0 * fM[c] + h.m3 * fI[c]) + h.m6 * pD;
        const double e = emission<FAST>(ew.get(c), qy, h.e_match, h.e_mis);
        const double pMn = (c + 1 < C) ? fM[c + 1] : nM, pIn = (c + 1 < C) ? fI[c + 1] : nI;
        const double newM = e * S;
        fI[c] = SPX_EI * (h.m1 * pMn + h.m4 * pIn);
        if constexpr (MASKC) {
            const bool vld = c < nc_own && !(ew.get(c) & SPX_CODE_OUT);
            D.set(c, vld ? h.m2 * prevM : 0.0);
            m8v[c] = vld ? h.m8 : 0.0;
        } else
            D.set(c, h.m2 * prevM);
        fM[c] = newM;
        prevM = newM;
    }
    {
        double pl = shfl_up1<G>(prevM);
        if (g == 0 || g > tlast) pl = 0.0; /* lanes beyond the band keep exact zeros */
        if constexpr (MASKC) {
            const bool vld = 0 < nc_own && !(ew.get(0) & SPX_CODE_OUT);
            D.set(0, vld ? h.m2 * pl : 0.0);
            m8v[0] = vld ? h.m8 : 0.0;
        } else
            D.set(0, h.m2 * pl);
    }
    /* serial phase: D recurrence and row sum in column order, one lane of the group at a time.
     * Straight-line code only (uniform selects, no branches inside the unrolled register arrays). */
    /* Lanes whose slots all lie left of column 1 (first rows: t < t_first) hold exact zeros and are skipped; so are,
     * on masked rows, the lanes right of column R for every problem of the wave (t > t_stop): their recurrence
     * inputs were zeroed above.  Both bounds are wave-uniform. */
    const int t_end = MASKC ? min(t_stop, tlast) : tlast;
    double carryD = 0.0, carryS = 0.0, mysum = 0.0;
    for (int t = t_first; t <= t_end; ++t) {
        if (g == t) {
            double d = carryD, s = carryS;
            /* LDS mode: fetch the whole m2*M(i,k-1) row of this lane first (independent reads, pipelined),
             * so that the recurrence below never waits on the LDS */
            double av[DS::lds ? C : 1];
            if constexpr (DS::lds) {
#pragma unroll
                for (int c = 0; c < C; ++c) av[c] = D.get(c);
            }
            auto Aget = [&](int c) { if constexpr (DS::lds) return av[c]; else return D.get(c); };
            if constexpr (MASKC) {
#pragma unroll
                for (int c = 0; c < C; ++c) { /* a slot that is not a band cell: 0 + 0*d = 0, and it adds (0+0)+0 to the sum */
                    d = Aget(c) + m8v[c] * d;
                    D.set(c, d);
                    s = s + ((fM[c] + fI[c]) + d);
                }
            } else if (FAST && t < tlast) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    d = Aget(c) + h.m8 * d;
                    D.set(c, d);
                    s = s + ((fM[c] + fI[c]) + d);
                }
            } else {
                /* W0 != 0: the band width is a compile-time constant of this instantiation */
                const int nc = W0 ? (t < (W0 - 1) / C ? C : W0 - ((W0 - 1) / C) * C) : min(C, Wu - t * C);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    /* slots beyond the band (c >= nc) keep D = 0: it feeds M of that slot in the next row */
                    const bool valid = (c < nc) && (FAST || !(ew.get(c) & SPX_CODE_OUT));
                    const double dn = Aget(c) + h.m8 * d;
                    d = valid ? dn : 0.0;
                    D.set(c, d);
                    /* no select on the sum: a slot that is not a band cell holds M = I = 0 (emission 0 / zero inputs) and
                     * its D was just forced to 0, so it adds exactly +0 */
                    s = s + ((fM[c] + fI[c]) + d);
                }
            }
            carryD = d; carryS = s; mysum = s;
        }
        if (t < t_end) {
            carryD = shfl_up1<G>(carryD);
            carryS = shfl_up1<G>(carryS);
        }
    }
    const double tot = __shfl(mysum, t_end, G);
    const double inv = 1.0 / tot;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        fM[c] *= inv; fI[c] *= inv;
        if constexpr (!DS::lds) D.v[c] *= inv;
    }
    inv_out = inv;
    return tot;
}

/* one backward row (1 <= i <= L-1), in place: on entry bM,bI = scaled row i+1; on exit scaled row i.
 * ew holds the code of column k+1 (ref index i - bw + j) per slot. */
template <int G, int C, bool FAST, int W0, class DS>
__device__ __forceinline__ void bwd_row(double (&bM)[C], double (&bI)[C], DS &D, const CodeWin<C> &ew,
                                        uint32_t qy, const HmmC &h, int g, int Wu, int tlast, double inv, bool first_row, bool any_first)
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
        D.set(c, e * h.m6);
        bM[c] = u;
        bI[c] = v;
    }
    /* serial phase: D recurrence over descending columns; row 1 has y = 0 (its D row is all zero) */
    {
        double carryD = 0.0;
        for (int t = tlast; t >= 0; --t) {
            if (g == t) {
                double d = carryD;
                double xv[DS::lds ? C : 1];
                if constexpr (DS::lds) {
#pragma unroll
                    for (int c = 0; c < C; ++c) xv[c] = D.get(c);
                }
                auto Xget = [&](int c) { if constexpr (DS::lds) return xv[c]; else return D.get(c); };
                if (t < tlast) {
#pragma unroll
                    for (int c = C - 1; c >= 0; --c) {
                        d = Xget(c) + h.m8 * d;
                        D.set(c, d);
                    }
                } else {
                    const int nc = W0 ? (W0 - ((W0 - 1) / C) * C) : min(C, Wu - t * C);
#pragma unroll
                    for (int c = C - 1; c >= 0; --c) {
                        const double x = (c < nc) ? Xget(c) : 0.0; /* D of a column that does not exist stays 0 */
                        d = x + h.m8 * d;
                        D.set(c, d);
                    }
                }
                carryD = d;
            }
            if (t > 0) carryD = shfl_down1<G>(carryD);
        }
        if (any_first) { /* wave-uniform: some problem of the wave is on its row 1 */
            /* the reference MULTIPLIES the D values of row 1 by y = 0 (probaln.c: `(e*m[6] + m[8]*bi[v01+2]) * y`): where the
             * backward values have overflowed -- unrelated sequences over a thousand rows -- that is inf * 0 = NaN, not 0,
             * and the NaN reaches M(1,k) and the MAP state of row 1 (found by the GPU fuzz: state 0 here, 1 there) */
            const double y = first_row ? 0.0 : 1.0;
#pragma unroll
            for (int c = 0; c < C; ++c) D.set(c, D.get(c) * y);
        }
    }
    /* parallel phase B: M += m2*D(i,k+1); scale */
    double hD = shfl_down1<G>(D.get(0));
    if (g == G - 1 || g >= tlast) hD = 0.0; /* the slot above the band has D = 0 */
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double dn = (c + 1 < C) ? D.get(c + 1) : hD; /* slots beyond the band hold D = 0 (see below) */
        bM[c] = (bM[c] + h.m2 * dn) * inv;
        bI[c] = bI[c] * inv;
    }
}


#ifndef SPX_WAVES_F
#define SPX_WAVES_F 2
#endif
#ifndef SPX_WAVES_B
#define SPX_WAVES_B 2
#endif

/* ====================================================================== */
/* forward pass: rows 1..L, saves 1/s[i] (i < L), s[L], s[L+1] and the scaled M,I rows at the wanted rows */
template <int G, int C, int W0, bool LDSD>
__global__ __launch_bounds__(64, SPX_WAVES_F) void baq_fwd_kernel(spx_dev_batch B)
{
    __shared__ double sD[LDSD ? C : 1][64];
    constexpr int SLOTS = G * C;
    const int lane = threadIdx.x & 63;
    const int g = lane % G;
    HmmC h;
    int hasN;
    const Prob P = load_problem<G>(B, lane, h, hasN);
    const bool act = P.act;
    const int L = P.L, R = P.R, bw = P.bw;
    const int Wu = W0 ? W0 : wave_max(act ? 2 * bw + 1 : 0);
    const int Lw = wave_max(L);
    if (Lw == 0) return;
    /* forward row i is interior iff i + bw <= R, for every problem of the wave, and no N anywhere */
    const int anyN = wave_max(hasN);
    const int fast_end = anyN ? 1 : min(wave_min(act ? R - bw : 0x7fffffff), Lw);
    const int tlast = (Wu - 1) / C;
    const int jbase = g * C;
    const int bwu = (Wu - 1) / 2, Rmax = wave_max(act ? R : 0);

    double fM[C], fI[C];
    typename std::conditional<LDSD, DLds<C>, DRegs<C>>::type D;
    if constexpr (LDSD) D.base = &sD[0][lane];
    double dinv = 1.0; /* LDS mode: factor still to be applied to the stored D row */
    CodeWin<C> cw, padw; /* padw: SPX_CODE_OUT in the slots beyond the band (j >= W), fixed per problem */
    double *sinv = B.sinv + (act ? B.s_off[P.pid] : 0);
    double *fsave = B.fsave + (act ? B.fsave_off[P.pid] : 0);
    const int64_t fstride = B.fsave_stride;
    const int nrows = P.nrows, row0 = P.row0;

    double s_cur = 1.0;
    /* 1/s[i] of eight rows is written as one 64-byte line (see baq_fwd1_kernel) -- where 16 VGPRs can be spared */
    constexpr bool STAGE = C <= 16;
    double ib0 = 0, ib1 = 0, ib2 = 0, ib3 = 0, ib4 = 0, ib5 = 0, ib6 = 0, ib7 = 0;
    auto put_inv = [&](double v) { ib0 = ib1; ib1 = ib2; ib2 = ib3; ib3 = ib4; ib4 = ib5; ib5 = ib6; ib6 = ib7; ib7 = v; };
    auto flush_inv = [&](int i) { /* rows i-7 .. i; rows < 0 of a short query land in the problem's lead pad */
        double2 *dst = reinterpret_cast<double2 *>(sinv + (i - 7));
        dst[0] = make_double2(ib0, ib1); dst[1] = make_double2(ib2, ib3);
        dst[2] = make_double2(ib4, ib5); dst[3] = make_double2(ib6, ib7);
    };
    int wnext = 0;
    int next_row = nrows > 0 ? B.rows[row0] : 0x7fffffff;
    auto save_row = [&]() {
        double *dst = fsave + (int64_t)wnext * fstride + jbase;
#pragma unroll
        for (int c = 0; c < C; ++c) { dst[c] = fM[c]; dst[SLOTS + c] = fI[c]; }
        wnext++;
        next_row = wnext < nrows ? B.rows[row0 + wnext] : 0x7fffffff;
    };
    /* row 1: f(1,k) = e*bM, EI*bI for k in [1, min(R, bw+1)], divided by the row sum */
    {
        const double bM = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_BM] : 0.0;
        const double bI = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_BI] : 0.0;
        const uint32_t qy = act ? fetch_code(B.qry4, P.qry0, 0, L) : 0;
#pragma unroll
        for (int k = 0; k < CodeWin<C>::NW; ++k) { cw.w[k] = 0; padw.w[k] = 0; }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c; /* slot j <-> ref idx 1 - bw + j - 1 */
            cw.set(c, act ? fetch_code(B.ref4, P.ref0, j - bw, R) : SPX_CODE_OUT);
            padw.set(c, j < Wu ? 0u : (uint32_t)SPX_CODE_OUT);
        }
        CodeWin<C> ew;
#pragma unroll
        for (int k = 0; k < CodeWin<C>::NW; ++k) ew.w[k] = cw.w[k] | padw.w[k];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint32_t code = ew.get(c);
            const bool valid = !(code & SPX_CODE_OUT);
            const double e = emission<false>(code, qy, h.e_match, h.e_mis);
            fM[c] = valid ? e * bM : 0.0;
            fI[c] = valid ? SPX_EI * bI : 0.0;
            D.set(c, 0.0);
        }
        double carry = 0.0, mysum = 0.0;
        for (int t = 0; t <= tlast; ++t) {
            if (g == t) {
                double s = carry;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const bool valid = !(ew.get(c) & SPX_CODE_OUT);
                    const double tt = fM[c] + fI[c];
                    s = valid ? s + tt : s;
                }
                mysum = s;
                carry = s;
            }
            carry = shfl_up1<G>(carry);
        }
        const double tot = __shfl(mysum, tlast, G);
        s_cur = tot;
        if (act) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                fM[c] = fM[c] / tot;
                fI[c] = fI[c] / tot;
            }
            if constexpr (STAGE) {
                put_inv(1.0 / tot);
                if (g == 0 && L == 1) flush_inv(1);
            } else if (g == 0)
                sinv[1] = 1.0 / tot;
            if (B.s_raw && g == 0) B.s_raw[(sinv - B.sinv) + 1] = tot;
            if (next_row == 1) save_row();
        }
    }
    /* rows 2..L: an interior stretch without masks, then the rows whose band touches column R */
    /* codes arrive 8 rows at a time (one dword of query codes, one funnel-shifted dword of reference codes
     * for the slot entering the band at the top), fetched one chunk ahead */
    const int top = jbase + C - 1; /* the slot that receives a new column each row */
    auto ref_chunk = [&](int ib) { return fetch8(B.ref4, P.ref0 + (ib - bw + top - 1)); };   /* rows ib..ib+7 */
    auto qry_chunk = [&](int ib) { return fetch8(B.qry4, P.qry0 + (ib - 1)); }; /* windows start anywhere inside a recoded read */
    uint32_t qwin = act ? qry_chunk(1) : 0, rwin = act ? ref_chunk(1) : 0;
    uint32_t qwin_n = act ? qry_chunk(9) : 0, rwin_n = act ? ref_chunk(9) : 0;
    auto row = [&](int i, auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        if (act && i <= L) {
            const uint32_t t4 = (uint32_t)((i - 1) & 7) * 4u;
            if (t4 == 0) {
                qwin = qwin_n; rwin = rwin_n;
                qwin_n = qry_chunk(i + 8); rwin_n = ref_chunk(i + 8);
            }
            const uint32_t qy = (qwin >> t4) & 0xfu;
            uint32_t rc = (rwin >> t4) & 0xfu;
            /* also on interior rows: the last lane's top slot lies (slots - W) columns beyond the band, so the
             * column it receives can be > R while i + bw <= R still holds; that code slides into a real band
             * slot a few rows later, when validity is read off the code */
            if ((unsigned)(i - bw + top - 1) >= (unsigned)R) rc = SPX_CODE_OUT;
            cw.shift_down(rc);
            CodeWin<C> ew;
#pragma unroll
            for (int k = 0; k < CodeWin<C>::NW; ++k) ew.w[k] = cw.w[k] | padw.w[k];
            double inv;
            /* wave-uniform lane range that can hold band cells on this row: columns 1 .. max R of the wave */
            const int t_first = max(0, bwu + 1 - i) / C, t_stop = max(0, Rmax - i + bwu) / C;
            s_cur = fwd_row<G, C, FAST, W0>(fM, fI, D, dinv, ew, qy, h, g, Wu, tlast, t_first, t_stop, inv);
            dinv = inv;
            if constexpr (STAGE) {
                put_inv(inv);
                if (g == 0 && ((i & 7) == 7 || i == L)) flush_inv(i);
            } else if (g == 0)
                sinv[i] = inv;
            if (B.s_raw && g == 0) B.s_raw[(sinv - B.sinv) + i] = s_cur;
            if (i == next_row) save_row();
        }
    };
    int i = 2;
    for (; i <= fast_end; ++i) row(i, std::true_type{});
    for (; i <= Lw; ++i) row(i, std::false_type{});
    /* terminal: s[L+1] = sum_k f(L,k).M*sM + f(L,k).I*sI in column order */
    {
        const double sM = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_SM] : 0.0;
        const double sI = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_SI] : 0.0;
        double carry = 0.0, mysum = 0.0;
        for (int t = 0; t <= tlast; ++t) {
            const int nc = min(C, Wu - t * C);
            if (g == t) {
                double s = carry;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int k = L - bw + jbase + c;
                    const bool valid = c < nc && k >= 1 && k <= P.Rt;
                    const double tt = fM[c] * sM + fI[c] * sI;
                    s = valid ? s + tt : s;
                }
                carry = s; mysum = s;
            }
            carry = shfl_up1<G>(carry);
        }
        const double sL1 = __shfl(mysum, tlast, G);
        if (act && g == 0) { sinv[L] = s_cur; sinv[L + 1] = sL1; } /* raw s[L], s[L+1] for the backward start */
    }
}

/* ====================================================================== */
/* backward pass: rows L..(first wanted row), MAP + phred + write-back rule at the wanted rows.
 * Rows below the first wanted row have no observable effect and are not computed. */
template <int G, int C, int W0, bool LDSD>
__global__ __launch_bounds__(64, SPX_WAVES_B) void baq_bwd_kernel(spx_dev_batch B)
{
    __shared__ double sD[LDSD ? C : 1][64];
    constexpr int SLOTS = G * C;
    const int lane = threadIdx.x & 63;
    const int g = lane % G;
    HmmC h;
    int hasN;
    const Prob P = load_problem<G>(B, lane, h, hasN, true);
    const bool act = P.act;
    const int L = P.L, R = P.R, bw = P.bw;
    const int Wu = W0 ? W0 : wave_max(act ? 2 * bw + 1 : 0);
    const int Lw = wave_max(L);
    if (Lw == 0) return;
    const int nrows = P.nrows, row0 = P.row0;
    const int stop = act ? B.rows[row0] : 0x7fffffff; /* first (smallest) wanted row */
    const int anyN = wave_max(hasN);
    const int tlast = (Wu - 1) / C;
    const int jbase = g * C;

    double bM[C], bI[C];
    typename std::conditional<LDSD, DLds<C>, DRegs<C>>::type D;
    if constexpr (LDSD) D.base = &sD[0][lane];
    CodeWin<C> cw, padw;
    const double *sinv = B.sinv + (act ? B.s_off[P.pid] : 0);
    const int64_t fstride = B.fsave_stride;
    /* row L */
    {
        const double sM = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_SM] : 0.0;
        const double sI = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_SI] : 0.0;
        const double sL = act ? sinv[L] : 1.0, sL1 = act ? sinv[L + 1] : 1.0;
        const double vM = (sM / sL) / sL1, vI = (sI / sL) / sL1;
#pragma unroll
        for (int k = 0; k < CodeWin<C>::NW; ++k) { cw.w[k] = 0; padw.w[k] = 0; }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c, k = L - bw + j;
            const bool valid = act && j < Wu && k >= 1 && k <= P.Rt;
            bM[c] = valid ? vM : 0.0;
            bI[c] = valid ? vI : 0.0;
            D.set(c, 0.0);
            padw.set(c, j < Wu ? 0u : (uint32_t)SPX_CODE_OUT);
            /* window for row L-1: code of ref idx (L-1) - bw + j (= column k+1 of that row) */
            cw.set(c, (act && L >= 2) ? fetch_code(B.ref4, P.ref0, (L - 1) - bw + j, R) : SPX_CODE_OUT);
        }
    }
    /* at a wanted row the saved forward row is replaced in place by z = f*b (M and I states); map_kernel
     * finishes the job (arg-max, ordered sum, phred, write-back rule) */
    double *fsave = B.fsave + (act ? B.fsave_off[P.pid] : 0);
    int wprev = nrows - 1;
    int prev_row = wprev >= 0 ? B.rows[row0 + wprev] : -1;
    /* exact classes (W0 != 0): the one-lane forward kernel saved row i >= 2 unscaled, so z = (f * 1/s[i]) * b */
    auto save_row = [&](double fscale) {
        double *dst = fsave + (int64_t)wprev * fstride + jbase;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if constexpr (W0 != 0) { dst[c] = (dst[c] * fscale) * bM[c]; dst[SLOTS + c] = (dst[SLOTS + c] * fscale) * bI[c]; }
            else { dst[c] = dst[c] * bM[c]; dst[SLOTS + c] = dst[SLOTS + c] * bI[c]; }
        }
        wprev--;
        prev_row = wprev >= 0 ? B.rows[row0 + wprev] : -1;
    };
    if (act && prev_row == L) save_row(L == 1 ? 1.0 : 1.0 / sinv[L]); /* sinv[L] holds s[L] itself */
    /* row i uses query idx i and lets ref idx i - bw + jbase enter at slot 0.  Codes come 8 STEPS at a time
     * (step t = row L-1-t, so the chunk phase is the same for every problem of the wave), one chunk ahead:
     * the chunk of steps t0..t0+7 holds rows i0-7..i0 (i0 = L-1-t0) in ascending nibble order */
    auto ref_chunk = [&](int i0) { return fetch8(B.ref4, P.ref0 + ((i0 - 7) - bw + jbase)); };
    auto qry_chunk = [&](int i0) { return fetch8(B.qry4, P.qry0 + (i0 - 7)); };
    uint32_t qwin = 0, rwin = 0;
    uint32_t qwin_n = (act && L >= 2) ? qry_chunk(L - 1) : 0, rwin_n = (act && L >= 2) ? ref_chunk(L - 1) : 0;
    double inv_p = (act && L >= 2) ? sinv[L - 1] : 0.0;
    /* every problem walks its own rows L-1, L-2, ... down to its first wanted row: step t of the wave is row
     * L-1-t of each problem, so problems of different length stay busy together (the launch order groups
     * problems by the number of rows they walk) and the masked general path is needed only for the first
     * steps, where the band still touches column R */
    const int nb = act ? max(L - stop, 0) : 0;
    const int nbw = wave_max(nb);
    const int n_slow = anyN ? nbw : min(nbw, wave_max(act ? min(nb, max(0, (L - 1) - (R - bw - 1))) : 0));
    auto row = [&](int t, auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int i = L - 1 - t;
        const bool on = act && t < nb;
        const bool any_first = __any(on && i == 1);
        if (on) {
            const uint32_t t4 = (uint32_t)(7 - (t & 7)) * 4u; /* wave-uniform */
            if ((t & 7) == 0) {
                qwin = qwin_n; rwin = rwin_n;
                qwin_n = qry_chunk(i - 8); /* rows below 1 read the lead pad: never used */
                rwin_n = ref_chunk(i - 8);
            }
            const uint32_t qy = (qwin >> t4) & 0xfu;
            const double inv = inv_p;
            if (t != 0) {
                uint32_t rc = (rwin >> t4) & 0xfu;
                if (!FAST) {
                    if ((unsigned)(i - bw + jbase) >= (unsigned)R) rc = SPX_CODE_OUT;
                }
                cw.shift_up(rc);
            }
            CodeWin<C> ew;
#pragma unroll
            for (int k = 0; k < CodeWin<C>::NW; ++k) ew.w[k] = cw.w[k] | padw.w[k];
            if (i >= 2) inv_p = sinv[i - 1]; /* prefetch for row i-1 */
            bwd_row<G, C, FAST, W0>(bM, bI, D, ew, qy, h, g, Wu, tlast, inv, i == 1, any_first);
            if (i == prev_row) save_row(i == 1 ? 1.0 : inv);
        }
    };
    int t = 0;
    for (; t < n_slow; ++t) row(t, std::false_type{});
    for (; t < nbw; ++t) row(t, std::true_type{});
}

/* ====================================================================== */
/* One lane per problem (G = 1) for the exact band width W = C: no serial passes with idle lanes.
 * Forward: M and I of the previous row stay in VGPRs UNSCALED (the scale factor is applied when a value is
 * read, which is the same multiplication the reference does when it rescales the row); the D row lives in
 * LDS ([slot][lane], conflict free).  (A one-lane backward kernel was measured too: slower than two lanes per
 * problem, because the wanted rows of 64 problems diverge; it is not kept.) */

template <int C>
__global__ __launch_bounds__(64, 2) void baq_fwd1_kernel(spx_dev_batch B)
{
    /* D row of the previous/current row, [slot][lane].  Slot 0 is the leftmost band cell: its D is
     * m2*M(k-1) + m8*D(k-1) with both neighbours outside the band, i.e. always +0 -- it is not stored.  At most 40
     * slots go to LDS (40 x 512 B = 20 KB per wave: 8 waves per CU fit the 160 KB instead of 7); the first CR slots
     * of the wider classes stay in VGPRs (same-box A/B: +0.6 %) */
    constexpr int CR = (C - 1 > 40) ? C - 1 - 40 : 0;
    __shared__ double sD[C - 1 - CR][64];
    double dR[CR > 0 ? CR : 1];
    const int lane = threadIdx.x & 63;
    HmmC h;
    int hasN;
    const Prob P = load_problem<1>(B, lane, h, hasN);
    const bool act = P.act;
    const int L = P.L, R = P.R, bw = P.bw;
    const int Lw = wave_max(L);
    if (Lw == 0) return;
    const int anyN = wave_max(hasN);
    const int fast_end = anyN ? 1 : min(wave_min(act ? R - bw : 0x7fffffff), Lw);
    double fM[C], fI[C];
    NibWin<C> cw;
    double *sinv = B.sinv + (act ? B.s_off[P.pid] : 0);
    double *fsave = B.fsave + (act ? B.fsave_off[P.pid] : 0);
    const int64_t fstride = B.fsave_stride;
    const int SLOTS = (int)(fstride >> 1);
    const int nrows = P.nrows, row0 = P.row0;
    int wnext = 0;
    int next_row = nrows > 0 ? B.rows[row0] : 0x7fffffff;
    double inv_prev = 1.0, s_cur = 1.0;
    /* 1/s[i] is collected for eight rows and written as one aligned 64-byte line per problem: a lane-strided
     * 8-byte store per row costs a partial-line write (and the read that fills the line) every time */
    double ib0 = 0, ib1 = 0, ib2 = 0, ib3 = 0, ib4 = 0, ib5 = 0, ib6 = 0, ib7 = 0;
    auto put_inv = [&](double v) { /* straight-line shift: ib7 is the newest row */
        ib0 = ib1; ib1 = ib2; ib2 = ib3; ib3 = ib4; ib4 = ib5; ib5 = ib6; ib6 = ib7; ib7 = v;
    };
    auto flush_inv = [&](int i) { /* rows i-7 .. i (a lead pad of one line per problem takes rows < 0 of a short query) */
        double2 *dst = reinterpret_cast<double2 *>(sinv + (i - 7));
        dst[0] = make_double2(ib0, ib1); dst[1] = make_double2(ib2, ib3);
        dst[2] = make_double2(ib4, ib5); dst[3] = make_double2(ib6, ib7);
    };
    /* rows >= 2 are saved UNSCALED (row 1 is stored already divided): the backward kernel of the exact classes applies
     * 1/s[i] before it multiplies by b -- the same two multiplications, 82 of them moved out of this kernel, where
     * every row pays them for the one or two lanes that save (same-box A/B: +0.6 %) */
    auto save_row = [&]() {
        double *dst = fsave + (int64_t)wnext * fstride;
#pragma unroll
        for (int c = 0; c + 1 < C; c += 2) { /* 16-byte stores (rows start on 16-byte boundaries, SLOTS is even): +1.2 % in a same-box A/B */
            *reinterpret_cast<double2 *>(dst + c) = make_double2(fM[c], fM[c + 1]);
            *reinterpret_cast<double2 *>(dst + SLOTS + c) = make_double2(fI[c], fI[c + 1]);
        }
        if (C & 1) { dst[C - 1] = fM[C - 1]; dst[SLOTS + C - 1] = fI[C - 1]; }
        wnext++;
        next_row = wnext < nrows ? B.rows[row0 + wnext] : 0x7fffffff;
    };
    /* row 1 */
    {
        const double bM = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_BM] : 0.0;
        const double bI = act ? B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_BI] : 0.0;
        const uint32_t qy = act ? fetch_code(B.qry4, P.qry0, 0, L) : 0;
#pragma unroll
        for (int k = 0; k < NibWin<C>::NW; ++k) cw.w[k] = 0;
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint32_t code = act ? fetch_code(B.ref4, P.ref0, c - bw, R) : SPX_CODE_OUT;
            cw.set(c, code);
            const bool valid = !(code & SPX_CODE_OUT);
            const double e = emission<false>(code, qy, h.e_match, h.e_mis);
            fM[c] = valid ? e * bM : 0.0;
            fI[c] = valid ? SPX_EI * bI : 0.0;
            if (c > 0) { if (c - 1 < CR) dR[c - 1 < CR ? c - 1 : 0] = 0.0; else sD[c - 1 - CR][lane] = 0.0; }
            const double tt = fM[c] + fI[c];
            s = valid ? s + tt : s;
        }
        s_cur = s;
        if (act) {
#pragma unroll
            for (int c = 0; c < C; ++c) { fM[c] = fM[c] / s; fI[c] = fI[c] / s; }
            put_inv(1.0 / s);
            if (L == 1) flush_inv(1);
            inv_prev = 1.0; /* row 1 is stored already divided, as the reference does */
            if (B.s_raw) B.s_raw[(sinv - B.sinv) + 1] = s;
            if (next_row == 1) save_row();
        }
    }
    auto ref_chunk = [&](int ib) { return fetch8(B.ref4, P.ref0 + (ib - bw + (C - 1) - 1)); };
    auto qry_chunk = [&](int ib) { return fetch8(B.qry4, P.qry0 + (ib - 1)); }; /* windows start anywhere inside a recoded read */
    uint32_t qwin = act ? qry_chunk(1) : 0, rwin = act ? ref_chunk(1) : 0;
    uint32_t qwin_n = act ? qry_chunk(9) : 0, rwin_n = act ? ref_chunk(9) : 0;
    auto row = [&](int i, auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        if (act && i <= L) {
            const uint32_t t4 = (uint32_t)((i - 1) & 7) * 4u;
            if (t4 == 0) {
                qwin = qwin_n; rwin = rwin_n;
                qwin_n = qry_chunk(i + 8); rwin_n = ref_chunk(i + 8);
            }
            const uint32_t qy = (qwin >> t4) & 0xfu;
            uint32_t rc = (rwin >> t4) & 0xfu;
            if (!FAST) {
                if ((unsigned)(i - bw + (C - 1) - 1) >= (unsigned)R) rc = SPX_CODE_OUT;
            }
            cw.shift_down(rc);
            /* interior rows: "reference base == query base" for the eight codes of a window word at once (exact
             * zero-nibble test), then a bitwise select of the emission: 1.5 ops per cell instead of 5 */
            uint32_t eq[NibWin<C>::NW];
            if constexpr (FAST) {
                const uint32_t qrep = qy * 0x11111111u;
#pragma unroll
                for (int k = 0; k < NibWin<C>::NW; ++k) {
                    const uint32_t x = cw.w[k] ^ qrep;
                    eq[k] = ~(((x & 0x77777777u) + 0x77777777u) | x) & 0x88888888u;
                }
            }
            const double ip = inv_prev;
            double pM = fM[0] * ip, pI = fI[0] * ip; /* scaled row i-1 at the current slot */
            double d = 0.0, s = 0.0, prevM = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                double S = h.m0 * pM + h.m3 * pI;
                if (c > 0) /* slot 0: + m6 * 0 */
                    S = S + h.m6 * ((c - 1 < CR ? dR[c - 1 < CR ? c - 1 : 0] : sD[c - 1 < CR ? 0 : c - 1 - CR][lane]) * ip);
                const uint32_t code = cw.get(c);
                double e;
                if constexpr (FAST) {
                    const int32_t m = (int32_t)(eq[c >> 3] << (28 - 4 * (c & 7))) >> 31; /* all ones on a match */
                    e = select_bits(m, h.e_match, h.e_mis);
                } else
                    e = emission<false>(code, qy, h.e_match, h.e_mis);
                const double newM = e * S;
                double nM = 0.0, nI = 0.0; /* scaled row i-1 at slot c+1 (the slot above the band is empty) */
                if (c + 1 < C) { nM = fM[c + 1] * ip; nI = fI[c + 1] * ip; }
                const double newI = SPX_EI * (h.m1 * nM + h.m4 * nI);
                const bool valid = FAST || !(code & SPX_CODE_OUT);
                if (c > 0) {
                    const double dn = h.m2 * prevM + h.m8 * d;
                    d = valid ? dn : 0.0;
                    if (c - 1 < CR) dR[c - 1 < CR ? c - 1 : 0] = d; else sD[c - 1 - CR][lane] = d;
                }
                const double tt = c > 0 ? (newM + newI) + d : newM + newI; /* slot 0: + 0 */
                s = valid ? s + tt : s;
                fM[c] = newM; fI[c] = newI;
                prevM = newM;
                pM = nM; pI = nI;
            }
            const double inv = 1.0 / s;
            s_cur = s;
            inv_prev = inv;
            put_inv(inv);
            if ((i & 7) == 7 || i == L) flush_inv(i);
            if (B.s_raw) B.s_raw[(sinv - B.sinv) + i] = s;
            if (i == next_row) save_row();
        }
    };
    int i = 2;
    for (; i <= fast_end; ++i) row(i, std::true_type{});
    for (; i <= Lw; ++i) row(i, std::false_type{});
    /* terminal: s[L+1] over the scaled row L */
    if (act) {
        const double sM = B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_SM], sI = B.hmm[(int64_t)P.pid * SPX_H_N + SPX_H_SI];
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int k = L - bw + c;
            const double tt = (fM[c] * inv_prev) * sM + (fI[c] * inv_prev) * sI;
            s = (k >= 1 && k <= P.Rt) ? s + tt : s;
        }
        sinv[L] = s_cur;
        sinv[L + 1] = s;
    }
}

/* ====================================================================== */
/* MAP + phred + write-back rule, one lane per wanted row.  z = f*b over the M and I states of the
 * row in column order: first strictly greatest wins, the sum is sequential (probaln_glocal's MAP loop);
 * then min(raw, q) with the CIGAR/MAP consistency check (ptMarker.c:778-779,786). */
/* CQMAX: most slots per lane this instantiation keeps in registers (12: 61 VGPRs, 8 waves per SIMD; 32: 150 VGPRs) */
template <int CQMAX, int LPR>
__global__ __launch_bounds__(256) void map_kernel(spx_dev_batch B, int32_t n_rows_total)
{
    /* LPR adjacent lanes per wanted row (4 for the narrow bands, 8 for the wide ones), each owning a contiguous
     * share of the slots (coalesced reads of the 2 x slots doubles a row holds); products and the running argmax
     * are lane-local, the sum is passed from lane to lane in column order */
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = B.row_base + (int)(tid / LPR), g = (int)(tid & (LPR - 1));
    bool on = r < B.row_base + n_rows_total;
    const int rr = on ? r : B.row_base;
    const int p = B.row_prob[rr];
    if (B.tier && B.tier_want != SPX_TIER_ALL && B.tier[p] != B.tier_want) on = false; /* two-tier DP: another pass' row */
    const int i = B.rows[rr], bw = B.bw[p], R = B.R[p];
    const int W = 2 * bw + 1, slots = B.prob_slots[p], Cq = (slots + LPR - 1) / LPR;
    const int64_t off = B.fsave_off[p] + (int64_t)(rr - B.row_off[p]) * 2 * slots;
    const double *zM = B.fsave + off, *zI = zM + slots; /* z = f*b, written by the backward kernel */
    const int j0 = max(0, bw + 1 - i), j1 = min(W - 1, R - i + bw); /* 1 <= k = i - bw + j <= R */
    const int ja = max(j0, g * Cq), jb = min(j1, g * Cq + Cq - 1);
    double best = 0.0, carry = 0.0, mysum = 0.0;
    int best_k = -1;
    /* narrow and medium bands: all loads of a row are issued before the first add, so a wave pays one memory round
     * trip instead of one per slot; slots outside [ja,jb] read as +0.0, which changes neither the sequential sum nor
     * the strict arg-max.  CQ = slots per lane held in registers; wider rows take the loop below */
    auto batched = [&](auto cq_tag) {
        constexpr int CQ = decltype(cq_tag)::value;
        double zm[CQ], zi[CQ];
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            const int j = g * Cq + c;
            const bool in = on && c < Cq && j >= ja && j <= jb;
            zm[c] = in ? zM[j] : 0.0;
            zi[c] = in ? zI[j] : 0.0;
        }
        const int kbase = i - bw + g * Cq;
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            if (zm[c] > best) { best = zm[c]; best_k = ((kbase + c - 1) << 2) | 0; }
            if (zi[c] > best) { best = zi[c]; best_k = ((kbase + c - 1) << 2) | 1; }
        }
        for (int t = 0; t < LPR; ++t) {
            double sacc = carry;
#pragma unroll
            for (int c = 0; c < CQ; ++c) { sacc += zm[c]; sacc += zi[c]; }
            if (g == t) mysum = sacc;
            carry = __shfl_up(mysum, 1, LPR);
        }
    };
    const int cq_wave = wave_max(on ? Cq : 0);
    if (cq_wave <= CQMAX) batched(std::integral_constant<int, CQMAX>{});
    else
    for (int t = 0; t < LPR; ++t) {
        if (g == t && on) {
            double s = carry;
            for (int j = ja; j <= jb; ++j) {
                const int k = i - bw + j;
                double z = zM[j];
                if (z > best) { best = z; best_k = ((k - 1) << 2) | 0; }
                s += z;
                z = zI[j];
                if (z > best) { best = z; best_k = ((k - 1) << 2) | 1; }
                s += z;
            }
            mysum = s;
        }
        carry = __shfl_up(mysum, 1, LPR); /* lane t+1 continues from lane t's partial sum */
    }
    const double sum = __shfl(mysum, LPR - 1, LPR);
    /* first strictly greatest in column order: the lower lane wins ties */
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
        const double ob = __shfl_up(best, o, LPR);
        const int ok = __shfl_up(best_k, o, LPR);
        if (g >= o && ob >= best && ok >= 0) { best = ob; best_k = ok; }
    }
    if (g == LPR - 1 && on) {
        const double mx = best / sum;
        const uint32_t q = phred_from_x(1.0 - mx, B.qthr);
        if (B.out_state) B.out_state[r] = best_k;
        if (B.out_q) B.out_q[r] = (uint8_t)q;
        if (B.out_bq) {
            const int expect = B.row_expect[r];
            const uint32_t raw = B.row_rawq[r];
            const uint32_t bq = ((best_k & 3) != 0 || (best_k >> 2) != expect) ? 0u : (raw < q ? raw : q);
            B.out_bq[r] = (uint8_t)(bq < 94 ? bq : 93);
        }
    }
}

extern "C" hipError_t spx_launch_map(const spx_dev_batch *B, int32_t n_rows_total, int wide, hipStream_t st)
{
    if (n_rows_total <= 0) return hipSuccess;
    /* eight lanes per row; wide: most rows belong to bands of more than 48 slots (ONT): up to 16 slots per lane */
    if (wide) hipLaunchKernelGGL((map_kernel<16, 8>), dim3(((int64_t)n_rows_total * 8 + 255) / 256), dim3(256), 0, st, *B, n_rows_total);
    else hipLaunchKernelGGL((map_kernel<6, 8>), dim3(((int64_t)n_rows_total * 8 + 255) / 256), dim3(256), 0, st, *B, n_rows_total);
    return hipGetLastError();
}

/* ---------------------------------------------------------------------- */
/* marker filter + score + deterministic part of the decision
 * (ptMarker.c:110-153 filter_lowq_markers, :307-325 calc_alignment_score; ptAlignment.c:137-177), three
 * small kernels: per read position the minimum quality over its n markers; per (group, alignment) the
 * score accumulated in position order (the order the reference adds them in); per group the decision. */
__global__ __launch_bounds__(256) void posmin_kernel(spx_dev_groups Gd, int32_t n_markers, uint8_t *__restrict__ posmin)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_markers) return;
    const int n = Gd.markers[m].first_of_pos; /* number of markers of this position, 0 if not the first */
    if (n == 0) return;
    int minq = 100;
    for (int k = 0; k < n; ++k) {
        const spx_dev_marker mk = Gd.markers[m + k];
        const int q = mk.row >= 0 ? Gd.out_bq[mk.row] : mk.qfix;
        minq = q < minq ? q : minq;
    }
    posmin[m] = (uint8_t)minq;
}

__global__ __launch_bounds__(256) void score_kernel(spx_dev_groups Gd, const uint8_t *__restrict__ posmin)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int gi = t / 10, a = t - gi * 10;
    if (gi >= Gd.n_groups) return;
    const int n = Gd.n_aln[gi];
    if (a >= n) return;
    const int m0 = Gd.mk_first[gi], m1 = Gd.mk_first[gi + 1];
    double sc = 0.0;
    for (int m = m0; m < m1; m += n) {
        const int minq = posmin[m];
        if (minq > Gd.min_q) sc += Gd.markers[m + a].is_match ? Gd.match_tbl[minq] : Gd.mis_tbl[minq];
    }
    Gd.score[(int64_t)gi * 10 + a] = sc;
}

__global__ __launch_bounds__(256) void decide_kernel(spx_dev_groups Gd)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= Gd.n_groups) return;
    const int n = Gd.n_aln[gi];
    const uint32_t sec = Gd.sec_mask[gi];
    const double *sc = Gd.score + (int64_t)gi * 10;
    double max_score = -1.7976931348623157e308, prim_score = -1.7976931348623157e308;
    int max_idx = -1, prim_idx = -1;
    for (int a = 0; a < n; ++a) {
        if (!((sec >> a) & 1)) { prim_idx = a; prim_score = sc[a]; }
        else if (max_score < sc[a]) { max_idx = a; max_score = sc[a]; }
    }
    uint32_t tie = 0;
    for (int a = 0; a < n; ++a)
        if (((sec >> a) & 1) && max_score <= sc[a]) tie |= 1u << a;
    Gd.prim_idx[gi] = (uint8_t)prim_idx;
    Gd.max_idx[gi] = (uint8_t)max_idx;
    Gd.tie_mask[gi] = (uint16_t)tie;
    Gd.pass[gi] = !(prim_idx == -1 || max_score <= (prim_score + Gd.prim_margin) || max_score < Gd.min_score);
}

/* fixed-size decision records for the cross-rank gather (one RCCL collective): everything the rand() replay of
 * get_best_record_index needs (ptAlignment.c:163-176), 16 bytes per dispatched group.  Groups with an error
 * (n_aln = 0 on the device) draw nothing and are marked n_aln = 0. */
__global__ __launch_bounds__(256) void pack_kernel(spx_dev_groups Gd, const int32_t *__restrict__ grp_index,
                                                   int32_t group_base, spx_decision *__restrict__ out)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= Gd.n_groups) return;
    spx_decision d;
    d.group = (uint32_t)(grp_index[gi] + group_base);
    const int n = Gd.n_aln[gi];
    d.n_aln = (int8_t)n;
    const int prim = (int8_t)Gd.prim_idx[gi], mx = (int8_t)Gd.max_idx[gi];
    d.prim_idx = (int8_t)prim;
    d.max_idx = (int8_t)mx;
    d.pass = Gd.pass[gi] ? 1 : 0;
    d.tie_mask = Gd.tie_mask[gi];
    d.reserved = 0;
    const double *sc = Gd.score + (int64_t)gi * 10;
    const double max_score = mx >= 0 ? sc[mx] : -1.7976931348623157e308, prim_score = prim >= 0 ? sc[prim] : -1.7976931348623157e308;
    const double dd = max_score - prim_score;
    int v = (dd > -2147483649.0 && dd < 2147483648.0) ? (int)dd : (int)0x80000000; /* cvttsd2si */
    if (v < 0 && v != (int)0x80000000) v = -v;
    d.absdiff = n >= 2 ? v : 0;
    out[gi] = d;
}

/* one packed result record per dispatched group (what spx_collect copies back in one piece) */
__global__ __launch_bounds__(256) void results_kernel(spx_dev_groups Gd, const spx_group_info *__restrict__ info,
                                                      const int32_t *__restrict__ rfe, spx_group_out *__restrict__ out)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= Gd.n_groups) return;
    const spx_group_info in = info[gi];
    spx_group_out o;
    const int n = in.err ? 0 : in.n_aln;
    for (int a = 0; a < 10; ++a) {
        o.score[a] = a < n ? Gd.score[(int64_t)gi * 10 + a] : 0.0;
        o.rfe[a] = a < n ? rfe[(int64_t)gi * 10 + a] : 0;
    }
    o.n_aln = (int8_t)(in.err ? in.err : in.n_aln);
    o.prim_idx = in.err ? (int8_t)-1 : (int8_t)Gd.prim_idx[gi];
    o.max_idx = in.err ? (int8_t)-1 : (int8_t)Gd.max_idx[gi];
    o.pass = in.err ? (int8_t)0 : (int8_t)Gd.pass[gi];
    o.tie_mask = in.err ? (uint16_t)0 : Gd.tie_mask[gi];
    o.best_idx = -1;
    o.relabel = 0;
    o.n_problems = in.n_prob;
    o.n_markers = in.n_mk;
    o.dp_cells = in.cells;
    out[gi] = o;
}

extern "C" hipError_t spx_launch_results(const spx_dev_groups *Gd, const spx_group_info *info, const int32_t *rfe,
                                         spx_group_out *out, hipStream_t st)
{
    if (Gd->n_groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(results_kernel, dim3((Gd->n_groups + 255) / 256), dim3(256), 0, st, *Gd, info, rfe, out);
    return hipGetLastError();
}

extern "C" hipError_t spx_launch_pack(const spx_dev_groups *Gd, const int32_t *grp_index, int32_t group_base,
                                      spx_decision *out, hipStream_t st)
{
    if (Gd->n_groups <= 0) return hipSuccess;
    int blocks = (Gd->n_groups + 255) / 256;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, st, *Gd, grp_index, group_base, out);
    return hipGetLastError();
}

/* ---------------------------------------------------------------------- */
/* phase 0 = forward kernel, 1 = backward kernel, 2 = both (back to back on the same stream) */
extern "C" hipError_t spx_launch_baq(int cls, int phase, const spx_dev_batch *B, hipStream_t st)
{
    if (B->n_order <= 0 && B->n_order_bwd <= 0) return hipSuccess;
    /* A class' launch may go out in several pieces (SPX_DP_PIECES, default 1): the waves of a DP kernel fill every SIMD's
     * register file and the kernels of the NEXT list's preparation, queued on other streams, get no slot until the whole
     * launch has drained; between two pieces of a stream the chip drains for a moment and they get in. */
    static const int n_pieces = [] { const char *e = getenv("SPX_DP_PIECES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    auto pieces = [&](int blocks, int ppw, bool bwd, auto &&launch) {
        if (blocks <= 0) return;
        int per = (blocks + n_pieces - 1) / n_pieces;
        if (per < 2048) per = blocks < 2048 ? blocks : 2048; /* (a piece should still fill the chip a few times) */
        for (int b0 = 0; b0 < blocks; b0 += per) {
            spx_dev_batch P = *B;
            if (bwd) { P.order_bwd += (int64_t)b0 * ppw; P.n_order_bwd -= b0 * ppw; }
            else { P.order += (int64_t)b0 * ppw; P.n_order -= b0 * ppw; }
            launch(P, blocks - b0 < per ? blocks - b0 : per);
        }
    };
#define SPX_LAUNCH(G_, C_, W0_, LDS_)                                                                                  \
    {                                                                                                                  \
        int ppw = 64 / G_, blocks = (B->n_order + ppw - 1) / ppw, blocks_b = (B->n_order_bwd + ppw - 1) / ppw;         \
        if (phase != 1) pieces(blocks, ppw, false, [&](const spx_dev_batch &P, int nb) { hipLaunchKernelGGL((baq_fwd_kernel<G_, C_, W0_, LDS_>), dim3(nb), dim3(64), 0, st, P); }); \
        if (phase != 0) pieces(blocks_b, ppw, true, [&](const spx_dev_batch &P, int nb) { hipLaunchKernelGGL((baq_bwd_kernel<G_, C_, W0_, LDS_>), dim3(nb), dim3(64), 0, st, P); }); \
    }                                                                                                                  \
    break;
    /* exact-width classes (the HiFi preset: bw = 20 + |R-L|): forward with one lane per problem (no half-idle serial
     * passes), backward with two lanes per problem -- with 64 problems per wave the divergent row saves at the
     * wanted rows cost more than the serial passes do */
#define SPX_LAUNCH_EXACT(W_, CB_)                                                                                      \
    {                                                                                                                  \
        if (phase != 1) pieces((B->n_order + 63) / 64, 64, false, [&](const spx_dev_batch &P, int nb) { hipLaunchKernelGGL((baq_fwd1_kernel<W_>), dim3(nb), dim3(64), 0, st, P); }); \
        if (phase != 0)                                                                                                \
            pieces((B->n_order_bwd + 31) / 32, 32, true, [&](const spx_dev_batch &P, int nb) { hipLaunchKernelGGL((baq_bwd_kernel<2, CB_, W_, false>), dim3(nb), dim3(64), 0, st, P); }); \
    }                                                                                                                  \
    break;
    switch (cls) { /* keep in step with spx_prep.cpp kClass* */
    case 0: SPX_LAUNCH_EXACT(41, 21)
    case 1: SPX_LAUNCH_EXACT(43, 22)
    case 2: SPX_LAUNCH_EXACT(45, 23)
    case 3: SPX_LAUNCH_EXACT(47, 24)
    case 4: SPX_LAUNCH(2, 24, 0, false)
    case 5: SPX_LAUNCH(4, 16, 0, false)
    /* (4,28)/(4,32) with the D row in LDS (DLds) were measured for the ONT bands: parity-clean but slower than
     * these -- four SIMDs share one LDS pipe and the row traffic saturates it */
    case 6: SPX_LAUNCH(4, 26, 0, false)
    case 7: /* forward (8,16); backward (4,32): half the serial passes, and unlike the forward kernel (134 spilled
             * VGPRs at (4,32), 99 with the D row in LDS -- both slower) it fits the register file */
    {
        if (phase != 1) pieces((B->n_order + 7) / 8, 8, false, [&](const spx_dev_batch &P, int nb) { hipLaunchKernelGGL((baq_fwd_kernel<8, 16, 0, false>), dim3(nb), dim3(64), 0, st, P); });
        if (phase != 0) pieces((B->n_order_bwd + 15) / 16, 16, true, [&](const spx_dev_batch &P, int nb) { hipLaunchKernelGGL((baq_bwd_kernel<4, 32, 0, false>), dim3(nb), dim3(64), 0, st, P); });
    }
    break;
    case 8: SPX_LAUNCH(16, 16, 0, false)
    case 9: SPX_LAUNCH(32, 16, 0, false)
    case 10: SPX_LAUNCH(64, 16, 0, false)
    case 11: SPX_LAUNCH(64, 32, 0, false)
    case 12: SPX_LAUNCH(4, 28, 0, false)
    case 13: SPX_LAUNCH(4, 30, 0, false)
    default: return hipErrorInvalidValue;
    }
#undef SPX_LAUNCH_EXACT
#undef SPX_LAUNCH
    return hipGetLastError();
}

extern "C" hipError_t spx_launch_score(const spx_dev_groups *Gd, int32_t n_markers, uint8_t *posmin, hipStream_t st)
{
    if (Gd->n_groups <= 0) return hipSuccess;
    if (n_markers > 0)
        hipLaunchKernelGGL(posmin_kernel, dim3((n_markers + 255) / 256), dim3(256), 0, st, *Gd, n_markers, posmin);
    hipLaunchKernelGGL(score_kernel, dim3((Gd->n_groups * 10 + 255) / 256), dim3(256), 0, st, *Gd, posmin);
    hipLaunchKernelGGL(decide_kernel, dim3((Gd->n_groups + 255) / 256), dim3(256), 0, st, *Gd);
    return hipGetLastError();
}
