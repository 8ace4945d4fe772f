/*
 * spx_fast_kernels.hip -- the FAST tier of the two-tier banded-HMM DP (gfx950).
 *
 *  fast_fwd_kernel<G,C,W0>   forward pass in the (U, V) formulation, G lanes x C band slots per problem
 *  fast_bwd_kernel<G,C,W0>   backward pass; at the wanted rows the saved forward rows are multiplied by the backward rows
 *  fast_map_kernel           per wanted row: emission + state ratio applied, arg-max, second largest, sum of the others,
 *                            and the CERTIFICATE that the exact tier's (state, q) equals the fast tier's
 *
 * Same real-number model as the exact tier (spx_kernels.hip = htslib-1.17 probaln_glocal as called at
 * /root/reference/programs/submodules/ptMarker/ptMarker.c:755-757; MAP rule ptMarker.c:778-779,786), different arithmetic:
 * only (state, q) of the wanted rows leave probaln_glocal, both integers, and both are functions of the ROW-NORMALISED
 * posterior products z(i,k)/sum_k z(i,k) -- invariant under any per-row factor.  So the fast tier
 *   - contracts multiply-adds (fma), 8-9 instructions per band cell instead of 25-54;
 *   - carries two combined rows  Ut = (m0 M + m3 I + m6 D)/(m6 m2),  Vt = next row's I/(EI m1)  instead of M, I, D
 *     (tools/fastdp/fastdp_model.c derives the recurrences; that C model is what these kernels follow);
 *   - evaluates the D recurrence D_k = M_{k-1} + m8 D_{k-1} as a lane-local chain plus ONE carry per lane, added as
 *     pw[c] * carry when the value is next read (all lanes busy; the exact tier walks the columns G times);
 *   - computes no row sums: a power-of-two factor every 16 rows keeps the rows in range (exact: no rounding);
 *   - leaves emission and the z = f*b products of the M state to the MAP kernel (the forward kernel stores its raw rows).
 * Every operation adds or multiplies non-negative numbers, so a value's relative error is bounded by the number of roundings
 * along the deepest lattice path: delta = 48 (L + R + W + 16) 2^-53 covers fast + exact tier with a factor 2 to spare.
 * fast_map_kernel certifies a row iff (a) the two largest z are more than delta apart (the arg-max is the exact tier's) and
 * (b) x = (sum of the other z)/(sum of all z), widened by 2 delta relative and (2W+8) 2^-53 absolute (the exact tier's
 * rounding of 1 - fl(max/fl(sum))), lies inside one phred bin.  A row that is not certified, a row whose values span more
 * than 2^range_bits, an ambiguous base or a degenerate constant set tier[problem] = 2, and the EXACT kernels then run that
 * problem (never the host, never the oracle).  tools/fastdp_study.py: 0 uncertified-but-different rows in 1.44 G rows of the presets and 3.77 G of the fuzz ranges.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/spx.h"
#include "spx_device.h"
#include "spx_dp_dev.h"

#define FAST_EI 0.25
#ifndef SPX_FAST_WAVES
#define SPX_FAST_WAVES 2
#endif
#ifndef SPX_FAST_HIFI_WAVES
#define SPX_FAST_HIFI_WAVES 3 /* waves per SIMD the two-lane backward kernels of the HiFi classes are compiled for (forward: 4; measured: forward 13.3 -> 10.8 ms
                               * at four, backward 9.4 -> 12.4: it spills there) */
#endif
#define FAST_RESCALE_MASK 15 /* rows between two rescales / range checks - 1 */
/* the unrolled slot loops are fenced every SPX_FAST_FENCE slots: left alone, the scheduler hoists the emission selects and the loads of
 * many slots to the top of a row and the register file spills */
#ifndef SPX_FAST_FENCE
#define SPX_FAST_FENCE 1
#endif
#define FAST_FENCE(c) do { if (FENCE > 0 && ((c) % (FENCE > 0 ? FENCE : 1)) == 0) __builtin_amdgcn_sched_barrier(0); } while (0)

/* the model conditions of the fast tier (fastdp_model.c FDP_F_MODEL): no ambiguous base; the problem's constants are the launch's
 * (the same expressions as spxl::hmm_constants, so equality is exact); positive and finite; the static range condition */
__device__ __forceinline__ bool fast_eligible(const HmmC &h, double bM, double bI, double sM, double sI, int hasN, const spx_fast_consts &K)
{
    bool ok = !hasN && sM == sI;
    ok = ok && h.m0 == K.m0h * (1 - sM) && h.m1 == K.m1h * (1 - sM) && h.m2 == h.m1 && h.m3 == K.m3h * (1 - sI) && h.m4 == K.m4h * (1 - sI);
    ok = ok && h.m6 == K.m6 && h.m8 == K.m8 && h.e_match == K.e_match && h.e_mis == K.e_mis;
    const double cs[13] = {h.m0, h.m1, h.m2, h.m3, h.m4, h.m6, h.m8, bM, bI, sM, sI, h.e_match, h.e_mis};
#pragma unroll
    for (int t = 0; t < 13; ++t) ok = ok && cs[t] > 1e-30 && cs[t] < 1e30;
    /* the smallest factor a value can take per row: between two range checks a row's spread grows by at most that per row */
    const double mu = fmin(K.m0h * K.e_mis, FAST_EI * K.m4h);
    ok = ok && mu >= __hiloint2double((1023 - K.mu_bits) << 20, 0);
    return ok;
}

__device__ __forceinline__ double zero_if(double v, int32_t m) /* m all ones: 0, m zero: v */
{
    uint32_t um = (uint32_t)m;
    asm("" : "+v"(um));
    return __hiloint2double((int)((uint32_t)__double2hiint(v) & ~um), (int)((uint32_t)__double2loint(v) & ~um));
}

/* bit 4n+3 of eq[k] set iff code n of word k equals qy (exact zero-nibble test) */
template <int NW>
__device__ __forceinline__ void eq_masks(const uint32_t (&w)[NW], uint32_t qy, uint32_t (&eq)[NW])
{
    const uint32_t qrep = qy * 0x11111111u;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t x = w[k] ^ qrep;
        eq[k] = ~(((x & 0x77777777u) + 0x77777777u) | x) & 0x88888888u;
    }
}
/* all ones iff bit 3 of nibble c is set */
template <int NW>
__device__ __forceinline__ int32_t nib_bit3(const uint32_t (&w)[NW], int c)
{
    return __builtin_amdgcn_sbfe((int32_t)w[c >> 3], 4 * (c & 7) + 3, 1);
}

template <int G>
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}

/* range check of the row A + power-of-two rescale of A and Bv, one problem = G lanes x C slots.  Slots [c_lo, c_hi] of this lane hold real
 * cells; exact zeros are structural (a positive value cannot reach zero between two checks that passed: fast_eligible's mu condition) and
 * do not count.  A alone is checked: the other row is within constant factors of it cell by cell and column sum by column sum (forward:
 * V >= M and U <= (cU0 + cU1/c4 + 1/(1-m8)) max V; backward: Bi >= X and Bm <= (1 + cB1/c4 + cB2/(1-m8)) max Bi) -- part of the 100 bits
 * fast_constants leaves.  Returns true when the problem must be flagged: non-finite or denormal values, or a spread over 2^range_bits. */
template <int G, int C, bool MASKED>
__device__ __forceinline__ bool range_rescale(double (&A)[C], double (&Bv)[C], int c_lo, int c_hi, int range_bits)
{
    uint32_t mx = 0, mn1 = 0xffffffffu; /* mn1 = (smallest non-zero high word) - 1 */
#pragma unroll
    for (int c = 0; c < C; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        uint32_t ha = (uint32_t)__double2hiint(A[c]);
        if constexpr (MASKED) ha = (c >= c_lo && c <= c_hi) ? ha : 0u;
        mx = max(mx, ha);
        mn1 = min(mn1, ha - 1u);
    }
    __builtin_amdgcn_sched_barrier(0);
    mx = group_max_u32<G>(mx);
    mn1 = group_min_u32<G>(mn1);
    const uint32_t mn = mn1 + 1u; /* 0: no non-zero value at all */
    const bool bad = mx >= 0x7fe00000u || mx < 0x00100000u || mn < 0x00100000u || (int)(mx >> 20) - (int)(mn >> 20) > range_bits;
    if (mx >= 0x00100000u && mx < 0x7fe00000u) {
        const double sc = __hiloint2double((int)((2046u - (mx >> 20)) << 20), 0); /* 2^-(exponent of the largest value) */
#pragma unroll
        for (int c = 0; c < C; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            A[c] *= sc; Bv[c] *= sc;
        }
    }
    return bad;
}

/* ====================================================================== */
template <int G, int C, int W0, int WAVES = SPX_FAST_WAVES, int FENCE = SPX_FAST_FENCE>
__global__ __launch_bounds__(64, WAVES) void fast_fwd_kernel(spx_dev_batch B, spx_fast_consts K)
{
    constexpr int NW = NibWin<C>::NW;
    /* FAST rows mask the emission of the last PADMAX slots of a lane (slots beyond the band in the last lane); waves with more
     * padding than that run every row through the masked variant */
    constexpr int PADMAX = (G == 1) ? 0 : (W0 ? G * C - W0 : (C < 8 ? C : 8));
    const int lane = threadIdx.x & 63;
    const int g = lane % G;
    HmmC h;
    int hasN;
    Prob P = load_problem<G>(B, lane, h, hasN);
    bool act = P.act;
    double bM = 0, bI = 0;
    if (act) {
        const double *hp = B.hmm + (int64_t)P.pid * SPX_H_N;
        bM = hp[SPX_H_BM]; bI = hp[SPX_H_BI];
        const bool ok = fast_eligible(h, bM, bI, hp[SPX_H_SM], hp[SPX_H_SI], hasN, K) && 2 * P.bw + 1 <= G * C;
        if (g == 0) {
            B.tier[P.pid] = ok ? SPX_TIER_FAST : SPX_TIER_RERUN;
            if (!ok && B.tier_counts) atomicAdd(&B.tier_counts[1], 1);
        }
        act = ok;
    }
    /* rows behind the LAST wanted row have no observable effect (the exact tier walks them for its scaling factors, which this tier
     * does not need): L below = the rows this kernel walks */
    const int L = act ? B.rows[P.row0 + P.nrows - 1] : 0, R = P.R, bw = P.bw;
    const int Wu = W0 ? W0 : wave_max(act ? 2 * bw + 1 : 0);
    const int Lw = wave_max(L);
    if (Lw == 0) return;
    int fast_end = Lw;
    if (G * C - Wu > PADMAX) fast_end = 1;
    const int jbase = g * C;
    const int SLOTS = (int)(B.fsave_stride >> 2); /* a wanted row of the fast tier: [U | V | Bm | Bi][slots] */

    double U[C], V[C];
    NibWin<C> cw;
    uint32_t padn[G > 1 ? NW : 1]; /* SPX_CODE_OUT in the slots beyond the band (last lanes of a multi-lane problem) */
    int32_t padm[PADMAX > 0 ? PADMAX : 1];
    double *fsave = B.fsave + (act ? B.fsave_off[P.pid] : 0);
    const int64_t fstride = B.fsave_stride;
    const int nrows = act ? P.nrows : 0, row0 = P.row0;
    int wnext = 0;
    int next_row = nrows > 0 ? B.rows[row0] : 0x7fffffff;
    bool flagged = false;

    /* row 1: M = e bM, I = EI bI on columns 1 .. min(R, bw + 1); no D */
    {
        const uint32_t qy = act ? fetch_code(B.qry4, P.qry0, 0, L) : 0;
        const double It1 = act ? (FAST_EI * bI) / K.gam : 0.0;
        const double U0 = act ? bM / K.ups : 0.0; /* what the MAP kernel multiplies by e m6 m2 */
#pragma unroll
        for (int k = 0; k < NW; ++k) { cw.w[k] = 0; if (G > 1) padn[k] = 0; }
        double *dst = fsave + jbase;
        const bool sv = act && next_row == 1;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c;
            const uint32_t code = act ? fetch_code(B.ref4, P.ref0, j - bw, R) : (uint32_t)SPX_CODE_OUT;
            cw.set(c, code);
            if (G > 1 && j >= Wu) padn[c >> 3] |= (uint32_t)SPX_CODE_OUT << (4 * (c & 7));
            const bool valid = !(code & SPX_CODE_OUT) && j < Wu;
            const double e = code == qy ? K.e_match : K.e_mis;
            const double M = valid ? e * bM : 0.0, It = valid ? It1 : 0.0;
            U[c] = fma(K.cU0, M, K.cU1 * It);
            V[c] = fma(K.c4, It, M);
            if (sv) { dst[c] = valid ? U0 : 0.0; dst[SLOTS + c] = It; }
        }
        if (PADMAX > 0) {
#pragma unroll
            for (int t = 0; t < PADMAX; ++t) padm[t] = (jbase + C - PADMAX + t) < Wu ? 0 : -1;
        }
        if (sv) { wnext++; next_row = wnext < nrows ? B.rows[row0 + wnext] : 0x7fffffff; }
    }
    const int top = jbase + C - 1; /* the slot that receives a new column each row */
    auto ref_chunk = [&](int ib) { return fetch8(B.ref4, P.ref0 + (ib - bw + top - 1)); }; /* rows ib..ib+7 */
    auto qry_chunk = [&](int ib) { return fetch8(B.qry4, P.qry0 + (ib - 1)); };
    uint32_t qwin = act ? qry_chunk(1) : 0, rwin = act ? ref_chunk(1) : 0;
    uint32_t qwin_n = act ? qry_chunk(9) : 0, rwin_n = act ? ref_chunk(9) : 0;
    /* every 16 rows: dynamic range of row i-1 (V: exactly zero wherever there is no cell), rescale by a power of two */
    auto rescale = [&](int i) {
        const bool bad = range_rescale<G, C, false>(V, U, 0, C - 1, K.range_bits);
        if (act && i <= L && bad) flagged = true;
    };
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
            if ((unsigned)(i - bw + top - 1) >= (unsigned)R) rc = SPX_CODE_OUT;
            cw.shift_down(rc);
            uint32_t ew[NW], eq[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) ew[k] = G > 1 ? (cw.w[k] | padn[k]) : cw.w[k];
            eq_masks<NW>(ew, qy, eq);
            double Vn = shfl_down1<G>(V[0]);
            if (g == G - 1) Vn = 0.0;
            /* wanted row: the MAP kernel needs M(i,k) = e (m6 m2) U(i-1,k-1) and It(i,k) = V(i-1,k): store the two raw rows */
            if (i == next_row) {
                double *dst = fsave + (int64_t)wnext * fstride + jbase;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    dst[c] = U[c];
                    dst[SLOTS + c] = (c + 1 < C) ? V[c + 1] : Vn;
                }
                wnext++;
                next_row = wnext < nrows ? B.rows[row0 + wnext] : 0x7fffffff;
            }
            double Dloc = 0.0, Mprev = 0.0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                FAST_FENCE(c);
                const double e = select_bits(nib_bit3<NW>(eq, c), K.emU, K.exU);
                double M = e * U[c];
                if constexpr (!FAST) M = zero_if(M, nib_bit3<NW>(ew, c)); /* no such cell: column > R, or a slot beyond the band */
                else if constexpr (PADMAX > 0) { if (c >= C - PADMAX) M = zero_if(M, padm[c - (C - PADMAX)]); }
                const double It = (c + 1 < C) ? V[c + 1] : Vn;
                if (c > 0) Dloc = fma(K.m8, Dloc, Mprev);
                U[c] = fma(K.cU0, M, fma(K.cU1, It, Dloc));
                V[c] = fma(K.c4, It, M);
                Mprev = M;
            }
            if constexpr (G > 1) {
                /* D carry across the lanes of the problem: the lane-local chains started from 0; lane g's slots lack pw[c] * (true D at the
                 * slot in front of its first one) */
                const double E = fma(K.m8, Dloc, Mprev); /* local D at the slot behind this lane's last one */
                double cin = 0.0;
#pragma unroll
                for (int t = 1; t < G; ++t) {
                    const double up = shfl_up1<G>(fma(K.pw[C], cin, E));
                    if (g == t) cin = up;
                }
#pragma unroll
                for (int c = 0; c < C; ++c) U[c] = fma(K.pw[c], cin, U[c]);
            }
        }
    };
    /* A block of 16 rows takes the masked variant when some problem of the wave that is still walking reaches, inside the block, the rows
     * whose band touches column R (the launch order goes by the rows walked, so a wave mixes reference lengths: one wave-wide bound would
     * mask most rows), or when the wave's padding is wider than the FAST variant handles.  (Chosen per block, in two separate loops: a
     * per-row choice inside one loop made the register allocator spill hundreds of values.) */
    const bool all_masked = fast_end <= 1;
    for (int i0 = 2; i0 <= Lw; i0 += FAST_RESCALE_MASK + 1) {
        rescale(i0);
        const int i1 = min(i0 + FAST_RESCALE_MASK, Lw);
        const int if1 = (all_masked || __any(act && i0 <= L && min(i1, L) + bw > R)) ? i0 - 1 : i1; /* (two loops one behind the other with
                                                                                                       * bounds, not an if / else: register allocation) */
        int i = i0;
        for (; i <= if1; ++i) row(i, std::true_type{});
        for (; i <= i1; ++i) row(i, std::false_type{});
    }
    if (act && flagged && g == 0) {
        B.tier[P.pid] = SPX_TIER_RERUN;
        if (B.tier_counts) atomicAdd(&B.tier_counts[2], 1);
    }
}

/* ====================================================================== */
template <int G, int C, int W0, int WAVES = SPX_FAST_WAVES, int FENCE = SPX_FAST_FENCE>
__global__ __launch_bounds__(64, WAVES) void fast_bwd_kernel(spx_dev_batch B, spx_fast_consts K)
{
    constexpr int NW = NibWin<C>::NW;
    constexpr int PADMAX = (G == 1) ? 0 : (W0 ? G * C - W0 : (C < 8 ? C : 8));
    const int lane = threadIdx.x & 63;
    const int g = lane % G;
    HmmC h;
    int hasN;
    Prob P = load_problem<G>(B, lane, h, hasN, true);
    bool act = P.act && B.tier[P.pid] == SPX_TIER_FAST; /* (problems the forward kernel found outside the model, or flagged) */
    const int L = act ? P.L : 0, R = P.R, bw = P.bw;
    const int Wu = W0 ? W0 : wave_max(act ? 2 * bw + 1 : 0);
    const int Lw = wave_max(L);
    if (Lw == 0) return;
    const int nrows = act ? P.nrows : 0, row0 = P.row0;
    const int stop = act ? B.rows[row0] : 0x7fffffff; /* first (smallest) wanted row */
    const int jbase = g * C;
    const int SLOTS = (int)(B.fsave_stride >> 2);
    const int64_t fstride = B.fsave_stride;

    double Bm[C], Bi[C];
    NibWin<C> cw;
    uint32_t padn[G > 1 ? NW : 1];
    int32_t padm[PADMAX > 0 ? PADMAX : 1];
    bool flagged = false;
    /* row L */
    {
        const double vM = act ? 1.0 / K.m0h : 0.0, vI = act ? 1.0 / K.m3h : 0.0; /* bM = sM, bI = sI (sM == sI: fast_eligible), in units of m0, m3 */
#pragma unroll
        for (int k = 0; k < NW; ++k) { cw.w[k] = 0; if (G > 1) padn[k] = 0; }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = jbase + c, k = L - bw + j;
            const bool valid = act && j < Wu && k >= 1 && k <= P.Rt;
            Bm[c] = valid ? vM : 0.0;
            Bi[c] = valid ? vI : 0.0;
            if (G > 1 && j >= Wu) padn[c >> 3] |= (uint32_t)SPX_CODE_OUT << (4 * (c & 7));
            /* window for row L-1: code of ref idx (L-1) - bw + j (= column k+1 of that row) */
            cw.set(c, (act && L >= 2) ? fetch_code(B.ref4, P.ref0, (L - 1) - bw + j, R) : (uint32_t)SPX_CODE_OUT);
        }
        if (PADMAX > 0) {
#pragma unroll
            for (int t = 0; t < PADMAX; ++t) padm[t] = (jbase + C - PADMAX + t) < Wu ? 0 : -1;
        }
    }
    double *fsave = B.fsave + (act ? B.fsave_off[P.pid] : 0);
    int wprev = nrows - 1;
    int prev_row = wprev >= 0 ? B.rows[row0 + wprev] : -1;
    /* wanted row: the backward rows go BESIDE the saved forward rows ([row][U | V | Bm | Bi][slots]); the MAP kernel forms the products.
     * (Round 6: multiplying in place -- load, multiply, store -- left these kernels 72 % of their wave cycles in s_waitcnt: the forward rows
     * were written long ago and come from HBM.  Touching the next wanted row's lines ahead of time did not help: they do not survive in L2.) */
    auto save_row = [&]() {
        double *dst = fsave + (int64_t)wprev * fstride + 2 * SLOTS + jbase;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            dst[c] = Bm[c];
            dst[SLOTS + c] = Bi[c];
        }
        wprev--;
        prev_row = wprev >= 0 ? B.rows[row0 + wprev] : -1;
    };
    if (act && prev_row == L) save_row();
    auto ref_chunk = [&](int i0) { return fetch8(B.ref4, P.ref0 + ((i0 - 7) - bw + jbase)); };
    auto qry_chunk = [&](int i0) { return fetch8(B.qry4, P.qry0 + (i0 - 7)); };
    uint32_t qwin = 0, rwin = 0;
    uint32_t qwin_n = (act && L >= 2) ? qry_chunk(L - 1) : 0, rwin_n = (act && L >= 2) ? ref_chunk(L - 1) : 0;
    const int nb = act ? max(L - stop, 0) : 0;
    const int nbw = wave_max(nb);
    /* masked rows: while the band still touches column R (first steps), and -- every step -- when the wave's padding is wide */
    int n_slow = min(nbw, wave_max(act ? min(nb, max(0, (L - 1) - (R - bw - 1))) : 0));
    if (G * C - Wu > PADMAX) n_slow = nbw;
    /* every 16 steps: dynamic range of row L - t over its real cells (columns < 1 and slots beyond the band hold values that never
     * reach a real cell, but are not zero), rescale by a power of two */
    auto rescale = [&](int t) {
        const int k0 = (L - t) - bw + jbase; /* column of slot 0 */
        const bool bad = range_rescale<G, C, true>(Bi, Bm, max(0, 1 - k0), min(min(C - 1, R - k0), Wu - 1 - jbase), K.range_bits);
        if (act && t < nb && bad) flagged = true;
    };
    auto row = [&](int t, auto fast_tag, auto first_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value; /* some problem of the wave is on its row 1 (D row = 0) */
        const int i = L - 1 - t;
        const bool on = act && t < nb;
        if (on) {
            const uint32_t t4 = (uint32_t)(7 - (t & 7)) * 4u;
            if ((t & 7) == 0) {
                qwin = qwin_n; rwin = rwin_n;
                qwin_n = qry_chunk(i - 8);
                rwin_n = ref_chunk(i - 8);
            }
            const uint32_t qy = (qwin >> t4) & 0xfu;
            if (t != 0) {
                uint32_t rc = (rwin >> t4) & 0xfu;
                if ((unsigned)(i - bw + jbase) >= (unsigned)R) rc = SPX_CODE_OUT;
                cw.shift_up(rc);
            }
            uint32_t ew[NW], eq[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) ew[k] = G > 1 ? (cw.w[k] | padn[k]) : cw.w[k];
            eq_masks<NW>(ew, qy, eq);
            double Yh = shfl_up1<G>(Bi[C - 1]);
            if (g == 0) Yh = 0.0;
            const double y = (FIRST && i == 1) ? 0.0 : 1.0;
            double Dloc = 0.0;
#pragma unroll
            for (int c = C - 1; c >= 0; --c) {
                FAST_FENCE(c);
                const double e = select_bits(nib_bit3<NW>(eq, c), K.emB, K.exB);
                double X = e * Bm[c];
                if constexpr (!FAST) X = zero_if(X, nib_bit3<NW>(ew, c)); /* column k+1 > R, or a slot beyond the band */
                else if constexpr (PADMAX > 0) { if (c >= C - PADMAX) X = zero_if(X, padm[c - (C - PADMAX)]); }
                const double Y = c > 0 ? Bi[c - 1] : Yh;
                Bm[c] = fma(K.cB1, Y, fma(K.cB2, Dloc, X));
                Bi[c] = fma(K.c4, Y, X);
                Dloc = fma(K.m8, Dloc, X);
                if constexpr (FIRST) Dloc *= y;
            }
            if constexpr (G > 1) {
                /* D carry from the lane above: Bm[c] lacks cB2 * pw[C-1-c] * (true D at slot 0 of the lane above) */
                double cin = 0.0;
#pragma unroll
                for (int tt = G - 2; tt >= 0; --tt) {
                    const double dn = shfl_down1<G>(fma(K.pw[C], cin, Dloc));
                    if (g == tt) cin = dn;
                }
                if constexpr (FIRST) cin *= y;
                const double cb = K.cB2 * cin;
#pragma unroll
                for (int c = 0; c < C; ++c) Bm[c] = fma(K.pw[C - 1 - c], cb, Bm[c]);
            }
            if (i == prev_row) save_row();
        }
    };
    /* a block of 16 steps in which some problem of the wave computes its row 1 (step L - 2) takes the FIRST variant (masked, D row times
     * y = 0 on that row); the first n_slow steps the masked one; every other block the FAST one -- three separate loops, see the forward kernel */
    for (int t0 = 0; t0 < nbw; t0 += FAST_RESCALE_MASK + 1) {
        rescale(t0);
        const int t1 = min(t0 + FAST_RESCALE_MASK + 1, nbw);
        const bool blk_first = __any(act && nb == L - 1 && nb > 0 && L - 2 >= t0 && L - 2 < t1), blk_slow = t0 < n_slow;
        const int ta = blk_first ? t0 : (blk_slow ? t1 : t0), tb = blk_first ? t0 : t1; /* [t0,ta) masked, [ta,tb) fast, [tb,t1) first: one is not empty */
        int t = t0;
        for (; t < ta; ++t) row(t, std::false_type{}, std::false_type{});
        for (; t < tb; ++t) row(t, std::true_type{}, std::false_type{});
        for (; t < t1; ++t) row(t, std::false_type{}, std::true_type{});
    }
    if (act && flagged && g == 0) {
        B.tier[P.pid] = SPX_TIER_RERUN;
        if (B.tier_counts) atomicAdd(&B.tier_counts[2], 1);
    }
}

/* ====================================================================== */
/* MAP + certificate of the fast tier: LPR adjacent lanes per wanted row of a tier-1 problem.  z_M(j) = e(i,k) m6 m2 * [U Bm](j),
 * z_I(j) = rho * [It Bi](j) with rho = EI m1 m3 / m0 (the products in brackets were left by fast_bwd_kernel). */
template <int CQ, int LPR>
__global__ __launch_bounds__(256) void fast_map_kernel(spx_dev_batch B, spx_fast_consts K, int32_t n_rows_total)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = B.row_base + (int)(tid / LPR), g = (int)(tid & (LPR - 1));
    bool on = r < B.row_base + n_rows_total;
    const int rr = on ? r : B.row_base;
    const int p = B.row_prob[rr];
    if (B.tier[p] != SPX_TIER_FAST) on = false;
    const int i = B.rows[rr], bw = B.bw[p], R = B.R[p], L = B.L[p];
    const int W = 2 * bw + 1, slots = B.prob_slots[p], Cq = (slots + LPR - 1) / LPR;
    const int64_t off = B.fsave_off[p] + (int64_t)(rr - B.row_off[p]) * 4 * slots;
    const double *zM = B.fsave + off, *zI = zM + slots; /* forward rows U, V; the backward rows Bm, Bi follow at 2 * slots */
    const int j0 = max(0, bw + 1 - i), j1 = min(W - 1, R - i + bw); /* 1 <= k = i - bw + j <= R */
    const double emU = K.emU, exU = K.exU, rho = K.rho;
    const uint32_t qy = on ? fetch_code(B.qry4, B.qry_nib[p], i - 1, L) : 0;
    const int64_t ref0 = B.ref_nib[p];
    /* this lane's z values (M, I per slot), 0 outside [j0, j1].  The LPR lanes of a row take the slots INTERLEAVED, slot j = c * LPR + g: every
     * load of the row is then one 64-byte segment (a contiguous share per lane -- 128 B apart for the 112-slot ONT rows -- made the kernel as
     * long as the class' forward kernel: 19.7 ms per ONT slice).  The certificate needs no column order: ties flag the row.  The reference codes of
     * eight consecutive slots arrive as one word, the same for the lanes of a row (slot j of row i <-> reference index i - bw + j - 1; the pool's
     * lead pad covers the indices in front of a window) */
    double z[2 * CQ];
    const int cq_wave = wave_max(on ? Cq : 0);
    double best = 0.0, second = 0.0;
    int best_t = -1; /* index into z[] */
    uint32_t codes[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) codes[c] = (on && c < Cq) ? fetch8(B.ref4, ref0 + (i - bw - 1) + LPR * c) : 0;
    static_assert(LPR == 8, "one code word (eight codes) per round of the row's lanes");
    auto load = [&](int c, int j) {
        const bool in = on && c < Cq && j >= j0 && j <= j1;
        double m = 0.0, ii = 0.0;
        if (in) {
            const uint32_t code = (codes[c] >> (4 * g)) & 0xfu;
            m = (zM[j] * zM[2 * slots + j]) * (code == qy ? emU : exU);
            ii = (zI[j] * zI[2 * slots + j]) * rho;
        }
        z[2 * c] = m; z[2 * c + 1] = ii;
    };
    if (cq_wave > CQ) { /* (bands wider than this instantiation holds: not a fast class) */
        if (on && g == 0) atomicExch(&B.tier[p], SPX_TIER_RERUN);
        return;
    }
#pragma unroll
    for (int c = 0; c < CQ; ++c) load(c, c * LPR + g);
    bool bad = false;
#pragma unroll
    for (int t = 0; t < 2 * CQ; ++t) {
        const double v = z[t];
        if (!(v >= 0.0) || v > 1.7e308) bad = true;
        if (v > best) { second = best; best = v; best_t = t; }
        else if (v > second) second = v;
    }
    /* across the lanes of the row: largest (lowest lane / index wins ties: flagged anyway), second largest */
    /* merge across the lanes of the row: (largest, second largest) of disjoint sets; equal maxima end up as second = best, i.e. flagged */
    int best_lane = g;
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
        const double ob = __shfl_xor(best, o, LPR), os = __shfl_xor(second, o, LPR);
        const int ot = __shfl_xor(best_t, o, LPR), ol = __shfl_xor(best_lane, o, LPR);
        bad = bad || __shfl_xor((int)bad, o, LPR);
        if (ob > best) { second = fmax(os, best); best = ob; best_t = ot; best_lane = ol; }
        else second = fmax(second, ob);
    }
    double others = 0.0;
#pragma unroll
    for (int t = 0; t < 2 * CQ; ++t) others += (g == best_lane && t == best_t) ? 0.0 : z[t];
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) others += __shfl_xor(others, o, LPR);
    if (g == 0 && on) {
        const double u53 = 1.1102230246251565e-16;
        const double delta = 48.0 * ((double)L + R + W + 16) * u53, A = (2.0 * W + 8.0) * u53;
        const double x = others / (best + others);
        bool ok = !bad && best_t >= 0 && best * (1.0 - delta) > second * (1.0 + delta);
        const uint32_t q_hi = phred_from_x(x * (1.0 - 2.0 * delta) - A, B.qthr), q_lo = phred_from_x(x * (1.0 + 2.0 * delta) + A, B.qthr);
        ok = ok && q_hi == q_lo;
        if (!ok) {
            if (atomicCAS(&B.tier[p], SPX_TIER_FAST, SPX_TIER_RERUN) == SPX_TIER_FAST && B.tier_counts) atomicAdd(&B.tier_counts[0], 1);
            if (B.tier_counts) atomicAdd(&B.tier_counts[3], 1);
        } else {
            const int j = (best_t >> 1) * LPR + best_lane;
            const int best_k = ((i - bw + j - 1) << 2) | (best_t & 1);
            const uint32_t q = q_lo;
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
}

extern "C" hipError_t spx_launch_fast_map(const spx_dev_batch *B, const spx_fast_consts *K, int32_t n_rows_total, int max_slots, hipStream_t st)
{
    if (n_rows_total <= 0) return hipSuccess;
    /* eight lanes per row, CQ rounds of eight slots: the instantiation that holds the list's widest fast class (<= 48 slots: the HiFi classes;
     * 64: with (4,16); 112 / 120: ONT's (4,26), (4,28) / (4,30); 128: every fast class).  A wave of a narrower instantiation that meets a wider row hands
     * its problem to the exact tier, so max_slots must cover every fast class of the list.  (The rounds are unrolled: on the mixed workload the
     * 16-round kernel took 9-10 ms per slice, as long as the forward kernels, where 48-slot rows with some 64-slot ones need 8.) */
    const dim3 grid(((int64_t)n_rows_total * 8 + 255) / 256), block(256);
    if (max_slots <= 48) hipLaunchKernelGGL((fast_map_kernel<6, 8>), grid, block, 0, st, *B, *K, n_rows_total);
    else if (max_slots <= 64) hipLaunchKernelGGL((fast_map_kernel<8, 8>), grid, block, 0, st, *B, *K, n_rows_total);
    else if (max_slots <= 112) hipLaunchKernelGGL((fast_map_kernel<14, 8>), grid, block, 0, st, *B, *K, n_rows_total);
    else if (max_slots <= 120) hipLaunchKernelGGL((fast_map_kernel<15, 8>), grid, block, 0, st, *B, *K, n_rows_total);
    else hipLaunchKernelGGL((fast_map_kernel<16, 8>), grid, block, 0, st, *B, *K, n_rows_total);
    return hipGetLastError();
}

/* band classes with a fast tier (keep in step with spx_launch_baq / spx_logic.h band_class) */
extern "C" int spx_fast_class(int cls)
{
    switch (cls) {
    case 0: case 1: case 2: case 3: case 4: case 5: case 6: case 12: case 13: return 1;
    default: return 0;
    }
}

/* phase 0 = forward, 1 = backward, 2 = both */
extern "C" hipError_t spx_launch_fast(int cls, int phase, const spx_dev_batch *B, const spx_fast_consts *K, hipStream_t st)
{
    if (B->n_order <= 0 && B->n_order_bwd <= 0) return hipSuccess;
    spx_dev_batch P = *B;
    P.fsave_stride = 2 * B->fsave_stride; /* four rows of slots per wanted row (B holds the exact tier's stride: two) */
    P.tier_want = SPX_TIER_ALL; /* the fast kernels take every problem of the class and sort them into tiers themselves */
#define SPX_FAST(G_, C_, WF_, WB_)                                                                                          \
    {                                                                                                                       \
        const int ppw = 64 / G_;                                                                                            \
        const int blocks = (B->n_order + ppw - 1) / ppw, blocks_b = (B->n_order_bwd + ppw - 1) / ppw;                       \
        if (phase != 1 && blocks > 0) hipLaunchKernelGGL((fast_fwd_kernel<G_, C_, 0, WF_>), dim3(blocks), dim3(64), 0, st, P, *K);   \
        if (phase != 0 && blocks_b > 0) hipLaunchKernelGGL((fast_bwd_kernel<G_, C_, 0, WB_>), dim3(blocks_b), dim3(64), 0, st, P, *K); \
    }                                                                                                                       \
    break;
    /* a class' launch may go out in several pieces (SPX_FAST_PIECES, default 1): between two pieces the chip drains for a moment and the
     * kernels of the next list's preparation, queued on other streams, get slots (spx_kernels.hip SPX_DP_PIECES; measured with the fast
     * kernels: 4 pieces -2 %, 8 and 16 +-0) */
    static const int n_pieces = [] { const char *e = getenv("SPX_FAST_PIECES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    auto pieces = [&](int blocks, int ppw, bool bwd, auto &&launch) {
        if (blocks <= 0) return;
        int per = (blocks + n_pieces - 1) / n_pieces;
        if (per < 2048) per = blocks < 2048 ? blocks : 2048;
        for (int b0 = 0; b0 < blocks; b0 += per) {
            spx_dev_batch Q = P;
            if (bwd) { Q.order_bwd += (int64_t)b0 * ppw; Q.n_order_bwd -= b0 * ppw; }
            else { Q.order += (int64_t)b0 * ppw; Q.n_order -= b0 * ppw; }
            launch(Q, blocks - b0 < per ? blocks - b0 : per);
        }
    };
    /* The exact band widths of the HiFi preset (W = 41, 43, 45, 47): TWO lanes per problem, C = (W + 1) / 2 slots each, W known at compile time
     * (one slot beyond the band in the second lane).  ONE lane per problem -- no lane exchange, no carry, 28 % fewer instructions per cell --
     * was measured too: its 2 x W doubles of state fill the register file at two waves per SIMD, the forward kernel spills its bookkeeping
     * (54 spill instructions, one reload per row) and ran 11.4 ms against 9.1 for 32 768 groups; the backward kernel (no spills) 6.55 against
     * 6.82 once both were store-only: +0.8 % on the step, within the boxes' spread -- not kept.  Waves per SIMD: forward W = 41 four (128 VGPRs), every other kernel three. */
#define SPX_FAST_HIFI(W_, C2_, WF_)                                                                                         \
    {                                                                                                                       \
        if (phase != 1 && B->n_order > 0)                                                                                   \
            pieces((B->n_order + 31) / 32, 32, false, [&](const spx_dev_batch &Q, int nb) { hipLaunchKernelGGL((fast_fwd_kernel<2, C2_, W_, WF_>), dim3(nb), dim3(64), 0, st, Q, *K); }); \
        if (phase != 0 && B->n_order_bwd > 0)                                                                               \
            pieces((B->n_order_bwd + 31) / 32, 32, true, [&](const spx_dev_batch &Q, int nb) { hipLaunchKernelGGL((fast_bwd_kernel<2, C2_, W_, SPX_FAST_HIFI_WAVES>), dim3(nb), dim3(64), 0, st, Q, *K); }); \
    }                                                                                                                       \
    break;
    switch (cls) {
    case 0: SPX_FAST_HIFI(41, 21, 4)
    case 1: SPX_FAST_HIFI(43, 22, 3)
    case 2: SPX_FAST_HIFI(45, 23, 3)
    case 3: SPX_FAST_HIFI(47, 24, 3)
    case 4: SPX_FAST(2, 24, 3, 2)
    case 5: SPX_FAST(4, 16, 4, 3)
    case 6: SPX_FAST(4, 26, 3, 2)
    case 12: SPX_FAST(4, 28, 3, 2)
    case 13: SPX_FAST(4, 30, 2, 2)
    default: return hipErrorInvalidValue;
    }
#undef SPX_FAST
#undef SPX_FAST_HIFI
    return hipGetLastError();
}
