/*
 * spx_dp_dev.h -- device helpers shared by the DP kernels: spx_kernels.hip (the EXACT tier: bit-exact with the CPU order of
 * operations) and spx_fast_kernels.hip (the FAST tier: same model, FMA / no row sums / certified).  Internal.
 */
#ifndef SPX_DP_DEV_H
#define SPX_DP_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spx_device.h"

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

/* 8 consecutive 4-bit codes starting at nibble address a (any alignment), as one dword */
__device__ __forceinline__ uint32_t fetch8(const uint8_t *__restrict__ pool, int64_t a)
{
    /* a may point a few codes in front of a window: both pools carry a lead pad (codes there are never used) */
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(pool) + (a >> 3);
    const uint32_t lo = p32[0], hi = p32[1];
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(a & 7) * 4u);
}
/* same for an address that is a multiple of 8 nibbles */
__device__ __forceinline__ uint32_t fetch8_aligned(const uint8_t *__restrict__ pool, int64_t a)
{
    return reinterpret_cast<const uint32_t *>(pool)[a >> 3];
}

/* m all ones: a, m zero: b */
__device__ __forceinline__ double select_bits(int32_t m, double a, double b)
{
    uint32_t um = (uint32_t)m;
    asm("" : "+v"(um)); /* keep it a bit mask: otherwise the compiler turns this back into compare + 2 v_cndmask */
    const uint32_t lo = ((uint32_t)__double2loint(a) & um) | ((uint32_t)__double2loint(b) & ~um);
    const uint32_t hi = ((uint32_t)__double2hiint(a) & um) | ((uint32_t)__double2hiint(b) & ~um);
    return __hiloint2double((int)hi, (int)lo);
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

/* per-lane view of one problem */
struct Prob {
    int pid, L, R, bw, nrows, row0;
    int Rt; /* last column of the termination sum / backward start: R, or R - 1 under the "row" reading of the terminal guard (SPX_H_TDROP) */
    int64_t ref0, qry0;
    bool act;
};

template <int G>
__device__ __forceinline__ Prob load_problem(const spx_dev_batch &B, int lane, HmmC &h, int &hasN, bool bwd = false)
{
    constexpr int PPW = 64 / G;
    Prob P;
    const int oslot = blockIdx.x * PPW + lane / G;
    P.pid = bwd ? (oslot < B.n_order_bwd ? B.order_bwd[oslot] : -1) : (oslot < B.n_order ? B.order[oslot] : -1);
    P.L = P.R = P.bw = P.nrows = P.row0 = P.Rt = 0;
    P.ref0 = P.qry0 = 0;
    hasN = 0;
    h = HmmC{0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (P.pid >= 0) {
        P.nrows = B.n_rows[P.pid];
        if (P.nrows <= 0) P.pid = -1; /* no marker row wants a value: nothing observable to compute */
        else if (B.tier && B.tier_want != SPX_TIER_ALL && B.tier[P.pid] != B.tier_want) P.pid = -1; /* two-tier DP: not this pass' problem */
    }
    P.act = P.pid >= 0;
    if (P.act) {
        const int pid = P.pid;
        P.L = B.L[pid]; P.R = B.R[pid]; P.bw = B.bw[pid];
        P.ref0 = B.ref_nib[pid]; P.qry0 = B.qry_nib[pid];
        P.row0 = B.row_off[pid];
        const double *hp = B.hmm + (int64_t)pid * SPX_H_N;
        h.m0 = hp[SPX_H_M0]; h.m1 = hp[SPX_H_M1]; h.m2 = hp[SPX_H_M2]; h.m3 = hp[SPX_H_M3]; h.m4 = hp[SPX_H_M4];
        h.m6 = hp[SPX_H_M6]; h.m8 = hp[SPX_H_M8]; h.e_match = hp[SPX_H_EMATCH]; h.e_mis = hp[SPX_H_EMIS];
        hasN = hp[SPX_H_PAD0] != 0.0; /* host flag: window or query holds an ambiguous base */
        P.Rt = hp[SPX_H_TDROP] != 0.0 ? P.R - 1 : P.R;
    }
    return P;
}

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}

template <int C>
struct NibWin { /* C 4-bit codes */
    static constexpr int NW = (C + 7) / 8;
    uint32_t w[NW];
    __device__ __forceinline__ uint32_t get(int c) const { return (w[c >> 3] >> (4 * (c & 7))) & 0xfu; }
    __device__ __forceinline__ void set(int c, uint32_t v)
    {
        w[c >> 3] = (w[c >> 3] & ~(0xfu << (4 * (c & 7)))) | (v << (4 * (c & 7)));
    }
    __device__ __forceinline__ void shift_down(uint32_t v) /* slot c <- slot c+1, slot C-1 <- v */
    {
#pragma unroll
        for (int k = 0; k < NW - 1; ++k) w[k] = (w[k] >> 4) | (w[k + 1] << 28);
        w[NW - 1] >>= 4;
        set(C - 1, v);
    }
    __device__ __forceinline__ void shift_up(uint32_t v) /* slot c <- slot c-1, slot 0 <- v */
    {
#pragma unroll
        for (int k = NW - 1; k > 0; --k) w[k] = (w[k] << 4) | (w[k - 1] >> 28);
        w[0] = (w[0] << 4) | v;
    }
};
#endif
