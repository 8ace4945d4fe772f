/*
 * spx_inflate_kernels.hip -- BGZF inflate on gfx950 (a BGZF block is an independent DEFLATE stream of at most 64 KB).
 *
 * What replaces what: htslib inflates every block on the reading thread (bgzf_read_block under sam_read1,
 * /root/reference/programs/src/secphase.c:268); the host reader of spx_io.cpp does it on a thread pool.  The MI355X boxes
 * give a container ~16 cores of CPU time, which caps host inflate at ~10 GB/s of inflated bytes (~150 k HiFi groups/s)
 * beside a device that scores 800 k groups/s -- so the compressed bytes cross PCIe (26 KB per group instead of 57) and
 * are inflated here.
 *
 * Three generations live in this file (DESIGN.md 3.4 has the measurements; tools/inflate_bench.py runs each of them alone):
 *   1. bgzf_inflate_kernel<LR,DR>            round 3: ONE WAVEFRONT PER BLOCK, the decode on the scalar unit (below)         16.7 GB/s
 *   2. bgzf_inflate_g_kernel<G,LR,DR,R,W>    round 4: 64 / G blocks per wavefront, the decode on the vector ALU, ring in LDS  26 GB/s
 *   3. bgzf_decode2_kernel<G,W> + bgzf_resolve_kernel + bgzf_crc_kernel   round 4, the default: Huffman decode (literals placed,
 *      matches left as holes), LZ77 copies 64 at a time, CRC -- spx_launch_bgzf_inflate2                                     50 GB/s
 * 1 and 2 run the decoder core of spx_inflate.h (shared with the host build that the CPU tests check against zlib); 3 has a
 * decoder of its own, written for the lane group (TokDec).
 *
 * Generation 1, mapping onto a wavefront:
 *   - the bit-serial Huffman decode is WAVE-UNIFORM: every lane executes the same decode of the same block, so the
 *     compiler keeps bit buffer, counters and table indices in SGPRs / on the scalar unit; the root tables (11-bit
 *     literal/length, 9-bit distance) live in LDS and are read back through v_readfirstlane;
 *   - the 64 lanes share the data-parallel parts: zeroing and filling the decode tables, LZ77 copies (lane i copies
 *     bytes i, i + 64, ...: an overlapping match is a repeat of its first `dist` bytes, so the lanes are independent),
 *     flushing finished output to HBM, and the CRC-32 (64 stripes, combined with the GF(2) shift operator);
 *   - every output byte goes through an LDS ring of the last kRing bytes: near matches (distance <= kRing / 2 = 1 KB, the
 *     bulk of what zlib finds in quality strings) never touch HBM; far matches (a secondary alignment repeats the
 *     primary's SEQ / QUAL ~23 KB back) read what earlier flushes wrote (flush = stores + agent-scope fence; the
 *     loads bypass the vector L1).
 * LDS per wave: tables 6.7 KB + ring 2 KB (the CRC's byte table re-uses the literal table): 18 waves per CU.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "spx_inflate.h"

namespace {

constexpr int kRing = 2048;
constexpr int kFlush = 512; /* flush when this many bytes wait in the ring (kFlush + 64 + 258 <= kRing / 2) */

struct BlockDesc {
    int64_t in_off;   /* byte offset of the DEFLATE data in the compressed buffer */
    int64_t out_off;  /* byte offset of the block's inflated bytes in the output buffer */
    uint32_t clen, ulen, crc, pad;
};

/* lane `n` of v <- the wave-uniform value c (v_writelane_b32; on gfx9 the lane select shares the constant bus with the
 * value, so it travels through M0) */
__device__ __forceinline__ int writelane(int v, int c, int n)
{
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(c), "s"(n) : "m0");
    return v;
}

template <int LR, int DR>
struct DevEnv {
    static constexpr int kLit = LR, kDist = DR;
    using Tab = spxz::TablesT<LR, DR>;
    const uint32_t *in_al;
    uint32_t in_shift; /* bits: 0, 8, 16, 24 */
    uint32_t n_dw;     /* dwords of in_al that hold bytes of this block: reads beyond them return 0 (the decoder's contract; a damaged
                        * stream behind valid BGZF framing may ask for ~6 bytes per output byte before the end-of-block check stops it) */
    uint8_t *out;
    uint32_t limit;    /* bytes the block may produce (ISIZE): nothing is written to HBM beyond it */
    uint32_t pos, flushed;
    uint32_t npend;    /* literals decoded but not yet in the ring: byte k of the run sits in lane k of litv */
    int litv;
    Tab *T;
    uint8_t *ring;
    int lane_;

    __device__ __forceinline__ uint32_t in32(uint32_t k) const
    {
        const uint32_t lo = k < n_dw ? in_al[k] : 0u, hi = k + 1 < n_dw ? in_al[k + 1] : 0u;
        return in_shift ? (lo >> in_shift) | (hi << (32 - in_shift)) : lo;
    }
    __device__ __forceinline__ Tab &tables() { return *T; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ int lane() const { return lane_; }
    __device__ __forceinline__ int lanes() const { return 64; }
    __device__ __forceinline__ int uniform(int v) const { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ uint32_t uniform_u32(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    /* pos = bytes COMMITTED to the ring; the literals in litv come on top */
    __device__ __forceinline__ uint32_t out_pos() const { return pos + npend; }

    /* pending literals (one per lane) -> ring: ONE LDS store for up to 64 bytes */
    __device__ __forceinline__ void commit()
    {
        if (npend) {
            if ((uint32_t)lane_ < npend) ring[(pos + (uint32_t)lane_) & (kRing - 1)] = (uint8_t)litv;
            pos += npend;
            npend = 0;
        }
    }
    /* ring -> HBM, all lanes; afterwards every byte below `pos` is visible to loads that bypass the vector L1.
     * (The loops below have wave-uniform trip counts with the lane test inside: a divergent loop bound would make the
     * compiler structurize the whole decoder loop around it.) */
    __device__ __forceinline__ void flush()
    {
        commit();
        __syncthreads();
        const uint32_t end = pos < limit ? pos : limit;
        for (uint32_t base = flushed; base < end; base += 64) {
            const uint32_t i = base + (uint32_t)lane_;
            if (i < end) out[i] = ring[i & (kRing - 1)];
        }
        flushed = pos;
        __threadfence();
    }
    /* the literal loop of the decoder: append (the decoder is wave-uniform: the byte is an SGPR value, v_writelane drops
     * it into lane npend) ... */
    __device__ __forceinline__ bool lit_full() const { return npend == 64; }
    __device__ __forceinline__ void lit_push(uint8_t c)
    {
        litv = writelane(litv, (int)c, (int)npend);
        ++npend;
    }
    /* ... and make room: 64 literals into the ring, on to HBM when enough has gathered; false = the block overruns */
    __device__ __forceinline__ bool lit_commit()
    {
        commit();
        if (pos - flushed >= (uint32_t)kFlush) flush();
        return pos <= limit;
    }
    __device__ __forceinline__ bool put_literal(uint8_t c) /* (outside the literal loop: stored blocks, codes beyond the root table) */
    {
        if (npend == 64 && !lit_commit()) return false; /* the loop may leave the register full */
        lit_push(c);
        return true;
    }
    __device__ __forceinline__ void copy_match(int len, int dist)
    {
        commit();
        const uint32_t src0 = pos - (uint32_t)dist;
        __syncthreads();
        if (dist <= kRing / 2) {
            /* near: ring -> ring.  Lane i produces bytes i, i + 64, ...; source byte = first `dist` bytes of the match
             * region repeated, all of them older than pos: no lane depends on another */
            for (int base = 0; base < len; base += 64) {
                const int i = base + lane_;
                if (i < len) {
                    const uint32_t s = src0 + (uint32_t)(dist >= len ? i : i % dist);
                    ring[(pos + (uint32_t)i) & (kRing - 1)] = ring[s & (kRing - 1)];
                }
            }
        } else {
            /* far: the source lies below `flushed` (kFlush + 64 + 258 <= kRing / 2), read it from HBM past the L1 */
            for (int base = 0; base < len; base += 64) {
                const int i = base + lane_;
                if (i < len) {
                    const uint8_t v = __hip_atomic_load(out + src0 + (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ring[(pos + (uint32_t)i) & (kRing - 1)] = v;
                }
            }
        }
        pos += (uint32_t)len;
        if (pos - flushed >= (uint32_t)kFlush) flush();
        else __syncthreads();
    }
};

template <int LR, int DR>
__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const uint8_t *__restrict__ comp, const BlockDesc *__restrict__ blocks, int32_t n_blocks,
                                                          uint8_t *__restrict__ outbuf, int32_t *__restrict__ status, int check_crc)
{
    __shared__ spxz::TablesT<LR, DR> T;
    __shared__ uint8_t ring[kRing];
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    DevEnv<LR, DR> env;
    env.in_al = reinterpret_cast<const uint32_t *>(comp + (d.in_off & ~(int64_t)3));
    env.in_shift = (uint32_t)(d.in_off & 3) * 8u;
    env.n_dw = ((uint32_t)(d.in_off & 3) + d.clen + 3u) >> 2;
    env.out = outbuf + d.out_off;
    env.limit = d.ulen;
    env.pos = 0;
    env.flushed = 0;
    env.npend = 0;
    env.litv = 0;
    env.T = &T;
    env.ring = ring;
    env.lane_ = (int)threadIdx.x;
    int rc = 0;
    if (d.ulen > 0) {
        rc = spxz::inflate_stream(env, (int64_t)d.clen * 8, d.ulen);
        env.flush();
    }
    rc = __builtin_amdgcn_readfirstlane(rc);
    if (rc == 0 && check_crc && d.ulen > 0) {
        /* CRC-32 of the block: 64 stripes (byte-table recurrence per lane), folded with the GF(2) shift operator */
        /* (the decode tables are dead now: the byte table of the CRC takes the literal table's place) */
        uint32_t *crc_tab = reinterpret_cast<uint32_t *>(T.lit);
        __syncthreads();
        for (int k = (int)threadIdx.x; k < 256; k += 64) crc_tab[k] = spxz::crc_table_entry((uint32_t)k);
        __syncthreads();
        const uint32_t n = d.ulen, step = (n + 63) / 64;
        const uint32_t a = min(n, step * threadIdx.x), e = min(n, a + step);
        uint32_t c = 0xffffffffu;
        const uint8_t *p = env.out;
        for (uint32_t k = a; k < e; ++k) c = crc_tab[(c ^ p[k]) & 0xff] ^ (c >> 8);
        c ^= 0xffffffffu;
        uint32_t len = e - a;
        /* tree: lane l absorbs lane l + s (the bytes that FOLLOW its own) */
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t oc = (uint32_t)__shfl_down((int)c, s), ol = (uint32_t)__shfl_down((int)len, s);
            if ((threadIdx.x & (2 * s - 1)) == 0) {
                if (ol > 0) c = len > 0 ? spxz::crc_combine(c, oc, ol) : oc;
                len += ol;
            }
        }
        if (threadIdx.x == 0 && c != d.crc) rc = -4;
        rc = __builtin_amdgcn_readfirstlane(rc);
    }
    if (threadIdx.x == 0) status[b] = rc;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Round 4: SEVERAL BLOCKS PER WAVEFRONT.  The kernel above decodes on the scalar unit -- one per CU, shared by every resident
 * wave: 45 scalar instructions per symbol (literals ~22, the 13 % of symbols that are matches the other half) at 60 % of
 * its issue slots is where 13.5 GB/s comes from, and with the input side on the device (spx_devin.cpp) that kernel is what
 * an end-to-end run waits for.  Here a wavefront holds 64 / G blocks: G adjacent lanes share one block and execute its
 * decode REDUNDANTLY on the vector ALU (four SIMDs per CU, two wave-instructions per cycle), so that one instruction
 * stream advances 64 / G blocks; the G lanes of a block split what is data parallel (table fills, LZ77 copies, flushes,
 * CRC stripes) exactly as the 64 lanes do above.  Control flow diverges between the blocks of a wave -- a match here, a
 * literal there -- and is handled by the hardware's execution mask: zlib cuts its DEFLATE blocks after a fixed number of
 * symbols, so the blocks of a wave reach their table builds together.  Decoder core: the same spx_inflate.h. */

template <int G, int LR, int DR, int R>
struct GrpEnv {
    static constexpr int kLit = LR, kDist = DR;
    /* near matches (distance <= R / 2) are served from the ring; a far match must find its source flushed: R / 2 - 258 >= kFlushG + 3 */
    static constexpr int kRingG = R, kFlushG = (R / 4 < R / 2 - 264) ? R / 4 : R / 2 - 264;
    using Tab = spxz::TablesT<LR, DR>;
    const uint32_t *in_al;
    uint32_t in_shift, n_dw;
    uint8_t *out;
    uint32_t limit, pos, flushed;
    Tab *T;
    uint8_t *ring;
    int lane_;

    __device__ __forceinline__ uint32_t in32(uint32_t k) const
    {
        const uint32_t lo = k < n_dw ? in_al[k] : 0u, hi = k + 1 < n_dw ? in_al[k + 1] : 0u;
        return in_shift ? (lo >> in_shift) | (hi << (32 - in_shift)) : lo;
    }
    __device__ __forceinline__ Tab &tables() { return *T; }
    /* the lanes of a block run in lockstep inside one wave and LDS operations of a wave complete in order: a "barrier" only
     * has to keep the compiler from moving LDS accesses across it */
    __device__ __forceinline__ void sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ int lane() const { return lane_; }
    __device__ __forceinline__ int lanes() const { return G; }
    /* a value only lane 0 of the block has computed (return codes of its serial parts); what every lane READS is the same anyway */
    __device__ __forceinline__ int uniform(int v) const { return __shfl(v, 0, G); }
    __device__ __forceinline__ uint32_t uniform_u32(uint32_t v) const { return v; }
    __device__ __forceinline__ uint32_t out_pos() const { return pos; }

    /* ring -> HBM in whole dwords (the ring's bytes [flushed, pos) wait; `flushed` stays a multiple of 4 until the last
     * flush, so the LDS reads are aligned; the global stores are byte-aligned, which gfx9 global memory allows) */
    __device__ __forceinline__ void flush(bool last)
    {
        sync();
        const uint32_t end = pos < limit ? pos : limit;
        const uint32_t e4 = last ? end : (end & ~3u);
        if (e4 > flushed) {
            for (uint32_t base = flushed; base + 4 <= e4; base += 4u * G) {
                const uint32_t i = base + 4u * (uint32_t)lane_;
                if (i + 4 <= e4) {
                    const uint32_t v = *reinterpret_cast<const uint32_t *>(ring + (i & (kRingG - 1)));
                    __builtin_memcpy(out + i, &v, 4);
                }
            }
            const uint32_t tail = flushed + ((e4 - flushed) & ~3u);
            if (tail + (uint32_t)lane_ < e4) out[tail + lane_] = ring[(tail + lane_) & (kRingG - 1)];
            flushed = last ? e4 : (pos < limit ? e4 : pos & ~3u);
        } else if (pos > limit)
            flushed = pos & ~3u;
        /* far matches read what was stored here: the stores must have reached the L2 (the loads go past the vector L1) */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    __device__ __forceinline__ bool lit_full() const { return pos - flushed >= (uint32_t)kFlushG; }
    __device__ __forceinline__ void lit_push(uint8_t c)
    {
        if (lane_ == 0) ring[pos & (kRingG - 1)] = c;
        ++pos;
    }
    __device__ __forceinline__ bool lit_commit()
    {
        flush(false);
        return pos <= limit;
    }
    __device__ __forceinline__ bool put_literal(uint8_t c)
    {
        if (lit_full() && !lit_commit()) return false;
        lit_push(c);
        return true;
    }
    __device__ __forceinline__ void copy_match(int len, int dist)
    {
        const uint32_t src0 = pos - (uint32_t)dist;
        if (dist <= kRingG / 2) {
            if (dist >= len) {
                for (int base = 0; base < len; base += G) {
                    const int i = base + lane_;
                    if (i < len) ring[(pos + (uint32_t)i) & (kRingG - 1)] = ring[(src0 + (uint32_t)i) & (kRingG - 1)];
                }
            } else { /* overlapping: the first `dist` bytes repeated */
                for (int base = 0; base < len; base += G) {
                    const int i = base + lane_;
                    if (i < len) ring[(pos + (uint32_t)i) & (kRingG - 1)] = ring[(src0 + (uint32_t)(i % dist)) & (kRingG - 1)];
                }
            }
        } else {
            /* far: the source lies below `flushed` (kFlushG + 4 + 258 <= kRingG / 2) */
            for (int base = 0; base < len; base += G) {
                const int i = base + lane_;
                if (i < len) {
                    const uint8_t v = __hip_atomic_load(out + src0 + (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ring[(pos + (uint32_t)i) & (kRingG - 1)] = v;
                }
            }
        }
        pos += (uint32_t)len;
        if (pos - flushed >= (uint32_t)kFlushG) flush(false);
    }
};

template <int G, int LR, int DR, int R, int WPE = 4>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void bgzf_inflate_g_kernel(const uint8_t *__restrict__ comp, const BlockDesc *__restrict__ blocks, int32_t n_blocks,
                                                            uint8_t *__restrict__ outbuf, int32_t *__restrict__ status, int check_crc)
{
    constexpr int NB = 64 / G;
    __shared__ spxz::TablesT<LR, DR> T[NB];
    __shared__ __attribute__((aligned(16))) uint8_t ring[NB][R];
    const int q = (int)threadIdx.x / G, lane = (int)threadIdx.x % G;
    const int b = (int)blockIdx.x * NB + q;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    GrpEnv<G, LR, DR, R> env;
    env.in_al = reinterpret_cast<const uint32_t *>(comp + (d.in_off & ~(int64_t)3));
    env.in_shift = (uint32_t)(d.in_off & 3) * 8u;
    env.n_dw = ((uint32_t)(d.in_off & 3) + d.clen + 3u) >> 2;
    env.out = outbuf + d.out_off;
    env.limit = d.ulen;
    env.pos = 0;
    env.flushed = 0;
    env.T = &T[q];
    env.ring = ring[q];
    env.lane_ = lane;
    int rc = 0;
    if (d.ulen > 0) {
        rc = spxz::inflate_stream(env, (int64_t)d.clen * 8, d.ulen);
        env.flush(true);
    }
    if (rc == 0 && check_crc && d.ulen > 0) {
        /* CRC-32: G stripes per block (byte-table recurrence per lane), folded with the GF(2) shift operator; the byte table
         * takes the place of the block's literal table */
        uint32_t *crc_tab = reinterpret_cast<uint32_t *>(T[q].lit);
        env.sync();
        for (int k = lane; k < 256; k += G) crc_tab[k] = spxz::crc_table_entry((uint32_t)k);
        env.sync();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t n = d.ulen, step = (n + G - 1) / G;
        const uint32_t a = min(n, step * (uint32_t)lane), e = min(n, a + step);
        uint32_t c = 0xffffffffu;
        const uint8_t *p = env.out;
        for (uint32_t k = a; k < e; ++k) {
            const uint8_t byte = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = crc_tab[(c ^ byte) & 0xff] ^ (c >> 8);
        }
        c ^= 0xffffffffu;
        uint32_t len = e - a;
        for (int s = 1; s < G; s <<= 1) {
            const uint32_t oc = (uint32_t)__shfl_down((int)c, s, G), ol = (uint32_t)__shfl_down((int)len, s, G);
            if ((lane & (2 * s - 1)) == 0) {
                if (ol > 0) c = len > 0 ? spxz::crc_combine(c, oc, ol) : oc;
                len += ol;
            }
        }
        c = (uint32_t)__shfl((int)c, 0, G);
        if (c != d.crc) rc = -4;
    }
    if (lane == 0) status[b] = rc;
}


/* CRC-32 of every inflated block against the BGZF trailer's: one wavefront per block, 64 stripes of 16-byte loads, four bytes per step
 * through four 256-entry tables (slicing by 4: the look-ups of a dword do not depend on each other), the stripes folded with the GF(2)
 * shift operator; status[b] 0 -> -4 on a mismatch */
__global__ __launch_bounds__(256) void bgzf_crc_kernel(const BlockDesc *__restrict__ blocks, int32_t n_blocks, const uint8_t *__restrict__ outbuf,
                                                       int32_t *__restrict__ status)
{
    __shared__ uint32_t tab[4][256]; /* tab[k][x] = CRC of byte x followed by k zero bytes */
    {
        const uint32_t t0 = spxz::crc_table_entry(threadIdx.x);
        tab[0][threadIdx.x] = t0;
        __syncthreads();
        uint32_t c = t0;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            c = tab[0][c & 0xffu] ^ (c >> 8);
            tab[k][threadIdx.x] = c;
        }
        __syncthreads();
    }
    const int b = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    if (d.ulen == 0 || status[b] != 0) return;
    const uint8_t *p = outbuf + d.out_off;
    /* stripes of whole 16-byte pieces (aligned in memory), the unaligned head goes to lane 0, the tail to the last stripe */
    const uint32_t n = d.ulen;
    const uint32_t head = min(n, (uint32_t)((16u - (uint32_t)((uintptr_t)p & 15u)) & 15u));
    const uint32_t pieces = (n - head) / 16u, per = (pieces + 63u) / 64u;
    const uint32_t a_pc = min(pieces, per * (uint32_t)lane), e_pc = min(pieces, a_pc + per);
    const uint32_t a = head + 16u * a_pc, e = head + 16u * e_pc;
    uint32_t c = 0xffffffffu;
    uint32_t len = e - a;
    if (lane == 0) {
        for (uint32_t k = 0; k < head; ++k) c = tab[0][(c ^ p[k]) & 0xff] ^ (c >> 8);
        len += head;
    }
    const uint4 *p16 = reinterpret_cast<const uint4 *>(p + a);
    for (uint32_t k = 0; k < e_pc - a_pc; ++k) {
        const uint4 w = p16[k];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c ^= ws[j];
            c = tab[3][c & 0xffu] ^ tab[2][(c >> 8) & 0xffu] ^ tab[1][(c >> 16) & 0xffu] ^ tab[0][c >> 24];
        }
    }
    /* the last stripe that has bytes takes the tail; with no whole piece at all that is lane 0 */
    const uint32_t tail0 = head + 16u * pieces;
    const uint32_t last_lane = pieces ? (pieces - 1u) / per : 0u;
    if ((uint32_t)lane == last_lane) {
        for (uint32_t k = tail0; k < n; ++k) c = tab[0][(c ^ p[k]) & 0xff] ^ (c >> 8);
        len += n - tail0;
    }
    c ^= 0xffffffffu;
    /* tree: lane l absorbs lane l + s (the bytes that FOLLOW its own): crc(A || B) = x^(8 |B|) * crc(A) + crc(B).  Every lane carries the
     * operator of its own length along (one exponentiation per lane, then one multiplication per level) instead of raising x to the
     * partner's length at every level: the exponentiations were two thirds of this kernel */
    uint32_t pw = spxz::gf2_xpow8n(len);
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t oc = (uint32_t)__shfl_down((int)c, s), ol = (uint32_t)__shfl_down((int)len, s), opw = (uint32_t)__shfl_down((int)pw, s);
        if ((lane & (2 * s - 1)) == 0) {
            if (ol > 0) c = len > 0 ? (spxz::gf2_mul(opw, c) ^ oc) : oc;
            len += ol;
            pw = spxz::gf2_mul(pw, opw);
        }
    }
    if (lane == 0 && c != d.crc) status[b] = -4;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Round 4, second step: DECODE AND COPY IN TWO KERNELS.  In BAM payload one symbol in eight is a match, 95 % of the matches are
 * 3-4 bytes long and four out of five lie further back than a ring in LDS can reach (tools/deflate_stats.cpp): every one of them
 * was a trip to the L2 -- or, with 10 000 blocks x 32 KB of window in flight, to HBM -- in the middle of a chain that is
 * sequential by nature.  The Huffman decoder does not need the bytes a match copies, only its length:
 *   kernel 1 (bgzf_decode2_kernel) decodes the symbols of a block and writes every LITERAL at its final position of the output
 *     (it knows the position: literals count 1, matches their length).  A match leaves a hole of >= 3 bytes: the hole's first
 *     three bytes take the match itself (distance - 1 in 16 bits, length - 3 in 8), and a bitmap (1 bit per output byte, 8 KB
 *     per block) marks where holes start.  No window, no ring, no flush: LDS holds the decode tables and 256 bytes of staged
 *     input per block, nothing in the symbol loop waits for global memory;
 *   kernel 2 (bgzf_resolve_kernel) fills the holes, one wavefront per block, 64 matches at a time: a match may be copied as soon
 *     as its source lies below the first hole that is still open (in the batch the holes are ordered; the first open one always
 *     qualifies), so a batch takes a few rounds of 64-wide gathers instead of 64 dependent trips; long matches (a secondary
 *     alignment repeating the primary's SEQ / QUAL) are copied by the whole wavefront when their turn has come;
 *   kernel 3 (bgzf_crc_kernel) checks the CRC-32.
 */
constexpr int kBitmapWords = 2048; /* per block: 65536 positions / 32 */

/* ---------------------------------------------------------------------------------------------------------------------
 * The decode kernel's own DEFLATE decoder (the kernels above run the shared core of spx_inflate.h; this one is the whole stream, written
 * for the lane group):
 *   - ONE bit reader everywhere: two input dwords in registers, v_alignbit_b32 for the next 32 bits, input staged through LDS in
 *     128-byte chunks;
 *   - block headers in PARALLEL: the code-length code through a 128-entry table, the canonical order of the 286 + 30 symbols by
 *     ballots (rank among the symbols of one length = population count of the lanes before me), the root tables filled by all
 *     lanes -- the serial version of these loops was a third of the kernel's vector instructions;
 *   - literals gather in a register (lane n takes the n-th) and leave as ONE store per run;
 *   - LDS per block: 2.2 KB of tables (the code lengths of a header live where the literal table is built afterwards) + 256
 *     bytes of input.
 * Same contract as the core: 0, -1 corrupt stream, -2 output overrun, -3 input overrun. */
template <int G>
struct TokDec {
    static constexpr int LR = 9, DR = 8;
    static constexpr int kNW = 32 / G;             /* dwords of a 128-byte input chunk per lane */
    static constexpr int kNC = (288 + G - 1) / G;  /* literal/length symbols per lane */
    static constexpr int kND = (32 + G - 1) / G;
    static_assert(G == 32 || G == 16, "lanes per block");
    struct Lds {
        union {
            uint16_t lit[1 << LR];
            uint8_t lens[320];           /* a header's code lengths: dead once the symbols are in canonical order */
        };
        union {
            uint16_t dist[1 << DR];
            struct {
                uint8_t dlens[32];       /* the distance code lengths, set aside while the literal table is built over `lens` */
                uint16_t first[16], offs[16]; /* canonical code / index in `sorted` of every length's first symbol */
                uint8_t cltab[128];      /* the code-length code: (length << 5) | symbol by the next 7 bits, 0xff = no code */
            } h;
        };
        uint16_t lit_sorted[288], dist_sorted[32]; /* (length << 9) | symbol in (length, symbol) order */
        /* codes longer than the root tables: limit[l] = (first code of length l + number of codes) << (15 - l), i.e. the first 15 bits
         * of the stream (most significant first) lie below limit[l] for every code of at most l bits; base[l] = index in `sorted` of
         * the length's first symbol - its first code */
        uint16_t lit_limit[16], dist_limit[16];
        int16_t lit_base[16], dist_base[16];
        uint32_t ibuf[64];
        uint32_t bm_word, bm_idx; /* the bitmap word being gathered: bits of output positions [32 * bm_idx, 32 * bm_idx + 32).  (In LDS, not in
                                   * registers: a match is one symbol in eight, and at seven waves per SIMD the compiler kept these two in scratch
                                   * memory -- a load the match path then waited for) */
    };
    const uint8_t *in;
    uint32_t clen, k_end;
    uint8_t *out;
    uint32_t *bitmap;        /* of the launch; this block's words start at bm_at() (recomputed where needed: one register less in the symbol loop) */
    __device__ __forceinline__ static uint32_t bm_at() { return ((uint32_t)blockIdx.x * (64u / G) + (uint32_t)threadIdx.x / G) * (uint32_t)kBitmapWords; }
    uint32_t limit, pos;
    uint32_t staged;
    uint32_t stage_v[kNW];
    uint32_t w0, w1, k, off; /* the reader: w0 = input dword k, w1 = dword k + 1, `off` bits of w0 consumed (may pass 32: step()) */
    uint32_t lit_room;       /* (see build()) */
    uint32_t npend;          /* literals waiting in litv: the n-th in lane n; they end at `pos` */
    uint32_t litv;
    Lds *L;
    int lane;

    /* table entries (16 bits, read sign-extended: a literal is >= 0):
     *   literal           bits 0-7 the byte, bits 8-11 code length
     *   length / end      bit 15, bits 0-7 length base - 3, bits 8-11 code length, bits 12-14 number of extra bits (7 = end of block)
     *   distance          bits 0-1 h, bits 2-5 number of extra bits x, bits 8-11 code length: base = 1 + (h << x)
     *   kNone             bit 15 alone: not a root code */
    static constexpr uint32_t kNone = 0x8000u;
    __device__ __forceinline__ static uint32_t lit_entry(uint32_t nbits, uint32_t sym)
    {
        if (sym < 256u) return sym | (nbits << 8);
        if (sym == 256u) return 0x8000u | (nbits << 8) | (7u << 12);
        return 0x8000u | (uint32_t)(spxz::len_base((int)sym) - 3) | (nbits << 8) | ((uint32_t)spxz::len_extra((int)sym) << 12);
    }
    __device__ __forceinline__ static uint32_t dist_entry(uint32_t nbits, uint32_t sym)
    {
        const uint32_t h = sym < 2u ? sym : 2u + (sym & 1u);
        return h | ((uint32_t)spxz::dist_extra((int)sym) << 2) | (nbits << 8);
    }
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    typedef uint16_t __attribute__((aligned(1))) u16_unaligned;
    __device__ __forceinline__ static void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ uint32_t gballot(bool p) const
    {
        const unsigned long long m = __ballot(p);
        return G == 32 ? (uint32_t)(m >> (threadIdx.x & 32u)) : ((uint32_t)(m >> (threadIdx.x & 48u)) & 0xffffu);
    }
    /* ---- input ---- */
    __device__ __forceinline__ void load_chunk(uint32_t c)
    {
#pragma unroll
        for (int j = 0; j < kNW; ++j) {
            const uint32_t o = 128u * c + 4u * (uint32_t)(lane * kNW + j);
            const uint32_t oc = o < clen ? o : clen; /* (at most the 4 bytes behind the data: the CRC32 of the BGZF trailer) */
            stage_v[j] = *reinterpret_cast<const u32_unaligned *>(in + oc);
        }
    }
    __device__ __forceinline__ void store_chunk(uint32_t c)
    {
#pragma unroll
        for (int j = 0; j < kNW; ++j) {
            const uint32_t o = 128u * c + 4u * (uint32_t)(lane * kNW + j);
            const uint32_t left = clen > o ? clen - o : 0u;
            const uint32_t v = left >= 4u ? stage_v[j] : (stage_v[j] & ((1u << (8u * left)) - 1u));
            L->ibuf[(c & 1u) * 32u + (uint32_t)(lane * kNW + j)] = v;
        }
    }
    /* the reader at bit `p` of the input */
    __device__ __forceinline__ void seek(uint32_t p)
    {
        k = p >> 5;
        off = p & 31u;
        const uint32_t c = k >> 5;
        load_chunk(c);
        store_chunk(c);
        load_chunk(c + 1u);
        store_chunk(c + 1u);
        load_chunk(c + 2u);
        staged = c + 2u;
        lds_fence();
        w0 = L->ibuf[k & 63u];
        w1 = L->ibuf[(k + 1u) & 63u];
    }
    /* off >= 32: on to the next dword; false = beyond the input */
    __device__ __forceinline__ bool step()
    {
        off -= 32u;
        ++k;
        w0 = w1;
        if (((k + 1u) >> 5) >= staged) {
            if (k > k_end) return false; /* (checked once per chunk: up to 128 bytes of zeros beyond the data may be looked at) */
            store_chunk(staged);
            ++staged;
            load_chunk(staged);
            lds_fence();
        }
        w1 = L->ibuf[(k + 1u) & 63u];
        return true;
    }
    __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(w1, w0, off); }
    /* n <= 16 bits; -1 (as uint32) can not be a value: the input is exhausted */
    __device__ __forceinline__ bool bits(uint32_t n, uint32_t *v)
    {
        if (off >= 32u && !step()) return false;
        *v = peek() & ((1u << n) - 1u);
        off += n;
        return true;
    }
    /* ---- output ---- */
    /* (pos = output bytes up to the last match or flush; the literals in litv come on top) */
    __device__ __forceinline__ void push(uint32_t c)
    {
        litv = (uint32_t)lane == npend ? c : litv;
        ++npend;
    }
    __device__ __forceinline__ void flush_literals()
    {
        if (npend) {
            const uint32_t p = pos + (uint32_t)lane;
            if ((uint32_t)lane < npend && p < limit) out[p] = (uint8_t)litv; /* (beyond the block's size: counted, not written) */
            pos += npend;
            npend = 0;
        }
    }
    __device__ __forceinline__ void token(uint32_t len, uint32_t dist)
    {
        const uint32_t w = pos >> 5;
        uint32_t word = L->bm_word;
        const uint32_t idx = L->bm_idx;
        if (w != idx) {
            if (lane == 0 && word) bitmap[bm_at() + idx] = word;
            word = 0;
            if (lane == 0) L->bm_idx = w;
        }
        word |= 1u << (pos & 31u);
        if (lane == 0) L->bm_word = word;
        if (lane == 0) {
            *reinterpret_cast<u16_unaligned *>(out + pos) = (uint16_t)(dist - 1u);
            out[pos + 2] = (uint8_t)(len - 3u);
        }
        pos += len;
    }

    /* ---- canonical order of `n` symbols (their lengths: mine[c] = length of symbol c * G + lane, 0 = unused) into sorted / count,
     * first / offs; -1 for an over-subscribed set or an incomplete one of more than one code.  `used` = coded symbols ---- */
    template <int NC>
    __device__ __forceinline__ int canonical(const uint32_t (&mine)[NC], uint16_t *sorted, uint16_t *limit_, int16_t *base_, int *used, uint32_t *min_len)
    {
        const uint32_t lt = (1u << lane) - 1u;
        int left = 1, bad = 0;
        *min_len = 16;
        uint32_t run = 0, code = 0, n_prev = 0;
        for (uint32_t l = 1; l < 16; ++l) {
            code = (code + n_prev) << 1;
            uint32_t at = run;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const bool m = mine[c] == l;
                const uint32_t mask = gballot(m);
                if (m) sorted[at + (uint32_t)__popc(mask & lt)] = (uint16_t)((l << 9) | (uint32_t)(c * G + lane));
                at += (uint32_t)__popc(mask);
            }
            const uint32_t n_l = at - run;
            if (n_l && *min_len == 16u) *min_len = l;
            left = (left << 1) - (int)n_l;
            if (left < 0) bad = 1;
            if (lane == 0) {
                L->h.first[l] = (uint16_t)code;
                L->h.offs[l] = (uint16_t)run;
                const uint32_t lim = (code + n_l) << (15u - l);
                limit_[l] = (uint16_t)(lim > 0xffffu ? 0xffffu : lim); /* (an over-subscribed set is refused below) */
                base_[l] = (int16_t)((int)run - (int)code);
            }
            run = at;
            n_prev = n_l;
        }
        *used = (int)run;
        if (left > 0 && run > 1u) bad = 1;
        lds_fence();
        return bad ? -1 : 0;
    }
    /* root table from the canonical order: symbol number i has code first[len] + (i - offs[len]), replicated over the high index bits */
    template <bool LIT>
    __device__ __forceinline__ void fill(uint16_t *tab, int root, const uint16_t *sorted, int used)
    {
        constexpr int NR = LIT ? kNC : kND;
        uint32_t packed[LIT ? 1 : NR]; /* entry | start index << 16 | length << 25 */
        /* the distance table goes where first / offs lie: everything is read before anything is written.  (The literal table lies
         * over the code lengths, which are dead by now.) */
        auto one = [&](int i) -> uint32_t {
            if (i >= used) return 0u;
            const uint32_t v = sorted[i], s = v & 511u, ll = v >> 9;
            if ((int)ll > root || (LIT ? s > 285u : s > 29u)) return 0u;
            const uint32_t e = LIT ? lit_entry(ll, s) : dist_entry(ll, s);
            const uint32_t r = __brev((uint32_t)L->h.first[ll] + ((uint32_t)i - (uint32_t)L->h.offs[ll])) >> (32u - ll);
            return e | (r << 16) | (ll << 25) | 0x40000000u; /* (never 0) */
        };
        auto put = [&](uint32_t pk) {
            if (pk)
                for (uint32_t q = (pk >> 16) & 511u; q < (1u << root); q += 1u << ((pk >> 25) & 15u)) tab[q] = (uint16_t)pk;
        };
        if (!LIT) {
#pragma unroll
            for (int c = 0; c < NR; ++c) packed[c] = one(c * G + lane);
            lds_fence();
        }
        uint32_t *t32 = reinterpret_cast<uint32_t *>(tab);
        for (int q = lane; q < (1 << root) / 2; q += G) t32[q] = kNone * 0x10001u;
        lds_fence();
        if (LIT) {
            for (int i = lane; i < used; i += G) put(one(i));
        } else {
#pragma unroll
            for (int c = 0; c < NR; ++c) put(packed[c]);
        }
        lds_fence();
    }
    /* both tables from L->lens[0 .. nlit) and L->lens[nlit .. nlit + ndist) */
    __device__ __forceinline__ int build(int nlit, int ndist)
    {
        for (int j = lane; j < 32; j += G) L->h.dlens[j] = j < ndist ? L->lens[nlit + j] : (uint8_t)0;
        uint32_t mine[kNC];
#pragma unroll
        for (int c = 0; c < kNC; ++c) {
            const int s = c * G + lane;
            mine[c] = s < nlit ? (uint32_t)L->lens[s] : 0u;
        }
        lds_fence();
        int used = 0;
        uint32_t min_len;
        if (canonical<kNC>(mine, L->lit_sorted, L->lit_limit, L->lit_base, &used, &min_len) != 0) return -1;
        /* literals gather in a register of G lanes: a trip of the symbol loop adds up to three and makes room first; at a step of the
         * reader the register is emptied once it is half full (one store per ~16 literals instead of one per literal) */
        lit_room = (uint32_t)G / 2u;
        (void)min_len;
        fill<true>(L->lit, LR, L->lit_sorted, used);
        uint32_t dmine[kND];
#pragma unroll
        for (int c = 0; c < kND; ++c) dmine[c] = (uint32_t)L->h.dlens[c * G + lane];
        lds_fence();
        if (canonical<kND>(dmine, L->dist_sorted, L->dist_limit, L->dist_base, &used, &min_len) != 0) return -1;
        fill<false>(L->dist, DR, L->dist_sorted, used);
        return 0;
    }
    /* a code longer than the root table (ROOT bits), from the next bits of the stream: (length << 16) | symbol, or 0xffffffff.  The
     * first 15 bits, most significant first, are compared with the lengths' limits */
    template <int ROOT>
    __device__ __forceinline__ static uint32_t slow(uint32_t bits, const uint16_t *limit_, const int16_t *base_, const uint16_t *sorted)
    {
        const uint32_t c15 = __brev(bits) >> 17;
        if (c15 < (uint32_t)limit_[ROOT]) return 0xffffffffu; /* a code of <= ROOT bits without a root entry: symbols 286, 287 / 30, 31 */
        uint32_t len = 0;
#pragma unroll
        for (int l = 15; l > ROOT; --l)
            if (c15 < (uint32_t)limit_[l]) len = (uint32_t)l;
        if (!len) return 0xffffffffu;
        const int idx = (int)base_[len] + (int)(c15 >> (15u - len));
        return (len << 16) | ((uint32_t)sorted[idx] & 511u);
    }

    /* ---- a dynamic block's header: the code lengths into L->lens ---- */
    __device__ __forceinline__ int header(int *nlit_out, int *ndist_out)
    {
        uint32_t v;
        if (!bits(14, &v)) return -3;
        const int nlit = (int)(v & 31u) + 257, ndist = (int)((v >> 5) & 31u) + 1, ncode = (int)(v >> 10) + 4;
        if (nlit > 286 || ndist > 30) return -1;
        /* the code-length code: 3 bits per symbol, packed */
        unsigned long long cl = 0;
        {
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#pragma unroll
            for (int q = 0; q < 19; ++q) {
                if (q < ncode) {
                    if (!bits(3, &v)) return -3;
                    cl |= (unsigned long long)v << (3 * order[q]);
                }
            }
        }
        unsigned long long cnt = 0; /* symbols per length, 8 bits each */
#pragma unroll
        for (int j = 0; j < 19; ++j) cnt += 1ull << (8u * (uint32_t)((cl >> (3 * j)) & 7u));
        int left = 1;
        unsigned long long first = 0; /* canonical first code per length, 8 bits each */
        {
            uint32_t code = 0, n_prev = 0;
#pragma unroll
            for (uint32_t l = 1; l < 8; ++l) {
                code = (code + n_prev) << 1;
                first |= (unsigned long long)(code & 0xffu) << (8u * l);
                n_prev = (uint32_t)(cnt >> (8u * l)) & 0xffu;
                left = (left << 1) - (int)n_prev;
                if (left < 0) return -1;
            }
        }
        const int used = 19 - (int)(cnt & 0xffu);
        if (left > 0 && used > 1) return -1;
        for (int q = lane; q < 32; q += G) reinterpret_cast<uint32_t *>(L->h.cltab)[q] = 0xffffffffu;
        lds_fence();
        for (int j = lane; j < 19; j += G) {
            const uint32_t l = (uint32_t)(cl >> (3 * j)) & 7u;
            if (l) {
                uint32_t rank = 0;
#pragma unroll
                for (int q = 0; q < 18; ++q)
                    if (q < j && ((uint32_t)(cl >> (3 * q)) & 7u) == l) ++rank;
                const uint32_t code = ((uint32_t)(first >> (8u * l)) & 0xffu) + rank;
                for (uint32_t q = __brev(code) >> (32u - l); q < 128u; q += 1u << l) L->h.cltab[q] = (uint8_t)((l << 5) | (uint32_t)j);
            }
        }
        lds_fence();
        const int total = nlit + ndist;
        int idx = 0;
        uint32_t prev = 0;
        while (idx < total) {
            if (off >= 32u && !step()) return -3;
            uint32_t b = peek();
            const uint32_t e = L->h.cltab[b & 127u];
            if (e == 0xffu) return -1;
            const uint32_t sym = e & 31u;
            off += e >> 5;
            b >>= e >> 5;
            if (sym < 16u) {
                if (lane == 0) L->lens[idx] = (uint8_t)sym;
                prev = sym;
                ++idx;
                continue;
            }
            int rep;
            uint32_t val = 0;
            if (sym == 16u) {
                if (idx == 0) return -1;
                val = prev;
                rep = 3 + (int)(b & 3u);
                off += 2u;
            } else if (sym == 17u) {
                rep = 3 + (int)(b & 7u);
                off += 3u;
            } else {
                rep = 11 + (int)(b & 127u);
                off += 7u;
            }
            if (idx + rep > total) return -1;
            for (int q = lane; q < rep; q += G) L->lens[idx + q] = (uint8_t)val;
            idx += rep;
            prev = val;
        }
        lds_fence();
        if (L->lens[256] == 0) return -1; /* no end-of-block code */
        *nlit_out = nlit;
        *ndist_out = ndist;
        return 0;
    }

    /* ---- the symbols of one block, to its end-of-block code ---- */
    __device__ __forceinline__ int symbols()
    {
        const int16_t *lit = reinterpret_cast<const int16_t *>(L->lit), *dis = reinterpret_cast<const int16_t *>(L->dist);
        int rc = 1;
        for (;;) {
            if (off >= 32u) {
                if (npend >= lit_room) flush_literals();
                if (!step()) { rc = -3; break; }
            }
            uint32_t b = peek();
            int e = lit[b & ((1u << LR) - 1u)];
            if (e >= 0) { /* a literal -- and, seven times out of eight, another one behind it: the 32 bits at hand hold three codes of <= 9 bits */
                if (npend > (uint32_t)G - 3u) flush_literals(); /* room for three */
                push((uint32_t)e);
                uint32_t used = (uint32_t)e >> 8;
                const int e2 = lit[(b >> used) & ((1u << LR) - 1u)];
                if (e2 >= 0) {
                    push((uint32_t)e2);
                    used += (uint32_t)e2 >> 8;
                    const int e3 = lit[(b >> used) & ((1u << LR) - 1u)];
                    if (e3 >= 0) {
                        push((uint32_t)e3);
                        used += (uint32_t)e3 >> 8;
                    }
                }
                off += used;
                continue;
            }
            uint32_t ue = (uint32_t)e & 0xffffu;
            if (ue == kNone) {
                const uint32_t r = slow<LR>(b, L->lit_limit, L->lit_base, L->lit_sorted);
                if (r == 0xffffffffu) { rc = -1; break; }
                const uint32_t sym = r & 0xffffu;
                if (sym > 285u) { rc = -1; break; }
                if (sym < 256u) {
                    if (npend > (uint32_t)G - 3u) flush_literals();
                    push(sym);
                    off += r >> 16;
                    continue;
                }
                ue = lit_entry(r >> 16, sym);
            }
            uint32_t n = (ue >> 8) & 15u, x = (ue >> 12) & 7u;
            if (x == 7u) {
                off += n;
                rc = 0;
                break;
            }
            const uint32_t len = (ue & 0xffu) + 3u + ((b >> n) & ((1u << x) - 1u));
            off += n + x;
            flush_literals();
            if (off >= 32u && !step()) { rc = -3; break; }
            b = peek();
            uint32_t de = (uint32_t)dis[b & ((1u << DR) - 1u)] & 0xffffu;
            if (de == kNone) {
                const uint32_t r = slow<DR>(b, L->dist_limit, L->dist_base, L->dist_sorted);
                if (r == 0xffffffffu) { rc = -1; break; }
                if ((r & 0xffffu) > 29u) { rc = -1; break; }
                de = dist_entry(r >> 16, r & 0xffffu);
            }
            n = de >> 8;
            x = (de >> 2) & 15u;
            const uint32_t dist = 1u + ((de & 3u) << x) + ((b >> n) & ((1u << x) - 1u));
            off += n + x;
            if (dist > pos) { rc = -1; break; }
            if (pos + len > limit) { rc = -2; break; }
            token(len, dist);
        }
        flush_literals();
        return rc;
    }

    __device__ __forceinline__ int run()
    {
        npend = 0;
        litv = 0;
        if (lane == 0) { L->bm_word = 0; L->bm_idx = 0; }
        pos = 0;
        k_end = ((clen + 3u) >> 2) + 1u;
        seek(0);
        for (;;) {
            uint32_t hdr;
            if (!bits(3, &hdr)) return -3;
            const int final_blk = (int)(hdr & 1u), type = (int)(hdr >> 1);
            if (type == 0) {
                /* stored: LEN, NLEN at the next byte boundary, then LEN bytes straight from the input */
                const uint32_t at = (32u * k + off + 7u) >> 3; /* byte offset of LEN */
                if (at + 4u > clen) return -3;
                const uint32_t len = (uint32_t)in[at] | ((uint32_t)in[at + 1] << 8), nlen = (uint32_t)in[at + 2] | ((uint32_t)in[at + 3] << 8);
                if ((len ^ 0xffffu) != nlen) return -1;
                if (at + 4u + len > clen) return -3;
                if (pos + len > limit) return -2;
                for (uint32_t q = (uint32_t)lane; q < len; q += (uint32_t)G) out[pos + q] = in[at + 4u + q];
                pos += len;
                seek(8u * (at + 4u + len));
            } else if (type == 1 || type == 2) {
                int nlit = 288, ndist = 32;
                if (type == 1) {
                    for (int s = lane; s < 288; s += G) L->lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                    for (int s = lane; s < 32; s += G) L->lens[288 + s] = 5;
                    lds_fence();
                } else {
                    const int rc = header(&nlit, &ndist);
                    if (rc != 0) return rc;
                }
                if (build(nlit, ndist) != 0) return -1;
                const int rc = symbols();
                if (rc != 0) return rc;
            } else
                return -1;
            if ((unsigned long long)32u * k + off > (unsigned long long)clen * 8u) return -3;
            if (final_blk) break;
        }
        lds_fence();
        if (lane == 0 && L->bm_word) bitmap[bm_at() + L->bm_idx] = L->bm_word;
        return pos == limit ? 0 : -2;
    }
};

template <int G, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void bgzf_decode2_kernel(const uint8_t *__restrict__ comp, const BlockDesc *__restrict__ blocks, int32_t n_blocks,
                                                            uint8_t *__restrict__ outbuf, uint32_t *__restrict__ bitmap, int32_t *__restrict__ status)
{
    constexpr int NB = 64 / G;
    __shared__ typename TokDec<G>::Lds lds[NB];
    const int q = (int)threadIdx.x / G, lane = (int)threadIdx.x % G;
    const int b = (int)blockIdx.x * NB + q;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    TokDec<G> dec;
    dec.in = comp + d.in_off;
    dec.clen = d.clen;
    dec.out = outbuf + d.out_off;
    dec.bitmap = bitmap; /* (the block's words: dec.bm_at(), b * kBitmapWords) */
    dec.limit = d.ulen;
    dec.L = &lds[q];
    dec.lane = lane;
    int rc = 0;
    if (d.ulen > 0) rc = dec.run();
    if (lane == 0) status[b] = rc;
}

constexpr int kLongMatch = 16; /* longer matches are copied by the whole wavefront */

__global__ __launch_bounds__(64) void bgzf_resolve_kernel(const BlockDesc *__restrict__ blocks, int32_t n_blocks, uint8_t *__restrict__ outbuf,
                                                          const uint32_t *__restrict__ bitmap, const int32_t *__restrict__ status)
{
    __shared__ uint16_t list[2048 / 3 + 4];
    const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    if (d.ulen == 0 || status[b] != 0) return;
    uint8_t *out = outbuf + d.out_off;
    const uint32_t *bm = bitmap + (size_t)b * kBitmapWords;
    const uint32_t nwords = (d.ulen + 31u) >> 5;
    for (uint32_t w0 = 0; w0 < nwords; w0 += 64) {
        const uint32_t w = w0 + (uint32_t)lane < nwords ? bm[w0 + lane] : 0u;
        const int cnt = __popc(w);
        int incl = cnt;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int t = __shfl_up(incl, s);
            if (lane >= s) incl += t;
        }
        const int total = __shfl(incl, 63);
        if (total == 0) continue;
        __syncthreads(); /* the previous chunk's list is done with */
        {
            uint32_t ww = w;
            int j = incl - cnt;
            while (ww) {
                const int bit = __ffs((int)ww) - 1;
                ww &= ww - 1u;
                list[j++] = (uint16_t)(((w0 + (uint32_t)lane) << 5) + (uint32_t)bit);
            }
        }
        __syncthreads();
        for (int i0 = 0; i0 < total; i0 += 64) {
            const int i = i0 + lane;
            bool pending = i < total;
            uint32_t pos = 0, len = 0, dist = 1;
            if (pending) {
                pos = list[i];
                const uint32_t t0 = out[pos], t1 = out[pos + 1], t2 = out[pos + 2];
                dist = (t0 | (t1 << 8)) + 1u;
                len = t2 + 3u;
            }
            const uint32_t src = pos - dist;
            for (;;) {
                const unsigned long long pm = __ballot(pending);
                if (!pm) break;
                const int first = __ffsll((long long)pm) - 1;
                const uint32_t hwm = (uint32_t)__shfl((int)pos, first), flen = (uint32_t)__shfl((int)len, first);
                if (flen > (uint32_t)kLongMatch) {
                    /* the first open hole is a long one: everything below it is final, the whole wavefront copies it (an overlapping
                     * match repeats its first `dist` bytes) */
                    const uint32_t fsrc = (uint32_t)__shfl((int)src, first), fdist = (uint32_t)__shfl((int)dist, first);
                    for (uint32_t k = (uint32_t)lane; k < flen; k += 64u) {
                        const uint32_t idx = fdist >= flen ? k : k % fdist;
                        out[hwm + k] = __hip_atomic_load(out + fsrc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (lane == first) pending = false;
                } else {
                    const bool ready = pending && (lane == first || (len <= (uint32_t)kLongMatch && src + len <= hwm));
                    if (ready) {
                        for (uint32_t k0 = 0; k0 < len; k0 += 4u) {
                            uint8_t v[4];
#pragma unroll
                            for (uint32_t j = 0; j < 4u; ++j) {
                                const uint32_t k = k0 + j;
                                v[j] = 0;
                                if (k < len) v[j] = __hip_atomic_load(out + src + (dist >= len ? k : k % dist), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
#pragma unroll
                            for (uint32_t j = 0; j < 4u; ++j)
                                if (k0 + j < len) out[pos + k0 + j] = v[j];
                        }
                        pending = false;
                    }
                }
                /* what this round stored is the next round's source: the stores must have reached the L2 (the loads go past the L1) */
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
        }
    }
}

} // namespace

/* the earlier kernels, kept for comparisons (tools/inflate_bench.py) and as the path without scratch memory:
 * SPX_INFLATE_LANES = 32 (default): round 4's first kernel, two blocks per wavefront (9 / 8-bit root tables, 1 KB ring, five waves per SIMD);
 * 64: round 3's kernel, one block per wavefront on the scalar unit (11 / 9-bit root tables) */
extern "C" hipError_t spx_launch_bgzf_inflate(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status,
                                              int check_crc, hipStream_t st)
{
    if (n_blocks <= 0) return hipSuccess;
    static const int lanes = [] { const char *e = getenv("SPX_INFLATE_LANES"); const int v = e ? atoi(e) : 32; return v == 64 ? 64 : 32; }();
    const BlockDesc *bd = (const BlockDesc *)blocks;
    if (lanes == 64) hipLaunchKernelGGL((bgzf_inflate_kernel<11, 9>), dim3((unsigned)n_blocks), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc);
    else hipLaunchKernelGGL((bgzf_inflate_g_kernel<32, 9, 8, 1024, 5>), dim3((unsigned)((n_blocks + 1) / 2)), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc);
    return hipGetLastError();
}

/* the decode kernel's bitmap: 8 KB per block */
extern "C" size_t spx_bgzf_inflate_scratch_bytes(int32_t n_blocks) { return (size_t)(n_blocks > 0 ? n_blocks : 0) * kBitmapWords * 4u; }

/* Inflate `n_blocks` BGZF blocks: decode, copy, CRC (three kernels on `st`).  `scratch`: spx_bgzf_inflate_scratch_bytes(n_blocks) bytes.
 * SPX_INFLATE_TOK = 32 (default: 32 lanes per block, seven waves per SIMD) / 16 (16 lanes per block) / 0 (the one-kernel path above) */
extern "C" hipError_t spx_launch_bgzf_inflate2(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status, int check_crc,
                                               void *scratch, hipStream_t st)
{
    if (n_blocks <= 0) return hipSuccess;
    static const int g = [] { const char *e = getenv("SPX_INFLATE_TOK"); const int v = e ? atoi(e) : 32; return (v == 16 || v == 0) ? v : 32; }();
    if (g == 0 || !scratch) return spx_launch_bgzf_inflate(comp, blocks, n_blocks, out, status, check_crc, st);
    const BlockDesc *bd = (const BlockDesc *)blocks;
    uint32_t *bitmap = (uint32_t *)scratch;
    static const int stage = [] { const char *e = getenv("SPX_INFLATE_TOK_STAGE"); return e ? atoi(e) : 3; }(); /* timing experiments: 1 = decode only, 2 = + copy */
    if (hipMemsetAsync(bitmap, 0, spx_bgzf_inflate_scratch_bytes(n_blocks), st) != hipSuccess) return hipGetLastError();
    if (g == 16) hipLaunchKernelGGL((bgzf_decode2_kernel<16, 4>), dim3((unsigned)((n_blocks + 3) / 4)), dim3(64), 0, st, comp, bd, n_blocks, out, bitmap, status);
    else hipLaunchKernelGGL((bgzf_decode2_kernel<32, 7>), dim3((unsigned)((n_blocks + 1) / 2)), dim3(64), 0, st, comp, bd, n_blocks, out, bitmap, status);
    if (stage >= 2) hipLaunchKernelGGL(bgzf_resolve_kernel, dim3((unsigned)n_blocks), dim3(64), 0, st, bd, n_blocks, out, bitmap, status);
    if (stage >= 3 && check_crc) hipLaunchKernelGGL(bgzf_crc_kernel, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, st, bd, n_blocks, out, status);
    return hipGetLastError();
}
