/*
 * spx_inflate_kernels.hip -- BGZF inflate on gfx950: ONE WAVEFRONT PER BGZF BLOCK (a block is an independent DEFLATE
 * stream of at most 64 KB), the decoder core of spx_inflate.h.
 *
 * What replaces what: htslib inflates every block on the reading thread (bgzf_read_block under sam_read1,
 * /root/reference/programs/src/secphase.c:268); the host reader of spx_io.cpp does it on a thread pool.  The MI355X boxes
 * give a container ~16 cores of CPU time, which caps host inflate at ~10 GB/s of inflated bytes (~150 k HiFi groups/s)
 * beside a device that scores 780 k groups/s -- so the compressed bytes cross PCIe (26 KB per group instead of 57) and
 * are inflated here.
 *
 * Mapping onto a wavefront:
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
    static constexpr bool kFlat = false;
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
    __device__ __forceinline__ uint32_t in32_fix(uint32_t v, uint32_t) const { return v; }
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
    static constexpr bool kFlat = false;
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
    __device__ __forceinline__ uint32_t in32_fix(uint32_t v, uint32_t) const { return v; }
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


/* ---------------------------------------------------------------------------------------------------------------------
 * Round 4, second step: MORE BLOCKS PER WAVEFRONT, ONE SYMBOL PER TRIP.  What the counters of the kernel above say
 * (profiles/r04_counters_inflate.json): vector ALU and scalar unit are each ~half busy, a wavefront issues one instruction every
 * ~18 cycles -- a latency chain (LDS look-ups, and a trip to the L2 for every match that lies further back than the ring: in BAM
 * payload 79 % of the matches do, most of them 3-4 bytes long), paid per SYMBOL and shared by only two blocks.  Here
 *   - G = 8 (or 4 / 16) lanes per block, so that one instruction stream advances 8 blocks;
 *   - the symbol loop is flat (E::kFlat in spx_inflate.h): one symbol per trip whatever its kind, the hardware's execution mask
 *     runs the literal path and the match path of a trip one after the other -- a block with a match does not wait for the
 *     others' literal runs to end;
 *   - a short far match is DEFERRED: the load from the L2 is issued, the block goes on decoding, the bytes reach the ring when
 *     the next match, the next flush or the end of the block needs them (literals never read the ring);
 *   - the CRC-32 moved into its own kernel (bgzf_crc_kernel below): 64 lanes per block with coalesced reads instead of G serial
 *     stripes at the tail of every block's decode.
 * Decoder core: the same spx_inflate.h. */
template <int G, int LR, int DR, int R, bool FLAT = true>
struct FlatEnv {
    static constexpr int kLit = LR, kDist = DR;
    static constexpr bool kFlat = FLAT;
    static constexpr int kRingG = R, kFlushG = (R / 4 < R / 2 - 264) ? R / 4 : R / 2 - 264;
    using Tab = spxz::TablesT<LR, DR>;
    const uint8_t *in;       /* the block's DEFLATE bytes (any alignment: gfx9 global memory takes unaligned dwords) */
    uint32_t clen;
    uint8_t *out;
    uint32_t limit, pos, flushed;
    uint32_t pend_n, pend_at; /* a deferred far match: byte k of it (k < pend_n <= G) arrives in lane k's pend_v and belongs at ring[pend_at + k] */
    uint32_t pend_v;
    Tab *T;
    uint8_t *ring;
    int lane_;

    /* one unconditional load per refill, completed (the zeros beyond the block's end) when the reader consumes the word: nothing
     * depends on the load while it is in flight.  The clamped address reads at most the 4 bytes after the DEFLATE data: a BGZF
     * block ends in CRC32 + ISIZE, so they belong to the block */
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    __device__ __forceinline__ uint32_t in32(uint32_t k) const
    {
        const uint32_t o = 4u * k < clen ? 4u * k : clen;
        return *reinterpret_cast<const u32_unaligned *>(in + o);
    }
    __device__ __forceinline__ uint32_t in32_fix(uint32_t raw, uint32_t k) const
    {
        const uint32_t o = 4u * k, left = clen > o ? clen - o : 0u;
        return left >= 4u ? raw : (raw & ((1u << (8u * left)) - 1u));
    }
    __device__ __forceinline__ Tab &tables() { return *T; }
    __device__ __forceinline__ void sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ int lane() const { return lane_; }
    __device__ __forceinline__ int lanes() const { return G; }
    __device__ __forceinline__ int uniform(int v) const { return __shfl(v, 0, G); }
    __device__ __forceinline__ uint32_t uniform_u32(uint32_t v) const { return v; }
    __device__ __forceinline__ uint32_t out_pos() const { return pos; }

    __device__ __forceinline__ void retire()
    {
        if (pend_n) {
            if ((uint32_t)lane_ < pend_n) ring[(pend_at + (uint32_t)lane_) & (kRingG - 1)] = (uint8_t)pend_v;
            pend_n = 0;
        }
    }
    __device__ __forceinline__ void flush(bool last)
    {
        retire();
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
        retire();
        const uint32_t src0 = pos - (uint32_t)dist;
        if (dist <= kRingG / 2) {
            if (dist >= len) {
                for (int base = 0; base < len; base += G) {
                    const int i = base + lane_;
                    if (i < len) ring[(pos + (uint32_t)i) & (kRingG - 1)] = ring[(src0 + (uint32_t)i) & (kRingG - 1)];
                }
            } else {
                for (int base = 0; base < len; base += G) {
                    const int i = base + lane_;
                    if (i < len) ring[(pos + (uint32_t)i) & (kRingG - 1)] = ring[(src0 + (uint32_t)(i % dist)) & (kRingG - 1)];
                }
            }
        } else if (len <= G) {
            /* far and short: the source lies below `flushed` (kFlushG + 4 + 258 <= kRingG / 2); the load goes past the vector L1 */
            if (lane_ < len) pend_v = __hip_atomic_load(out + src0 + (uint32_t)lane_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pend_at = pos;
            pend_n = (uint32_t)len;
        } else {
            /* far and long (a secondary alignment repeating the primary's SEQ / QUAL): 16 bytes per lane and round, the four loads
             * of a round in flight together.  (A lane may read up to 3 bytes beyond the match: they lie below pos - dist + 261 < pos.) */
            for (int base = 0; base < len; base += 16 * G) {
                const int i = base + 16 * lane_;
                if (i < len) {
                    uint32_t v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[q] = i + 4 * q < len ? __hip_atomic_load(reinterpret_cast<const uint32_t *>(out + src0 + (uint32_t)(i + 4 * q)), __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT)
                                               : 0u;
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (i + q < len) ring[(pos + (uint32_t)(i + q)) & (kRingG - 1)] = (uint8_t)(v[q >> 2] >> (8 * (q & 3)));
                }
            }
        }
        pos += (uint32_t)len;
        if (pos - flushed >= (uint32_t)kFlushG) flush(false);
    }
};

template <int G, int LR, int DR, int R, int WPE, bool FLAT = true>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void bgzf_inflate_flat_kernel(const uint8_t *__restrict__ comp, const BlockDesc *__restrict__ blocks, int32_t n_blocks,
                                                               uint8_t *__restrict__ outbuf, int32_t *__restrict__ status)
{
    constexpr int NB = 64 / G;
    __shared__ spxz::TablesT<LR, DR> T[NB];
    __shared__ __attribute__((aligned(16))) uint8_t ring[NB][R];
    const int q = (int)threadIdx.x / G, lane = (int)threadIdx.x % G;
    const int b = (int)blockIdx.x * NB + q;
    if (b >= n_blocks) return;
    const BlockDesc d = blocks[b];
    FlatEnv<G, LR, DR, R, FLAT> env;
    env.in = comp + d.in_off;
    env.clen = d.clen;
    env.out = outbuf + d.out_off;
    env.limit = d.ulen;
    env.pos = 0;
    env.flushed = 0;
    env.pend_n = 0;
    env.pend_at = 0;
    env.pend_v = 0;
    env.T = &T[q];
    env.ring = ring[q];
    env.lane_ = lane;
    int rc = 0;
    if (d.ulen > 0) {
        rc = spxz::inflate_stream(env, (int64_t)d.clen * 8, d.ulen);
        env.flush(true);
    }
    if (lane == 0) status[b] = rc;
}

/* CRC-32 of every inflated block against the BGZF trailer's: one wavefront per block, 64 stripes (byte-table recurrence per lane, 16
 * bytes per load), folded with the GF(2) shift operator; status[b] 0 -> -4 on a mismatch */
__global__ __launch_bounds__(256) void bgzf_crc_kernel(const BlockDesc *__restrict__ blocks, int32_t n_blocks, const uint8_t *__restrict__ outbuf,
                                                       int32_t *__restrict__ status)
{
    __shared__ uint32_t crc_tab[256];
    crc_tab[threadIdx.x] = spxz::crc_table_entry(threadIdx.x);
    __syncthreads();
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
    uint32_t a = head + 16u * a_pc, e = head + 16u * e_pc;
    uint32_t c = 0xffffffffu;
    uint32_t len = e - a;
    if (lane == 0) {
        for (uint32_t k = 0; k < head; ++k) c = crc_tab[(c ^ p[k]) & 0xff] ^ (c >> 8);
        len += head;
    }
    const uint4 *p16 = reinterpret_cast<const uint4 *>(p + a);
    for (uint32_t k = 0; k < e_pc - a_pc; ++k) {
        const uint4 w = p16[k];
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c ^= ws[j];
#pragma unroll
            for (int t = 0; t < 4; ++t) c = crc_tab[c & 0xff] ^ (c >> 8);
        }
    }
    /* the last stripe that has bytes takes the tail; with no whole piece at all that is lane 0 */
    const uint32_t tail0 = head + 16u * pieces;
    const uint32_t last_lane = pieces ? (pieces - 1u) / per : 0u;
    if ((uint32_t)lane == last_lane) {
        for (uint32_t k = tail0; k < n; ++k) c = crc_tab[(c ^ p[k]) & 0xff] ^ (c >> 8);
        len += n - tail0;
    }
    c ^= 0xffffffffu;
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t oc = (uint32_t)__shfl_down((int)c, s), ol = (uint32_t)__shfl_down((int)len, s);
        if ((lane & (2 * s - 1)) == 0) {
            if (ol > 0) c = len > 0 ? spxz::crc_combine(c, oc, ol) : oc;
            len += ol;
        }
    }
    if (lane == 0 && c != d.crc) status[b] = -4;
}

} // namespace

static int inflate_root_bits()
{
    /* SPX_INFLATE_ROOT: bits of the literal/length root table: 9 (with an 8-bit distance table; default), 10 (10 / 8), 11 (11 / 9: round 3) */
    static const int r = [] { const char *e = getenv("SPX_INFLATE_ROOT"); const int v = e ? atoi(e) : 9; return (v == 10 || v == 11) ? v : 9; }();
    return r;
}

#define SPX_LAUNCH_G(G, LR, DR, R) \
    hipLaunchKernelGGL((bgzf_inflate_g_kernel<G, LR, DR, R>), dim3((unsigned)((n_blocks + (64 / G) - 1) / (64 / G))), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc)

extern "C" hipError_t spx_launch_bgzf_inflate_grouped(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status,
                                                      int check_crc, int lanes_per_block, hipStream_t st)
{
    if (n_blocks <= 0) return hipSuccess;
    const BlockDesc *bd = (const BlockDesc *)blocks;
    const int root = inflate_root_bits();
    static const int ring = [] { const char *e = getenv("SPX_INFLATE_RING"); const int v = e ? atoi(e) : 1024; return (v == 2048 || v == 1025 || v == 4096) ? v : 1024; }();
    if (lanes_per_block == 16) {
        if (root == 11) SPX_LAUNCH_G(16, 11, 9, 2048); else if (root == 10) SPX_LAUNCH_G(16, 10, 8, 2048); else SPX_LAUNCH_G(16, 9, 8, 2048);
    } else {
        if (root == 11) SPX_LAUNCH_G(32, 11, 9, 2048);
        else if (root == 10) SPX_LAUNCH_G(32, 10, 8, 2048);
        else if (ring == 4096) SPX_LAUNCH_G(32, 9, 8, 4096);
        else if (ring == 1024)
            hipLaunchKernelGGL((bgzf_inflate_g_kernel<32, 9, 8, 1024, 5>), dim3((unsigned)((n_blocks + 1) / 2)), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc);
        else if (ring == 1025) /* experiment: 8-bit literal root, six waves per SIMD */
            hipLaunchKernelGGL((bgzf_inflate_g_kernel<32, 8, 8, 1024, 6>), dim3((unsigned)((n_blocks + 1) / 2)), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc);
        else SPX_LAUNCH_G(32, 9, 8, 2048);
    }
    return hipGetLastError();
}

extern "C" hipError_t spx_launch_bgzf_inflate(const uint8_t *comp, const void *blocks, int32_t n_blocks, uint8_t *out, int32_t *status,
                                              int check_crc, hipStream_t st)
{
    if (n_blocks <= 0) return hipSuccess;
    /* SPX_INFLATE_LANES: 32 (default) / 16 = lanes per block of the grouped kernel (2 / 4 blocks per wave, decode on the vector ALU;
     * SPX_INFLATE_RING = 1024 (default: five waves per SIMD) / 2048 / 4096 bytes of LDS ring per block), 64 = round 3's kernel: one block
     * per wave on the scalar unit */
    static const int lanes = [] { const char *e = getenv("SPX_INFLATE_LANES"); const int v = e ? atoi(e) : 32; return (v == 16 || v == 64) ? v : 32; }();
    /* SPX_INFLATE_FLAT = 4 / 8 / 16 / 32: the flat kernel with that many lanes per block (9 / 8-bit root tables, 1 KB ring) + the CRC kernel */
    static const int flat = [] { const char *e = getenv("SPX_INFLATE_FLAT"); const int v = e ? atoi(e) : 0; return (v == 4 || v == 8 || v == 16 || v == 32 || v == 132) ? v : 0; }();
    if (flat) {
        const BlockDesc *bd = (const BlockDesc *)blocks;
#define SPX_LAUNCH_FLAT(G, WPE) \
    hipLaunchKernelGGL((bgzf_inflate_flat_kernel<G, 9, 8, 1024, WPE>), dim3((unsigned)((n_blocks + (64 / G) - 1) / (64 / G))), dim3(64), 0, st, comp, bd, n_blocks, out, status)
        if (flat == 4) SPX_LAUNCH_FLAT(4, 1);
        else if (flat == 8) SPX_LAUNCH_FLAT(8, 2);
        else if (flat == 16) SPX_LAUNCH_FLAT(16, 3);
        else if (flat == 132) /* 32 lanes, the symbol loop with the inner literal loop */
            hipLaunchKernelGGL((bgzf_inflate_flat_kernel<32, 9, 8, 1024, 5, false>), dim3((unsigned)((n_blocks + 1) / 2)), dim3(64), 0, st, comp, bd, n_blocks, out, status);
        else SPX_LAUNCH_FLAT(32, 5);
        if (check_crc) hipLaunchKernelGGL(bgzf_crc_kernel, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, st, bd, n_blocks, out, status);
        return hipGetLastError();
    }
    if (lanes != 64) return spx_launch_bgzf_inflate_grouped(comp, blocks, n_blocks, out, status, check_crc, lanes, st);
    const BlockDesc *bd = (const BlockDesc *)blocks;
    const int root = inflate_root_bits();
    if (root == 11) hipLaunchKernelGGL((bgzf_inflate_kernel<11, 9>), dim3((unsigned)n_blocks), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc);
    else if (root == 10) hipLaunchKernelGGL((bgzf_inflate_kernel<10, 8>), dim3((unsigned)n_blocks), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc);
    else hipLaunchKernelGGL((bgzf_inflate_kernel<9, 8>), dim3((unsigned)n_blocks), dim3(64), 0, st, comp, bd, n_blocks, out, status, check_crc);
    return hipGetLastError();
}
