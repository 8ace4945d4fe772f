/*
 * spx_inflate.h -- DEFLATE (RFC 1951) decoder core for BGZF blocks, written ONCE for both sides: the gfx950 kernel of
 * spx_inflate_kernels.hip (one wavefront per BGZF block) and a host build that the CPU tests run against zlib.
 *
 * Why it exists: the reference inflates on ONE thread inside htslib (sam_read1, src/secphase.c:268); round 3's host
 * reader inflates on a thread pool, but the MI355X boxes give a container ~16 cores of CPU time for a GPU that scores
 * 780 k groups/s -- inflate alone (57 KB per HiFi group at ~0.6 GB/s per core) caps the host at ~150 k groups/s.  On the
 * device a BGZF block is one independent DEFLATE stream of at most 64 KB, so blocks map to wavefronts: the bit-serial
 * Huffman decode of a block is wave-UNIFORM work (every lane computes the same thing: the compiler keeps it on the
 * scalar unit, decode tables live in LDS), and the 64 lanes do the parts that are data parallel -- table fill, LZ77
 * copies, flushing literals, CRC32.
 *
 * The decoder is parameterised by an environment E that provides
 *     E::in32(k)              compressed dword k of the block (little endian; reads past the end return 0)
 *     E::put_literal(b)       one output byte (checked)
 *     E::lit_full(), E::lit_push(b), E::lit_commit()   the same in three steps for the literal loop: no room for another
 *                             literal right now / append one (unchecked) / make room (false: the output is full)
 *     E::copy_match(len,dist) LZ77 copy at the current output position
 *     E::out_pos()            bytes produced so far
 *     E::tables()             scratch for the decode tables (LDS on the device)
 *     E::sync()               wave barrier around cooperative table fills (no-op on the host)
 *     E::lane(), E::lanes()   for the cooperative loops
 * Returns 0 or a negative code (-1 corrupt stream, -2 output overrun, -3 input overrun).
 */
#ifndef SPX_INFLATE_H
#define SPX_INFLATE_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SPXZ_HD __host__ __device__ inline
#define SPXZ_COLD __host__ __device__ __attribute__((noinline)) /* keeps a rare path (and its control flow) out of the hot loop */
#else
#define SPXZ_HD inline
#define SPXZ_COLD __attribute__((noinline))
#endif

namespace spxz {

/* Bits of the root tables are a property of the ENVIRONMENT (E::kLit, E::kDist): the tables live in LDS on the device and
 * their size sets how many blocks a CU decodes at once.  Round 3 used 11 / 9 bits; the literals of BAM payload turn out to
 * have short codes (synthetic HiFi / ONT BAMs at zlib level 6: 94 % of the literals <= 7 bits, 98.4 % <= 9), so round 4's
 * kernels use 9 / 8 (1.5 KB instead of 5 KB) and send the rest through the canonical walk. */
/* root-table entry (16 bits): bits 0-8 symbol (bit 8 set = not a literal), bits 9-12 code length; 0 = not a root code -> canonical walk.  Length and
 * distance base / extra bits are computed from the symbol (len_base ..): the rarer path pays, the table stays small */
SPXZ_HD uint16_t mk_entry(int nbits, int sym) { return (uint16_t)((unsigned)sym | ((unsigned)nbits << 9)); }
constexpr uint16_t kNoEntry = 0x100; /* "not a literal", length 0: the literal loop needs ONE bit test */

template <int LR, int DR>
struct TablesT {
    static constexpr int kLitBits = LR, kDistBits = DR;
    uint16_t lit[1 << LR];
    uint16_t dist[1 << DR];
    /* canonical description (walked for codes longer than the root, and while building) */
    uint16_t lit_count[16], dist_count[16];   /* codes per length */
    uint16_t lit_sorted[288], dist_sorted[32]; /* symbols ordered by (length, symbol) */
    uint8_t lens[288 + 32];                    /* code lengths of the block being set up */
    uint16_t first[16], offs[16];              /* of the set being filled: canonical code / index in `sorted` of every length's first symbol */
};
using Tables = TablesT<11, 9>;

SPXZ_HD int len_base(int sym) /* sym 257..285 */
{
    const int k = sym - 257;
    if (k < 8) return 3 + k;
    if (k == 28) return 258;
    const int e = (k - 4) >> 2;
    return 3 + ((4 + (k & 3)) << e);
}
SPXZ_HD int len_extra(int sym)
{
    const int k = sym - 257;
    if (k < 8 || k == 28) return 0;
    return (k - 4) >> 2;
}
SPXZ_HD int dist_base(int sym) /* 0..29 */
{
    if (sym < 4) return 1 + sym;
    const int e = (sym - 2) >> 1;
    return 1 + ((2 + (sym & 1)) << e);
}
SPXZ_HD int dist_extra(int sym) { return sym < 4 ? 0 : (sym - 2) >> 1; }

SPXZ_HD uint32_t rev_bits(uint32_t code, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

/* bit reader over E::in32: at least 32 valid bits after refill().  The next input word is fetched one refill AHEAD
 * (`pre`): on the device the scalar load's latency then overlaps a symbol's worth of decoding. */
template <class E>
struct Bits {
    E &env;
    uint64_t buf = 0;
    int cnt = 0;
    uint32_t next = 0; /* dword index of the word in `pre` */
    uint32_t pre;
    SPXZ_HD explicit Bits(E &e) : env(e) { pre = env.in32(0); }
    SPXZ_HD void refill()
    {
        if (cnt <= 32) {
            buf |= (uint64_t)pre << cnt;
            ++next;
            cnt += 32;
            pre = env.in32(next);
        }
    }
    SPXZ_HD uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); } /* n <= 16 */
    SPXZ_HD void drop(int n) { buf >>= n; cnt -= n; }
    SPXZ_HD uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
    /* bits consumed so far (for the input-overrun check) */
    SPXZ_HD int64_t consumed() const { return (int64_t)next * 32 - cnt; }
};

/* canonical walk (Mark Adler's puff): one bit at a time; used for codes longer than the root tables and for the
 * code-length code.  Does NOT consume: returns (code length << 16) | symbol, or 0xffffffff.  (The tables it reads live in
 * LDS / scratch on the device, i.e. in vector registers: the caller makes the result wave-uniform before it touches the
 * bit reader, so that the reader's state stays on the scalar unit.) */
static SPXZ_COLD uint32_t decode_slow_bits(uint32_t bits, const uint16_t *count, const uint16_t *sorted, int max_len)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= max_len; ++len) {
        code |= (int)((bits >> (len - 1)) & 1);
        const int c = count[len];
        if (code - c < first) return ((uint32_t)len << 16) | sorted[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return 0xffffffffu;
}
template <class E>
SPXZ_HD uint32_t decode_slow(const Bits<E> &b, const uint16_t *count, const uint16_t *sorted, int max_len)
{
    return decode_slow_bits((uint32_t)b.buf, count, sorted, max_len);
}

/* counts + sorted symbols from lens[0..n), and (first / offs0 != NULL) the canonical code and the index in `sorted` of every
 * length's first symbol; returns 0, or -1 for an over-subscribed / incomplete set (an incomplete set is allowed when it has a
 * single code, as zlib allows for distance trees) */
SPXZ_HD int canon_build(const uint8_t *lens, int n, uint16_t *count, uint16_t *sorted, uint16_t *first, uint16_t *offs0)
{
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[lens[s] & 15]++;
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return -1; /* over-subscribed */
    }
    uint16_t offs[16];
    offs[0] = 0; offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    if (first) {
        int code = 0;
        first[0] = 0;
        offs0[0] = 0;
        for (int l = 1; l < 16; ++l) { /* RFC 1951 3.2.2 with bl_count[0] = 0 */
            code = (code + (l > 1 ? count[l - 1] : 0)) << 1;
            first[l] = (uint16_t)code;
            offs0[l] = offs[l];
        }
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s] & 15;
        if (!l) continue;
        sorted[offs[l]++] = (uint16_t)s;
    }
    const int used = n - count[0];
    if (left > 0 && used > 1) return -1; /* incomplete: only a set of at most one code may be (zlib accepts those too) */
    return 0;
}

/* root-table fill for the coded symbols number i = first_i, first_i + stride, ... in (length, symbol) order: the canonical
 * code of sorted[i] is first[len] + (i - offs[len]); every code of length <= root is replicated over the high index bits;
 * longer codes leave their (shared) root slots at "walk" */
template <bool LIT>
SPXZ_HD void fill_root(uint16_t *tab, int root, const uint8_t *lens, const uint16_t *sorted, const uint16_t *first, const uint16_t *offs, int used,
                       int first_i, int stride)
{
    for (int i = first_i; i < used; i += stride) {
        const int s = sorted[i];
        const int l = lens[s];
        if (l > root) continue;
        if (LIT ? s > 285 : s > 29) continue; /* 286, 287 / 30, 31: never valid (their slots stay "walk" -> the walk rejects them) */
        const uint16_t e = mk_entry(l, s);
        const uint32_t r = rev_bits((uint32_t)first[l] + (uint32_t)(i - offs[l]), l);
        for (uint32_t k = r; k < (1u << root); k += (1u << l)) tab[k] = e;
    }
}

template <class E>
SPXZ_HD int build_tables(E &env, int nlit, int ndist)
{
    auto &T = env.tables();
    const int lane = env.lane(), lanes = env.lanes();
    for (int k = lane; k < (1 << E::kLit); k += lanes) T.lit[k] = kNoEntry;
    for (int k = lane; k < (1 << E::kDist); k += lanes) T.dist[k] = kNoEntry;
    /* (a set that is refused leaves `sorted` / `first` / `offs` half-written: nothing is filled from them) */
    int rc = 0;
    if (lane == 0) rc = canon_build(T.lens, nlit, T.lit_count, T.lit_sorted, T.first, T.offs);
    env.sync();
    rc = env.uniform(rc);
    if (rc == 0) fill_root<true>(T.lit, E::kLit, T.lens, T.lit_sorted, T.first, T.offs, nlit - (int)T.lit_count[0], lane, lanes);
    env.sync();
    int rc2 = 0;
    if (lane == 0 && rc == 0) rc2 = canon_build(T.lens + nlit, ndist, T.dist_count, T.dist_sorted, T.first, T.offs);
    env.sync();
    rc2 = env.uniform(rc2);
    if (rc == 0 && rc2 == 0) fill_root<false>(T.dist, E::kDist, T.lens + nlit, T.dist_sorted, T.first, T.offs, ndist - (int)T.dist_count[0], lane, lanes);
    env.sync();
    return (rc | rc2) ? -1 : 0;
}

/* one DEFLATE stream (all its blocks).  out_limit: bytes the caller expects (ISIZE) */
template <class E>
SPXZ_HD int inflate_stream(E &env, int64_t in_bits_limit, uint32_t out_limit)
{
    Bits<E> b(env);
    auto &T = env.tables();
    for (;;) {
        b.refill();
        const uint32_t hdr = b.take(3);
        const int final_blk = (int)(hdr & 1), type = (int)(hdr >> 1);
        if (type == 0) { /* stored */
            b.drop(b.cnt & 7); /* to the byte boundary */
            b.refill();
            const uint32_t len = b.take(16);
            b.refill();
            const uint32_t nlen = b.take(16);
            if ((len ^ 0xffffu) != nlen) return -1;
            if (env.out_pos() + len > out_limit) return -2;
            for (uint32_t k = 0; k < len; ++k) {
                b.refill();
                if (!env.put_literal((uint8_t)b.take(8))) return -2;
            }
        } else if (type == 1 || type == 2) {
            int nlit, ndist;
            if (type == 1) {
                const int lane = env.lane(), lanes = env.lanes();
                for (int s = lane; s < 288; s += lanes) T.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                for (int s = lane; s < 32; s += lanes) T.lens[288 + s] = 5;
                env.sync();
                nlit = 288; ndist = 32;
                /* the fixed distance code has 32 codes of 5 bits; symbols 30, 31 never occur in valid data */
            } else {
                b.refill();
                nlit = (int)b.take(5) + 257;
                ndist = (int)b.take(5) + 1;
                const int ncode = (int)b.take(4) + 4;
                if (nlit > 286 || ndist > 30) return -1;
                /* code-length code: 19 symbols of <= 7 bits; decoded by the canonical walk (it is used ~300 times) */
                uint8_t cl[19];
                const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                for (int k = 0; k < 19; ++k) cl[k] = 0;
                for (int k = 0; k < ncode; ++k) { b.refill(); cl[order[k]] = (uint8_t)b.take(3); }
                uint16_t ccount[16], csorted[19];
                if (env.uniform(canon_build(cl, 19, ccount, csorted, nullptr, nullptr)) != 0) return -1;
                int idx = 0;
                int bad = 0;
                while (idx < nlit + ndist) {
                    b.refill();
                    const uint32_t r = env.uniform_u32(decode_slow(b, ccount, csorted, 7));
                    if (r == 0xffffffffu) { bad = 1; break; }
                    b.drop((int)(r >> 16));
                    const int sym = (int)(r & 0xffff);
                    if (sym < 16) { if (env.lane() == 0) T.lens[idx] = (uint8_t)sym; ++idx; continue; }
                    int rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) { bad = 1; break; }
                        val = -1; /* previous length */
                        rep = 3 + (int)b.take(2);
                    } else if (sym == 17) rep = 3 + (int)b.take(3);
                    else rep = 11 + (int)b.take(7);
                    if (idx + rep > nlit + ndist) { bad = 1; break; }
                    if (env.lane() == 0) {
                        const uint8_t v = val < 0 ? T.lens[idx - 1] : 0;
                        for (int k = 0; k < rep; ++k) T.lens[idx + k] = v;
                    }
                    idx += rep;
                }
                if (bad) return -1;
                env.sync();
                if (env.uniform((int)T.lens[256]) == 0) return -1; /* no end-of-block code */
                /* the distance lengths follow the literal lengths directly: build_tables expects them at lens + nlit */
            }
            if (build_tables(env, nlit, ndist) != 0) return -1;
            /* ---- the symbol loop: a tight inner loop over runs of literals (one table look-up, one v_writelane each on
             * the device), everything else outside it ---- */
            for (;;) {
                uint32_t e;
                for (;;) {
                    b.refill();
                    e = env.uniform_u32(T.lit[b.peek(E::kLit)]);
                    /* a length / end-of-block code, or no root code at all -- or no room for one more literal (the device
                     * keeps up to 64 in a register; committing them, flushing and the overrun check stay OUT of this loop:
                     * inlined into it they cost a dozen scalar moves per literal) */
                    if ((e & 0x100u) || env.lit_full()) break;
                    b.drop((int)(e >> 9));
                    env.lit_push((uint8_t)e);
                }
                if (!(e & 0x100u)) {
                    if (!env.lit_commit()) return -2; /* false: more bytes than the block may hold */
                    continue;
                }
                int sym;
                if (e != kNoEntry) {
                    b.drop((int)(e >> 9));
                    sym = (int)(e & 511);
                } else {
                    const uint32_t r = env.uniform_u32(decode_slow(b, T.lit_count, T.lit_sorted, 15));
                    if (r == 0xffffffffu) return -1;
                    b.drop((int)(r >> 16));
                    sym = (int)(r & 0xffff);
                    if (sym > 285) return -1;
                    if (sym < 256) {
                        if (!env.put_literal((uint8_t)sym)) return -2;
                        continue;
                    }
                }
                if (sym == 256) break;
                b.refill();
                const int len = len_base(sym) + (int)b.take(len_extra(sym));
                b.refill();
                const uint32_t d = env.uniform_u32(T.dist[b.peek(E::kDist)]);
                int dsym;
                if (d != kNoEntry) {
                    b.drop((int)(d >> 9));
                    dsym = (int)(d & 511);
                } else {
                    const uint32_t r = env.uniform_u32(decode_slow(b, T.dist_count, T.dist_sorted, 15));
                    if (r == 0xffffffffu) return -1;
                    b.drop((int)(r >> 16));
                    dsym = (int)(r & 0xffff);
                    if (dsym > 29) return -1;
                }
                b.refill();
                const int dist = dist_base(dsym) + (int)b.take(dist_extra(dsym));
                if ((uint32_t)dist > env.out_pos()) return -1;
                if (env.out_pos() + (uint32_t)len > out_limit) return -2;
                env.copy_match(len, dist);
            }
            if (env.out_pos() > out_limit) return -2;
        } else
            return -1;
        if (b.consumed() > in_bits_limit) return -3;
        if (final_blk) break;
    }
    if (b.consumed() > in_bits_limit) return -3;
    return env.out_pos() == out_limit ? 0 : -2;
}

/* ---- CRC-32 (IEEE 802.3, reflected; what gzip / BGZF store) ---- */
SPXZ_HD uint32_t crc_table_entry(uint32_t n)
{
    uint32_t c = n;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
    return c;
}
/* a * b mod P over GF(2), reflected representation (bit 31 = x^0) */
SPXZ_HD uint32_t gf2_mul(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
    for (int k = 0; k < 32; ++k) {
        if (a & 0x80000000u) r ^= b;
        a <<= 1;
        b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return r;
}
/* x^(8 * n_bytes) mod P */
SPXZ_HD uint32_t gf2_xpow8n(uint64_t n_bytes)
{
    uint32_t r = 0x80000000u;        /* x^0 */
    uint32_t p = 0x00800000u;        /* x^8 */
    while (n_bytes) {
        if (n_bytes & 1) r = gf2_mul(r, p);
        p = gf2_mul(p, p);
        n_bytes >>= 1;
    }
    return r;
}
/* crc(A || B) from crc(A), crc(B), len(B) (both "finished" CRCs, i.e. with the final complement) */
SPXZ_HD uint32_t crc_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) { return gf2_mul(gf2_xpow8n(len_b), crc_a) ^ crc_b; }

/* ---- host environment: plain buffers ---- */
template <int LR, int DR>
struct HostEnvT {
    static constexpr int kLit = LR, kDist = DR;
    const uint8_t *in;
    size_t in_len;
    uint8_t *out;
    uint32_t pos = 0;
    TablesT<LR, DR> T;
    uint32_t in32(uint32_t k) const
    {
        uint32_t v = 0;
        for (int q = 0; q < 4; ++q) {
            const size_t at = (size_t)k * 4 + (size_t)q;
            if (at < in_len) v |= (uint32_t)in[at] << (8 * q);
        }
        return v;
    }
    uint32_t cap = 0xffffffffu; /* bytes out[] can take */
    bool put_literal(uint8_t c) { if (pos >= cap) return false; out[pos++] = c; return true; }
    bool lit_full() const { return pos >= cap; }
    void lit_push(uint8_t c) { out[pos++] = c; }
    bool lit_commit() const { return pos < cap; }
    void copy_match(int len, int dist) { for (int k = 0; k < len; ++k, ++pos) out[pos] = out[pos - dist]; }
    uint32_t out_pos() const { return pos; }
    TablesT<LR, DR> &tables() { return T; }
    void sync() {}
    int lane() const { return 0; }
    int lanes() const { return 1; }
    int uniform(int v) const { return v; }
    uint32_t uniform_u32(uint32_t v) const { return v; }
};
using HostEnv = HostEnvT<11, 9>;

} // namespace spxz
#endif
