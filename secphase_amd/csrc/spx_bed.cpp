/*
 * spx_bed.cpp -- BED side outputs of secphase.
 *
 *   <prefix>.modified_read_blocks.markers.bed   reference extents of the old primary and of the promoted
 *        secondary of every relabelled read, merged per contig into disjoint segments carrying the number
 *        of alignments covering them (src/secphase.c:201-203 ptBlock_add_alignment(..., true);
 *        :719-721 merge_and_save_blocks(..., true))
 *   <prefix>.marker_blocks.bed                  reference positions of the surviving markers of those two
 *        alignments (src/secphase.c:205-212, submodules/ptMarker/ptMarker.c:844-865), merged without counts
 *
 * merge = ptBlock_merge_blocks_v2 (submodules/ptBlock/ptBlock.c:274-428) on blocks sorted by start
 * (:228-236): the union of the blocks cut at every block start and at every (block end + 1), each piece
 * carrying the summed count of the blocks covering it; pieces are NOT re-joined.  Here as a sweep over the
 * sorted break points instead of the reference's list rewriting; tests/test_blocks.py checks it against the
 * oracle's restatement and against the reference's own known-answer vectors.
 * Writer = ptBlock_save_in_bed (:573-602): contigs in strcmp order, "ctg\tstart\tend+1[\tcount]".
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/mman.h>
#include <errno.h>
#include <fcntl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/spx.h"

struct Blk3 { int32_t s, e, c; };

/* std::sort on `threads` threads: sorted runs, then pairwise merges level by level (a contig of a small assembly can hold millions of
 * marker positions while the save has only a handful of contigs to spread over its threads) */
template <class T, class Cmp>
static void parallel_sort(std::vector<T> &v, Cmp cmp, unsigned threads)
{
    const size_t n = v.size();
    if (threads <= 1 || n < 65536) { std::sort(v.begin(), v.end(), cmp); return; }
    unsigned runs = 1;
    while (runs * 2 <= threads && n / (runs * 2) >= 16384) runs *= 2;
    std::vector<size_t> cut(runs + 1);
    for (unsigned r = 0; r <= runs; ++r) cut[r] = n * r / runs;
    {
        std::vector<std::thread> th;
        for (unsigned r = 0; r < runs; ++r) th.emplace_back([&, r]() { std::sort(v.begin() + (ptrdiff_t)cut[r], v.begin() + (ptrdiff_t)cut[r + 1], cmp); });
        for (auto &t : th) t.join();
    }
    for (unsigned w = 1; w < runs; w *= 2) {
        std::vector<std::thread> th;
        for (unsigned r = 0; r + w < runs; r += 2 * w)
            th.emplace_back([&, r, w]() {
                std::inplace_merge(v.begin() + (ptrdiff_t)cut[r], v.begin() + (ptrdiff_t)cut[r + w], v.begin() + (ptrdiff_t)cut[std::min(runs, r + 2 * w)], cmp);
            });
        for (auto &t : th) t.join();
    }
}

static void merge_count_sorted(std::vector<Blk3> v, bool has_count, std::vector<Blk3> &out, unsigned threads);

/* coverage segmentation of blocks (any order); has_count=0: c ignored, output c = 0 */
static void merge_count(const std::vector<Blk3> &v_in, bool has_count, std::vector<Blk3> &out, unsigned threads = 1)
{
    out.clear();
    if (v_in.empty()) return;
    const std::vector<Blk3> &v = v_in;
    if (!has_count) {
        /* single-base blocks without counts (marker positions): the segmentation is the set of distinct positions */
        bool points = true;
        for (const Blk3 &b : v)
            if (b.s != b.e) { points = false; break; }
        if (points) {
            /* dense sets (a run that relabels most reads puts tens of millions of positions on a few contigs): one bit per reference
             * position between the smallest and the largest, set from all threads, read back in order -- O(n + range) instead of a sort */
            int32_t lo = v[0].s, hi = v[0].s;
            for (const Blk3 &b : v) { lo = std::min(lo, b.s); hi = std::max(hi, b.s); }
            const uint64_t range = (uint64_t)((int64_t)hi - (int64_t)lo) + 1;
            if (v.size() >= 65536 && range <= (uint64_t)v.size() * 64u) {
                const size_t words = (size_t)((range + 63) / 64);
                std::vector<std::atomic<uint64_t>> bits(words);
                for (auto &w : bits) w.store(0, std::memory_order_relaxed);
                const unsigned T = std::max(1u, threads);
                auto run = [&](auto fn) {
                    if (T == 1) { fn(0u); return; }
                    std::vector<std::thread> th;
                    for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t);
                    for (auto &x : th) x.join();
                };
                run([&](unsigned t) {
                    for (size_t i = v.size() * t / T, e = v.size() * (t + 1) / T; i < e; ++i) {
                        const uint64_t d = (uint64_t)((int64_t)v[i].s - (int64_t)lo), bit = 1ull << (d & 63);
                        std::atomic<uint64_t> &w = bits[(size_t)(d >> 6)];
                        if (!(w.load(std::memory_order_relaxed) & bit)) w.fetch_or(bit, std::memory_order_relaxed); /* (most positions repeat) */
                    }
                });
                std::vector<size_t> first(T + 1, 0);
                run([&](unsigned t) {
                    size_t n = 0;
                    for (size_t w = words * t / T, e = words * (t + 1) / T; w < e; ++w) n += (size_t)__builtin_popcountll(bits[w].load(std::memory_order_relaxed));
                    first[t + 1] = n;
                });
                for (unsigned t = 0; t < T; ++t) first[t + 1] += first[t];
                out.resize(first[T]);
                run([&](unsigned t) {
                    size_t at = first[t];
                    for (size_t w = words * t / T, e = words * (t + 1) / T; w < e; ++w) {
                        uint64_t x = bits[w].load(std::memory_order_relaxed);
                        while (x) {
                            const int32_t p = (int32_t)((int64_t)lo + (int64_t)(w * 64 + (size_t)__builtin_ctzll(x)));
                            out[at++] = {p, p, 0};
                            x &= x - 1;
                        }
                    }
                });
                return;
            }
            std::vector<int32_t> p(v.size());
            for (size_t i = 0; i < v.size(); ++i) p[i] = v[i].s;
            parallel_sort(p, std::less<int32_t>(), threads);
            p.erase(std::unique(p.begin(), p.end()), p.end());
            out.resize(p.size());
            for (size_t i = 0; i < p.size(); ++i) out[i] = {p[i], p[i], 0};
            return;
        }
    }
    return merge_count_sorted(std::vector<Blk3>(v_in), has_count, out, threads);
}

static void merge_count_sorted(std::vector<Blk3> v, bool has_count, std::vector<Blk3> &out, unsigned threads)
{
    parallel_sort(v, [](const Blk3 &a, const Blk3 &b) { return a.s != b.s ? a.s < b.s : a.e < b.e; }, threads);
    /* break points: every start, every end+1 */
    std::vector<int64_t> bp;
    bp.reserve(v.size() * 2);
    for (const Blk3 &b : v) {
        if (b.e < b.s) continue; /* empty blocks take no part in the reference's loop either way */
        bp.push_back(b.s);
        bp.push_back((int64_t)b.e + 1);
    }
    parallel_sort(bp, std::less<int64_t>(), threads);
    bp.erase(std::unique(bp.begin(), bp.end()), bp.end());
    /* coverage difference array over the break points */
    std::vector<int64_t> dc(bp.size() + 1, 0), dn(bp.size() + 1, 0);
    for (const Blk3 &b : v) {
        if (b.e < b.s) continue;
        size_t i0 = std::lower_bound(bp.begin(), bp.end(), (int64_t)b.s) - bp.begin();
        size_t i1 = std::lower_bound(bp.begin(), bp.end(), (int64_t)b.e + 1) - bp.begin();
        dc[i0] += b.c; dc[i1] -= b.c;
        dn[i0] += 1; dn[i1] -= 1;
    }
    int64_t cov = 0, cnt = 0;
    for (size_t i = 0; i + 1 < bp.size(); ++i) {
        cov += dn[i];
        cnt += dc[i];
        if (cov > 0) out.push_back({(int32_t)bp[i], (int32_t)(bp[i + 1] - 1), has_count ? (int32_t)cnt : 0});
    }
}

extern "C" int spx_merge_blocks_count(int32_t n, const int32_t *s, const int32_t *e, const int32_t *c, int32_t *os,
                                      int32_t *oe, int32_t *oc, int32_t cap)
{
    if (n < 0 || !s || !e || !os || !oe) return SPX_EINVAL;
    std::vector<Blk3> v(n), out;
    for (int32_t i = 0; i < n; ++i) v[i] = {s[i], e[i], c ? c[i] : 0};
    merge_count(v, c != nullptr, out);
    if ((int32_t)out.size() > cap) return SPX_EINVAL;
    for (size_t i = 0; i < out.size(); ++i) { os[i] = out[i].s; oe[i] = out[i].e; if (oc) oc[i] = out[i].c; }
    return (int32_t)out.size();
}

/* single-base blocks without counts (marker positions) are kept as a SET while they arrive: one bit per reference position, in chunks
 * of 65 536 positions that exist once a position falls into them.  A run that relabels most reads adds tens of millions of positions
 * (each marker of each relabelled alignment) that collapse to a few million distinct ones: the bit costs less than the 12-byte block
 * did, and what is left for the save is reading the bits in order. */
struct PointSet {
    static constexpr int kShift = 16, kWords = 1 << (kShift - 6);
    std::vector<std::unique_ptr<uint64_t[]>> chunk;
    int64_t added = 0;
    void add(int32_t pos)
    {
        const size_t c = (size_t)((uint32_t)pos >> kShift);
        if (c >= chunk.size()) chunk.resize(c + 1);
        if (!chunk[c]) { chunk[c].reset(new uint64_t[kWords]); memset(chunk[c].get(), 0, sizeof(uint64_t) * kWords); }
        const uint32_t d = (uint32_t)pos & ((1u << kShift) - 1u);
        chunk[c][d >> 6] |= 1ull << (d & 63);
        ++added;
    }
    bool empty() const { return added == 0; }
};
struct BedContig {
    std::vector<Blk3> list; /* blocks with counts or of more than one base, and positions below 0 */
    PointSet points;
};

struct spx_bedset {
    std::map<std::string, BedContig> per_contig; /* std::map iterates in strcmp order for plain ASCII names */
    /* callers pass the same name POINTERS over and over (the contig table of the assembly): pointer -> entry, in front of
     * the string-keyed map (std::map nodes never move) */
    std::unordered_map<const char *, std::pair<const std::string *, BedContig *>> by_ptr;
    BedContig &entry(const char *contig)
    {
        auto it = by_ptr.find(contig);
        /* (the text is compared too: an address may be re-used for another name by a caller with short-lived strings) */
        if (it != by_ptr.end() && strcmp(it->second.first->c_str(), contig) == 0) return *it->second.second;
        auto node = per_contig.find(contig);
        if (node == per_contig.end()) node = per_contig.emplace(contig, BedContig()).first;
        by_ptr[contig] = {&node->first, &node->second};
        return node->second;
    }
};

extern "C" int spx_bedset_create(spx_bedset **out)
{
    if (!out) return SPX_EINVAL;
    *out = new spx_bedset();
    return SPX_OK;
}
extern "C" void spx_bedset_free(spx_bedset *b) { delete b; }
extern "C" int spx_bedset_add(spx_bedset *b, const char *contig, int32_t start, int32_t end, int32_t count)
{
    if (!b || !contig) return SPX_EINVAL;
    b->entry(contig).list.push_back({start, end, count});
    return SPX_OK;
}
/* n single-base blocks on one contig (the marker positions of one relabelled alignment): one look-up of the contig
 * instead of one per position */
extern "C" int spx_bedset_add_points(spx_bedset *b, const char *contig, const int32_t *pos, int32_t n)
{
    if (!b || !contig || (!pos && n > 0)) return SPX_EINVAL;
    if (n <= 0) return SPX_OK;
    BedContig &e = b->entry(contig);
    for (int32_t k = 0; k < n; ++k) {
        if (pos[k] >= 0) e.points.add(pos[k]);
        else e.list.push_back({pos[k], pos[k], 0});
    }
    return SPX_OK;
}
extern "C" int64_t spx_bedset_size(const spx_bedset *b)
{
    int64_t n = 0;
    if (b) for (const auto &kv : b->per_contig) n += (int64_t)kv.second.list.size() + kv.second.points.added;
    return n;
}

/* the positions of a point set in ascending order, as blocks {p, p, 0}: chunks are read side by side */
static void points_in_order(const PointSet &ps, std::vector<Blk3> &out, unsigned threads)
{
    const size_t nc = ps.chunk.size();
    std::vector<size_t> first(nc + 1, 0);
    const unsigned T = std::max(1u, std::min<unsigned>(threads, (unsigned)std::max<size_t>(1, nc)));
    auto run = [&](auto fn) {
        if (T == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
    };
    run([&](unsigned t) {
        for (size_t c = nc * t / T; c < nc * (t + 1) / T; ++c) {
            size_t n = 0;
            if (ps.chunk[c])
                for (int w = 0; w < PointSet::kWords; ++w) n += (size_t)__builtin_popcountll(ps.chunk[c][w]);
            first[c + 1] = n;
        }
    });
    for (size_t c = 0; c < nc; ++c) first[c + 1] += first[c];
    const size_t base = out.size();
    out.resize(base + first[nc]);
    run([&](unsigned t) {
        for (size_t c = nc * t / T; c < nc * (t + 1) / T; ++c) {
            if (!ps.chunk[c]) continue;
            size_t at = base + first[c];
            for (int w = 0; w < PointSet::kWords; ++w) {
                uint64_t x = ps.chunk[c][w];
                while (x) {
                    const int32_t p = (int32_t)((c << PointSet::kShift) + (size_t)w * 64 + (size_t)__builtin_ctzll(x));
                    out[at++] = {p, p, 0};
                    x &= x - 1;
                }
            }
        }
    });
}


/* merge_and_save_blocks (src/secphase.c:59-72): merge per contig, write the BED.  The file is created even
 * when there is nothing to write (the WDLs glob for it, wdls/workflows/secphase.wdl:100-107). */
static char *put_i32(char *o, int32_t v)
{
    char b[16];
    int n = 0;
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) b[n++] = '-';
    while (n) *o++ = b[--n];
    return o;
}

static inline int digits_i32(int32_t v)
{
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    int n = v < 0 ? 2 : 1;
    while (u >= 10) { u /= 10; ++n; }
    return n;
}

extern "C" int spx_bedset_save(const spx_bedset *b, const char *path, int print_count)
{
    if (!b || !path) return SPX_EINVAL;
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return SPX_EINVAL;
    const bool timing = getenv("SPX_TIMING") != nullptr;
    const auto t_a = std::chrono::steady_clock::now();
    /* 1. contigs are independent: merged on threads (a handful of contigs: several threads inside each) */
    std::vector<const std::pair<const std::string, BedContig> *> items;
    for (const auto &kv : b->per_contig) items.push_back(&kv);
    std::vector<std::vector<Blk3>> merged(items.size());
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const unsigned inner = (unsigned)std::max<size_t>(1, hw / std::max<size_t>(1, items.size()));
    auto on_threads = [&](size_t n_tasks, const std::function<void(size_t)> &task) {
        std::atomic<size_t> next(0);
        auto loop = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n_tasks) break;
                task(k);
            }
        };
        const unsigned nthr = (unsigned)std::max<size_t>(1, std::min<size_t>(hw, n_tasks));
        if (nthr <= 1) { loop(); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nthr; ++t) th.emplace_back(loop);
        for (auto &t : th) t.join();
    };
    on_threads(items.size(), [&](size_t k) {
        const BedContig &bc = items[k]->second;
        if (bc.list.empty()) points_in_order(bc.points, merged[k], inner); /* marker positions only: the set IS the merged list */
        else if (bc.points.empty()) merge_count(bc.list, print_count != 0, merged[k], inner);
        else { /* both kinds on one contig: the general merge over all of them */
            std::vector<Blk3> all(bc.list);
            points_in_order(bc.points, all, inner);
            merge_count(all, print_count != 0, merged[k], inner);
        }
    });
    /* 2. the text in pieces of <= 128 k lines: their exact sizes first, so that every piece knows its place in the file */
    struct Piece { size_t k, a, b, bytes, at; };
    std::vector<Piece> pieces;
    for (size_t k = 0; k < items.size(); ++k)
        for (size_t a = 0; a < merged[k].size(); a += (size_t)1 << 17) pieces.push_back({k, a, std::min(merged[k].size(), a + ((size_t)1 << 17)), 0, 0});
    on_threads(pieces.size(), [&](size_t q) {
        Piece &pc = pieces[q];
        const size_t fixed = items[pc.k]->first.size() + 3 + (print_count ? 1 : 0);
        size_t n = 0;
        for (size_t i = pc.a; i < pc.b; ++i) {
            const Blk3 &m = merged[pc.k][i];
            if (m.e < m.s) continue;
            n += fixed + (size_t)digits_i32(m.s) + (size_t)digits_i32(m.e + 1) + (print_count ? (size_t)digits_i32(m.c) : 0);
        }
        pc.bytes = n;
    });
    size_t total = 0;
    for (Piece &pc : pieces) { pc.at = total; total += pc.bytes; }
    const auto t_b = std::chrono::steady_clock::now();
    /* 3. every piece is formatted into a buffer of its own and written at its place with pwrite(): a full disk or an exceeded quota comes
     * back as an error of the call (the page-cache copy costs a parallel memcpy; a shared mapping of the file would save it, but there
     * ENOSPC arrives as SIGBUS inside the formatting loop and a deferred write error is never seen) */
    std::atomic<bool> ok_all(true);
    if (total) {
        on_threads(pieces.size(), [&](size_t q) {
            const Piece &pc = pieces[q];
            if (!pc.bytes || !ok_all.load(std::memory_order_relaxed)) return;
            const std::string &name = items[pc.k]->first;
            char *buf = (char *)malloc(pc.bytes);
            if (!buf) { ok_all.store(false); return; }
            char *o = buf;
            for (size_t i = pc.a; i < pc.b; ++i) {
                const Blk3 &m = merged[pc.k][i];
                if (m.e < m.s) continue;
                memcpy(o, name.data(), name.size()); o += name.size();
                *o++ = '\t';
                o = put_i32(o, m.s); *o++ = '\t';
                o = put_i32(o, m.e + 1);
                if (print_count) { *o++ = '\t'; o = put_i32(o, m.c); }
                *o++ = '\n';
            }
            size_t done = 0;
            while (done < pc.bytes) {
                const ssize_t w = pwrite(fd, buf + done, pc.bytes - done, (off_t)(pc.at + done));
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) { ok_all.store(false); break; }
                done += (size_t)w;
            }
            free(buf);
        });
    }
    bool ok = ok_all.load();
    if (close(fd) != 0) ok = false;
    if (timing)
        fprintf(stderr, "[spx timing] BED %s: merge + sizes %.3f s, text into the file %.3f s (%zu bytes)\n", path, std::chrono::duration<double>(t_b - t_a).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_b).count(), total);
    return ok ? SPX_OK : SPX_EINVAL;
}
