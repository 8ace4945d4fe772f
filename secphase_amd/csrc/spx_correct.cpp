/*
 * spx_correct.cpp -- the consumer of the relabel list: what the reference's correct_bam does with
 * `<prefix>.out.log`, on this repository's own BAM reader.
 *
 *   spx_relabel_table_*   the parser of the list: get_phased_read_table, /root/reference/programs/src/correct_bam.c:32-91
 *                         (keys on the first character of a line: `$` read name, `*` old primary, `@` promoted secondary;
 *                         columns 3-4 = contig and 0-based start; a record whose two locations coincide is ignored, :64-68,77-82)
 *   spx_correct_bam       the record loop, correct_bam.c:347-376: unmapped / excluded reads dropped, BAM_FSECONDARY cleared on the
 *                         record the table names and set on every other record of that read (is_prim, :93-109), --primaryOnly,
 *                         read-length / alignment-length filters (:189-220), MAPQ table (:111-140,166-184), --maxMapq, `de` divergence
 *                         filter, --noTag; output as BAM (BGZF, like sam_open(path, "wb")) or as SAM text.
 *
 * Host code only: this is file plumbing behind the hot path, no kernel is involved.  Where the reference leaves behaviour undefined
 * the choice made here is marked U:.
 */
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/spx.h"

extern "C" int spx_internal_bam_record(const spx_bam_reader *src, const spx_batch *bt, int32_t a, const uint8_t **rec, int32_t *block_size);
extern "C" int spx_internal_bam_header(const spx_bam_reader *src, const char **text, int64_t *text_len, int32_t *n_targets);
extern "C" int64_t spx_internal_bam_target_len(const spx_bam_reader *src, int32_t i);
extern "C" int spx_internal_format_sam(const spx_bam_reader *src, const uint8_t *rec, int32_t block_size, void *std_string_out);
extern "C" void spx_internal_set_error(const char *msg);

namespace {

struct Location {
    std::string contig;
    int32_t start;
    uint8_t mapq;
};

int fail(int rc, const std::string &msg)
{
    spx_internal_set_error(msg.c_str());
    return rc;
}

/* getline + strtok(line, "\t"): consecutive tabs count as one separator */
std::vector<std::string> tab_tokens(const std::string &line)
{
    std::vector<std::string> t;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && line[i] == '\t') ++i;
        if (i >= line.size()) break;
        size_t j = line.find('\t', i);
        if (j == std::string::npos) j = line.size();
        t.push_back(line.substr(i, j - i));
        i = j;
    }
    return t;
}

bool read_lines(const char *path, std::vector<std::string> &lines)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return false;
    char *buf = nullptr;
    size_t cap = 0;
    ssize_t n;
    while ((n = getline(&buf, &cap, fp)) != -1) lines.emplace_back(buf, (size_t)n);
    free(buf);
    fclose(fp);
    return true;
}

} // namespace

struct spx_relabel_table {
    std::unordered_map<std::string, Location> by_name;
    std::vector<std::string> names; /* sorted, for iteration */
};

extern "C" int spx_relabel_table_load(const char *path, spx_relabel_table **out)
{
    if (!out) return SPX_EINVAL;
    *out = nullptr;
    spx_relabel_table *t = new spx_relabel_table();
    if (path) {
        std::vector<std::string> lines;
        if (!read_lines(path, lines)) { delete t; return fail(SPX_EINVAL, std::string("cannot open ") + path); }
        std::string read_name, contig_new, contig_old;
        int start_new = -1, start_old = -1;
        bool have_name = false;
        for (std::string &line : lines) {
            if (!line.empty() && line.back() == '\n') line.pop_back();
            if (line.empty()) continue;
            if (line[0] == '$') {
                const std::vector<std::string> f = tab_tokens(line);
                read_name = f.size() > 1 ? f[1] : std::string(); /* U: a `$` line without a name (the reference dereferences NULL) */
                have_name = f.size() > 1;
                start_new = start_old = -1;
            } else if (line[0] == '@' || line[0] == '*') {
                const std::vector<std::string> f = tab_tokens(line);
                if (f.size() < 4 || !have_name) continue; /* U: short line (the reference dereferences NULL) */
                if (line[0] == '@') { contig_new = f[2]; start_new = atoi(f[3].c_str()); }
                else { contig_old = f[2]; start_old = atoi(f[3].c_str()); }
                /* both locations seen: the read enters the table unless they coincide (:64-68,77-82); a later record of the same
                 * read name replaces the earlier one (stHash_insert) */
                if (start_new != -1 && start_old != -1 && (start_old != start_new || contig_old != contig_new))
                    t->by_name[read_name] = Location{contig_new, start_new, 0};
            }
        }
    }
    t->names.reserve(t->by_name.size());
    for (const auto &kv : t->by_name) t->names.push_back(kv.first);
    std::sort(t->names.begin(), t->names.end());
    *out = t;
    return SPX_OK;
}

extern "C" int64_t spx_relabel_table_size(const spx_relabel_table *t) { return t ? (int64_t)t->names.size() : 0; }

extern "C" int spx_relabel_table_get(const spx_relabel_table *t, int64_t i, const char **qname, const char **contig, int32_t *start)
{
    if (!t || i < 0 || (size_t)i >= t->names.size()) return SPX_EINVAL;
    const Location &l = t->by_name.at(t->names[(size_t)i]);
    if (qname) *qname = t->names[(size_t)i].c_str();
    if (contig) *contig = l.contig.c_str();
    if (start) *start = l.start;
    return SPX_OK;
}

extern "C" int spx_relabel_table_find(const spx_relabel_table *t, const char *qname, const char **contig, int32_t *start)
{
    if (!t || !qname) return SPX_EINVAL;
    const auto it = t->by_name.find(qname);
    if (it == t->by_name.end()) return 0;
    if (contig) *contig = it->second.contig.c_str();
    if (start) *start = it->second.start;
    return 1;
}

extern "C" void spx_relabel_table_free(spx_relabel_table *t) { delete t; }

extern "C" void spx_correct_default_options(spx_correct_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->min_read_length = 5000;      /* correct_bam.c:248 */
    o->min_alignment_length = 5000; /* :249 */
    o->max_divergence = 0.12;       /* :250 */
    o->threads = 2;                 /* :251 */
    o->max_mapq = 100;              /* :252 */
}

namespace {

/* ---- BGZF writer: blocks of <= 0xff00 data bytes, a record that still fits the block is not split (htslib's bgzf_flush_try), blocks
 * compressed on threads in groups, EOF marker at the end ---- */
struct BgzfOut {
    FILE *fp = nullptr;
    int threads = 1;
    std::vector<std::vector<uint8_t>> pending; /* full blocks waiting for a compression round */
    std::vector<uint8_t> cur;
    bool ok = true;
    static constexpr size_t kBlock = 0xff00;

    static bool deflate_block(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        out.resize(18 + deflateBound(&zs, (uLong)in.size()) + 8);
        zs.next_in = const_cast<Bytef *>(in.data());
        zs.avail_in = (uInt)in.size();
        zs.next_out = out.data() + 18;
        zs.avail_out = (uInt)(out.size() - 18 - 8);
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END || 18 + clen + 8 > 65536) return false; /* (0xff00 bytes of input always fit: deflateBound < 64 KB) */
        const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 0, 0};
        memcpy(out.data(), hdr, 18);
        const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
        out[16] = (uint8_t)(bsize & 0xff); out[17] = (uint8_t)(bsize >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in.data(), (uInt)in.size()), isize = (uint32_t)in.size();
        uint8_t *t = out.data() + 18 + clen;
        for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(isize >> (8 * k)); }
        out.resize(18 + clen + 8);
        return true;
    }
    void round()
    {
        if (pending.empty()) return;
        std::vector<std::vector<uint8_t>> comp(pending.size());
        std::atomic<size_t> next(0);
        std::atomic<bool> good(true);
        auto loop = [&] {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= pending.size()) break;
                if (!deflate_block(pending[k], comp[k])) good.store(false);
            }
        };
        const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, pending.size()));
        std::vector<std::thread> th;
        for (int k = 1; k < nt; ++k) th.emplace_back(loop);
        loop();
        for (auto &x : th) x.join();
        if (!good.load()) ok = false;
        for (const auto &c : comp)
            if (ok && fwrite(c.data(), 1, c.size(), fp) != c.size()) ok = false;
        pending.clear();
    }
    void flush_block()
    {
        if (cur.empty()) return;
        pending.emplace_back();
        pending.back().swap(cur);
        if (pending.size() >= 64) round();
    }
    void write(const uint8_t *p, size_t n, bool atomic_unit)
    {
        if (atomic_unit && n <= kBlock && cur.size() + n > kBlock) flush_block();
        while (n) {
            const size_t take = std::min(n, kBlock - cur.size());
            cur.insert(cur.end(), p, p + take);
            p += take; n -= take;
            if (cur.size() == kBlock) flush_block();
        }
    }
    bool close()
    {
        flush_block();
        round();
        static const uint8_t eof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (ok && fwrite(eof, 1, sizeof eof, fp) != sizeof eof) ok = false;
        if (fclose(fp) != 0) ok = false;
        fp = nullptr;
        return ok;
    }
};

inline int32_t rd32(const uint8_t *p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
inline void wr32(std::vector<uint8_t> &v, int32_t x) { for (int k = 0; k < 4; ++k) v.push_back((uint8_t)((uint32_t)x >> (8 * k))); }

/* bam_aux2f(bam_aux_get(b, "de")): a float / double tag as is, an integer tag converted; U: a record without the tag counts as
 * divergence 0 (the reference passes NULL to bam_aux2f and crashes) */
double de_tag(const uint8_t *aux, const uint8_t *end)
{
    while (aux + 3 <= end) {
        const char ty = (char)aux[2];
        const uint8_t *v = aux + 3;
        const bool hit = aux[0] == 'd' && aux[1] == 'e';
        size_t n = 0;
        switch (ty) {
        case 'A': case 'c': case 'C': n = 1; break;
        case 's': case 'S': n = 2; break;
        case 'i': case 'I': case 'f': n = 4; break;
        case 'd': n = 8; break;
        case 'Z': case 'H': { const uint8_t *z = v; while (z < end && *z) ++z; n = (size_t)(z - v) + 1; break; }
        case 'B': {
            if (v + 5 > end) return 0.0;
            const char sub = (char)v[0];
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            n = 5 + es * (size_t)(uint32_t)rd32(v + 1);
            break;
        }
        default: return 0.0;
        }
        if (v + n > end) return 0.0;
        if (hit) {
            switch (ty) {
            case 'f': { float f; memcpy(&f, v, 4); return f; }
            case 'd': { double d; memcpy(&d, v, 8); return d; }
            case 'c': return (int8_t)v[0];
            case 'C': return v[0];
            case 's': return (int16_t)(v[0] | (v[1] << 8));
            case 'S': return (uint16_t)(v[0] | (v[1] << 8));
            case 'i': return rd32(v);
            case 'I': return (uint32_t)rd32(v);
            default: return 0.0; /* (bam_aux2f sets errno = EINVAL and returns 0) */
            }
        }
        aux = v + n;
    }
    return 0.0;
}

} // namespace

extern "C" int spx_correct_bam(const char *in_path, const char *out_path, const spx_correct_options *opt_in, spx_correct_stats *stats)
{
    if (!in_path || !out_path) return SPX_EINVAL;
    spx_correct_options opt;
    if (opt_in) opt = *opt_in; else spx_correct_default_options(&opt);
    spx_correct_stats st;
    memset(&st, 0, sizeof st);
    /* the three side inputs */
    spx_relabel_table *table = nullptr;
    int rc = spx_relabel_table_load(opt.phasing_log, &table);
    if (rc != SPX_OK) return rc;
    st.table_reads = (int64_t)table->by_name.size();
    std::unordered_map<std::string, std::vector<Location>> mapq_table; /* get_mapq_table, :111-146 */
    if (opt.mapq_table) {
        std::vector<std::string> lines;
        if (!read_lines(opt.mapq_table, lines)) { spx_relabel_table_free(table); return fail(SPX_EINVAL, std::string("cannot open ") + opt.mapq_table); }
        for (std::string &line : lines) {
            if (!line.empty() && line.back() == '\n') line.pop_back();
            const std::vector<std::string> f = tab_tokens(line);
            if (f.size() < 4) continue; /* U: short line */
            mapq_table[f[0]].push_back(Location{f[1], atoi(f[2].c_str()) - 1, (uint8_t)atoi(f[3].c_str())});
        }
    }
    std::unordered_set<std::string> exclude; /* get_read_set, :149-164: the LAST character of every line is cut off, newline or not */
    if (opt.exclude) {
        std::vector<std::string> lines;
        if (!read_lines(opt.exclude, lines)) { spx_relabel_table_free(table); return fail(SPX_EINVAL, std::string("cannot open ") + opt.exclude); }
        for (std::string &line : lines) {
            if (!line.empty()) line.pop_back();
            exclude.insert(line);
        }
    }
    spx_bam_options bo;
    spx_bam_default_options(&bo);
    bo.threads = std::max(1, opt.threads);
    spx_bam_reader *rd = nullptr;
    rc = spx_bam_open_opts(in_path, &bo, &rd);
    if (rc != SPX_OK) { spx_relabel_table_free(table); return fail(rc, spx_io_last_error()); }
    const char *htext = nullptr;
    int64_t hlen = 0;
    int32_t nt = 0;
    spx_internal_bam_header(rd, &htext, &hlen, &nt);
    FILE *fp = fopen(out_path, "wb");
    if (!fp) { spx_bam_close(rd); spx_relabel_table_free(table); return fail(SPX_EINVAL, std::string("cannot create ") + out_path); }
    BgzfOut bz;
    bool ok = true;
    if (opt.sam_text) {
        if (hlen) { ok = fwrite(htext, 1, (size_t)hlen, fp) == (size_t)hlen; if (ok && htext[hlen - 1] != '\n') ok = fputc('\n', fp) != EOF; }
        const std::string t(htext ? htext : "", (size_t)hlen);
        if (!(t.compare(0, 4, "@SQ\t") == 0 || t.find("\n@SQ\t") != std::string::npos))
            for (int32_t i = 0; i < nt && ok; ++i) ok = fprintf(fp, "@SQ\tSN:%s\tLN:%lld\n", spx_bam_target_name(rd, i), (long long)spx_internal_bam_target_len(rd, i)) > 0;
    } else {
        bz.fp = fp;
        bz.threads = std::max(1, opt.threads);
        std::vector<uint8_t> h;
        h.insert(h.end(), {'B', 'A', 'M', 1});
        wr32(h, (int32_t)hlen);
        h.insert(h.end(), (const uint8_t *)htext, (const uint8_t *)htext + hlen);
        wr32(h, nt);
        for (int32_t i = 0; i < nt; ++i) {
            const char *nm = spx_bam_target_name(rd, i);
            const size_t ln = strlen(nm) + 1;
            wr32(h, (int32_t)ln);
            h.insert(h.end(), (const uint8_t *)nm, (const uint8_t *)nm + ln);
            wr32(h, (int32_t)spx_internal_bam_target_len(rd, i));
        }
        bz.write(h.data(), h.size(), false);
        bz.flush_block(); /* the header ends its block, as bam_hdr_write's bgzf_flush does */
    }
    std::vector<uint8_t> recbuf;
    std::string line;
    const spx_batch *bt = nullptr;
    int ng = 0;
    bool specific = false; /* a specific message has been recorded (spx_last_error): the generic one below must not replace it */
    while (ok && (ng = spx_bam_next_batch(rd, 16384, &bt)) > 0) {
        for (int32_t g = 0; g < bt->n_groups && ok; ++g) {
            const char *qname = bt->qnames + bt->qname_off[g];
            const bool excluded = !exclude.empty() && exclude.count(qname) != 0;
            const auto tit = table->by_name.empty() ? table->by_name.end() : table->by_name.find(qname);
            const auto mit = mapq_table.empty() ? mapq_table.end() : mapq_table.find(qname);
            for (int32_t a = bt->grp_first[g]; a < bt->grp_first[g + 1] && ok; ++a) {
                ++st.records_in;
                const uint8_t *rec = nullptr;
                int32_t bs = 0;
                if (spx_internal_bam_record(rd, bt, a, &rec, &bs) != SPX_OK) { ok = false; specific = true; break; }
                uint32_t flag = (uint32_t)(rec[14] | (rec[15] << 8));
                if (flag & SPX_FUNMAP) continue;  /* :349 */
                if (excluded) continue;           /* :350 */
                const int32_t refid = rd32(rec), pos = rd32(rec + 4);
                const char *contig = (refid >= 0 && refid < nt) ? spx_bam_target_name(rd, refid) : "*";
                bool prim;                        /* is_prim, :93-109 */
                if (tit != table->by_name.end()) prim = tit->second.contig == contig && tit->second.start == pos;
                else prim = (flag & SPX_FSECONDARY) == 0;
                const uint32_t flag_in = flag;
                if (prim) flag &= ~(uint32_t)SPX_FSECONDARY;
                else {
                    if (opt.primary_only) continue;
                    flag |= SPX_FSECONDARY;
                }
                /* get_read_length / get_alignment_length (:189-220) walk the CIGAR through ptCigarIt_next, which splits M ops by the cs
                 * tag into = and X pieces whose lengths add up to the op's: the sums are those of the CIGAR itself */
                const uint32_t l_name = rec[8], ncig = (uint32_t)(rec[12] | (rec[13] << 8));
                const int32_t lseq = rd32(rec + 16);
                const uint8_t *cig = rec + 32 + l_name;
                int64_t read_len = 0, aln_len = 0;
                const uint32_t *cw = bt->cigar + bt->cigar_off[a]; /* (the real CIGAR when the record keeps it in a CG tag) */
                for (int32_t k = 0; k < bt->n_cigar[a]; ++k) {
                    const uint32_t op = cw[k] & 15, len = cw[k] >> 4;
                    if (op == SPX_CMATCH || op == SPX_CEQUAL || op == SPX_CDIFF) { read_len += len; aln_len += len; }
                    else if (op == SPX_CINS || op == SPX_CSOFT_CLIP || op == SPX_CHARD_CLIP) read_len += len;
                }
                if (read_len < opt.min_read_length || aln_len < opt.min_alignment_length) continue; /* :359-362 */
                uint32_t mapq = rec[9];           /* get_mapq, :166-184 */
                if (mit != mapq_table.end())
                    for (const Location &l : mit->second)
                        if (l.contig == contig && l.start == pos) { mapq = l.mapq; break; }
                if (opt.max_mapq < (int32_t)mapq) continue; /* :364 */
                const uint8_t *aux = cig + 4 * (size_t)ncig + ((size_t)lseq + 1) / 2 + (size_t)lseq, *end = rec + bs;
                if (aux > end) { ok = false; specific = true; fail(SPX_EINVAL, "corrupt BAM record"); break; }
                if (opt.max_divergence < de_tag(aux, end)) continue; /* :365-366 */
                /* :367 -- bam1_t.l_data -= l_aux.  A record of more than 65 535 CIGAR operations keeps its real CIGAR in a CG:B,I tag behind a
                 * placeholder; htslib's sam_read1 moves it back in front of the tags before the reference cuts them and bam_write1 emits CG again,
                 * so the reference's output keeps it: here the CG tag alone survives the cut */
                int32_t bs_out = opt.no_tag ? (int32_t)(aux - rec) : bs;
                const uint8_t *cg = nullptr;
                size_t cg_len = 0;
                if (opt.no_tag && (uint32_t)bt->n_cigar[a] != ncig) {
                    for (const uint8_t *t = aux; t + 3 <= end;) { /* walk the tags to CG:B,I */
                        const uint8_t *t0 = t;
                        const char ty = (char)t[2];
                        t += 3;
                        size_t sz = 0;
                        if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                        else if (ty == 's' || ty == 'S') sz = 2;
                        else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                        else if (ty == 'Z' || ty == 'H') { while (t < end && *t) ++t; sz = 1; }
                        else if (ty == 'B' && t + 5 <= end) {
                            const char sub = (char)t[0];
                            const uint32_t cnt = (uint32_t)rd32(t + 1);
                            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                            sz = 5 + es * (size_t)cnt;
                        } else break;
                        if (t + sz > end) break;
                        t += sz;
                        if (t0[0] == 'C' && t0[1] == 'G' && ty == 'B') { cg = t0; cg_len = (size_t)(t - t0); break; }
                    }
                }
                recbuf.clear();
                wr32(recbuf, bs_out + (int32_t)cg_len);
                recbuf.insert(recbuf.end(), rec, rec + bs_out);
                if (cg) { recbuf.insert(recbuf.end(), cg, cg + cg_len); bs_out += (int32_t)cg_len; }
                recbuf[4 + 9] = (uint8_t)mapq;
                recbuf[4 + 14] = (uint8_t)(flag & 0xff);
                recbuf[4 + 15] = (uint8_t)(flag >> 8);
                if (opt.sam_text) {
                    line.clear();
                    if (spx_internal_format_sam(rd, recbuf.data() + 4, bs_out, &line) != SPX_OK) { ok = false; specific = true; break; }
                    if (fwrite(line.data(), 1, line.size(), fp) != line.size()) ok = false;
                } else
                    bz.write(recbuf.data(), recbuf.size(), true);
                ++st.records_out;
                if ((flag_in & SPX_FSECONDARY) && !(flag & SPX_FSECONDARY)) ++st.made_primary;
                if (!(flag_in & SPX_FSECONDARY) && (flag & SPX_FSECONDARY)) ++st.made_secondary;
            }
        }
        spx_bam_release_batch(rd, bt);
    }
    if (ng < 0) { ok = false; specific = true; fail(ng, spx_io_last_error()); }
    if (opt.sam_text) { if (fclose(fp) != 0) ok = false; }
    else if (!bz.close() || !bz.ok) ok = false;
    spx_bam_close(rd);
    spx_relabel_table_free(table);
    if (stats) *stats = st;
    if (!ok) return specific ? SPX_EINVAL : fail(SPX_EINVAL, std::string("correct_bam: could not read ") + in_path + " or write " + out_path);
    return SPX_OK;
}
