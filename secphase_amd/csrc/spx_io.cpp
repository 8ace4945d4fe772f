/*
 * spx_io.cpp -- input side of the drop-in: name-grouped BAM -> spx_batch blocks, FASTA -> spx_ref.
 *
 * The reference reads through htslib (sam_open/sam_read1 at src/secphase.c:236-268, fai_load/fai_fetch at
 * src/secphase.c:101, submodules/ptMarker/ptMarker.c:739-744).  htslib is not available in this environment,
 * so this is a small self-contained reader on zlib: BGZF blocks are inflated in parallel, BAM records are
 * copied field by field into the flat record format (SEQ and CIGAR keep BAM's own packing).
 * Grouping follows src/secphase.c:273-279: consecutive records with the same read name form a group.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx.h"

namespace {

thread_local std::string g_io_err;

struct Block { size_t coff, clen, uoff, ulen; };

struct Reader {
    FILE *fp = nullptr;
    std::vector<uint8_t> cbuf;  /* compressed blocks of the current chunk */
    std::vector<uint8_t> ubuf;  /* inflated bytes not yet consumed + current chunk */
    size_t upos = 0;
    bool eof = false;
    int threads = 4;
    /* header */
    std::vector<std::string> tname;
    std::vector<int64_t> tlen;
    std::vector<int32_t> tmap; /* BAM tid -> contig index of the reference handed to the scorer (-1 unknown) */
    /* batch storage */
    std::vector<int32_t> grp_first, tid, pos, l_qseq, n_cigar;
    std::vector<int64_t> qname_off, cigar_off, seq_off, qual_off, cs_off;
    std::vector<uint16_t> flag;
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> seq4, qual;
    std::vector<char> qnames, cs;
    spx_batch view;
    std::string last_name;
    bool have_last = false;
    int64_t n_records = 0, n_groups_total = 0;
};

bool read_chunk(Reader *r)
{
    /* read up to 256 BGZF blocks, inflate them in parallel, append to ubuf */
    if (r->eof) return false;
    if (r->upos > 0) {
        r->ubuf.erase(r->ubuf.begin(), r->ubuf.begin() + (long)r->upos);
        r->upos = 0;
    }
    r->cbuf.clear();
    std::vector<Block> blocks;
    size_t utot = r->ubuf.size();
    for (int n = 0; n < 256; ++n) {
        uint8_t hdr[18];
        size_t got = fread(hdr, 1, 18, r->fp);
        if (got == 0) { r->eof = true; break; }
        if (got != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { g_io_err = "not a BGZF block"; r->eof = true; return false; }
        const unsigned xlen = hdr[10] | (hdr[11] << 8);
        /* the BC subfield is first in every BAM written by htslib/samtools; general case: scan the extra field */
        std::vector<uint8_t> extra(xlen);
        memcpy(extra.data(), hdr + 12, std::min<size_t>(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, r->fp) != xlen - 6) { g_io_err = "truncated BGZF header"; r->eof = true; return false; }
        int bsize = -1;
        for (size_t o = 0; o + 4 <= xlen;) {
            const unsigned slen = extra[o + 2] | (extra[o + 3] << 8);
            if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2) bsize = extra[o + 4] | (extra[o + 5] << 8);
            o += 4 + slen;
        }
        if (bsize < 0) { g_io_err = "BGZF block without BC field"; r->eof = true; return false; }
        const size_t clen = (size_t)bsize + 1 - 12 - xlen; /* deflate data + crc32 + isize */
        const size_t coff = r->cbuf.size();
        r->cbuf.resize(coff + clen);
        if (fread(r->cbuf.data() + coff, 1, clen, r->fp) != clen || clen < 8) { g_io_err = "truncated BGZF block"; r->eof = true; return false; }
        const uint8_t *t = r->cbuf.data() + coff + clen - 4;
        const size_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((size_t)t[3] << 24);
        blocks.push_back({coff, clen - 8, utot, isize});
        utot += isize;
    }
    if (blocks.empty()) return false;
    r->ubuf.resize(utot);
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    auto work = [&]() {
        for (;;) {
            size_t k = next.fetch_add(1);
            if (k >= blocks.size()) break;
            const Block &b = blocks[k];
            if (b.ulen == 0) continue;
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; continue; }
            zs.next_in = r->cbuf.data() + b.coff; zs.avail_in = (uInt)b.clen;
            zs.next_out = r->ubuf.data() + b.uoff; zs.avail_out = (uInt)b.ulen;
            int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0) bad = 1;
            inflateEnd(&zs);
        }
    };
    int nt = std::max(1, std::min<int>(r->threads, (int)blocks.size()));
    if (nt == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work);
        for (auto &t : th) t.join();
    }
    if (bad) { g_io_err = "inflate failed"; r->eof = true; return false; }
    return true;
}

/* make sure n bytes are available at upos; false at clean EOF / error */
bool need(Reader *r, size_t n)
{
    while (r->ubuf.size() - r->upos < n)
        if (!read_chunk(r)) return r->ubuf.size() - r->upos >= n;
    return true;
}

inline int32_t le32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }

void clear_batch(Reader *r)
{
    r->grp_first.clear(); r->tid.clear(); r->pos.clear(); r->l_qseq.clear(); r->n_cigar.clear(); r->qname_off.clear();
    r->cigar_off.clear(); r->seq_off.clear(); r->qual_off.clear(); r->cs_off.clear(); r->flag.clear(); r->cigar.clear();
    r->seq4.clear(); r->qual.clear(); r->qnames.clear(); r->cs.clear();
}

/* find the cs:Z tag in the aux block */
const char *find_cs(const uint8_t *aux, const uint8_t *end)
{
    while (aux + 3 <= end) {
        const char t0 = (char)aux[0], t1 = (char)aux[1], ty = (char)aux[2];
        const uint8_t *v = aux + 3;
        size_t len;
        switch (ty) {
        case 'A': case 'c': case 'C': len = 1; break;
        case 's': case 'S': len = 2; break;
        case 'i': case 'I': case 'f': len = 4; break;
        case 'Z': case 'H': {
            const uint8_t *z = v;
            while (z < end && *z) ++z;
            if (t0 == 'c' && t1 == 's' && ty == 'Z') return (const char *)v;
            len = (size_t)(z - v) + 1;
            break;
        }
        case 'B': {
            if (v + 5 > end) return nullptr;
            const char sub = (char)v[0];
            const uint32_t cnt = (uint32_t)le32(v + 1);
            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            len = 5 + es * cnt;
            break;
        }
        default: return nullptr;
        }
        aux = v + len;
    }
    return nullptr;
}

} // namespace

struct spx_bam_reader { Reader r; };
struct spx_fasta {
    std::vector<int64_t> name_off, seq_off;
    std::vector<char> names, bases;
    spx_ref ref;
};

extern "C" const char *spx_io_last_error(void) { return g_io_err.c_str(); }

extern "C" int spx_bam_open(const char *path, int threads, spx_bam_reader **out)
{
    if (!path || !out) return SPX_EINVAL;
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) { g_io_err = std::string("cannot open ") + path; return SPX_EINVAL; }
    spx_bam_reader *h = new spx_bam_reader();
    Reader *r = &h->r;
    r->fp = fp;
    r->threads = threads > 0 ? threads : 4;
    if (!need(r, 12) || memcmp(r->ubuf.data() + r->upos, "BAM\1", 4) != 0) { g_io_err = "not a BAM file"; fclose(fp); delete h; return SPX_EINVAL; }
    const int32_t l_text = le32(r->ubuf.data() + r->upos + 4);
    r->upos += 8;
    if (!need(r, (size_t)l_text + 4)) { g_io_err = "truncated BAM header"; fclose(fp); delete h; return SPX_EINVAL; }
    r->upos += (size_t)l_text;
    const int32_t n_ref = le32(r->ubuf.data() + r->upos);
    r->upos += 4;
    for (int32_t i = 0; i < n_ref; ++i) {
        if (!need(r, 4)) { g_io_err = "truncated BAM header"; fclose(fp); delete h; return SPX_EINVAL; }
        const int32_t ln = le32(r->ubuf.data() + r->upos);
        r->upos += 4;
        if (!need(r, (size_t)ln + 4)) { g_io_err = "truncated BAM header"; fclose(fp); delete h; return SPX_EINVAL; }
        r->tname.emplace_back((const char *)r->ubuf.data() + r->upos);
        r->upos += (size_t)ln;
        r->tlen.push_back(le32(r->ubuf.data() + r->upos));
        r->upos += 4;
    }
    r->tmap.assign(n_ref, -1);
    for (int32_t i = 0; i < n_ref; ++i) r->tmap[i] = i;
    *out = h;
    return SPX_OK;
}

extern "C" int32_t spx_bam_n_targets(const spx_bam_reader *h) { return h ? (int32_t)h->r.tname.size() : 0; }
extern "C" const char *spx_bam_target_name(const spx_bam_reader *h, int32_t i)
{
    return (h && i >= 0 && (size_t)i < h->r.tname.size()) ? h->r.tname[i].c_str() : nullptr;
}

/* BAM target ids -> contig indices of `ref` (by name); alignments on contigs the FASTA lacks get tid -1
 * and their group is rejected by the scorer with SPX_EINVAL */
extern "C" int spx_bam_bind_reference(spx_bam_reader *h, const spx_ref *ref)
{
    if (!h || !ref) return SPX_EINVAL;
    Reader *r = &h->r;
    int missing = 0;
    for (size_t i = 0; i < r->tname.size(); ++i) {
        r->tmap[i] = -1;
        for (int32_t c = 0; c < ref->n_contigs; ++c)
            if (r->tname[i] == ref->names + ref->name_off[c]) { r->tmap[i] = c; break; }
        if (r->tmap[i] < 0) ++missing;
    }
    return missing;
}

/* up to max_groups complete name groups; the batch stays valid until the next call.  Returns the number of
 * groups (0 at end of file) or SPX_E*. */
extern "C" int spx_bam_next_batch(spx_bam_reader *h, int32_t max_groups, const spx_batch **out)
{
    if (!h || !out || max_groups <= 0) return SPX_EINVAL;
    Reader *r = &h->r;
    clear_batch(r);
    int32_t ng = 0;
    bool open_group = false;
    for (;;) {
        if (!need(r, 4)) break;
        const int32_t bs = le32(r->ubuf.data() + r->upos);
        if (bs < 32 || !need(r, (size_t)bs + 4)) { if (bs >= 32) g_io_err = "truncated BAM record"; break; }
        const uint8_t *p = r->ubuf.data() + r->upos + 4;
        const int32_t refid = le32(p), posv = le32(p + 4);
        const uint32_t l_name = p[8];
        const uint32_t ncig = p[12] | (p[13] << 8), flg = p[14] | (p[15] << 8);
        const int32_t lseq = le32(p + 16);
        const char *name = (const char *)p + 32;
        /* group boundary on a name change (src/secphase.c:273-279) */
        const bool same = r->have_last && r->last_name == name;
        if (!same) {
            if (open_group && ng == max_groups) break; /* leave this record for the next batch */
            r->grp_first.push_back((int32_t)r->flag.size());
            r->qname_off.push_back((int64_t)r->qnames.size());
            r->qnames.insert(r->qnames.end(), name, name + strlen(name) + 1);
            r->last_name = name;
            r->have_last = true;
            open_group = true;
            ++ng;
        } else if (!open_group) {
            /* cannot happen: a batch always ends on a group boundary */
            r->grp_first.push_back((int32_t)r->flag.size());
            r->qname_off.push_back((int64_t)r->qnames.size());
            r->qnames.insert(r->qnames.end(), name, name + strlen(name) + 1);
            open_group = true;
            ++ng;
        }
        const uint8_t *cig = p + 32 + l_name, *sq = cig + 4 * (size_t)ncig, *ql = sq + ((size_t)lseq + 1) / 2, *aux = ql + lseq,
                      *end = p + bs;
        r->flag.push_back((uint16_t)flg);
        r->tid.push_back((refid >= 0 && (size_t)refid < r->tmap.size()) ? r->tmap[refid] : -1);
        r->pos.push_back(posv);
        r->l_qseq.push_back(lseq);
        r->n_cigar.push_back((int32_t)ncig);
        r->cigar_off.push_back((int64_t)r->cigar.size());
        for (uint32_t k = 0; k < ncig; ++k) r->cigar.push_back((uint32_t)le32(cig + 4 * k));
        r->seq_off.push_back((int64_t)r->seq4.size());
        r->seq4.insert(r->seq4.end(), sq, sq + ((size_t)lseq + 1) / 2);
        r->qual_off.push_back((int64_t)r->qual.size());
        r->qual.insert(r->qual.end(), ql, ql + lseq);
        const char *csz = aux <= end ? find_cs(aux, end) : nullptr;
        if (csz) {
            r->cs_off.push_back((int64_t)r->cs.size());
            r->cs.insert(r->cs.end(), csz, csz + strlen(csz) + 1);
        } else r->cs_off.push_back(-1);
        r->upos += (size_t)bs + 4;
        r->n_records++;
    }
    r->grp_first.push_back((int32_t)r->flag.size());
    r->seq4.push_back(0); r->qual.push_back(0); r->cs.push_back(0); r->qnames.push_back(0); r->cigar.push_back(0);
    spx_batch &b = r->view;
    b.n_groups = ng;
    b.n_alns = (int32_t)r->flag.size();
    b.grp_first = r->grp_first.data(); b.qname_off = r->qname_off.data(); b.qnames = r->qnames.data();
    b.flag = r->flag.data(); b.tid = r->tid.data(); b.pos = r->pos.data(); b.l_qseq = r->l_qseq.data();
    b.n_cigar = r->n_cigar.data(); b.cigar_off = r->cigar_off.data(); b.seq_off = r->seq_off.data();
    b.qual_off = r->qual_off.data(); b.cs_off = r->cs_off.data(); b.cigar = r->cigar.data(); b.seq4 = r->seq4.data();
    b.qual = r->qual.data(); b.cs = r->cs.data();
    *out = &r->view;
    r->n_groups_total += ng;
    return ng;
}

extern "C" void spx_bam_close(spx_bam_reader *h)
{
    if (!h) return;
    if (h->r.fp) fclose(h->r.fp);
    delete h;
}

/* whole FASTA into RAM (the scorer keeps its own 4-bit copy in HBM; this one feeds spx_set_reference and
 * the contig names of the relabel list) */
extern "C" int spx_fasta_load(const char *path, spx_fasta **out)
{
    if (!path || !out) return SPX_EINVAL;
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) { g_io_err = std::string("cannot open ") + path; return SPX_EINVAL; }
    spx_fasta *f = new spx_fasta();
    std::vector<char> buf(1 << 22);
    bool in_name = false, name_done = false, bol = true;
    size_t got;
    while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
        for (size_t i = 0; i < got; ++i) {
            const char c = buf[i];
            if (in_name) {
                if (c == '\n') { f->names.push_back(0); in_name = false; bol = true; }
                else if (!name_done) {
                    if (c == ' ' || c == '\t' || c == '\r') name_done = true;
                    else f->names.push_back(c);
                }
                continue;
            }
            if (c == '\n') { bol = true; continue; }
            if (bol && c == '>') {
                f->name_off.push_back((int64_t)f->names.size());
                f->seq_off.push_back((int64_t)f->bases.size());
                in_name = true; name_done = false; bol = false;
                continue;
            }
            bol = false;
            if (c == '\r' || c == ' ' || c == '\t') continue;
            if (f->seq_off.empty()) continue; /* junk before the first header */
            f->bases.push_back(c);
        }
    }
    fclose(fp);
    if (in_name) f->names.push_back(0);
    f->seq_off.push_back((int64_t)f->bases.size());
    f->bases.push_back(0);
    f->ref.n_contigs = (int32_t)f->name_off.size();
    f->ref.name_off = f->name_off.data();
    f->ref.names = f->names.data();
    f->ref.seq_off = f->seq_off.data();
    f->ref.bases = f->bases.data();
    *out = f;
    return SPX_OK;
}
extern "C" const spx_ref *spx_fasta_ref(const spx_fasta *f) { return f ? &f->ref : nullptr; }
extern "C" void spx_fasta_free(spx_fasta *f) { delete f; }
