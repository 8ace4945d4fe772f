/*
 * spx_devin_kernels.hip -- the reference's record loop on the device: BAM record chain, fields, cs / MD / CG tags, name
 * groups, dispatch filter and the gather into the staged layout, all on inflated bytes that never leave HBM.
 *
 * What replaces what (/root/reference/programs): sam_read1 + the group scan of src/secphase.c:230-351 (records with the
 * same read name, in file order, form a group :273-279; unmapped records are skipped after the boundary check :336;
 * at most 11 records are kept :337), the dispatch filter :285-288, bam_aux_get for the cs / MD tag
 * (submodules/cigar_it/cigar_it.c:46-63) and, from htslib, the CG:B,I long-CIGAR convention sam_read1 undoes.
 *
 * Kernels (one segment = ~1 GB of inflated bytes, see spx_devin.h):
 *   chain_spec     one lane per BGZF block: walk the length fields from the block's first byte AS IF a record started
 *                  there (htslib starts a record at a block start whenever it fits the block) until the walk leaves the block
 *   chain_resolve  one workgroup: follows the REAL chain from the known first record through the per-block results (tiles
 *                  of them staged in LDS: one LDS round trip per block instead of a dependent HBM load per record);
 *                  where the chain enters a block in its middle (a record longer than a block, the front of the carry) it
 *                  walks record by record
 *   chain_emit     one lane per block on the chain: the record offsets
 *   parse          one lane per record: fixed fields, length validation, name compare with the previous record, aux walk
 *   group_first / batch_bounds / group_filter / slot / totals   name groups, dispatch filter, sizes (prefix sums: hipcub)
 *   image_groups / image_records / image_slots   the staged image (spxl::Rec + CIGAR / SEQ / QUAL / text pools) and the
 *                  few bytes per group the host keeps
 * What bounds them: latency of dependent loads for the walks (like the preparation kernels), HBM bandwidth for
 * image_slots (one read + one write of every payload byte).
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "spx_devin.h"

namespace {

__device__ __forceinline__ uint32_t ld32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

/* ---------------------------------------------------------------- record chain ---- */
__global__ __launch_bounds__(256) void chain_spec_kernel(spx_din_args A)
{
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= A.n_blocks) return;
    int64_t p = A.bstart[b];
    const int64_t end_b = A.bstart[b + 1];
    int32_t cnt = 0, flag = 0;
    while (p < end_b) {
        if (p + 4 > A.n_end) { flag = 1; break; }
        const int64_t bs = (int64_t)(int32_t)ld32(A.buf + p);
        if (bs < 32 || bs > A.max_rec) { flag = 2; break; }
        if (p + 4 + bs > A.n_end) { flag = 1; break; }
        ++cnt;
        p += 4 + bs;
    }
    A.land[b] = p;
    A.cnt[b] = cnt;
    A.bflag[b] = flag;
    A.first_idx[b] = -1;
}

/* (a small tile: the workgroup must find room on a CU that the inflate kernel of the NEXT segment keeps filled to the last
 * KB of LDS -- with 48 KB of tiles it waited for that whole grid to drain, ~80 ms per segment) */
constexpr int kTile = 256;

__global__ __launch_bounds__(256) void chain_resolve_kernel(spx_din_args A)
{
    __shared__ int64_t s_land[kTile], s_bstart[kTile + 1];
    __shared__ int32_t s_cnt[kTile], s_flag[kTile];
    __shared__ int32_t s_done, s_tile;
    /* thread 0's cursor */
    int64_t cur = A.p0, nrec = 0, tail = -1, err_at = 0;
    int32_t err = 0, b = 0, steps_guard = 0;
    int tile_lo = 0;
    const int64_t bstart0 = A.n_blocks > 0 ? A.bstart[0] : A.n_end;
    for (;;) {
        __syncthreads();
        const int tile_hi = min(A.n_blocks, tile_lo + kTile);
        for (int k = tile_lo + (int)threadIdx.x; k < tile_hi; k += (int)blockDim.x) {
            s_land[k - tile_lo] = A.land[k];
            s_cnt[k - tile_lo] = A.cnt[k];
            s_flag[k - tile_lo] = A.bflag[k];
        }
        for (int k = tile_lo + (int)threadIdx.x; k <= tile_hi; k += (int)blockDim.x) s_bstart[k - tile_lo] = k <= A.n_blocks ? A.bstart[k] : A.n_end;
        __syncthreads();
        if (threadIdx.x == 0) {
            int done = 0, next_tile = tile_lo;
            for (;;) {
                if (cur + 4 > A.n_end) { tail = cur; done = 1; break; }
                bool regular = false;
                if (cur >= bstart0) {
                    if (b < tile_lo) b = tile_lo;
                    while (b < tile_hi && s_bstart[b + 1 - tile_lo] <= cur) ++b;
                    if (b >= tile_hi) { /* the block of `cur` lies behind this tile (b < n_blocks: cur < n_end) */
                        next_tile = b;
                        break;
                    }
                    regular = cur == s_bstart[b - tile_lo];
                }
                if (regular) {
                    const int t = b - tile_lo;
                    A.first_idx[b] = (int32_t)nrec;
                    nrec += s_cnt[t];
                    if (s_flag[t] == 2) { err = 1; err_at = s_land[t]; done = 1; break; }
                    if (s_flag[t] == 1) { tail = s_land[t]; done = 1; break; }
                    cur = s_land[t]; /* >= the next block's start: the loop moves b on */
                    continue;
                }
                /* a record that starts in the middle of a block (or in the carry): one step on the inflated bytes */
                const int64_t bs = (int64_t)(int32_t)ld32(A.buf + cur);
                if (bs < 32 || bs > A.max_rec) { err = 1; err_at = cur; done = 1; break; }
                if (cur + 4 + bs > A.n_end) { tail = cur; done = 1; break; }
                if (nrec < A.rec_cap) A.R.off[nrec] = cur;
                ++nrec;
                cur += 4 + bs;
                if (++steps_guard < 0) { err = 2; done = 1; break; }
            }
            s_done = done;
            s_tile = next_tile;
        }
        __syncthreads();
        if (s_done) break;
        tile_lo = s_tile;
    }
    if (threadIdx.x == 0) {
        A.counts->n_rec = nrec;
        A.counts->tail_start = tail;
        A.counts->err = err;
        A.counts->err_at = err_at;
    }
}

__global__ __launch_bounds__(256) void chain_emit_kernel(spx_din_args A)
{
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= A.n_blocks) return;
    const int64_t i0 = A.first_idx[b];
    if (i0 < 0) return;
    int64_t p = A.bstart[b];
    const int32_t n = A.cnt[b];
    for (int32_t c = 0; c < n; ++c) {
        if (i0 + c < A.rec_cap) A.R.off[i0 + c] = p;
        p += 4 + (int64_t)(int32_t)ld32(A.buf + p);
    }
}

/* ---------------------------------------------------------------- fields and tags ---- */
/* One pass over the aux fields (htslib's layout: 2-byte key, 1-byte type, value): the first cs:Z, the first MD:Z and the
 * first CG:B seen BEFORE a field that cannot be walked (unknown type, unterminated string) -- what separate searches from
 * the start of the block find (spx_io.cpp find_tag / find_tag_b). */
struct AuxHit {
    int64_t cs_at, md_at, cg_at; /* value offsets, -1: absent */
    int32_t cs_len, md_len;
    uint32_t cg_n;
    uint8_t cg_sub;
};
__device__ void walk_aux(const uint8_t *buf, int64_t aux, int64_t end, bool want_cg, AuxHit &h)
{
    h.cs_at = h.md_at = h.cg_at = -1;
    h.cs_len = h.md_len = -1;
    h.cg_n = 0;
    h.cg_sub = 0;
    while (aux + 3 <= end) {
        const uint8_t k0 = buf[aux], k1 = buf[aux + 1], ty = buf[aux + 2];
        const int64_t v = aux + 3;
        int64_t len;
        switch (ty) {
        case 'A': case 'c': case 'C': len = 1; break;
        case 's': case 'S': len = 2; break;
        case 'i': case 'I': case 'f': len = 4; break;
        case 'Z': case 'H': {
            int64_t z = v;
            while (z < end && buf[z] != 0) ++z;
            if (z >= end) return; /* unterminated: the aux block is corrupt from here on */
            if (ty == 'Z' && k0 == 'c' && k1 == 's' && h.cs_at < 0) {
                h.cs_at = v;
                h.cs_len = (int32_t)(z - v);
                if (!want_cg) return; /* nothing behind the cs tag matters */
            }
            if (ty == 'Z' && k0 == 'M' && k1 == 'D' && h.md_at < 0) { h.md_at = v; h.md_len = (int32_t)(z - v); }
            len = z - v + 1;
            break;
        }
        case 'B': {
            if (v + 5 > end) return;
            const uint8_t sub = buf[v];
            const uint32_t cnt = ld32(buf + v + 1);
            const int64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            len = 5 + es * (int64_t)cnt;
            if (k0 == 'C' && k1 == 'G' && h.cg_at < 0 && want_cg) {
                if (v + len <= end) { h.cg_at = v + 5; h.cg_n = cnt; h.cg_sub = sub; }
                want_cg = false; /* (the host search returns at the first CG field, usable or not) */
                if (h.cs_at >= 0) return;
            }
            break;
        }
        default: return;
        }
        aux = v + len;
    }
}

__global__ __launch_bounds__(128) void parse_kernel(spx_din_args A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_rec = A.counts->n_rec < A.rec_cap ? A.counts->n_rec : A.rec_cap;
    if (r >= n_rec) return;
    const uint8_t *buf = A.buf;
    const int64_t at = A.R.off[r];
    const int64_t bs = (int64_t)(int32_t)ld32(buf + at);
    const int64_t p = at + 4, end = p + bs;
    const int32_t refid = (int32_t)ld32(buf + p), posv = (int32_t)ld32(buf + p + 4);
    const uint32_t l_name = buf[p + 8];
    const uint32_t ncig = ld16(buf + p + 12), flg = ld16(buf + p + 14);
    const int32_t lseq = (int32_t)ld32(buf + p + 16);
    /* the fixed part announces the lengths of the variable part: none may reach past the record, the name is NUL-terminated */
    if (lseq < 0 || l_name < 1 || 32 + (uint64_t)l_name + 4 * (uint64_t)ncig + ((uint64_t)lseq + 1) / 2 + (uint64_t)lseq > (uint64_t)bs ||
        buf[p + 32 + l_name - 1] != 0) {
        if (atomicCAS(&A.counts->err, 0, 3) == 0) A.counts->err_at = at;
        A.R.isnew[r] = 1;
        A.R.flag[r] = 4; /* (never looked at: the batch is refused) */
        A.R.lname[r] = 1;
        return;
    }
    const int64_t cig = p + 32 + l_name, sq = cig + 4 * (int64_t)ncig, ql = sq + ((int64_t)lseq + 1) / 2, aux = ql + lseq;
    /* more than 65535 CIGAR operations: the record carries <l_seq>S<ref_len>N and the real CIGAR in CG:B,I (sam_read1 puts it back) */
    bool want_cg = false;
    if (ncig == 2) {
        const uint32_t c0 = ld32(buf + cig), c1 = ld32(buf + cig + 4);
        want_cg = (c0 & 0xf) == SPX_CSOFT_CLIP && (c0 >> 4) == (uint32_t)lseq && (c1 & 0xf) == SPX_CREF_SKIP;
    }
    AuxHit h;
    walk_aux(buf, aux, end, want_cg, h);
    int64_t cig_at = cig;
    int32_t n_cig = (int32_t)ncig;
    if (h.cg_at >= 0 && (h.cg_sub == 'I' || h.cg_sub == 'i') && h.cg_n > 0 && h.cg_at + 4 * (int64_t)h.cg_n <= end) {
        cig_at = h.cg_at;
        n_cig = (int32_t)h.cg_n;
    }
    A.R.flag[r] = (int32_t)flg;
    A.R.tid[r] = (refid >= 0 && refid < A.n_targets) ? A.tmap[refid] : -1;
    A.R.pos[r] = posv;
    A.R.lq[r] = lseq;
    A.R.ncig[r] = n_cig;
    A.R.cig_at[r] = cig_at;
    A.R.lname[r] = (int32_t)l_name;
    if (h.cs_at >= 0) { A.R.tag_at[r] = h.cs_at; A.R.cs_len[r] = h.cs_len; A.R.md_len[r] = -1; }
    else if (h.md_at >= 0) { A.R.tag_at[r] = h.md_at; A.R.cs_len[r] = -1; A.R.md_len[r] = h.md_len; }
    else { A.R.tag_at[r] = -1; A.R.cs_len[r] = -1; A.R.md_len[r] = -1; }
    /* group boundary on a name change (src/secphase.c:273-279); the chain of a segment starts at a group start */
    int32_t isnew = 1;
    if (r > 0) {
        const int64_t q = A.R.off[r - 1] + 4;
        const uint32_t ln2 = buf[q + 8];
        if (ln2 == l_name) {
            isnew = 0;
            for (uint32_t k = 0; k + 1 < l_name; ++k)
                if (buf[q + 32 + k] != buf[p + 32 + k]) { isnew = 1; break; }
        }
    }
    A.R.isnew[r] = isnew;
}

/* ---------------------------------------------------------------- groups ---- */
__global__ __launch_bounds__(256) void group_first_kernel(spx_din_args A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_rec = A.counts->n_rec;
    if (r >= n_rec) return;
    if (A.R.isnew[r]) A.grp_first[A.R.gid[r] - 1] = (int32_t)r;
    if (r == n_rec - 1) A.grp_first[A.R.gid[r]] = (int32_t)n_rec;
}

__global__ void batch_bounds_kernel(spx_din_args A)
{
    spx_din_counts &C = *A.counts;
    const int64_t n_rec = C.n_rec;
    const int64_t ng = n_rec > 0 ? A.R.gid[n_rec - 1] : 0;
    C.n_groups = ng;
    /* the last group may continue in the next segment: it is handed on with that one */
    const int64_t nb = A.is_final ? ng : (ng > 0 ? ng - 1 : 0);
    C.n_batch = nb;
    C.n_batch_rec = nb > 0 ? A.grp_first[nb] : 0;
    C.carry_start = nb < ng ? A.R.off[A.grp_first[nb]] : C.tail_start;
}

struct GAdd {
    __host__ __device__ spx_din_group_scan operator()(const spx_din_group_scan &a, const spx_din_group_scan &b) const
    {
        return spx_din_group_scan{a.disp + b.disp, a.slots + b.slots, a.name_bytes + b.name_bytes};
    }
};
struct SAdd {
    __host__ __device__ spx_din_slot_scan operator()(const spx_din_slot_scan &a, const spx_din_slot_scan &b) const
    {
        return spx_din_slot_scan{a.cw + b.cw, a.sb + b.sb, a.qb + b.qb, a.tb + b.tb, a.oc + b.oc, a.cc + b.cc, a.mc + b.mc};
    }
};

/* the records of group g that the reference keeps (mapped ones, at most 11: src/secphase.c:336-337) and the dispatch
 * filter over them (:285-288): 2..10 records, no supplementary one, exactly one that is not secondary */
__device__ int kept_records(const spx_din_args &A, int64_t g, int32_t *rec)
{
    int n = 0;
    for (int32_t a = A.grp_first[g]; a < A.grp_first[g + 1]; ++a) {
        if (A.R.flag[a] & SPX_FUNMAP) continue;
        if (n > 10) continue;
        rec[n++] = a;
    }
    return n;
}

__global__ __launch_bounds__(256) void group_filter_kernel(spx_din_args A, int64_t n_items)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_items) return;
    spx_din_group_scan v = {0, 0, 0};
    if (g < A.counts->n_batch) {
        int32_t rec[12];
        const int n = kept_records(A, g, rec);
        int supp = 0, prim = 0;
        for (int i = 0; i < n; ++i) {
            if (A.R.flag[rec[i]] & SPX_FSUPPLEMENTARY) ++supp;
            if (!(A.R.flag[rec[i]] & SPX_FSECONDARY)) ++prim;
        }
        const bool disp = n > 1 && n <= 10 && supp == 0 && prim == 1;
        v.disp = disp ? 1 : 0;
        v.slots = disp ? n : 0;
        v.name_bytes = A.R.lname[A.grp_first[g]];
    }
    A.gscan[g] = v;
}

__global__ __launch_bounds__(256) void slot_kernel(spx_din_args A)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.counts->n_batch) return;
    const spx_din_group_scan me = A.gscan[g], nx = A.gscan[g + 1];
    if (nx.disp == me.disp) return; /* not dispatched */
    int32_t rec[12];
    const int n = kept_records(A, g, rec);
    for (int i = 0; i < n; ++i) {
        const int64_t s = me.slots + i;
        const int32_t a = rec[i];
        A.slot_rec[s] = a;
        A.slot_grp[s] = (int32_t)me.disp;
        spxl::Rec r;
        r.n_cigar = A.R.ncig[a];
        r.cs_len = A.R.cs_len[a];
        r.md_len = A.R.md_len[a];
        r.l_qseq = A.R.lq[a];
        int32_t oc, cc, mc;
        spxl::aln_caps(r, oc, cc, mc);
        const int64_t lq = r.l_qseq > 0 ? r.l_qseq : 0;
        spx_din_slot_scan v;
        v.cw = r.n_cigar > 0 ? r.n_cigar : 0;
        v.sb = (((lq + 1) / 2) + 3) & ~(int64_t)3;
        v.qb = lq;
        v.tb = (r.cs_len >= 0 ? r.cs_len : r.md_len >= 0 ? r.md_len : 0) + 1;
        v.oc = oc; v.cc = cc; v.mc = mc;
        A.sscan[s] = v;
    }
}

__global__ void totals_kernel(spx_din_args A, int64_t n_items)
{
    spx_din_counts &C = *A.counts;
    const spx_din_group_scan g = A.gscan[n_items];
    const spx_din_slot_scan s = A.sscan[n_items];
    C.n_dgroups = g.disp;
    C.n_slots = g.slots;
    C.name_bytes = g.name_bytes;
    C.cigar_words = s.cw; C.seq_bytes = s.sb; C.qual_bytes = s.qb; C.text_bytes = s.tb;
    C.ops_bound = s.oc; C.conf_bound = s.cc; C.mm_bound = s.mc;
}

/* ---------------------------------------------------------------- the staged image ---- */
/* A segment's groups become ONE work list, or several (ranges of groups) when there are more than a list may hold. */
__global__ __launch_bounds__(256) void image_groups_kernel(spx_din_args A, spx_din_out O, spx_din_range Q)
{
    const int64_t g = Q.g0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g == Q.g0) {
        O.slot0[Q.gend.disp - Q.gbase.disp] = (int32_t)(Q.gend.slots - Q.gbase.slots);
        O.h_grp_first[Q.g1 - Q.g0] = (int32_t)(Q.r1 - Q.r0);
    }
    if (g >= Q.g1) return;
    const spx_din_group_scan me = A.gscan[g], nx = A.gscan[g + 1];
    const bool disp = nx.disp != me.disp;
    const int64_t k = g - Q.g0, nb0 = me.name_bytes - Q.gbase.name_bytes;
    O.grp_disp[k] = disp ? 1 : 0;
    O.name_off[k] = nb0;
    O.h_grp_first[k] = (int32_t)(A.grp_first[g] - Q.r0);
    const int64_t q = A.R.off[A.grp_first[g]] + 4 + 32;
    const int32_t ln = A.R.lname[A.grp_first[g]];
    for (int32_t c = 0; c < ln; ++c) O.names[nb0 + c] = (char)A.buf[q + c];
    if (disp) {
        O.gidx[me.disp - Q.gbase.disp] = (int32_t)k;
        O.slot0[me.disp - Q.gbase.disp] = (int32_t)(me.slots - Q.gbase.slots);
    }
}

__global__ __launch_bounds__(256) void image_records_kernel(spx_din_args A, spx_din_out O, spx_din_range Q)
{
    const int64_t r = Q.r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Q.r1) return;
    O.h_flag[r - Q.r0] = (uint16_t)A.R.flag[r];
    O.h_tid[r - Q.r0] = A.R.tid[r];
    O.h_pos[r - Q.r0] = A.R.pos[r];
}

/* n bytes from src (any alignment) to dst (any alignment) by the whole workgroup: 16-byte stores on the aligned middle */
__device__ void copy_bytes(uint8_t *dst, const uint8_t *src, int64_t n)
{
    const int t = (int)threadIdx.x, nt = (int)blockDim.x;
    int64_t head = (int64_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (head > n) head = n;
    if (t < head) dst[t] = src[t];
    const int64_t nv = (n - head) >> 4;
    for (int64_t i = t; i < nv; i += nt) {
        uint4 v;
        __builtin_memcpy(&v, src + head + 16 * i, 16);
        *reinterpret_cast<uint4 *>(dst + head + 16 * i) = v;
    }
    const int64_t done = head + 16 * nv;
    if (t < n - done) dst[done + t] = src[done + t];
}

__global__ __launch_bounds__(256) void image_slots_kernel(spx_din_args A, spx_din_out O, spx_din_range Q)
{
    const int64_t s = Q.s0 + blockIdx.x, k = blockIdx.x;
    if (s >= Q.s1) return;
    const int32_t a = A.slot_rec[s];
    spx_din_slot_scan at = A.sscan[s];
    at.cw -= Q.sbase.cw; at.sb -= Q.sbase.sb; at.qb -= Q.sbase.qb; at.tb -= Q.sbase.tb;
    const int64_t rec_at = A.R.off[a] + 4;
    const int32_t lq = A.R.lq[a] > 0 ? A.R.lq[a] : 0, ncig = A.R.ncig[a], cs_len = A.R.cs_len[a], md_len = A.R.md_len[a];
    if (threadIdx.x == 0) {
        spxl::Rec r;
        r.rec = (int32_t)(a - Q.r0); r.batch = 0; r.grp = (int32_t)(A.slot_grp[s] - Q.gbase.disp);
        r.flag = A.R.flag[a]; r.tid = A.R.tid[a]; r.pos = A.R.pos[a]; r.l_qseq = A.R.lq[a]; r.n_cigar = ncig;
        r.cs_len = cs_len; r.md_len = md_len;
        r.cigar_off = at.cw; r.seq_off = at.sb; r.qual_off = at.qb; r.tag_off = at.tb;
        r.pk_seq_off = at.sb; r.pk_qual_off = at.qb;
        r.alias_slot = -1; r.alias_shift = 0; r.alias_rev = 0; r.pad_ = 0;
        O.recs[k] = r;
    }
    /* CIGAR words (the source is byte-aligned only) */
    {
        const uint8_t *src = A.buf + A.R.cig_at[a];
        uint32_t *dst = O.cigar + at.cw;
        for (int32_t i = (int32_t)threadIdx.x; i < ncig; i += (int32_t)blockDim.x) dst[i] = ld32(src + 4 * (int64_t)i);
    }
    /* SEQ as BAM stores it, zero-padded to whole words; QUAL */
    const uint32_t l_name = A.buf[rec_at + 8], ncig_rec = ld16(A.buf + rec_at + 12);
    const uint8_t *sq = A.buf + rec_at + 32 + l_name + 4 * (int64_t)ncig_rec;
    const int64_t nb = ((int64_t)lq + 1) / 2, sbytes = (nb + 3) & ~(int64_t)3;
    copy_bytes(O.seq + at.sb, sq, nb);
    if ((int64_t)threadIdx.x < sbytes - nb) O.seq[at.sb + nb + threadIdx.x] = 0;
    copy_bytes(O.qual + at.qb, sq + nb, lq);
    /* cs (or MD) text with its terminator */
    const int32_t tl = cs_len >= 0 ? cs_len : md_len >= 0 ? md_len : 0;
    if (tl > 0) copy_bytes((uint8_t *)O.text + at.tb, A.buf + A.R.tag_at[a], tl);
    if (threadIdx.x == 0) O.text[at.tb + tl] = 0;
}

/* worst inflate status of a segment's blocks into the counts block */
__global__ __launch_bounds__(256) void inflate_status_kernel(const int32_t *status, int32_t n, spx_din_counts *C)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < n && status[k] != 0) atomicMax(&C->inflate_bad, -status[k]);
}

} // namespace

extern "C" size_t spx_din_scan_temp_bytes(int64_t n_items)
{
    size_t a = 0, b = 0, c = 0;
    const int n = (int)(n_items + 1);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, a, (const int32_t *)nullptr, (int32_t *)nullptr, n);
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, b, (spx_din_group_scan *)nullptr, (spx_din_group_scan *)nullptr, GAdd(), spx_din_group_scan{0, 0, 0}, n);
    (void)hipcub::DeviceScan::ExclusiveScan(nullptr, c, (spx_din_slot_scan *)nullptr, (spx_din_slot_scan *)nullptr, SAdd(), spx_din_slot_scan{0, 0, 0, 0, 0, 0, 0}, n);
    return std::max(a, std::max(b, c)) + 256;
}

/* record chain of a segment: counts->n_rec, tail_start, err; R.off filled (up to rec_cap entries) */
extern "C" hipError_t spx_din_chain(const spx_din_args *A, hipStream_t st)
{
    const unsigned nb = (unsigned)((A->n_blocks + 255) / 256);
    if (nb) hipLaunchKernelGGL(chain_spec_kernel, dim3(nb), dim3(256), 0, st, *A);
    hipLaunchKernelGGL(chain_resolve_kernel, dim3(1), dim3(256), 0, st, *A);
    if (nb) hipLaunchKernelGGL(chain_emit_kernel, dim3(nb), dim3(256), 0, st, *A);
    return hipGetLastError();
}

/* fields, tags, groups, dispatch filter, sizes: everything up to the counts the host needs to carve the staged image.
 * n_rec = the chain's record count (host copy); temp: spx_din_scan_temp_bytes(n_rec) */
extern "C" hipError_t spx_din_groups(const spx_din_args *A, int64_t n_rec, void *temp, size_t temp_bytes, hipStream_t st)
{
    if (n_rec <= 0) {
        hipLaunchKernelGGL(batch_bounds_kernel, dim3(1), dim3(1), 0, st, *A);
        return hipGetLastError();
    }
    const unsigned gr = (unsigned)((n_rec + 127) / 128), g256 = (unsigned)((n_rec + 1 + 255) / 256);
    hipLaunchKernelGGL(parse_kernel, dim3(gr), dim3(128), 0, st, *A);
    size_t tb = temp_bytes;
    hipError_t e = hipcub::DeviceScan::InclusiveSum(temp, tb, A->R.isnew, A->R.gid, (int)n_rec, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(group_first_kernel, dim3(g256), dim3(256), 0, st, *A);
    hipLaunchKernelGGL(batch_bounds_kernel, dim3(1), dim3(1), 0, st, *A);
    hipLaunchKernelGGL(group_filter_kernel, dim3(g256), dim3(256), 0, st, *A, n_rec + 1);
    tb = temp_bytes;
    e = hipcub::DeviceScan::ExclusiveScan(temp, tb, A->gscan, A->gscan, GAdd(), spx_din_group_scan{0, 0, 0}, (int)(n_rec + 1), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(A->sscan, 0, sizeof(spx_din_slot_scan) * (size_t)(n_rec + 1), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(slot_kernel, dim3(g256), dim3(256), 0, st, *A);
    tb = temp_bytes;
    e = hipcub::DeviceScan::ExclusiveScan(temp, tb, A->sscan, A->sscan, SAdd(), spx_din_slot_scan{0, 0, 0, 0, 0, 0, 0}, (int)(n_rec + 1), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(totals_kernel, dim3(1), dim3(1), 0, st, *A, n_rec);
    return hipGetLastError();
}

/* the staged image of the groups [Q.g0, Q.g1) + what the host keeps of them */
extern "C" hipError_t spx_din_image(const spx_din_args *A, const spx_din_out *O, const spx_din_range *Q, hipStream_t st)
{
    const int64_t ng = Q->g1 - Q->g0, nr = Q->r1 - Q->r0, ns = Q->s1 - Q->s0;
    hipLaunchKernelGGL(image_groups_kernel, dim3((unsigned)((ng + 1 + 255) / 256)), dim3(256), 0, st, *A, *O, *Q);
    if (nr > 0) hipLaunchKernelGGL(image_records_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, *A, *O, *Q);
    if (ns > 0) hipLaunchKernelGGL(image_slots_kernel, dim3((unsigned)ns), dim3(256), 0, st, *A, *O, *Q);
    return hipGetLastError();
}

extern "C" hipError_t spx_din_inflate_status(const int32_t *status, int32_t n, spx_din_counts *C, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(inflate_status_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, status, n, C);
    return hipGetLastError();
}
