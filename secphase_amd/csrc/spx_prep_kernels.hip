/*
 * spx_prep_kernels.hip -- work-list preparation ON THE DEVICE (gfx950): the integer part of the secphase marker path
 * (SURVEY.md section 8 rows A1-A8 and A10's control flow; /root/reference/programs/submodules/cigar_it/cigar_it.c,
 * ptAlignment/ptAlignment.c:42-95, ptMarker/ptMarker.c:42-107,156-295,328-667,670-831, src/secphase.c:162-170).
 * The functions are those of spx_logic.h -- the same source the host plan runs -- driven by these kernels:
 *
 *   recode_kernel        SEQ (BAM nt16, high nibble first) -> 0..4 codes the DP kernels read, coalesced
 *   aln_count_kernel     one thread per alignment: size of its op table, bounds of its mismatch / block lists
 *   aln_build_kernel     one thread per alignment: op table, aligned extents, confident blocks, mismatch list
 *   group_merge / aln_filter / group_blocks / aln_count_plan / group_sum kernels
 *                        marker columns, consensus windows, work-list sizes: the passes over one group alternate with
 *                        the passes over one alignment (the walks over ops and markers: most of the work)
 *   aln_emit / group_finish / problem_constants kernels
 *                        DP problems (with their HMM constants), wanted rows, marker table
 *   scan kernels         exclusive prefix sums that turn the counts into offsets (tiles of 1024 through LDS)
 *   order kernels        launch orders of the band classes: radix sort by (class, band, length) + padding to whole waves
 *
 * This is integer, pointer-chasing, latency-bound work: one lane walks one alignment's ops.  It costs a few per cent of
 * the DP kernels' time and needs no host cores, which is what lets 8 GPUs share one host.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>

#include "spx_logic.h"
#include "spx_prep_dev.h"

using namespace spxl;

/* The item a lane of a walking kernel takes.  Ordinary waves keep list order (neighbours share cache lines); an item that was EXTRACTED
 * (spx_prep_args: flag = its tier) idles in its ordinary lane and has a wave of its own in front of the ordinary waves.
 * item_of: kernels whose walk is one lane's -- the wave's lane 0 walks, tier 2 items only.
 * part_of: kernels whose walk can be shared -- every lane of the wave gets the item, with its lane as the share (part of nparts), tier >= 1. */
__device__ __forceinline__ int item_of(int i, int n, int n_heavy, const int32_t *__restrict__ heavy, const uint8_t *__restrict__ flag, int tier = 2)
{
    if (!heavy) return i < n ? i : -1;
    const int w = i >> 6, l = i & 63;
    if (w < n_heavy) { /* (an entry of the list that is not heavy enough was left in its ordinary lane) */
        if (l) return -1;
        const int it = heavy[w];
        return flag[it] >= tier ? it : -1;
    }
    const int j = i - (n_heavy << 6);
    if (j >= n) return -1;
    return flag[j] >= tier ? -1 : j;
}
__device__ __forceinline__ int part_of(int i, int n, int n_share, const int32_t *__restrict__ heavy, const uint8_t *__restrict__ flag, int &part, int &nparts)
{
    part = 0; nparts = 1;
    if (!heavy) return i < n ? i : -1;
    const int w = i >> 6;
    if (w < n_share) {
        const int it = heavy[w];
        if (flag[it] < 1) return -1;
        part = i & 63; nparts = 64;
        return it;
    }
    return item_of(i, n, n_share, heavy, flag, 1);
}
__device__ __forceinline__ int slot_of(const spx_prep_args &A, int i) { return item_of(i, A.n_slots, A.n_heavy_slots, A.slot_heavy, A.slot_flag); }
__device__ __forceinline__ int group_of(const spx_prep_args &A, int i) { return item_of(i, A.n_dgroups, A.n_heavy_groups, A.group_heavy, A.group_flag); }
__device__ __forceinline__ int slot_part(const spx_prep_args &A, int i, int &part, int &nparts)
{
    return part_of(i, A.n_slots, A.n_share_slots, A.slot_heavy, A.slot_flag, part, nparts);
}
__device__ __forceinline__ int group_part(const spx_prep_args &A, int i, int &part, int &nparts)
{
    return part_of(i, A.n_dgroups, A.n_share_groups, A.group_heavy, A.group_flag, part, nparts);
}
/* waves of 64 lanes a walking kernel needs */
static inline unsigned walk_waves(int n, int n_heavy, const void *heavy)
{
    return (unsigned)((heavy ? n_heavy : 0) + (n + 63) / 64);
}
/* ---------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void recode_kernel(const uint32_t *__restrict__ raw, uint32_t *__restrict__ code, int64_t n_words,
                                                     const Rec *__restrict__ recs, int32_t n_slots, AlnState *__restrict__ ast)
{
    /* grid-stride: a few thousand fat waves instead of a block per KB -- this kernel runs beside the DP kernels of the
     * previous batch, where every new workgroup waits for a wave slot */
    const uint64_t tbl = 0x4444444344424104ull; /* seq_nt16_int, one nibble per nt16 code */
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t x = raw[w];
        uint32_t out = 0, nmask = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t byte = (x >> (8 * b)) & 0xffu;
            const uint32_t hi = (uint32_t)(tbl >> (4 * (byte >> 4))) & 0xfu, lo = (uint32_t)(tbl >> (4 * (byte & 0xfu))) & 0xfu;
            out |= (hi | (lo << 4)) << (8 * b);
            if (hi > 3) nmask |= 1u << (2 * b);
            if (lo > 3) nmask |= 1u << (2 * b + 1);
        }
        code[w] = out;
        if (nmask) { /* rare: find the alignment this word belongs to and flag it if the base lies inside its SEQ */
            const int64_t byte0 = w * 4;
            int lo = 0, hi = n_slots - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (recs[mid].seq_off <= byte0) lo = mid; else hi = mid - 1;
            }
            const int64_t nib0 = (byte0 - recs[lo].seq_off) * 2;
            for (int k = 0; k < 8; ++k)
                if (((nmask >> k) & 1) && nib0 + k < recs[lo].l_qseq) { ast[lo].has_n = 1; break; }
        }
    }
}

/* ---- staged records: packed transfer layout -> the pools the walks read.  One workgroup per alignment.
 * unpack: a record whose SEQ / QUAL crossed PCIe: packed -> final place (whole dwords for SEQ, bytes for QUAL);
 * alias:  a record whose SEQ / QUAL repeat those of another record of its group (verified on the host, spx_prep.cpp): rebuilt from that record's
 *         final bytes -- base i = base (shift + i) of the primary, or the complement of base (shift - i) (nt16: the
 *         nibble's bits reversed) with the qualities in reverse order.  Runs after unpack: a source is never aliased itself. */
__global__ __launch_bounds__(256) void seqqual_unpack_kernel(const Rec *__restrict__ recs, int32_t n_slots, const uint8_t *__restrict__ pk_seq,
                                                            const uint8_t *__restrict__ pk_qual, uint8_t *__restrict__ seq, uint8_t *__restrict__ qual)
{
    const int s = blockIdx.x;
    if (s >= n_slots) return;
    const Rec r = recs[s];
    if (r.alias_slot >= 0 || r.l_qseq <= 0) return;
    const int64_t lq = r.l_qseq, nsw = (((lq + 1) / 2) + 3) / 4;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(pk_seq + r.pk_seq_off);
    uint32_t *dst = reinterpret_cast<uint32_t *>(seq + r.seq_off);
    for (int64_t k = threadIdx.x; k < nsw; k += blockDim.x) dst[k] = src[k];
    const uint8_t *qs = pk_qual + r.pk_qual_off;
    uint8_t *qd = qual + r.qual_off;
    /* the packed qualities start on a dword; the final place does not: dword loads, byte stores */
    const int64_t nqw = (lq + 3) / 4;
    for (int64_t k = threadIdx.x; k < nqw; k += blockDim.x) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(qs)[k];
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (4 * k + b < lq) qd[4 * k + b] = (uint8_t)(v >> (8 * b));
    }
}
__global__ __launch_bounds__(256) void seqqual_alias_kernel(const Rec *__restrict__ recs, int32_t n_slots, uint8_t *__restrict__ seq,
                                                           uint8_t *__restrict__ qual)
{
    const int s = blockIdx.x;
    if (s >= n_slots) return;
    const Rec r = recs[s];
    if (r.alias_slot < 0 || r.l_qseq <= 0) return;
    const Rec p = recs[r.alias_slot];
    const int64_t lq = r.l_qseq, shift = r.alias_shift;
    const bool rev = r.alias_rev != 0;
    const uint8_t *ps = seq + p.seq_off, *pq = qual + p.qual_off;
    uint8_t *ds = seq + r.seq_off, *dq = qual + r.qual_off;
    for (int64_t i = threadIdx.x; i < lq; i += blockDim.x) dq[i] = pq[rev ? shift - i : shift + i];
    auto base_of = [&](int64_t j) -> uint32_t { return (ps[j >> 1] >> ((~j & 1) << 2)) & 0xfu; };
    auto comp = [](uint32_t c) -> uint32_t { return ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3); };
    const int64_t nb = (((lq + 1) / 2) + 3) & ~(int64_t)3; /* whole padded bytes: the pad stays zero */
    for (int64_t b = threadIdx.x; b < nb; b += blockDim.x) {
        uint32_t v = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t i = 2 * b + h;
            if (i < lq) {
                const uint32_t c = rev ? comp(base_of(shift - i)) : base_of(shift + i);
                v |= c << (h ? 0 : 4); /* BAM: the first base of a byte is the high nibble */
            }
        }
        ds[b] = (uint8_t)v;
    }
}
extern "C" hipError_t spx_stage_expand(const void *recs, int32_t n_slots, const uint8_t *pk_seq, const uint8_t *pk_qual, uint8_t *seq, uint8_t *qual,
                                       int has_alias, hipStream_t st)
{
    if (n_slots <= 0) return hipSuccess;
    hipLaunchKernelGGL(seqqual_unpack_kernel, dim3((unsigned)n_slots), dim3(256), 0, st, (const Rec *)recs, n_slots, pk_seq, pk_qual, seq, qual);
    if (has_alias) hipLaunchKernelGGL(seqqual_alias_kernel, dim3((unsigned)n_slots), dim3(256), 0, st, (const Rec *)recs, n_slots, seq, qual);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void aln_count_kernel(spx_prep_args A)
{
    const int s = slot_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (s < 0) return;
    AlnState st;
    memset(&st, 0, sizeof st); /* (first kernel of the phase: the state starts here; recode_kernel, which flags has_n, runs after it) */
    const Rec r = A.recs[s];
    st.err = build_ops<false>(r, A.P, A.par.min_q, A.par.indel_threshold, st, nullptr);
    if (st.err) { st.n_ops = 0; st.mm_cap = 0; st.conf_cap = 0; }
    A.ast[s] = st;
}

/* the fast path of phase 1: table sizes from the record lengths (aln_caps), the tags are parsed once (aln_build_kernel) */
__global__ __launch_bounds__(256) void aln_caps_kernel(spx_prep_args A)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= A.n_slots) return;
    AlnState st;
    memset(&st, 0, sizeof st); /* (first kernel of the phase: the state starts here; recode_kernel, which flags has_n, runs after it) */
    aln_caps(A.recs[s], st.n_ops, st.conf_cap, st.mm_cap);
    if (A.tight_caps) { st.n_ops = st.n_ops / 8 + 2; st.mm_cap = st.mm_cap / 8 + 1; } /* tests: forces the fallback */
    A.ast[s] = st;
}

__global__ __launch_bounds__(64) void aln_build_kernel(spx_prep_args A)
{
    const int s = slot_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (s < 0) return;
    AlnState st = A.ast[s];
    if (st.err) return;
    if (st.ops_off + st.n_ops > A.ops_cap || st.conf_off + st.conf_cap > A.conf_cap || st.mm_off + st.mm_cap > A.mm_cap) {
        /* a pool sized from the host's bounds is too small: the host grows it to the exact totals (slots_apply_kernel
         * has them) and runs this phase again; until then the alignment counts as failed, so that nothing walks its
         * unwritten op table */
        atomicOr(&A.tot->overflow, 4);
        st.err = SPX_ENOMEM;
        A.ast[s] = st;
        return;
    }
    const Rec r = A.recs[s];
    Op *ops = A.P.ops + st.ops_off;
    const int32_t cap = A.exact_counts ? 0 : st.n_ops; /* bounded tables: n_ops holds the bound until build_ops sets the count */
    st.err = build_ops<true>(r, A.P, A.par.min_q, A.par.indel_threshold, st, ops, cap);
    if (!st.err) st.err = finish_alignment(r, A.P, A.par.min_q, A.par.indel_threshold, st, ops, A.P.conf + st.conf_off, A.P.mm + st.mm_off);
    if (st.err == SPX_ENOMEM && !A.exact_counts) atomicOr(&A.tot->overflow, 8); /* a bound did not hold: count exactly */
    A.ast[s] = st;
}

__global__ __launch_bounds__(64) void group_arena_kernel(spx_prep_args A)
{
    const int k = group_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (k < 0) return;
    const int s0 = A.slot0[k];
    GroupView G = {A.slot0[k + 1] - s0, A.recs + s0, A.ast + s0};
    A.ga_bytes[k] = group_arena_layout(G, A.par.all_rows != 0, A.slack).bytes;
}

/* ---- the passes of spx_logic.h: G* one read group per thread, A* one alignment per thread ---- */
struct GroupCtx {
    GroupView G;
    GroupScratch S;
    bool ok;
};
__device__ __forceinline__ GroupCtx group_ctx(const spx_prep_args &A, int k)
{
    GroupCtx c;
    const int s0 = A.slot0[k];
    c.G.n = A.slot0[k + 1] - s0;
    c.G.rec = A.recs + s0;
    c.G.st = A.ast + s0;
    const GroupArena ga = group_arena_layout(c.G, A.par.all_rows != 0, A.slack);
    c.ok = A.ga_off[k] + ga.bytes <= A.arena_cap;
    c.S = group_scratch(ga, A.arena + A.ga_off[k]);
    return c;
}

__global__ __launch_bounds__(64) void group_merge_kernel(spx_prep_args A)
{
    const int k = group_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (k < 0) return;
    GroupCtx c = group_ctx(A, k);
    GroupCount gc;
    if (!c.ok) {
        atomicOr(&A.tot->overflow, 1);
        count_clear(gc);
        gc.err = SPX_ENOMEM;
        A.gc[k] = gc;
        return;
    }
    group_pass_merge(c.G, A.P, A.rv, c.S, gc);
    A.gc[k] = gc;
}

__global__ __launch_bounds__(64) void aln_filter_kernel(spx_prep_args A)
{
    /* a heavy alignment's wave deals the marker columns to its 64 lanes in contiguous shares (round 5: the walk over the 45 000 columns of a
     * 100 kb read was one lane's); an ordinary lane takes all the columns of its alignment */
    int part = 0, nparts = 1;
    const int s = slot_part(A, blockIdx.x * blockDim.x + threadIdx.x, part, nparts);
    if (s < 0) return;
    const int k = A.recs[s].grp;
    const GroupCount gc = A.gc[k];
    if (gc.err || gc.n_cols == 0) return;
    GroupCtx c = group_ctx(A, k);
    aln_pass_filter(c.G, s - A.slot0[k], A.P, c.S, gc, part, nparts);
}

__global__ __launch_bounds__(64) void aln_compact_kernel(spx_prep_args A)
{
    const int s = slot_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (s < 0) return;
    const int k = A.recs[s].grp;
    const GroupCount gc = A.gc[k];
    if (gc.err || gc.n_cols == 0) return;
    GroupCtx c = group_ctx(A, k);
    aln_pass_compact(c.G, s - A.slot0[k], c.S, gc);
}

/* consensus windows: rounds run per group, the projections of the windows onto the alignments per alignment */
__global__ __launch_bounds__(64) void group_blocks_kernel(spx_prep_args A)
{
    const int k = group_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (k < 0) return;
    GroupCount gc = A.gc[k];
    if (gc.err || gc.n_cols == 0) return;
    GroupCtx c = group_ctx(A, k);
    group_pass_blocks_begin(c.G, A.P, A.par, c.S, gc);
    A.gc[k] = gc;
}
__global__ __launch_bounds__(64) void aln_project_kernel(spx_prep_args A)
{
    const int s = slot_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (s < 0) return;
    const int k = A.recs[s].grp;
    const GroupCount gc = A.gc[k];
    if (gc.err || gc.n_cols == 0) return;
    GroupCtx c = group_ctx(A, k);
    const BlocksState B = *c.S.bstate;
    blocks_project(c.G, s - A.slot0[k], A.P, A.par, c.S, B);
}
__global__ __launch_bounds__(64) void group_resume_kernel(spx_prep_args A)
{
    const int k = group_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (k < 0) return;
    const GroupCount gc = A.gc[k];
    if (gc.err || gc.n_cols == 0) return;
    GroupCtx c = group_ctx(A, k);
    if (c.S.bstate->phase != 1) return;
    BlocksState B = *c.S.bstate;
    blocks_after_projection(c.G, A.P, A.par, c.S, B);
    *c.S.bstate = B;
}
__global__ __launch_bounds__(64) void group_blocks_end_kernel(spx_prep_args A)
{
    const int k = group_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (k < 0) return;
    GroupCount gc = A.gc[k];
    if (gc.err || gc.n_cols == 0) return;
    GroupCtx c = group_ctx(A, k);
    group_pass_blocks_end(c.G, A.P, A.par, c.S, gc);
    if (gc.err == SPX_ENOMEM) atomicOr(&A.tot->overflow, 2); /* an interval list outgrew its estimate: repeat with more slack */
    A.gc[k] = gc;
}

/* wave-wide helpers of the shared walks (all 64 lanes of a heavy alignment's wave call them) */
__device__ __forceinline__ int64_t wave_sum64(int64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int64_t wave_exscan64(int64_t v)
{
    const int lane = threadIdx.x & 63;
    int64_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    return x - v;
}

__global__ __launch_bounds__(64) void aln_count_plan_kernel(spx_prep_args A)
{
    int part = 0, nparts = 1;
    const int s = slot_part(A, blockIdx.x * blockDim.x + threadIdx.x, part, nparts);
    if (s < 0) return;
    const int k = A.recs[s].grp;
    const GroupCount gc = A.gc[k];
    GroupCount ac;
    if (gc.err || !gc.scored) { count_clear(ac); if (part == 0) A.ac[s] = ac; return; }
    GroupCtx c = group_ctx(A, k);
    const int ai = s - A.slot0[k];
    if (nparts > 1 && plan_can_split(A.par, c.G.st[ai])) {
        /* a heavy alignment: the 64 lanes of its wave count contiguous shares of its blocks (spx_logic.h plan_baq_range); the shares' offsets go
         * to heavy_plan for the emitting pass, their counts are added up -- or the error of the FIRST share that has one is taken */
        PlanBase add;
        aln_pass_count_part(c.G, ai, A.P, A.rv, A.par, c.S, gc, ac, add, part, nparts);
        A.heavy_plan[(size_t)blockIdx.x * 64 + part] = add;
        const unsigned long long em = __ballot(ac.err != 0);
        GroupCount tot;
        count_clear(tot);
        if (em) tot.err = __shfl(ac.err, __ffsll((long long)em) - 1);
        else {
            tot.n_prob = (int32_t)wave_sum64(ac.n_prob); tot.n_rows = (int32_t)wave_sum64(ac.n_rows); tot.n_qe = (int32_t)wave_sum64(ac.n_qe);
            tot.cells = wave_sum64(ac.cells); tot.s_need = wave_sum64(ac.s_need); tot.f_need = wave_sum64(ac.f_need);
            for (int q = 0; q < SPX_N_CLASSES; ++q) { tot.cls_prob[q] = (int32_t)wave_sum64(ac.cls_prob[q]); tot.cls_cells[q] = wave_sum64(ac.cls_cells[q]); }
        }
        if (part == 0) {
            if (tot.err == SPX_ENOMEM) atomicOr(&A.tot->overflow, 2);
            A.ac[s] = tot;
        }
        return;
    }
    if (part != 0) return;
    aln_pass_count(c.G, ai, A.P, A.rv, A.par, c.S, gc, ac);
    if (ac.err == SPX_ENOMEM) atomicOr(&A.tot->overflow, 2);
    A.ac[s] = ac;
}
__global__ __launch_bounds__(64) void group_sum_kernel(spx_prep_args A)
{
    const int k = group_of(A, blockIdx.x * blockDim.x + threadIdx.x);
    if (k < 0) return;
    GroupCount gc = A.gc[k];
    GroupView G;
    const int s0 = A.slot0[k];
    G.n = A.slot0[k + 1] - s0;
    G.rec = A.recs + s0;
    G.st = A.ast + s0;
    group_pass_sum(G, gc, A.ac + s0);
    A.gc[k] = gc;
    for (int c = 0; c < SPX_N_CLASSES; ++c)
        if (gc.cls_prob[c]) {
            atomicAdd((unsigned long long *)&A.tot->cls_prob[c], (unsigned long long)gc.cls_prob[c]);
            atomicAdd((unsigned long long *)&A.tot->cls_cells[c], (unsigned long long)gc.cls_cells[c]);
        }
}

__global__ __launch_bounds__(64) void aln_emit_kernel(spx_prep_args A, spx_emit_args E)
{
    int part = 0, nparts = 1;
    const int s = slot_part(A, blockIdx.x * blockDim.x + threadIdx.x, part, nparts);
    if (s < 0) return;
    const int k = A.recs[s].grp;
    const GroupCount gc = A.gc[k];
    if (gc.err || !gc.scored) return;
    GroupCtx c = group_ctx(A, k);
    const int ai = s - A.slot0[k];
    if (nparts > 1 && plan_can_split(A.par, c.G.st[ai])) {
        /* (the same shares as in the counting pass; a share starts where the shares in front of it end) */
        const PlanBase add = A.heavy_plan[(size_t)blockIdx.x * 64 + part];
        PlanBase at = E.base[s];
        at.prob += wave_exscan64(add.prob); at.row += wave_exscan64(add.row); at.qe += wave_exscan64(add.qe);
        at.s_off += wave_exscan64(add.s_off); at.f_off += wave_exscan64(add.f_off);
        aln_pass_emit_part(c.G, ai, A.P, A.rv, A.par, c.S, gc, at, E.out, part, nparts);
        return;
    }
    if (part != 0) return;
    aln_pass_emit(c.G, ai, A.P, A.rv, A.par, c.S, gc, E.base[s], E.out);
}

__global__ __launch_bounds__(256) void rows_unpack_kernel(const RowRec *__restrict__ rr, int64_t n_rows, int32_t *__restrict__ rows,
                                                          int32_t *__restrict__ expect, int32_t *__restrict__ prob, uint8_t *__restrict__ rawq)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const RowRec r = rr[i];
    rows[i] = r.row; expect[i] = r.expect; prob[i] = r.prob; rawq[i] = (uint8_t)r.rawq;
}

__global__ __launch_bounds__(64) void group_finish_kernel(spx_prep_args A, spx_emit_args E)
{
    int part = 0, nparts = 1;
    const int k = group_part(A, blockIdx.x * blockDim.x + threadIdx.x, part, nparts);
    if (k < 0) return;
    const GroupCount gc = A.gc[k];
    GroupCtx c = group_ctx(A, k);
    const int n = c.G.n;
    const int64_t mk0 = E.mk_base[k];
    group_pass_markers(c.G, c.S, gc, E.markers + mk0, E.mk_ref_pos + mk0, part, nparts); /* (a heavy group: the cells dealt to the wave's 64 lanes) */
    if (part != 0) return;
    const bool ok = gc.err == 0;
    E.mk_first[k] = (int32_t)mk0;
    if (k == A.n_dgroups - 1) E.mk_first[k + 1] = (int32_t)(mk0 + (ok ? (int64_t)gc.n_cols * n : 0));
    E.n_aln[k] = ok ? (uint8_t)n : 0; /* a group with an error takes no part in scoring; its code travels in the info record */
    uint16_t sec = 0;
    for (int i = 0; i < 10; ++i) {
        const bool in = i < n;
        if (in && (c.G.rec[i].flag & SPX_FSECONDARY)) sec |= (uint16_t)(1u << i);
        E.rfe[k * 10 + i] = in ? c.G.st[i].rfe : 0;
        E.rfs[k * 10 + i] = in ? c.G.st[i].rfs : 0;
        E.atid[k * 10 + i] = in ? c.G.rec[i].tid : -1;
    }
    E.sec_mask[k] = sec;
    spx_group_info gi;
    gi.err = gc.err; gi.n_aln = n; gi.n_prob = ok ? gc.n_prob : 0; gi.n_mk = ok ? gc.n_cols * n : 0; gi.cells = ok ? gc.cells : 0;
    E.info[k] = gi;
}

__global__ __launch_bounds__(256) void problem_constants_kernel(Params par, int32_t n_prob, const int32_t *__restrict__ L,
                                                                const int32_t *__restrict__ R, const uint8_t *__restrict__ has_n,
                                                                double *__restrict__ hmm)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_prob) return;
    double h[SPX_H_N];
    problem_constants(par, L[p], R[p], has_n[p], h);
    double2 *dst = reinterpret_cast<double2 *>(hmm + (int64_t)p * SPX_H_N);
#pragma unroll
    for (int k = 0; k < SPX_H_N / 2; ++k) dst[k] = make_double2(h[2 * k], h[2 * k + 1]);
}

/* ---------------------------------------------------------------------- */
/* exclusive prefix sums by ONE workgroup of 1024 lanes: the arrays have 10^4 .. 10^5 entries (alignments, groups),
 * a single pass over them through LDS costs microseconds and needs no temporary storage protocol */
__device__ __forceinline__ int64_t block_exscan(int64_t v, int64_t *lds, int64_t &total)
{
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t a = t >= o ? lds[t - o] : 0;
        __syncthreads();
        lds[t] += a;
        __syncthreads();
    }
    const int64_t incl = lds[t];
    total = lds[1023];
    __syncthreads();
    return incl - v;
}

/* Multi-block exclusive scans of up to five int64 columns v[col][n] at once (n = alignments or groups of a batch,
 * 10^4 .. 10^6): scan_local scans every 1024-element tile in place and leaves the tile totals, scan_tops scans those
 * (one workgroup; n <= 2^20), and the consumers add tile offset + local value themselves. */
#define SPX_SCAN_TILE 1024
__global__ __launch_bounds__(1024) void scan_local_kernel(int64_t *__restrict__ v, int64_t n, int ncol, int64_t stride,
                                                          int64_t *__restrict__ tile_tot)
{
    __shared__ int64_t lds[1024];
    const int64_t i = (int64_t)blockIdx.x * SPX_SCAN_TILE + threadIdx.x;
    for (int c = 0; c < ncol; ++c) {
        const int64_t x = i < n ? v[c * stride + i] : 0;
        int64_t tot;
        const int64_t e = block_exscan(x, lds, tot);
        if (i < n) v[c * stride + i] = e;
        if (threadIdx.x == 0) tile_tot[c * SPX_SCAN_TILE + blockIdx.x] = tot;
    }
}
__global__ __launch_bounds__(1024) void scan_tops_kernel(int64_t *__restrict__ tile_tot, int n_tiles, int ncol, int64_t *__restrict__ grand)
{
    __shared__ int64_t lds[1024];
    for (int c = 0; c < ncol; ++c) {
        const int64_t x = (int)threadIdx.x < n_tiles ? tile_tot[c * SPX_SCAN_TILE + threadIdx.x] : 0;
        int64_t tot;
        const int64_t e = block_exscan(x, lds, tot);
        if ((int)threadIdx.x < n_tiles) tile_tot[c * SPX_SCAN_TILE + threadIdx.x] = e;
        if (threadIdx.x == 0) grand[c] = tot;
    }
}
__device__ __forceinline__ int64_t scanned(const spx_prep_args &A, int col, int64_t i)
{
    return A.scan_v[col * A.scan_stride + i] + A.scan_tile[col * SPX_SCAN_TILE + i / SPX_SCAN_TILE];
}
static hipError_t run_scan(const spx_prep_args *A, int64_t n, int ncol, hipStream_t st)
{
    const int tiles = (int)((n + SPX_SCAN_TILE - 1) / SPX_SCAN_TILE);
    if (tiles > SPX_SCAN_TILE) return hipErrorInvalidValue; /* > 2^20 entries: stage fewer groups at a time */
    hipLaunchKernelGGL(scan_local_kernel, dim3(tiles), dim3(1024), 0, st, A->scan_v, n, ncol, A->scan_stride, A->scan_tile);
    hipLaunchKernelGGL(scan_tops_kernel, dim3(1), dim3(1024), 0, st, A->scan_tile, tiles, ncol, A->scan_grand);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void slots_extract_kernel(spx_prep_args A)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= A.n_slots) return;
    A.scan_v[0 * A.scan_stride + s] = A.ast[s].n_ops;
    A.scan_v[1 * A.scan_stride + s] = A.ast[s].conf_cap;
    A.scan_v[2 * A.scan_stride + s] = A.ast[s].mm_cap;
}
__global__ __launch_bounds__(256) void slots_apply_kernel(spx_prep_args A)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0) { A.tot->n_ops = A.scan_grand[0]; A.tot->n_conf = A.scan_grand[1]; A.tot->n_mm = A.scan_grand[2]; }
    if (s >= A.n_slots) return;
    A.ast[s].ops_off = scanned(A, 0, s);
    A.ast[s].conf_off = scanned(A, 1, s);
    A.ast[s].mm_off = scanned(A, 2, s);
}

__global__ __launch_bounds__(256) void arena_extract_kernel(spx_prep_args A)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < A.n_dgroups) A.scan_v[k] = A.ga_bytes[k];
}
__global__ __launch_bounds__(256) void arena_apply_kernel(spx_prep_args A)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) A.tot->arena_bytes = A.scan_grand[0];
    if (k < A.n_dgroups) A.ga_off[k] = scanned(A, 0, k);
}

__global__ __launch_bounds__(256) void plan_extract_slots_kernel(spx_prep_args A)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= A.n_slots) return;
    const GroupCount &ac = A.ac[q];
    A.scan_v[0 * A.scan_stride + q] = ac.n_prob;
    A.scan_v[1 * A.scan_stride + q] = ac.n_rows;
    A.scan_v[2 * A.scan_stride + q] = ac.n_qe;
    A.scan_v[3 * A.scan_stride + q] = ac.s_need;
    A.scan_v[4 * A.scan_stride + q] = ac.f_need;
}
__global__ __launch_bounds__(256) void plan_apply_slots_kernel(spx_prep_args A, PlanBase *__restrict__ base_out)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) {
        A.tot->n_prob = A.scan_grand[0]; A.tot->n_rows = A.scan_grand[1]; A.tot->n_qe = A.scan_grand[2];
        A.tot->s_tot = A.scan_grand[3]; A.tot->f_tot = A.scan_grand[4];
    }
    if (q >= A.n_slots) return;
    PlanBase pb;
    pb.prob = scanned(A, 0, q); pb.row = scanned(A, 1, q); pb.qe = scanned(A, 2, q); pb.s_off = scanned(A, 3, q); pb.f_off = scanned(A, 4, q);
    base_out[q] = pb;
}
__global__ __launch_bounds__(256) void plan_extract_groups_kernel(spx_prep_args A)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= A.n_dgroups) return;
    const GroupCount &gc = A.gc[k];
    const bool ok = gc.err == 0;
    A.scan_v[0 * A.scan_stride + k] = ok ? (int64_t)gc.n_cols * (A.slot0[k + 1] - A.slot0[k]) : 0;
    A.scan_v[1 * A.scan_stride + k] = ok ? gc.cells : 0;
    A.scan_v[2 * A.scan_stride + k] = ok ? 1 : 0;
}
__global__ __launch_bounds__(256) void plan_apply_groups_kernel(spx_prep_args A, int64_t *__restrict__ mk_base)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { A.tot->n_mk = A.scan_grand[0]; A.tot->cells = A.scan_grand[1]; A.tot->n_ok = A.scan_grand[2]; }
    if (k < A.n_dgroups) mk_base[k] = scanned(A, 0, k);
}

/* ---------------------------------------------------------------------- */
/* launch orders: per band class, problems by (band width ascending, length descending), every band width padded to
 * whole waves (a wave must hold problems of ONE width: its band geometry is wave-uniform) */
__global__ __launch_bounds__(256) void order_keys_kernel(spx_order_args O)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= O.n_prob) return;
    const int bw = O.bw[p], L = O.L[p], nr = O.n_rows[p];
    const int cls = band_class(2 * bw + 1);
    const int fr = (O.fwd_by_last_row && nr > 0) ? O.rows[O.row_off[p] + nr - 1] : L; /* rows the forward kernel walks */
    const uint32_t lf = (uint32_t)fr > 0xfffffu ? 0xfffffu : (uint32_t)fr;
    /* rows the backward kernel walks: L down to the first wanted row */
    const int br = nr > 0 ? L - O.rows[O.row_off[p]] + 1 : 0;
    const uint32_t lb = (uint32_t)br > 0xfffffu ? 0xfffffu : (uint32_t)br;
    int sl = 0;
    while (sl + 1 < O.n_slices && p >= O.slice_prob[sl + 1]) ++sl; /* (at most 32 slices; the bounds sit in the kernel arguments) */
    const uint64_t hi = ((uint64_t)sl << 34) | ((uint64_t)cls << 30) | ((uint64_t)bw << 20);
    O.key_f[p] = hi | (0xfffffu - lf);
    O.key_b[p] = hi | (0xfffffu - lb);
    O.val[p] = p;
}

/* DP slices: where the problems / rows / scratch of the dispatched groups [ng * k / K, ng * (k + 1) / K) start -- the
 * prefix sums of the work list (PlanBase of a group's first alignment), K + 1 entries, the last = the totals */
__global__ void slice_bounds_kernel(const int32_t *__restrict__ slot0, const spxl::PlanBase *__restrict__ base, int32_t ng, int32_t K, spxl::PlanBase tot,
                                    spxl::PlanBase *__restrict__ out)
{
    const int k = (int)threadIdx.x;
    if (k > K) return;
    if (k == K) { out[k] = tot; return; }
    const int32_t g = (int32_t)((int64_t)ng * k / K);
    out[k] = base[slot0[g]];
}

extern "C" hipError_t spx_prep_slice_bounds(const int32_t *slot0, const spxl::PlanBase *base, int32_t ng, int32_t K, const spxl::PlanBase *tot,
                                            spxl::PlanBase *out, hipStream_t st)
{
    hipLaunchKernelGGL(slice_bounds_kernel, dim3(1), dim3(64), 0, st, slot0, base, ng, K, *tot, out);
    return hipGetLastError();
}

/* bin of a key: (slice, class, band width) */
__device__ __forceinline__ uint32_t order_bin(uint64_t key)
{
    const uint32_t sl = (uint32_t)(key >> 34), cls = (uint32_t)(key >> 30) & 0xfu, bw = (uint32_t)(key >> 20) & 0x3ffu;
    return ((sl * SPX_N_CLASSES + cls) << 10) | bw;
}
__global__ __launch_bounds__(256) void order_bins_kernel(const uint64_t *__restrict__ keys, int32_t n, int32_t *__restrict__ bin_start,
                                                         int32_t *__restrict__ bin_end)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = order_bin(keys[i]);
    if (i == 0 || order_bin(keys[i - 1]) != b) bin_start[b] = i;
    if (i == n - 1 || order_bin(keys[i + 1]) != b) bin_end[b] = i + 1;
}

/* one workgroup per slice: padded size of every (class, band) bin and its offset inside the class segment */
__global__ __launch_bounds__(1024) void order_pad_kernel(const int32_t *__restrict__ bin_start, const int32_t *__restrict__ bin_end,
                                                         int32_t *__restrict__ pad_base, int bwd)
{
    __shared__ int64_t lds[1024];
    const int sl = blockIdx.x;
    for (int cls = 0; cls < SPX_N_CLASSES; ++cls) { /* the 1024 possible band widths of one class */
        const int b = ((sl * SPX_N_CLASSES + cls) << 10) + threadIdx.x;
        const int ppw = 64 / (bwd ? class_lanes_bwd(cls) : class_lanes(cls));
        const int cnt = bin_start[b] >= 0 ? bin_end[b] - bin_start[b] : 0;
        const int64_t padded = (cnt + ppw - 1) / ppw * ppw;
        int64_t tot;
        pad_base[b] = (int32_t)block_exscan(padded, lds, tot);
    }
}

__global__ __launch_bounds__(256) void order_scatter_kernel(const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals, int32_t n,
                                                            const int32_t *__restrict__ bin_start, const int32_t *__restrict__ pad_base,
                                                            const spx_order_segs *__restrict__ segs, int32_t *__restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    const uint32_t b = order_bin(key);
    const int sl = (int)(key >> 34), cls = (int)(key >> 30) & 0xf;
    const int64_t dst = segs[sl].off[cls] + pad_base[b] + (i - bin_start[b]);
    if (dst < segs[sl].off[cls] + segs[sl].cap[cls]) order[dst] = vals[i];
}

/* ---------------------------------------------------------------------- */
/* work estimate of an alignment: what its walks are proportional to -- the characters of its cs / MD tag (one op per token) and its CIGAR
 * operations; of a group: the sum over its alignments.  Sorted descending (hipcub radix sort of (estimate, index) pairs). */
__device__ __forceinline__ int32_t work_estimate(const Rec &r)
{
    const int32_t t = r.cs_len >= 0 ? r.cs_len : (r.md_len >= 0 ? r.md_len : 0);
    return t + 4 * r.n_cigar + (r.l_qseq >> 6);
}
__global__ __launch_bounds__(256) void heavy_keys_kernel(spx_prep_args A, int32_t *__restrict__ key_s, int32_t *__restrict__ val_s, int32_t *__restrict__ key_g,
                                                         int32_t *__restrict__ val_g, uint8_t *__restrict__ flag_s, uint8_t *__restrict__ flag_g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n_slots) { key_s[i] = work_estimate(A.recs[i]); val_s[i] = i; flag_s[i] = 0; }
    if (i < A.n_dgroups) {
        int64_t w = 0;
        for (int s = A.slot0[i]; s < A.slot0[i + 1]; ++s) w += work_estimate(A.recs[s]);
        key_g[i] = (int32_t)(w > 0x7fffffff ? 0x7fffffff : w);
        val_g[i] = i;
        flag_g[i] = 0;
    }
}
/* the first n_share entries of the sorted list are extracted (tier 1; the first n_heavy of them tier 2) if they are heavy in absolute terms
 * (min_work) AND against the list's own median (4 x: a list of alignments that are all long -- ONT reads of one length -- gains nothing from
 * moving some of them) */
__global__ __launch_bounds__(256) void heavy_mark_kernel(const int32_t *__restrict__ key_sorted, const int32_t *__restrict__ heavy, int32_t n, int32_t n_heavy,
                                                         int32_t n_share, int32_t min_work, uint8_t *__restrict__ flag)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t med4 = 4 * (int64_t)key_sorted[n / 2];
    if (w < n_share && key_sorted[w] >= min_work && key_sorted[w] >= med4) flag[heavy[w]] = w < n_heavy ? 2 : 1;
}

extern "C" size_t spx_prep_heavy_temp_bytes(int32_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr,
                                                   n > 0 ? n : 1, 0, 31);
    return bytes;
}

/* with n = max(n_slots, n_dgroups): keys 3n, vals 2n int32; slot_heavy [n_slots] / group_heavy [n_dgroups] (the whole sorted index lists: the first
 * n_heavy_* entries are used); flags one byte per item */
extern "C" hipError_t spx_prep_heavy(const spx_prep_args *A, int32_t *keys, int32_t *vals, void *temp, size_t temp_bytes, int32_t *slot_heavy, int32_t *group_heavy,
                                     uint8_t *slot_flag, uint8_t *group_flag, int32_t min_work, hipStream_t st)
{
    const int n = std::max(A->n_slots, A->n_dgroups);
    if (n <= 0) return hipSuccess;
    int32_t *key_s = keys, *key_g = keys + n, *key_out = keys + 2 * (size_t)n, *val_s = vals, *val_g = vals + n;
    hipLaunchKernelGGL(heavy_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *A, key_s, val_s, key_g, val_g, slot_flag, group_flag);
    size_t tb = temp_bytes;
    hipError_t e = hipSuccess;
    if (A->n_slots > 0 && A->n_share_slots > 0) {
        e = hipcub::DeviceRadixSort::SortPairsDescending(temp, tb, (const int32_t *)key_s, key_out, (const int32_t *)val_s, slot_heavy, A->n_slots, 0, 31, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(heavy_mark_kernel, dim3((A->n_share_slots + 255) / 256), dim3(256), 0, st, key_out, slot_heavy, A->n_slots, A->n_heavy_slots, A->n_share_slots, min_work, slot_flag);
    }
    tb = temp_bytes;
    if (A->n_dgroups > 0 && A->n_share_groups > 0) {
        e = hipcub::DeviceRadixSort::SortPairsDescending(temp, tb, (const int32_t *)key_g, key_out, (const int32_t *)val_g, group_heavy, A->n_dgroups, 0, 31, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(heavy_mark_kernel, dim3((A->n_share_groups + 255) / 256), dim3(256), 0, st, key_out, group_heavy, A->n_dgroups, A->n_heavy_groups, A->n_share_groups, 4 * min_work, group_flag); /* (a group: the sum over its alignments) */
    }
    return hipGetLastError();
}

/* ---------------------------------------------------------------------- */
extern "C" hipError_t spx_prep_phase1(const spx_prep_args *A, const uint32_t *raw_seq, int64_t seq_words, hipStream_t st)
{
    if (A->n_slots <= 0) return hipSuccess;
    const dim3 ga(walk_waves(A->n_slots, A->n_heavy_slots, A->slot_heavy));
    if (A->exact_counts) hipLaunchKernelGGL(aln_count_kernel, ga, dim3(64), 0, st, *A);
    else hipLaunchKernelGGL(aln_caps_kernel, dim3((A->n_slots + 255) / 256), dim3(256), 0, st, *A);
    if (seq_words > 0)
        hipLaunchKernelGGL(recode_kernel, dim3((unsigned)std::min<int64_t>((seq_words + 255) / 256, 2048)), dim3(256), 0, st, raw_seq,
                           (uint32_t *)(A->code4_w + A->P.code_lead_bytes), seq_words, A->recs, A->n_slots, A->ast);
    hipLaunchKernelGGL(slots_extract_kernel, dim3((A->n_slots + 255) / 256), dim3(256), 0, st, *A);
    { hipError_t e = run_scan(A, A->n_slots, 3, st); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(slots_apply_kernel, dim3((A->n_slots + 255) / 256), dim3(256), 0, st, *A);
    hipLaunchKernelGGL(aln_build_kernel, ga, dim3(64), 0, st, *A);
    return hipGetLastError();
}

extern "C" hipError_t spx_prep_phase2(const spx_prep_args *A, spxl::PlanBase *base_out, int64_t *mk_base, hipStream_t st)
{
    if (A->n_dgroups <= 0) return hipSuccess;
    const dim3 gg(walk_waves(A->n_dgroups, A->n_heavy_groups, A->group_heavy)), ga(walk_waves(A->n_slots, A->n_heavy_slots, A->slot_heavy)), b64(64);
    const dim3 gas(walk_waves(A->n_slots, A->n_share_slots, A->slot_heavy)); /* the kernels that share an extracted alignment among the lanes of its wave */
    hipLaunchKernelGGL(group_arena_kernel, gg, b64, 0, st, *A);
    hipLaunchKernelGGL(arena_extract_kernel, dim3((A->n_dgroups + 255) / 256), dim3(256), 0, st, *A);
    { hipError_t e = run_scan(A, A->n_dgroups, 1, st); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(arena_apply_kernel, dim3((A->n_dgroups + 255) / 256), dim3(256), 0, st, *A);
    hipLaunchKernelGGL(group_merge_kernel, gg, b64, 0, st, *A);
    hipLaunchKernelGGL(aln_filter_kernel, gas, b64, 0, st, *A);
    hipLaunchKernelGGL(aln_compact_kernel, ga, b64, 0, st, *A);
    hipLaunchKernelGGL(group_blocks_kernel, gg, b64, 0, st, *A);
    for (int round = 0; round < 3; ++round) { /* nearly every group needs one or two projection rounds */
        hipLaunchKernelGGL(aln_project_kernel, ga, b64, 0, st, *A);
        hipLaunchKernelGGL(group_resume_kernel, gg, b64, 0, st, *A);
    }
    hipLaunchKernelGGL(group_blocks_end_kernel, gg, b64, 0, st, *A);
    hipLaunchKernelGGL(aln_count_plan_kernel, gas, b64, 0, st, *A);
    hipLaunchKernelGGL(group_sum_kernel, gg, b64, 0, st, *A);
    hipLaunchKernelGGL(plan_extract_slots_kernel, dim3((A->n_slots + 255) / 256), dim3(256), 0, st, *A);
    { hipError_t e = run_scan(A, A->n_slots, 5, st); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(plan_apply_slots_kernel, dim3((A->n_slots + 255) / 256), dim3(256), 0, st, *A, base_out);
    hipLaunchKernelGGL(plan_extract_groups_kernel, dim3((A->n_dgroups + 255) / 256), dim3(256), 0, st, *A);
    { hipError_t e = run_scan(A, A->n_dgroups, 3, st); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(plan_apply_groups_kernel, dim3((A->n_dgroups + 255) / 256), dim3(256), 0, st, *A, mk_base);
    return hipGetLastError();
}

extern "C" hipError_t spx_prep_emit(const spx_prep_args *A, const spx_emit_args *E, hipStream_t st)
{
    if (A->n_dgroups <= 0) return hipSuccess;
    hipLaunchKernelGGL(aln_emit_kernel, dim3(walk_waves(A->n_slots, A->n_share_slots, A->slot_heavy)), dim3(64), 0, st, *A, *E);
    if (E->out.rr && E->n_rows > 0)
        hipLaunchKernelGGL(rows_unpack_kernel, dim3((unsigned)((E->n_rows + 255) / 256)), dim3(256), 0, st, E->out.rr, E->n_rows, E->out.rows,
                           E->out.row_expect, E->out.row_prob, E->out.row_rawq);
    hipLaunchKernelGGL(group_finish_kernel, dim3(walk_waves(A->n_dgroups, A->n_share_groups, A->group_heavy)), dim3(64), 0, st, *A, *E);
    if (E->n_prob > 0)
        hipLaunchKernelGGL(problem_constants_kernel, dim3((E->n_prob + 255) / 256), dim3(256), 0, st, A->par, E->n_prob, E->out.L, E->out.R,
                           E->out.has_n, E->hmm);
    return hipGetLastError();
}

extern "C" size_t spx_order_temp_bytes(int32_t n_prob)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const int32_t *)nullptr,
                                       (int32_t *)nullptr, n_prob > 0 ? n_prob : 1, 0, 40);
    return bytes;
}

/* fills order_f / order_b (both pre-set to -1 by the caller's memset) of EVERY slice of the list: one pair of sorts, the slice is the top
 * field of the key.  bin_start holds 2 x n_slices x SPX_N_CLASSES x 1024 entries pre-set to -1 by the caller (forward half, backward half);
 * bin_end / pad_base likewise (not pre-set) */
extern "C" hipError_t spx_prep_orders(const spx_order_args *O, hipStream_t st)
{
    const int32_t n = O->n_prob;
    if (n <= 0) return hipSuccess;
    const size_t nb = (size_t)O->n_slices * SPX_N_CLASSES * 1024;
    int end_bit = 34;
    while ((1 << (end_bit - 34)) < O->n_slices) ++end_bit;
    hipLaunchKernelGGL(order_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, *O);
    for (int pass = 0; pass < 2; ++pass) {
        size_t tb = O->temp_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(O->temp, tb, pass ? O->key_b : O->key_f, O->key_sorted, O->val, O->val_sorted, n, 0, end_bit, st);
        if (e != hipSuccess) return e;
        int32_t *bs = O->bin_start + pass * nb, *be = O->bin_end + pass * nb, *pb = O->pad_base + pass * nb;
        hipLaunchKernelGGL(order_bins_kernel, dim3((n + 255) / 256), dim3(256), 0, st, O->key_sorted, n, bs, be);
        hipLaunchKernelGGL(order_pad_kernel, dim3(O->n_slices), dim3(1024), 0, st, bs, be, pb, pass);
        hipLaunchKernelGGL(order_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, O->key_sorted, O->val_sorted, n, bs, pb,
                           pass ? O->segs_b : O->segs_f, pass ? O->order_b : O->order_f);
    }
    return hipGetLastError();
}
