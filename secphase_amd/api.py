"""ctypes binding of libspx.so (include/spx.h) -- plumbing for tests and bench.

The library is the product; this file only loads it and mirrors its structs.
There is no Python or CPU fallback: if the shared object is missing the import
of `lib()` raises, and without a gfx950 device `Context()` raises SpxError.
"""
import ctypes as C
import os
import subprocess

from .records import SpxBatch, SpxParams, SpxRef

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPX_LIB") or os.path.join(_DIR, "libspx.so")

OK, ENODEVICE, EHIP, EINVAL, ENOMEM, EUNSUPPORTED, ENOREF, ENOTAG = 0, -1, -2, -3, -4, -5, -6, -7


class SpxError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().spx_strerror(code).decode()
        detail = lib().spx_last_error().decode()
        super().__init__(f"{where}: {msg} ({code}) {detail}")


class ProbalnPar(C.Structure):
    _fields_ = [("d", C.c_float), ("e", C.c_float), ("bw", C.c_int)]


class GroupOut(C.Structure):
    _fields_ = [
        ("score", C.c_double * 10), ("rfe", C.c_int32 * 10),
        ("n_aln", C.c_int8), ("prim_idx", C.c_int8), ("max_idx", C.c_int8), ("pass_", C.c_int8),
        ("tie_mask", C.c_uint16), ("best_idx", C.c_int8), ("relabel", C.c_int8),
        ("n_problems", C.c_int32), ("n_markers", C.c_int32), ("dp_cells", C.c_int64),
    ]


class Decision(C.Structure):
    _fields_ = [("group", C.c_uint32), ("n_aln", C.c_int8), ("prim_idx", C.c_int8), ("max_idx", C.c_int8), ("pass_", C.c_uint8),
                ("tie_mask", C.c_uint16), ("reserved", C.c_uint16), ("absdiff", C.c_int32)]


class RelabelRec(C.Structure):
    _fields_ = [("group", C.c_uint32), ("n_aln", C.c_int8), ("prim_idx", C.c_int8), ("pad_", C.c_int8 * 2),
                ("score", C.c_double * 10), ("rfe", C.c_int32 * 10), ("pos", C.c_int32 * 10), ("tid", C.c_int32 * 10),
                ("flag", C.c_uint16 * 10), ("qname", C.c_char * 260)]


class Stats(C.Structure):
    _fields_ = [
        ("n_groups", C.c_int64), ("n_dispatched", C.c_int64), ("n_problems", C.c_int64), ("n_rows", C.c_int64),
        ("dp_cells", C.c_int64), ("n_markers", C.c_int64), ("bytes_h2d", C.c_int64), ("bytes_d2h", C.c_int64),
        ("problems_per_class", C.c_int64 * 16),
        ("prep_seconds", C.c_double), ("h2d_seconds", C.c_double), ("kernel_seconds", C.c_double),
        ("d2h_seconds", C.c_double), ("baq_kernel_ms", C.c_double), ("score_kernel_ms", C.c_double),
        ("main_fwd_ms", C.c_double), ("main_bwd_ms", C.c_double), ("main_class_cells", C.c_int64),
        ("main_class", C.c_int32), ("main_class_lanes", C.c_int32), ("main_class_slots", C.c_int32),
        ("n_launches_averaged", C.c_int32), ("dp_slices", C.c_int32), ("reserved_", C.c_int32),
        ("tier_fast_problems", C.c_int64), ("tier_rerun_certificate", C.c_int64), ("tier_rerun_model", C.c_int64),
        ("tier_rerun_range", C.c_int64), ("tier_rows_uncertified", C.c_int64),
        ("dp_critical_ms", C.c_double), ("tail_span_ms", C.c_double),
    ]


_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_f64p = C.POINTER(C.c_double)


class DbamOptions(C.Structure):
    _fields_ = [("threads", C.c_int32), ("max_groups", C.c_int32), ("ahead", C.c_int32), ("flags", C.c_int32),
                ("host_inflate_percent", C.c_int32), ("reserved", C.c_int32), ("segment_bytes", C.c_int64), ("carry_bytes", C.c_int64), ("start_voffset", C.c_int64), ("end_voffset", C.c_int64)]


class PlanView(C.Structure):
    _fields_ = [
        ("n_problems", C.c_int32), ("n_rows", C.c_int32), ("n_groups", C.c_int32), ("n_markers", C.c_int32),
        ("L", _i32p), ("R", _i32p), ("bw", _i32p), ("ref_tid", _i32p), ("ref_rfs", _i32p),
        ("qry_nib", _i64p), ("qry4", _u8p), ("hmm", _f64p),
        ("row_off", _i32p), ("n_rows_of", _i32p), ("rows", _i32p), ("row_expect", _i32p), ("row_rawq", _u8p),
        ("grp_index", _i32p), ("mk_first", _i32p), ("mk_row", _i32p),
        ("mk_qfix", _u8p), ("mk_is_match", _u8p), ("mk_aln", _u8p), ("mk_first_of_pos", _u8p),
        ("n_aln", _u8p), ("sec_mask", _u16p), ("rfe", _i32p), ("grp_error", _i32p),
        ("n_qedits", C.c_int32), ("pad_", C.c_int32), ("qe_rec", _i32p), ("qe_pos", _i32p), ("qe_len", _i32p),
        ("qe_row0", _i32p),
    ]


EXPORTS = [
    "spx_strerror", "spx_last_error", "spx_device_count", "spx_create", "spx_destroy", "spx_set_reference",
    "spx_group_is_dispatched", "spx_score_batch", "spx_prepare", "spx_prepare_many", "spx_launch", "spx_sync", "spx_trim", "spx_bedset_add_points", "spx_collect",
    "spx_work_stats", "spx_work_free", "spx_finalize", "spx_write_relabel_log", "spx_probaln_glocal",
    "spx_probaln_batch", "spx_pack_decisions", "spx_plan_create", "spx_plan_get", "spx_plan_free", "spx_host_tables",
    "spx_finalizer_create", "spx_finalizer_apply", "spx_finalizer_free",
    "spx_bedset_create", "spx_bedset_free", "spx_bedset_add", "spx_bedset_size", "spx_bedset_save",
    "spx_merge_blocks_count", "spx_relabel_blocks",
    "spx_io_last_error", "spx_bam_open", "spx_bam_n_targets", "spx_bam_target_name", "spx_bam_bind_reference",
    "spx_bam_next_batch", "spx_bam_close", "spx_bam_open_opts", "spx_bam_default_options", "spx_bam_release_batch",
    "spx_bam_index_build", "spx_bam_index_save", "spx_bam_index_load", "spx_fasta_load",
    "spx_count_draws", "spx_finalizer_skip", "spx_format_relabel_text", "spx_free_text",
    "spx_inflate_bgzf_device", "spx_inflate_core_host", "spx_crc32_core_host", "spx_stage_transfer_stats", "spx_effective_cpus", "spx_bam_drop_pages",
    "spx_bam_attach_device_inflate", "spx_bam_inflate_counts", "spx_inflater_create", "spx_inflater_run", "spx_inflater_free", "spx_fasta_ref", "spx_fasta_free",
    "spx_probaln_posteriors", "spx_apply_quals", "spx_sam_open", "spx_sam_write_group", "spx_sam_close",
    "spx_stage", "spx_prepare_staged", "spx_work_export", "spx_work_release",
    "spx_pipe_create", "spx_pipe_submit", "spx_pipe_next", "spx_pipe_pending", "spx_pipe_destroy",
    "spx_work_device_bytes", "spx_dbam_default_options", "spx_dbam_open", "spx_dbam_header", "spx_dbam_start", "spx_dbam_next", "spx_dbam_release", "spx_dbam_stats", "spx_dbam_close",
    "spx_set_terminal_guard", "spx_get_terminal_guard", "spx_set_dp_tiers", "spx_get_dp_tiers", "spx_last_tier_stats",
    "spx_relabel_table_load", "spx_relabel_table_size", "spx_relabel_table_get", "spx_relabel_table_find", "spx_relabel_table_free",
    "spx_correct_default_options", "spx_correct_bam",
    "spx_sam_write_group_of", "spx_decisions_from_results", "spx_relabel_candidates", "spx_finalizer_apply_decisions", "spx_write_relabel_records",
]

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_DIR, "csrc")])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                          "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.spx_strerror.restype = C.c_char_p
    L.spx_strerror.argtypes = [C.c_int]
    L.spx_last_error.restype = C.c_char_p
    L.spx_device_count.restype = C.c_int
    L.spx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.spx_destroy.argtypes = [vp]
    L.spx_destroy.restype = None
    L.spx_set_reference.argtypes = [vp, C.POINTER(SpxRef)]
    L.spx_group_is_dispatched.argtypes = [C.POINTER(SpxBatch), C.c_int32]
    L.spx_score_batch.argtypes = [vp, C.POINTER(SpxBatch), C.POINTER(SpxParams), C.POINTER(GroupOut), C.POINTER(Stats)]
    L.spx_prepare.argtypes = [vp, C.POINTER(SpxBatch), C.POINTER(SpxParams), C.c_int, C.POINTER(vp)]
    L.spx_prepare_many.argtypes = [vp, C.POINTER(C.POINTER(SpxBatch)), C.c_int32, C.POINTER(SpxParams), C.c_int,
                                   C.POINTER(vp)]
    if hasattr(L, "spx_stage"):  # (tools/ab_bench.py may load an older build through SPX_LIB)
        L.spx_stage.argtypes = [vp, C.POINTER(C.POINTER(SpxBatch)), C.c_int32, C.POINTER(SpxParams), C.c_int, C.POINTER(vp)]
        L.spx_prepare_staged.argtypes = [vp, vp]
        L.spx_work_export.argtypes = [vp, vp, C.POINTER(vp)]
        L.spx_work_release.argtypes = [vp, vp]
    if hasattr(L, "spx_decisions_from_results"):
        L.spx_decisions_from_results.argtypes = [C.POINTER(GroupOut), C.c_int32, C.c_int32, C.POINTER(Decision), C.c_int32]
        L.spx_relabel_candidates.argtypes = [C.POINTER(SpxBatch), C.c_int32, C.POINTER(GroupOut), C.POINTER(SpxParams),
                                             C.POINTER(RelabelRec), C.c_int32]
        L.spx_finalizer_apply_decisions.argtypes = [vp, C.POINTER(SpxParams), C.POINTER(Decision), C.c_int32,
                                                    C.POINTER(C.c_int8), C.POINTER(C.c_int8)]
        L.spx_write_relabel_records.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(SpxRef), C.POINTER(RelabelRec), C.c_int32,
                                                C.POINTER(C.c_int8)]
    if hasattr(L, "spx_pipe_create"):
        L.spx_pipe_create.argtypes = [vp, C.POINTER(SpxParams), C.c_int, C.c_int, C.POINTER(vp)]
        L.spx_pipe_submit.argtypes = [vp, C.POINTER(C.POINTER(SpxBatch)), C.c_int32, vp, C.c_int32, vp]
        L.spx_pipe_next.argtypes = [vp, C.POINTER(GroupOut), C.c_int32, C.POINTER(vp), C.POINTER(vp)]
        L.spx_pipe_pending.argtypes = [vp]
        L.spx_pipe_destroy.argtypes = [vp]
        L.spx_pipe_destroy.restype = None
    L.spx_launch.argtypes = [vp, vp]
    L.spx_sync.argtypes = [vp]
    if hasattr(L, "spx_trim"):
        L.spx_trim.argtypes = [vp]
    L.spx_pack_decisions.argtypes = [vp, vp, C.c_int32, vp, C.c_int64]
    L.spx_collect.argtypes = [vp, vp, C.POINTER(GroupOut)]
    L.spx_work_stats.argtypes = [vp, C.POINTER(Stats)]
    if hasattr(L, "spx_work_device_bytes"):
        L.spx_work_device_bytes.argtypes = [vp, C.POINTER(C.c_int32)]
        L.spx_work_device_bytes.restype = C.c_int64
    L.spx_work_free.argtypes = [vp, vp]
    L.spx_work_free.restype = None
    L.spx_finalize.argtypes = [C.POINTER(SpxParams), C.c_uint, C.POINTER(GroupOut), C.c_int32]
    L.spx_write_relabel_log.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(SpxBatch), C.POINTER(SpxRef),
                                        C.POINTER(GroupOut)]
    L.spx_probaln_glocal.argtypes = [_u8p, C.c_int, _u8p, C.c_int, _u8p, C.POINTER(ProbalnPar), C.POINTER(C.c_int), _u8p]
    L.spx_probaln_batch.argtypes = [vp, C.c_int32, _u8p, _i64p, _u8p, _i64p, _i32p, C.POINTER(ProbalnPar), _i32p, _u8p,
                                    _f64p]
    L.spx_probaln_posteriors.argtypes = [vp, C.c_int32, _u8p, _i64p, _u8p, _i64p, _i32p, C.POINTER(ProbalnPar), C.c_int32,
                                         _f64p, _f64p, _f64p]
    L.spx_plan_create.argtypes = [C.POINTER(SpxRef), C.POINTER(SpxBatch), C.POINTER(SpxParams), C.POINTER(vp)]
    L.spx_plan_get.argtypes = [vp, C.POINTER(PlanView)]
    L.spx_plan_free.argtypes = [vp]
    L.spx_plan_free.restype = None
    L.spx_host_tables.argtypes = [_f64p, _f64p, _f64p]
    L.spx_host_tables.restype = None
    L.spx_finalizer_create.argtypes = [C.c_uint, C.POINTER(vp)]
    L.spx_finalizer_apply.argtypes = [vp, C.POINTER(SpxParams), C.POINTER(GroupOut), C.c_int32]
    L.spx_count_draws.argtypes = [C.POINTER(GroupOut), C.c_int32]
    L.spx_count_draws.restype = C.c_int64
    L.spx_finalizer_skip.argtypes = [vp, C.c_int64]
    L.spx_format_relabel_text.argtypes = [C.POINTER(SpxBatch), C.POINTER(SpxRef), C.POINTER(GroupOut), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.spx_free_text.argtypes = [C.c_void_p]
    L.spx_free_text.restype = None
    L.spx_finalizer_free.argtypes = [vp]
    L.spx_finalizer_free.restype = None
    L.spx_bedset_create.argtypes = [C.POINTER(vp)]
    L.spx_bedset_free.argtypes = [vp]
    L.spx_bedset_free.restype = None
    L.spx_bedset_add.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
    L.spx_bedset_size.argtypes = [vp]
    L.spx_bedset_size.restype = C.c_int64
    L.spx_bedset_save.argtypes = [vp, C.c_char_p, C.c_int]
    L.spx_merge_blocks_count.argtypes = [C.c_int32] + [_i32p] * 6 + [C.c_int32]
    L.spx_relabel_blocks.argtypes = [vp, C.POINTER(SpxRef), C.POINTER(GroupOut), vp, vp]
    L.spx_io_last_error.restype = C.c_char_p
    L.spx_bam_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.spx_bam_n_targets.argtypes = [vp]
    L.spx_bam_target_name.argtypes = [vp, C.c_int32]
    L.spx_bam_target_name.restype = C.c_char_p
    L.spx_bam_bind_reference.argtypes = [vp, C.POINTER(SpxRef)]
    L.spx_bam_next_batch.argtypes = [vp, C.c_int32, C.POINTER(C.POINTER(SpxBatch))]
    L.spx_bam_close.argtypes = [vp]
    L.spx_bam_close.restype = None
    L.spx_bam_release_batch.argtypes = [vp, C.POINTER(SpxBatch)]
    L.spx_fasta_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.spx_fasta_ref.argtypes = [vp]
    L.spx_fasta_ref.restype = C.POINTER(SpxRef)
    L.spx_fasta_free.argtypes = [vp]
    L.spx_fasta_free.restype = None
    if hasattr(L, "spx_dbam_open"):
        L.spx_dbam_default_options.argtypes = [C.POINTER(DbamOptions)]
        L.spx_dbam_default_options.restype = None
        L.spx_dbam_open.argtypes = [C.c_char_p, C.POINTER(DbamOptions), C.POINTER(vp)]
        L.spx_dbam_header.argtypes = [vp]
        L.spx_dbam_header.restype = vp
        L.spx_dbam_start.argtypes = [vp, C.POINTER(vp), C.c_int32, C.POINTER(SpxParams)]
        L.spx_dbam_next.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.POINTER(SpxBatch))]
        L.spx_dbam_release.argtypes = [vp, C.POINTER(SpxBatch)]
        L.spx_dbam_stats.argtypes = [vp, _i64p, _i64p, _f64p]
        L.spx_dbam_stats.restype = None
        L.spx_dbam_close.argtypes = [vp]
        L.spx_dbam_close.restype = None
    L.spx_apply_quals.argtypes = [vp, vp, C.c_int32, C.POINTER(SpxBatch), _u8p]
    L.spx_sam_open.argtypes = [C.c_char_p, vp, C.POINTER(vp)]
    L.spx_sam_write_group.argtypes = [vp, vp, C.c_int32, _u8p]
    L.spx_sam_close.argtypes = [vp]
    L.spx_set_terminal_guard.argtypes = [C.c_int]
    L.spx_get_terminal_guard.restype = C.c_int
    _lib = L
    return L


GUARD_BAND, GUARD_ROW = 0, 1


def set_terminal_guard(reading):
    """reading of probaln.c's terminal guard (include/spx.h: SPX_GUARD_BAND default, SPX_GUARD_ROW); process-wide, read
    when a work list is prepared"""
    _chk(lib().spx_set_terminal_guard(int(reading)), "spx_set_terminal_guard")


def set_dp_tiers(on):
    """two-tier DP (certified fast kernels + exact re-run of uncertified problems) on / off for lists prepared from now on"""
    _chk(lib().spx_set_dp_tiers(int(on)), "spx_set_dp_tiers")


def last_tier_stats():
    """(fast-class problems, re-run by certificate, by model, by range, rows not certified) of the latest collect / probaln_batch"""
    v = (C.c_int64 * 5)()
    _chk(lib().spx_last_tier_stats(v), "spx_last_tier_stats")
    return tuple(int(x) for x in v)


def get_dp_tiers():
    return lib().spx_get_dp_tiers()


def get_terminal_guard():
    return lib().spx_get_terminal_guard()


def _chk(rc, where):
    if rc != 0:
        raise SpxError(rc, where)


class Context:
    """One context per (process, device): stream, HBM-resident reference, tables."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(lib().spx_create(device, C.byref(self.h)), "spx_create")

    def set_reference(self, ref):
        _chk(lib().spx_set_reference(self.h, ref), "spx_set_reference")

    def score_batch(self, batch, params, finalize_seed=1):
        n = batch.contents.n_groups
        out = (GroupOut * n)()
        st = Stats()
        _chk(lib().spx_score_batch(self.h, batch, C.byref(params), out, C.byref(st)), "spx_score_batch")
        if finalize_seed is not None:
            _chk(lib().spx_finalize(C.byref(params), finalize_seed, out, n), "spx_finalize")
        return out, st

    def prepare(self, batch, params, host_threads=0):
        """batch: one POINTER(SpxBatch) or a list of them (merged into one work list)"""
        return Work(self, batch, params, host_threads)

    def stage(self, batch, params, host_threads=0):
        """records into HBM only; Work.prepare_staged() then builds the work list on the device"""
        return Work(self, batch, params, host_threads, stage_only=True)

    def probaln_batch(self, refs, queries, set_q, pars):
        """refs/queries: lists of uint8 numpy arrays of 0..4 codes; returns (states, qs, kernel_ms)."""
        import numpy as np
        n = len(refs)
        ro = np.zeros(n + 1, np.int64)
        qo = np.zeros(n + 1, np.int64)
        ro[1:] = np.cumsum([len(r) for r in refs])
        qo[1:] = np.cumsum([len(q) for q in queries])
        rcat = np.ascontiguousarray(np.concatenate(refs).astype(np.uint8))
        qcat = np.ascontiguousarray(np.concatenate(queries).astype(np.uint8))
        sq = np.ascontiguousarray(np.asarray(set_q, np.int32))
        P = (ProbalnPar * n)(*[ProbalnPar(*p) for p in pars])
        state = np.zeros(int(qo[-1]), np.int32)
        q = np.zeros(int(qo[-1]), np.uint8)
        ms = C.c_double(0)
        _chk(lib().spx_probaln_batch(self.h, n, rcat.ctypes.data_as(_u8p), ro.ctypes.data_as(_i64p),
                                     qcat.ctypes.data_as(_u8p), qo.ctypes.data_as(_i64p), sq.ctypes.data_as(_i32p), P,
                                     state.ctypes.data_as(_i32p), q.ctypes.data_as(_u8p), C.byref(ms)),
             "spx_probaln_batch")
        return ([state[qo[i]:qo[i + 1]] for i in range(n)], [q[qo[i]:qo[i + 1]] for i in range(n)], ms.value)

    def probaln_posteriors(self, refs, queries, set_q, pars, which=0):
        """diagnostics: (scale[L+2], zM[L,R], zI[L,R]) of problem `which` of the batch -- see spx.h.
        A single problem may be given as (ref, query, set_q, par)."""
        import numpy as np
        if not isinstance(refs, (list, tuple)):
            refs, queries, set_q, pars = [refs], [queries], [set_q], [pars]
        n = len(refs)
        ro = np.zeros(n + 1, np.int64)
        qo = np.zeros(n + 1, np.int64)
        ro[1:] = np.cumsum([len(r) for r in refs])
        qo[1:] = np.cumsum([len(q) for q in queries])
        rcat = np.ascontiguousarray(np.concatenate(refs).astype(np.uint8))
        qcat = np.ascontiguousarray(np.concatenate(queries).astype(np.uint8))
        sq = np.ascontiguousarray(np.asarray(set_q, np.int32))
        P = (ProbalnPar * n)(*[ProbalnPar(*p) for p in pars])
        L_, R_ = len(queries[which]), len(refs[which])
        scale = np.zeros(L_ + 2)
        zM = np.zeros((L_, R_))
        zI = np.zeros((L_, R_))
        _chk(lib().spx_probaln_posteriors(self.h, n, rcat.ctypes.data_as(_u8p), ro.ctypes.data_as(_i64p),
                                          qcat.ctypes.data_as(_u8p), qo.ctypes.data_as(_i64p), sq.ctypes.data_as(_i32p), P,
                                          which, scale.ctypes.data_as(_f64p), zM.ctypes.data_as(_f64p),
                                          zI.ctypes.data_as(_f64p)), "spx_probaln_posteriors")
        return scale, zM, zI

    def close(self):
        if self.h:
            lib().spx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Work:
    def __init__(self, ctx, batch, params, host_threads, stage_only=False):
        self.ctx = ctx
        self.params = params
        self.h = C.c_void_p()
        if not isinstance(batch, (list, tuple)):
            batch = [batch]
        self.n = sum(b.contents.n_groups for b in batch)
        arr = (C.POINTER(SpxBatch) * len(batch))(*batch)
        if stage_only:
            _chk(lib().spx_stage(ctx.h, arr, len(batch), C.byref(params), host_threads, C.byref(self.h)), "spx_stage")
        else:
            _chk(lib().spx_prepare_many(ctx.h, arr, len(batch), C.byref(params), host_threads, C.byref(self.h)),
                 "spx_prepare_many")

    def prepare_staged(self):
        """(re)build the work list on the device from the staged records"""
        _chk(lib().spx_prepare_staged(self.ctx.h, self.h), "spx_prepare_staged")

    def release(self):
        """drop the prepared list, keep the staged records"""
        _chk(lib().spx_work_release(self.ctx.h, self.h), "spx_work_release")

    def export_plan(self):
        """diagnostics: the device-built work list as a Plan-like object (view, close)"""
        return ExportedPlan(self)

    def launch(self):
        _chk(lib().spx_launch(self.ctx.h, self.h), "spx_launch")

    def sync(self):
        _chk(lib().spx_sync(self.ctx.h), "spx_sync")

    def pack_decisions(self, group_base, device_ptr, capacity):
        n = lib().spx_pack_decisions(self.ctx.h, self.h, group_base, C.c_void_p(device_ptr), capacity)
        if n < 0:
            raise SpxError(n, "spx_pack_decisions")
        return n

    def collect(self, finalize_seed=1):
        out = (GroupOut * self.n)()
        _chk(lib().spx_collect(self.ctx.h, self.h, out), "spx_collect")
        if finalize_seed is not None:
            _chk(lib().spx_finalize(C.byref(self.params), finalize_seed, out, self.n), "spx_finalize")
        return out

    def apply_quals(self, batch, qual, batch_index=0):
        """qual: writable uint8 numpy copy of the batch's qual[] (edited in place); needs params.flags & 1"""
        _chk(lib().spx_apply_quals(self.ctx.h, self.h, batch_index, batch, qual.ctypes.data_as(_u8p)),
             "spx_apply_quals")
        return qual

    def device_bytes(self):
        """(device bytes of the prepared list, DP slices)"""
        k = C.c_int32(0)
        return int(lib().spx_work_device_bytes(self.h, C.byref(k))), k.value

    def stats(self):
        st = Stats()
        _chk(lib().spx_work_stats(self.h, C.byref(st)), "spx_work_stats")
        return st

    def free(self):
        if self.h:
            lib().spx_work_free(self.ctx.h, self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Pipe:
    """in-order scoring pipeline (spx_pipe_*): submit batches or staged works, take results in submission order"""

    def __init__(self, ctx, params, depth=3, host_threads=0):
        self.ctx = ctx
        self.params = params
        self.h = C.c_void_p()
        _chk(lib().spx_pipe_create(ctx.h, C.byref(params), depth, host_threads, C.byref(self.h)), "spx_pipe_create")
        self._keep = []

    def submit(self, batch=None, staged=None):
        if staged is not None:
            _chk(lib().spx_pipe_submit(self.h, None, 0, staged.h, staged.n, None), "spx_pipe_submit")
            self._keep.append((staged.n, None))
            return
        if not isinstance(batch, (list, tuple)):
            batch = [batch]
        arr = (C.POINTER(SpxBatch) * len(batch))(*batch)
        n = sum(b.contents.n_groups for b in batch)
        _chk(lib().spx_pipe_submit(self.h, arr, len(batch), None, 0, None), "spx_pipe_submit")
        self._keep.append((n, arr))

    def next(self, out=None):
        """results of the oldest submission: (GroupOut array, n)"""
        n, _ = self._keep.pop(0)
        if out is None:
            out = (GroupOut * max(n, 1))()
        rc = lib().spx_pipe_next(self.h, out, n, None, None)
        if rc < 0:
            raise SpxError(rc, "spx_pipe_next")
        return out, rc

    def pending(self):
        return lib().spx_pipe_pending(self.h)

    def close(self):
        if self.h:
            lib().spx_pipe_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StagedWork(Work):
    """a work list staged by the device input (spx_dbam_next): same handle, owned by the caller"""

    def __init__(self, ctx, handle, n, params):
        self.ctx = ctx
        self.params = params
        self.h = handle
        self.n = n


class DeviceBam:
    """device-resident BAM input (spx_dbam_*): staged work lists + name batches in file order"""

    def __init__(self, path, ctxs, params, ref, **opts):
        L = lib()
        o = DbamOptions()
        L.spx_dbam_default_options(C.byref(o))
        for k, v in opts.items():
            setattr(o, k, v)
        self.h = C.c_void_p()
        _chk(L.spx_dbam_open(os.fsencode(path), C.byref(o), C.byref(self.h)), "spx_dbam_open")
        self.ctxs = list(ctxs)
        self.params = params
        self.missing = L.spx_bam_bind_reference(L.spx_dbam_header(self.h), ref)
        arr = (C.c_void_p * len(self.ctxs))(*[c.h for c in self.ctxs])
        _chk(L.spx_dbam_start(self.h, arr, len(self.ctxs), C.byref(params)), "spx_dbam_start")

    def next(self):
        """(StagedWork, ctx index, POINTER(SpxBatch) names, n groups) or None at the end of the file"""
        w, k, nb = C.c_void_p(), C.c_int32(0), C.POINTER(SpxBatch)()
        n = lib().spx_dbam_next(self.h, C.byref(w), C.byref(k), C.byref(nb))
        if n < 0:
            raise SpxError(n, "spx_dbam_next")
        if n == 0:
            return None
        return StagedWork(self.ctxs[k.value], w, n, self.params), k.value, nb, n

    def release(self, names):
        lib().spx_dbam_release(self.h, names)

    def close(self):
        if self.h:
            lib().spx_dbam_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ExportedPlan:
    def __init__(self, work):
        self.h = C.c_void_p()
        _chk(lib().spx_work_export(work.ctx.h, work.h, C.byref(self.h)), "spx_work_export")
        self.view = PlanView()
        _chk(lib().spx_plan_get(self.h, C.byref(self.view)), "spx_plan_get")

    def close(self):
        if self.h:
            lib().spx_plan_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """Host-only work list (no device)."""

    def __init__(self, ref, batch, params):
        self.h = C.c_void_p()
        _chk(lib().spx_plan_create(ref, batch, C.byref(params), C.byref(self.h)), "spx_plan_create")
        self.view = PlanView()
        _chk(lib().spx_plan_get(self.h, C.byref(self.view)), "spx_plan_get")

    def close(self):
        if self.h:
            lib().spx_plan_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_relabel_log(path, batch, ref, out, mode="w"):
    _chk(lib().spx_write_relabel_log(path.encode(), mode.encode(), batch, ref, out), "spx_write_relabel_log")
