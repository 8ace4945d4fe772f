"""ctypes mirrors of include/spx_records.h (the flat alignment-group format).

Plumbing only: lets Python tests / bench hand the same bytes to the C-ABI
library, the synthetic generator and the CPU oracle.
"""
import ctypes as C

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u16p = C.POINTER(C.c_uint16)
c_u32p = C.POINTER(C.c_uint32)
c_u8p = C.POINTER(C.c_uint8)

FUNMAP, FREVERSE, FSECONDARY, FSUPPLEMENTARY = 0x4, 0x10, 0x100, 0x800


class SpxBatch(C.Structure):
    _fields_ = [
        ("n_groups", C.c_int32), ("n_alns", C.c_int32),
        ("grp_first", c_i32p), ("qname_off", c_i64p), ("qnames", C.c_void_p),
        ("flag", c_u16p), ("tid", c_i32p), ("pos", c_i32p), ("l_qseq", c_i32p), ("n_cigar", c_i32p),
        ("cigar_off", c_i64p), ("seq_off", c_i64p), ("qual_off", c_i64p), ("cs_off", c_i64p),
        ("cigar", c_u32p), ("seq4", c_u8p), ("qual", c_u8p), ("cs", C.c_void_p),
        ("md_off", c_i64p), ("md", C.c_void_p),
    ]


class SpxRef(C.Structure):
    _fields_ = [
        ("n_contigs", C.c_int32), ("name_off", c_i64p), ("names", C.c_void_p),
        ("seq_off", c_i64p), ("bases", C.c_void_p),
    ]


class SpxParams(C.Structure):
    _fields_ = [
        ("baq_flag", C.c_int32), ("consensus", C.c_int32), ("indel_threshold", C.c_int32), ("min_q", C.c_int32),
        ("min_score", C.c_int32), ("set_q", C.c_int32), ("flank_margin", C.c_int32), ("flags", C.c_int32),
        ("prim_margin_score", C.c_double), ("prim_margin_random", C.c_double),
        ("conf_d", C.c_double), ("conf_e", C.c_double), ("conf_b", C.c_double),
    ]


def preset(name, bandwidth=None):
    """--hifi / --ont presets, /root/reference/programs/src/secphase.c:477-504."""
    p = SpxParams()
    p.baq_flag, p.consensus, p.min_q, p.min_score = 1, 1, 10, -10
    p.flank_margin, p.prim_margin_random, p.conf_e, p.conf_b = 500, 0.0, 0.1, 20.0
    if name == "hifi":
        p.indel_threshold, p.conf_d, p.set_q, p.prim_margin_score = 10, 1e-4, 40, 40.0
    elif name == "ont":
        p.indel_threshold, p.conf_d, p.set_q, p.prim_margin_score = 20, 1e-3, 20, 20.0
    else:
        raise ValueError(name)
    if bandwidth is not None:
        p.conf_b = float(bandwidth)
    return p
