"""ctypes wrapper of synth/libspxsynth.so (deterministic synthetic inputs)."""
import ctypes as C
import os
import subprocess

from .records import SpxBatch, SpxRef

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIFI, ONT, MIXED = 0, 1, 2


class SynthCfg(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_contigs", C.c_int32), ("contig_len", C.c_int32), ("n_paralogs", C.c_int32),
        ("platform", C.c_int32), ("read_len", C.c_int32), ("max_read_len", C.c_int32),
        ("min_secondaries", C.c_int32), ("max_secondaries", C.c_int32),
        ("softclip_frac", C.c_double), ("hardclip_frac", C.c_double),
        ("shuffle_records", C.c_int32), ("inverted_paralogs", C.c_int32),
        ("n_base_frac", C.c_double), ("snv_rate", C.c_double), ("indel_rate", C.c_double),
        ("paralog_snv_rate", C.c_double), ("tag_mode", C.c_int32), ("reserved", C.c_int32),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "synth")])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_ROOT, "synth", "libspxsynth.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.spx_synth_default_cfg.argtypes = [C.POINTER(SynthCfg), C.c_int]
        L.spx_synth_genome_create.restype = C.c_void_p
        L.spx_synth_genome_create.argtypes = [C.POINTER(SynthCfg)]
        L.spx_synth_genome_ref.restype = C.POINTER(SpxRef)
        L.spx_synth_genome_ref.argtypes = [C.c_void_p]
        L.spx_synth_genome_free.argtypes = [C.c_void_p]
        L.spx_synth_reads_create.restype = C.c_void_p
        L.spx_synth_reads_create.argtypes = [C.c_void_p, C.POINTER(SynthCfg), C.c_int64, C.c_int32]
        L.spx_synth_reads_batch.restype = C.POINTER(SpxBatch)
        L.spx_synth_reads_batch.argtypes = [C.c_void_p]
        L.spx_synth_reads_free.argtypes = [C.c_void_p]
        L.spx_synth_write_bam.restype = C.c_int64
        L.spx_synth_write_bam.argtypes = [C.c_char_p, C.POINTER(C.POINTER(SpxBatch)), C.c_int32, C.POINTER(SpxRef),
                                          C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]
        L.spx_synth_write_fasta.argtypes = [C.c_char_p, C.POINTER(SpxRef), C.c_int]
        _lib = L
    return _lib


def default_cfg(platform, **kw):
    cfg = SynthCfg()
    lib().spx_synth_default_cfg(C.byref(cfg), platform)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


class Genome:
    def __init__(self, cfg):
        self.cfg = cfg
        self.h = lib().spx_synth_genome_create(C.byref(cfg))
        self.ref = lib().spx_synth_genome_ref(self.h)

    def reads(self, first, n, cfg=None):
        return Reads(self, cfg or self.cfg, first, n)

    def close(self):
        if self.h:
            lib().spx_synth_genome_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Reads:
    def __init__(self, genome, cfg, first, n):
        self.genome = genome
        self.h = lib().spx_synth_reads_create(genome.h, C.byref(cfg), first, n)
        self.batch = lib().spx_synth_reads_batch(self.h)

    def close(self):
        if self.h:
            lib().spx_synth_reads_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def write_bam(path, batches, ref, contig_order=None, threads=8, level=6, extra_tags=True):
    """the record batches (ctypes pointers to spx_batch), in order, as one name-grouped BAM; returns the file size"""
    arr = (C.POINTER(SpxBatch) * len(batches))(*batches)
    order = None
    if contig_order is not None:
        order = (C.c_int32 * len(contig_order))(*contig_order)
    n = lib().spx_synth_write_bam(os.fsencode(path), arr, len(batches), ref, order, threads, level, 1 if extra_tags else 0)
    if n < 0:
        raise OSError(f"could not write {path}")
    return n


def write_fasta(path, ref, width=80):
    if lib().spx_synth_write_fasta(os.fsencode(path), ref, width) != 0:
        raise OSError(f"could not write {path}")
