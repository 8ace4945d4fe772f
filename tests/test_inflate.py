"""The DEFLATE core the gfx950 inflate kernel runs (secphase_amd/csrc/spx_inflate.h), compiled for the host, against zlib:
stored / fixed / dynamic blocks, every compression level, data shaped like BAM records, random and degenerate inputs,
damaged streams; the striped CRC-32 with the GF(2) combine against zlib.crc32."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from secphase_amd import api


def _lib():
    L = api.lib()
    L.spx_inflate_core_host.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
    L.spx_crc32_core_host.argtypes = [C.c_char_p, C.c_int64, C.c_int32]
    L.spx_crc32_core_host.restype = C.c_uint32
    return L


def _raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    return co.compress(data) + co.flush()


def _inflate(L, comp, n):
    out = C.create_string_buffer(max(n, 1))
    rc = L.spx_inflate_core_host(comp, len(comp), out, n)
    return rc, out.raw[:n]


def _cases():
    rng = np.random.default_rng(7)
    quals = np.clip(np.round(rng.normal(45, 12, 40000)), 2, 93).astype(np.uint8).tobytes()
    seq = rng.integers(0, 256, 20000, dtype=np.uint8).tobytes()
    bamish = (b"read_000123\0" + seq[:7500] + quals[:15000] + b"csZ:1200*ag:33+ac:17-t:5000\0") * 2 + seq[7500:9000]
    return {
        "empty": b"",
        "one byte": b"x",
        "zeros": bytes(65280),
        "run": b"ab" * 20000,
        "text": (b"the quick brown fox jumps over the lazy dog. " * 900)[:40000],
        "random": rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(),
        "quals": quals,
        "bam-like": bamish[:65280],
        "few symbols": bytes(rng.integers(0, 3, 50000, dtype=np.uint8)),
        "all bytes": bytes(range(256)) * 100,
    }


def test_core_equals_zlib(built):
    L = _lib()
    for name, data in _cases().items():
        for level in (0, 1, 3, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                comp = _raw(data, level, strategy)
                rc, got = _inflate(L, comp, len(data))
                assert rc == 0, (name, level, strategy, rc)
                assert got == data, (name, level, strategy)


def test_multi_block_streams_and_long_codes(built):
    """a stream of several DEFLATE blocks (Z_FULL_FLUSH between them: stored empty blocks in the middle), and a skewed
    alphabet whose rare symbols get codes longer than the 10-bit root table (the canonical walk)"""
    L = _lib()
    rng = np.random.default_rng(3)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(), b"A" * 3000, bytes(rng.integers(0, 4, 7000, dtype=np.uint8))]
    comp = b""
    for p in parts:
        comp += co.compress(p) + co.flush(zlib.Z_FULL_FLUSH)
    comp += co.flush()
    data = b"".join(parts)
    rc, got = _inflate(L, comp, len(data))
    assert rc == 0 and got == data
    # geometric symbol frequencies: code lengths up to 15
    p = 0.5 ** np.arange(1, 41)
    p = np.concatenate([p, np.full(216, (1 - p.sum()) / 216)])
    skew = rng.choice(256, size=60000, p=p / p.sum()).astype(np.uint8).tobytes()
    comp = _raw(skew, 9, zlib.Z_HUFFMAN_ONLY)
    rc, got = _inflate(L, comp, len(skew))
    assert rc == 0 and got == skew


def test_damaged_streams_are_rejected(built):
    L = _lib()
    data = _cases()["bam-like"]
    comp = bytearray(_raw(data))
    # wrong expected size
    assert _inflate(L, bytes(comp), len(data) - 1)[0] < 0
    assert _inflate(L, bytes(comp), len(data) + 1)[0] < 0
    # truncated input
    assert _inflate(L, bytes(comp[: len(comp) // 2]), len(data))[0] < 0
    # flipped bits: either an error or different bytes, never a crash
    rng = np.random.default_rng(11)
    for _ in range(200):
        c2 = bytearray(comp)
        c2[int(rng.integers(0, len(c2)))] ^= 1 << int(rng.integers(0, 8))
        rc, got = _inflate(L, bytes(c2), len(data))
        assert rc < 0 or got != data or bytes(c2) == bytes(comp)
    # reserved block type
    assert _inflate(L, bytes([0x07, 0, 0, 0]), 0)[0] < 0


def test_striped_crc_equals_zlib(built):
    L = _lib()
    rng = np.random.default_rng(5)
    for n in (0, 1, 63, 64, 65, 1000, 65280):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for pieces in (1, 2, 7, 64):
            assert L.spx_crc32_core_host(data, n, pieces) == (zlib.crc32(data) & 0xffffffff), (n, pieces)


# ---------------------------------------------------------------- the kernel itself
def _bgzf_file(payloads, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """BGZF blocks (one per payload) + their start offsets"""
    import struct
    blob, offs = b"", [0]
    for data in payloads:
        comp = _raw(data, level, strategy)
        hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25)
        blob += hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))
        offs.append(len(blob))
    return blob, offs


def _device_inflate(L, ctx, blob, offs):
    n = len(offs) - 1
    L.spx_inflate_bgzf_device.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.spx_inflate_bgzf_device.restype = C.c_int64
    cap = 65536 * max(n, 1)
    out = C.create_string_buffer(cap)
    st = (C.c_int32 * max(n, 1))()
    ms = C.c_double()
    got = L.spx_inflate_bgzf_device(ctx.h, blob, (C.c_int64 * (n + 1))(*offs), n, out, cap, st, C.byref(ms))
    return got, out.raw[:max(got, 0)], list(st)[:n], ms.value


@pytest.mark.gpu
def test_kernel_equals_zlib(built):
    L = _lib()
    ctx = api.Context(0)
    rng = np.random.default_rng(21)
    payloads = []
    for name, data in _cases().items():
        payloads.append(data[:65280])
    # block sizes around the ring / flush boundaries, and many small blocks
    for n in (1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 65279, 65280):
        payloads.append(rng.integers(0, 7, n, dtype=np.uint8).tobytes())
    payloads += [rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes() for _ in range(300)]
    for level in (1, 6, 9, 0):
        blob, offs = _bgzf_file(payloads, level)
        got, out, st, _ = _device_inflate(L, ctx, blob, offs)
        assert got == sum(len(p) for p in payloads), L.spx_last_error()
        assert all(s == 0 for s in st), [k for k, s in enumerate(st) if s][:5]
        assert out == b"".join(payloads)
    ctx.close()


@pytest.mark.gpu
def test_kernel_on_the_decode_kernels_special_paths(built):
    """what the decode / copy kernels treat specially: 1-bit literal codes (the literal register is emptied at every step of the
    reader), a code set of ONE code, fixed-code blocks, runs (matches that overlap themselves, distance 1..3), matches of 258 bytes
    30 KB back (copied by the whole wavefront), many short far matches (rounds of the copy kernel), codes longer than the root
    tables, blocks whose neighbours in the wavefront are empty or tiny"""
    L = _lib()
    ctx = api.Context(0)
    rng = np.random.default_rng(33)
    far = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    short_far = bytearray(rng.integers(0, 256, 8000, dtype=np.uint8).tobytes())
    while len(short_far) < 65000:
        at = int(rng.integers(0, len(short_far) - 4))
        short_far += short_far[at:at + int(rng.integers(3, 6))] + bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
    skew = np.minimum(rng.geometric(0.02, 60000), 255).astype(np.uint8).tobytes()  # code lengths 2..15
    sets = [
        (zlib.Z_HUFFMAN_ONLY, [bytes(rng.integers(0, 2, 60000, dtype=np.uint8)), bytes(65000), b"", b"a", bytes(rng.integers(0, 2, 7, dtype=np.uint8))]),
        (zlib.Z_FIXED, [far[:20000], b"abc" * 9000, b"", far[:5]]),
        (zlib.Z_RLE, [b"x" * 65000, b"xy" * 30000, b"xyz" * 20000, far[:100] + b"q" * 300 + far[:100]]),
        (zlib.Z_DEFAULT_STRATEGY, [far + far, bytes(short_far[:65000]), skew, far[:1000] * 60, b"", b"z"]),
    ]
    for strategy, payloads in sets:
        for level in (6, 9, 1):
            blob, offs = _bgzf_file(payloads, level, strategy)
            got, out, st, _ = _device_inflate(L, ctx, blob, offs)
            assert got == sum(len(p) for p in payloads), L.spx_last_error()
            assert all(s == 0 for s in st), (strategy, level, [k for k, s in enumerate(st) if s])
            assert out == b"".join(payloads), (strategy, level)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"SPX_INFLATE_TOK": "16"}, {"SPX_INFLATE_TOK": "0"}, {"SPX_INFLATE_TOK": "0", "SPX_INFLATE_LANES": "64"}],
                         ids=["decode kernel, 16 lanes per block", "round 4's first kernel", "round 3's kernel"])
def test_the_other_kernel_generations_still_agree_with_zlib(built, env):
    """the kernels kept for comparisons run the same payloads in a child process (the choice is read once per process)"""
    import subprocess
    import sys
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "test_kernel_equals_zlib or special_paths or damage_per_block"], env=dict(os.environ, **env), capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-1500:] + p.stderr[-500:]


@pytest.mark.gpu
def test_kernel_reports_damage_per_block(built):
    L = _lib()
    ctx = api.Context(0)
    data = _cases()["bam-like"]
    blob, offs = _bgzf_file([data, data, data])
    b = bytearray(blob)
    b[offs[1] + 18 + 200] ^= 0x10   # DEFLATE data of the second block
    b[offs[3] - 8] ^= 0x01          # stored CRC of the third block
    got, out, st, _ = _device_inflate(L, ctx, bytes(b), offs)
    assert got == 3 * len(data)
    assert st[0] == 0 and st[1] != 0 and st[2] == -4
    assert out[:len(data)] == data
    ctx.close()


@pytest.mark.gpu
def test_kernel_on_a_synthetic_bam(built, tmp_path):
    """the blocks of a BAM written by the bench-side writer (htslib block policy): byte-identical to zlib's output"""
    import gzip
    from common import small_genome
    from secphase_amd import synth
    L = _lib()
    ctx = api.Context(0)
    g = small_genome(synth.HIFI, read_len=15000, max_secondaries=2, n_paralogs=2)
    r = g.reads(0, 400)
    bam = str(tmp_path / "x.bam")
    synth.write_bam(bam, [r.batch], g.ref, threads=4)
    blob = open(bam, "rb").read()
    offs, at = [0], 0
    while at < len(blob):
        at += (blob[at + 16] | (blob[at + 17] << 8)) + 1
        offs.append(at)
    got, out, st, ms = _device_inflate(L, ctx, blob, offs)
    want = gzip.open(bam).read()
    assert got == len(want) and all(s == 0 for s in st)
    assert out == want
    ctx.close()
