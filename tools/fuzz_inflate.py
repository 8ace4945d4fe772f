"""GPU fuzz of the BGZF inflate kernel against zlib: random payloads (literal runs, matches at every distance up to 32 KB,
overlapping matches, stored / fixed / dynamic blocks, all levels and strategies, sizes around the ring and flush boundaries),
byte-identical output and status 0 for every block.  Usage: python tools/fuzz_inflate.py [seed] [seconds] [host]
("host": the same decoder core compiled for the CPU, spx_inflate_core_host -- no GPU needed; sanitizer runs.)"""
import ctypes as C
import os
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from secphase_amd import api  # noqa: E402


def payload(rng):
    n = int(rng.choice([0, 1, 2, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4096, 65279, 65280])) if rng.random() < 0.3 else int(rng.integers(0, 65281))
    kind = rng.random()
    if kind < 0.15:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind < 0.3:
        return bytes(rng.integers(0, int(rng.integers(1, 8)), n, dtype=np.uint8))
    out = bytearray()
    alpha = int(rng.choice([2, 4, 16, 64, 256]))
    while len(out) < n:
        r = rng.random()
        if r < 0.35 or len(out) < 4:  # literals
            out += rng.integers(0, alpha, int(rng.integers(1, 200)), dtype=np.uint8).tobytes()
        elif r < 0.9:  # a match: any distance up to 32 KB, any length (overlapping when length > distance)
            dist = int(rng.integers(1, min(len(out), 32768) + 1)) if rng.random() < 0.7 else int(rng.choice([1, 2, 3, 4, 1023, 1024, 1025, 2047, 2048, 2049]))
            dist = max(1, min(dist, len(out)))
            ln = int(rng.integers(3, 259)) if rng.random() < 0.8 else int(rng.integers(259, 3000))
            for _ in range(ln):
                out.append(out[-dist])
        else:  # a long run
            out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 5000))
    return bytes(out[:n])


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120
    host = len(sys.argv) > 3 and sys.argv[3] == "host"
    rng = np.random.default_rng(seed)
    L = api.lib()
    if host:
        L.spx_inflate_core_host.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
        t0, nblocks, nbytes = time.time(), 0, 0
        while time.time() - t0 < seconds:
            data = payload(rng)
            level = int(rng.choice([0, 1, 3, 6, 9]))
            strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
            co = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.choice([1, 8, 9])), strat)
            comp = co.compress(data) + co.flush()
            out = C.create_string_buffer(max(len(data), 1))
            rc = L.spx_inflate_core_host(comp, len(comp), out, len(data))
            if rc != 0 or out.raw[:len(data)] != data:
                sys.exit(f"inflate fuzz (host): seed {seed}: rc {rc}, {len(data)} bytes, level {level}, strategy {strat}")
            # a damaged copy must end with an error or with different bytes, never with a crash
            if len(comp) > 8 and rng.random() < 0.3:
                bad = bytearray(comp); bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
                L.spx_inflate_core_host(bytes(bad), len(bad), out, len(data))
            nblocks += 1; nbytes += len(data)
        print(f"inflate fuzz (host core): seed {seed}, {nblocks} streams, {nbytes / 1e6:.0f} MB, no mismatch, {time.time() - t0:.0f} s")
        return
    L.spx_inflate_bgzf_device.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.c_int32, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.spx_inflate_bgzf_device.restype = C.c_int64
    ctx = api.Context(0)
    t0, nblocks, nbytes = time.time(), 0, 0
    while time.time() - t0 < seconds:
        payloads, blob, offs = [], b"", [0]
        for _ in range(300):
            data = payload(rng)
            level = int(rng.choice([0, 1, 3, 6, 9]))
            strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
            co = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.choice([1, 8, 9])), strat)
            comp = co.compress(data) + co.flush()
            if len(comp) + 26 > 65536:
                continue
            payloads.append(data)
            blob += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + \
                struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))
            offs.append(len(blob))
        n = len(payloads)
        cap = 65536 * n
        out = C.create_string_buffer(cap)
        st = (C.c_int32 * n)()
        ms = C.c_double()
        got = L.spx_inflate_bgzf_device(ctx.h, blob, (C.c_int64 * (n + 1))(*offs), n, out, cap, st, C.byref(ms))
        want = b"".join(payloads)
        bad = [k for k in range(n) if st[k] != 0]
        if got != len(want) or bad or out.raw[:got] != want:
            k = bad[0] if bad else next(i for i in range(n) if out.raw[sum(map(len, payloads[:i])):sum(map(len, payloads[:i + 1]))] != payloads[i])
            open(os.path.join(ROOT, "gpurun_out", f"fuzz_inflate_fail_{seed}.bin"), "wb").write(payloads[k])
            sys.exit(f"inflate fuzz: seed {seed}: block {k} (status {st[k]}, {len(payloads[k])} bytes) differs; payload saved")
        nblocks += n
        nbytes += len(want)
    print(f"inflate fuzz: seed {seed}, {nblocks} blocks, {nbytes / 1e6:.0f} MB, no mismatch, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
