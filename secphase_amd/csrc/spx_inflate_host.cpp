/*
 * spx_inflate_host.cpp -- the DEFLATE core of spx_inflate.h compiled for the host: diagnostics entry points that let the
 * CPU tests check the decoder the gfx950 kernel runs (same source) against zlib, without a GPU.  Not a product path:
 * the host reader inflates with libdeflate / zlib (spx_io.cpp), the device path with spx_inflate_kernels.hip.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spx.h"
#include "spx_inflate.h"

template <int LR, int DR>
static int run_core(const uint8_t *in, int64_t in_len, uint8_t *out, int64_t out_len)
{
    spxz::HostEnvT<LR, DR> env;
    env.in = in;
    env.in_len = (size_t)in_len;
    env.out = out;
    env.cap = (uint32_t)out_len;
    return spxz::inflate_stream(env, in_len * 8, (uint32_t)out_len);
}

extern "C" int spx_inflate_core_host(const uint8_t *in, int64_t in_len, uint8_t *out, int64_t out_len)
{
    if (!in || (!out && out_len > 0) || in_len < 0 || out_len < 0 || out_len > 0xffffffffll) return SPX_EINVAL;
    /* the root-table sizes the device kernels are built with: SPX_INFLATE_ROOT = 9 (9 / 8 bits, round 4), 10 (10 / 8), 11 (11 / 9, round 3) */
    const char *e = getenv("SPX_INFLATE_ROOT");
    const int root = e ? atoi(e) : 9;
    if (root == 11) return run_core<11, 9>(in, in_len, out, out_len);
    if (root == 10) return run_core<10, 8>(in, in_len, out, out_len);
    return run_core<9, 8>(in, in_len, out, out_len);
}

extern "C" uint32_t spx_crc32_core_host(const uint8_t *p, int64_t n, int32_t pieces)
{
    /* the way the kernel computes it: `pieces` stripes, each with the byte-table recurrence, combined with the GF(2)
     * shift operator */
    uint32_t tab[256];
    for (uint32_t k = 0; k < 256; ++k) tab[k] = spxz::crc_table_entry(k);
    if (pieces < 1) pieces = 1;
    const int64_t step = (n + pieces - 1) / pieces;
    uint32_t crc = 0;
    bool first = true;
    for (int64_t a = 0; a < n || first; a += step) {
        const int64_t b = a + step < n ? a + step : n;
        uint32_t c = 0xffffffffu;
        for (int64_t k = a; k < b; ++k) c = tab[(c ^ p[k]) & 0xff] ^ (c >> 8);
        c ^= 0xffffffffu;
        crc = first ? c : spxz::crc_combine(crc, c, (uint64_t)(b - a));
        first = false;
        if (step == 0) break;
    }
    return crc;
}
