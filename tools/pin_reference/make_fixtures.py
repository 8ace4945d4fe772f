#!/usr/bin/env python3
"""Deterministic input fixtures for pinning the in-tree logic against a REAL secphase binary.

  python tools/pin_reference/make_fixtures.py OUTDIR      writes <name>.bam / <name>.fa for every fixture + manifest.json

The same function is imported by tests/test_reference_pin.py, which regenerates the inputs, checks their digests
against the manifest stored beside the goldens and compares the oracle's (CPU) and the HIP path's (-m gpu) out.log and
BEDs with what the reference wrote.  Nothing here reads /root/reference."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name -> (generator platform, generator overrides, groups, command-line preset flags of the reference run)
FIXTURES = {
    "hifi": ("HIFI", dict(n_contigs=2, contig_len=300000, max_secondaries=3, n_paralogs=3), 300, ["--hifi"]),
    "ont": ("ONT", dict(n_contigs=2, contig_len=300000, max_secondaries=3, n_paralogs=3, read_len=12000), 80, ["--ont"]),
    "edge": ("HIFI", dict(n_contigs=2, contig_len=200000, max_secondaries=4, n_paralogs=3, read_len=5000, hardclip_frac=0.5,
                          softclip_frac=0.6, shuffle_records=1, inverted_paralogs=1, min_secondaries=0), 250, ["--hifi"]),
    "md_only": ("HIFI", dict(n_contigs=2, contig_len=200000, max_secondaries=2, n_paralogs=2, read_len=6000, tag_mode=1), 200, ["--hifi"]),
    "ties": ("HIFI", dict(n_contigs=2, contig_len=200000, max_secondaries=4, n_paralogs=3, read_len=4000, min_secondaries=0,
                          paralog_snv_rate=0.0002), 300, ["--hifi", "-p", "5"]),
    "mixed": ("MIXED", dict(n_contigs=2, contig_len=400000, n_paralogs=8, max_read_len=40000), 150, ["--hifi"]),
}


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def genome_and_reads(name):
    from secphase_amd import synth
    plat, kw, n, _ = FIXTURES[name]
    cfg = synth.default_cfg(getattr(synth, plat), **kw)
    g = synth.Genome(cfg)
    return g, g.reads(0, n)


def write_fixture(name, outdir):
    """<outdir>/<name>.bam + .fa; returns {file: sha256}.  zlib level and block policy are fixed, so the BAM bytes are too."""
    from secphase_amd import synth
    g, r = genome_and_reads(name)
    fa, bam = os.path.join(outdir, name + ".fa"), os.path.join(outdir, name + ".bam")
    synth.write_fasta(fa, g.ref)
    synth.write_bam(bam, [r.batch], g.ref, threads=1, level=6)
    return {name + ".fa": sha256(fa), name + ".bam": sha256(bam)}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "pin_inputs"
    os.makedirs(out, exist_ok=True)
    man = {"fixtures": {}, "flags": {k: v[3] for k, v in FIXTURES.items()}}
    for name in FIXTURES:
        man["fixtures"].update(write_fixture(name, out))
    json.dump(man, open(os.path.join(out, "manifest.json"), "w"), indent=1, sort_keys=True)
    print(f"wrote {len(FIXTURES)} fixtures to {out}")


if __name__ == "__main__":
    main()
