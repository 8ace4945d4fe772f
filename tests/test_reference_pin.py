"""Pin of the in-tree logic against a REAL secphase build (SURVEY 8 rows A1-A14 + N1).

The goldens (tests/golden/ref_<name>.out.log / .modified.bed / .markers.bed + ref_manifest.json) are written by
tools/pin_reference/run.sh on a machine that has the reference built; this container cannot build it (htslib and sonLib
are absent), so until someone runs that script the comparisons SKIP.  What always runs: the fixtures are deterministic
(their digests do not change between two generations) and the oracle accepts every one of them."""
import filecmp
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "pin_reference"))
import make_fixtures as mf  # noqa: E402

from oracle import orc  # noqa: E402
from secphase_amd import records  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MANIFEST = os.path.join(GOLD, "ref_manifest.json")


def _params(name):
    flags = mf.FIXTURES[name][3]
    p = records.preset("ont" if "--ont" in flags else "hifi")
    if "-p" in flags:
        p.prim_margin_score = float(flags[flags.index("-p") + 1])
    return p


def test_fixtures_are_deterministic(built, tmp_path):
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    for name in ("hifi", "edge", "md_only"):
        assert mf.write_fixture(name, str(a)) == mf.write_fixture(name, str(b))


@pytest.mark.parametrize("name", sorted(mf.FIXTURES))
def test_oracle_against_reference_goldens(built, tmp_path, name):
    g, r = mf.genome_and_reads(name)
    log_o, bm_o, bk_o = (str(tmp_path / n) for n in ("o.log", "o.mod.bed", "o.mk.bed"))
    nre, res = orc.run_batch(r.batch, g.ref, _params(name), threads=2, seed=1, log_path=log_o, bed_modified=bm_o, bed_markers=bk_o)
    assert all(e.n_aln >= 0 for e in res), "a pin fixture must stay inside what the reference defines (DESIGN.md section 4, U1-U6)"
    assert nre > 0
    if not os.path.exists(MANIFEST):
        pytest.skip("no goldens from a real secphase build yet (tools/pin_reference/run.sh SECPHASE_BIN)")
    man = json.load(open(MANIFEST))
    digests = mf.write_fixture(name, str(tmp_path))
    for f, d in digests.items():
        assert man["fixtures"][f] == d, "the fixture generator changed since the goldens were made: run tools/pin_reference/run.sh again"
    assert filecmp.cmp(log_o, os.path.join(GOLD, f"ref_{name}.out.log"), shallow=False)
    assert filecmp.cmp(bm_o, os.path.join(GOLD, f"ref_{name}.modified.bed"), shallow=False)
    assert filecmp.cmp(bk_o, os.path.join(GOLD, f"ref_{name}.markers.bed"), shallow=False)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(mf.FIXTURES))
def test_command_line_against_reference_goldens(built, tmp_path, name):
    """the HIP path through the command-line drop-in on the very files the reference ran on"""
    import subprocess
    if not os.path.exists(MANIFEST):
        pytest.skip("no goldens from a real secphase build yet (tools/pin_reference/run.sh SECPHASE_BIN)")
    mf.write_fixture(name, str(tmp_path))
    exe = os.path.join(ROOT, "secphase_amd", "bin", "secphase")
    outd = str(tmp_path / "out")
    p = subprocess.run([exe] + mf.FIXTURES[name][3] + ["-@", "4", "-i", str(tmp_path / f"{name}.bam"), "-f", str(tmp_path / f"{name}.fa"),
                                                      "--outDir", outd, "--prefix", "t", "--groupsPerBatch", "64"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr
    assert filecmp.cmp(os.path.join(outd, "t.out.log"), os.path.join(GOLD, f"ref_{name}.out.log"), shallow=False)
    assert filecmp.cmp(os.path.join(outd, "t.modified_read_blocks.markers.bed"), os.path.join(GOLD, f"ref_{name}.modified.bed"), shallow=False)
    assert filecmp.cmp(os.path.join(outd, "t.marker_blocks.bed"), os.path.join(GOLD, f"ref_{name}.markers.bed"), shallow=False)


def test_pin_all_argument_handling(tmp_path):
    """tools/pin_all.sh REF_CHECKOUT: refuses what is not a checkout of the reference, and --dry-run prints the two docker
    commands (the reference's own Dockerfile; both pin scripts inside that image) without needing docker"""
    import subprocess
    tool = os.path.join(ROOT, "tools", "pin_all.sh")
    p = subprocess.run([tool], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr
    p = subprocess.run([tool, str(tmp_path), "--dry-run"], capture_output=True, text=True)
    assert p.returncode == 2 and "Dockerfile" in p.stderr
    ref = tmp_path / "ref"
    (ref / "programs" / "src").mkdir(parents=True)
    (ref / "Dockerfile").write_text("FROM scratch\n# htslib-1.17\n")
    (ref / "programs" / "src" / "secphase.c").write_text("\n")
    p = subprocess.run([tool, str(ref), "--dry-run"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 2 and lines[0].startswith("docker build") and str(ref) in lines[0]
    assert "tools/pin_htslib/run.sh /usr/local" in lines[1] and "tools/pin_reference/run.sh" in lines[1]
    p = subprocess.run([tool, str(ref), "--bogus"], capture_output=True, text=True)
    assert p.returncode == 2
