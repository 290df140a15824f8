"""CPU: the checker's two ways to the records agree — the oracle fed by the product's host reader and the oracle fed by
tests/pybam.py (gzip + struct: nothing of the product) write the same `.bamqc`, which is the committed config-1 fixture."""
import filecmp
import os

from bamqc_amd import hostio
from tests.cli_oracle import oracle_bamqualcheck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_independent_decoder_feeds_the_oracle_to_the_same_bytes(tmp_path):
    bam, fa = str(tmp_path / "c1.bam"), str(tmp_path / "c1.fa")
    hostio.synth_write(bam, fa, seed=1001, n_reads=10_000, ref_names=["chr1"], ref_lens=[1_000_000])
    a, b = str(tmp_path / "host_reader.bamqc"), str(tmp_path / "independent.bamqc")
    assert oracle_bamqualcheck(bam, fa, a, chroms="chr1") == 0
    assert oracle_bamqualcheck(bam, fa, b, chroms="chr1", independent=True) == 0
    assert filecmp.cmp(a, b, shallow=False)
    assert filecmp.cmp(b, os.path.join(ROOT, "tests", "golden", "config1.bamqc"), shallow=False)


def test_independent_decoder_multi_lane(tmp_path):
    bam, fa = str(tmp_path / "m.bam"), str(tmp_path / "m.fa")
    hostio.synth_write(bam, fa, seed=7, n_reads=8_000, ref_names=["chr1", "chr2", "chrUn_1"], ref_lens=[400_000, 300_000, 50_000], n_lanes=3)
    a, b = str(tmp_path / "host_reader.bamqc"), str(tmp_path / "independent.bamqc")
    assert oracle_bamqualcheck(bam, fa, a, isize=500, klist=(21,), qlist=(10,)) == 0
    assert oracle_bamqualcheck(bam, fa, b, isize=500, klist=(21,), qlist=(10,), independent=True) == 0
    assert filecmp.cmp(a, b, shallow=False)
