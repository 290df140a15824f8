"""CPU: the committed config-1 fixture (tests/golden/config1.bamqc) is reproduced by the oracle from
the seeded generator + host BAM/FASTA readers (guards generator / reader / oracle / writer drift) and
has the line grammar the reference's downstream parser expects (bamqc_summary.py:96-131)."""
import filecmp
import os

from bamqc_amd import hostio
from tests.cli_oracle import oracle_bamqualcheck

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "config1.bamqc")


def test_oracle_reproduces_committed_fixture(tmp_path):
    bam, fa = str(tmp_path / "c1.bam"), str(tmp_path / "c1.fa")
    hostio.synth_write(bam, fa, seed=1001, n_reads=10_000, ref_names=["chr1"], ref_lens=[1_000_000])
    out = str(tmp_path / "o.bamqc")
    assert oracle_bamqualcheck(bam, fa, out, chroms="chr1") == 0
    assert filecmp.cmp(out, GOLD, shallow=False)


def test_fixture_line_grammar():
    lines = open(GOLD).read().split("\n")
    assert lines[-1] == "" and len(lines) == 81
    keys = [ln.split(" ")[0] for ln in lines[:-1]]
    assert keys[:4] == ["sample_id", "lane", "total_read_pairs", "total_bps"]
    assert keys[15:17] == ["genome_coverage_histogram", "insert_size_histogram"]
    assert keys[-16] == "triplet_counts_A_1st_FW" and keys[-1] == "triplet_counts_T_2nd_RC"
    d = {ln.split(" ")[0]: ln.split(" ")[1:] for ln in lines[:-1]}
    assert len(d["genome_coverage_histogram"]) == 101 and len(d["insert_size_histogram"]) == 1001
    assert len(d["8mer_count"]) == 65536 and len(d["As_by_position_first"]) == 150
    assert len(d["triplet_counts_C_2nd_FW"]) == 64 and len(d["nr_1_most_abundant_8mer"]) == 2
    assert [k for k in keys if "mer_" in k and "abundant" not in k and k != "8mer_count"] == [
        "32mer_count_after_qual_clipping_17", "distinct_32mer_count_after_qual_clipping_17",
        "unique_32mer_count_after_qual_clipping_17", "32mer_F2_after_qual_clipping_17"]
    assert sum(int(x) for x in d["genome_coverage_histogram"]) % 1000 == 0
    float(d["average_base_qual_by_position_first"][0])
