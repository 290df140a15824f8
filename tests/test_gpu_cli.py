"""GPU: the `bamqualcheck` program end to end (BAM + FASTA in, `.bamqc` out) against the oracle."""
import filecmp
import os
import subprocess

import pytest

from bamqc_amd import hostio
from tests.cli_oracle import oracle_bamqualcheck

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bin", "bamqualcheck")


READER = {"BQC_GPU_DECODE": "0"}


@pytest.fixture(autouse=True, params=["host_reader", "gpu_reader"])
def reader(request):
    """Every test of this module runs twice: records decoded by the host reader, and by the reader on the card (forced also for
    these small files; what it does not decode itself — wild tags, unknown read groups — makes the program start over with the
    host reader: the output must not tell)."""
    READER["BQC_GPU_DECODE"] = "1" if request.param == "gpu_reader" else "0"
    yield request.param


def run_cli(*args):
    return subprocess.run([EXE] + list(args), capture_output=True, text=True, env=dict(os.environ, **READER))


def test_config1_10k_reads_bytes_identical(tmp_path):
    # BASELINE.json configs[0]: 10k-read 150 bp PE synthetic BAM vs 1 Mb FASTA, -c chr1
    bam, fa = str(tmp_path / "c1.bam"), str(tmp_path / "c1.fa")
    hostio.synth_write(bam, fa, seed=1001, n_reads=10_000, ref_names=["chr1"], ref_lens=[1_000_000])
    got, want = str(tmp_path / "gpu.bamqc"), str(tmp_path / "oracle.bamqc")
    r = run_cli("-r", fa, "-o", got, "-c", "chr1", bam)  # default -k 32 -q 17 -e 0.01 -s 1: sketch lines included
    assert r.returncode == 0, r.stderr
    assert oracle_bamqualcheck(bam, fa, want, chroms="chr1") == 0
    assert filecmp.cmp(got, want, shallow=False)
    assert filecmp.cmp(got, os.path.join(ROOT, "tests", "golden", "config1.bamqc"), shallow=False)  # committed fixture
    ind = str(tmp_path / "independent.bamqc")  # the oracle fed by tests/pybam.py: no product code on the checker's side
    assert oracle_bamqualcheck(bam, fa, ind, chroms="chr1", independent=True) == 0
    assert filecmp.cmp(got, ind, shallow=False)
    txt = open(got).read()
    assert txt.startswith("sample_id SYN\nlane L1\ntotal_read_pairs ") and "triplet_counts_T_2nd_RC" in txt


def test_multi_lane_default_chroms_small_batches(tmp_path):
    bam, fa = str(tmp_path / "m.bam"), str(tmp_path / "m.fa")
    names = ["chr1", "chr2", "chrX", "chrUn_1"]
    hostio.synth_write(bam, fa, seed=7, n_reads=30_000, ref_names=names, ref_lens=[400_000, 300_000, 200_000, 50_000], n_lanes=3)
    got, want = str(tmp_path / "gpu.bamqc"), str(tmp_path / "oracle.bamqc")
    r = run_cli("--reference", fa, "--output-file=" + got, "-i", "500", "--batch-reads", "7001", "-k", "21,32", "-q", "10", bam)
    assert r.returncode == 0, r.stderr
    assert oracle_bamqualcheck(bam, fa, want, isize=500, klist=(21, 32), qlist=(10,), batch_reads=4000) == 0
    assert filecmp.cmp(got, want, shallow=False)
    assert open(got).read().count("sample_id SYN") == 3


def test_fasta_missing_is_fatal_only_when_needed(tmp_path):
    bam, fa = str(tmp_path / "f.bam"), str(tmp_path / "f.fa")
    hostio.synth_write(bam, fa, seed=9, n_reads=2000, ref_names=["chr1"], ref_lens=[200_000])
    out = str(tmp_path / "o.bamqc")
    r = run_cli("-r", str(tmp_path / "nope.fa"), "-o", out, "-c", "chr1", "--no-sketch", bam)
    assert r.returncode == 1 and "Could not open fasta file" in r.stderr
    assert os.path.getsize(out) == 0  # output is opened (truncated) before the scan and left empty


def test_output_dir_missing(tmp_path):
    bam, fa = str(tmp_path / "f.bam"), str(tmp_path / "f.fa")
    hostio.synth_write(bam, fa, seed=9, n_reads=100, ref_names=["chr1"], ref_lens=[100_000])
    r = run_cli("-r", fa, "-o", str(tmp_path / "no" / "dir" / "o.bamqc"), bam)
    assert r.returncode == 1 and "Could not open output file" in r.stderr


def test_sam_from_stdin_matches_the_bam_run(tmp_path):
    """`bamqualcheck ... -` reads SAM text from stdin (bamqualcheck.cpp:252-260): same bytes as the BAM run."""
    from tests import pybam
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    hostio.synth_write(bam, fa, seed=77, n_reads=6_000, ref_names=["chr1", "chr2"], ref_lens=[300_000, 200_000], n_lanes=2)
    sam = pybam.bam_to_sam_text(bam).encode()
    out_b, out_s = str(tmp_path / "b.bamqc"), str(tmp_path / "s.bamqc")
    p = run_cli("-r", fa, "-o", out_b, "-c", "chr1,chr2", bam)
    assert p.returncode == 0, p.stderr
    q = subprocess.run([EXE, "-r", fa, "-o", out_s, "-c", "chr1,chr2", "-"], input=sam, capture_output=True)
    assert q.returncode == 0, q.stderr
    assert filecmp.cmp(out_b, out_s, shallow=False)


def test_wild_bam_end_to_end(tmp_path):
    """Records no aligner would write (tests/test_gpu_fuzz.py), as a BAM with junk tags and oddly cut BGZF blocks, through the
    program: the `.bamqc` text equals the oracle's byte for byte."""
    import numpy as np
    from tests.test_host_io import _wild_bam
    from tests.test_gpu_fuzz import wild_batch
    bam, fa = str(tmp_path / "w.bam"), str(tmp_path / "w.fa")
    _wild_bam(bam, 33, 4000)
    _, refs = wild_batch(33, 1)  # (the contigs depend on the seed only)
    with open(fa, "w") as f:
        for i, r in enumerate(refs):
            s = "".join("ACGTN"[c] for c in r)
            f.write(">chr%d some description\n" % (i + 1) + "\n".join(s[j:j + 70] for j in range(0, len(s), 70)) + "\n")
    got, want = str(tmp_path / "gpu.bamqc"), str(tmp_path / "oracle.bamqc")
    r = run_cli("-r", fa, "-o", got, "-i", "2000", "--batch-reads", "1500", "-k", "8,32", "-q", "17", bam)
    assert r.returncode == 0, r.stderr
    assert oracle_bamqualcheck(bam, fa, want, isize=2000, klist=(8, 32), qlist=(17,), batch_reads=999) == 0
    assert filecmp.cmp(got, want, shallow=False)


def test_long_reads_end_to_end_batches_cut_by_bases(tmp_path):
    """10 kb reads through the program with a small base limit per batch: batches end on the base limit, the reader's
    parallel record walk sizes its window from bases per record; output equals the oracle's."""
    bam, fa = str(tmp_path / "l.bam"), str(tmp_path / "l.fa")
    hostio.synth_write(bam, fa, seed=5, n_reads=40_000, ref_names=["chr1", "chr2"], ref_lens=[3_000_000, 2_000_000], read_len=10_000, long_reads=True)
    got, want = str(tmp_path / "gpu.bamqc"), str(tmp_path / "oracle.bamqc")
    r = run_cli("-r", fa, "-o", got, "-c", "chr1,chr2", "--no-sketch", "--max-read-len", "65536", bam)
    assert r.returncode == 0, r.stderr
    assert oracle_bamqualcheck(bam, fa, want, chroms="chr1,chr2", klist=(), qlist=(), batch_reads=5000) == 0
    assert filecmp.cmp(got, want, shallow=False)


def test_paired_config1_output_feeds_the_downstream_consumer(tmp_path):
    """SURVEY §8f N3 on the PRODUCT's output: the program's `.bamqc` for the properly paired config-1 reads passes the line rules
    of the reference's bamqc_summary.py:96-131 (restated in tests/test_summary_consumer.py), holds every key summarize() reads,
    and is byte for byte the oracle's file — the one tests/golden/config1_summary.json was computed from by the reference's own
    consumer."""
    from tests.test_summary_consumer import NEEDED, paired_config1_bamqc, read_rules
    want, n = paired_config1_bamqc(tmp_path)
    got = str(tmp_path / "product.bamqc")
    r = run_cli("-r", str(tmp_path / "c1.fa"), "-o", got, "-c", "chr1", str(tmp_path / "paired.bam"))
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(got, want, shallow=False)
    lanes = read_rules(got)
    assert len(lanes) == 1
    lane = lanes[0]
    assert lane["total_read_pairs"] == n and lane["read_length"] == 150
    assert not [k for k in NEEDED if k not in lane]
    assert len(lane["8mer_count"]) == 65536 and len(lane["genome_coverage_histogram"]) == 101


def test_optional_paths_write_the_same_bytes(tmp_path):
    """Switchable paths that are off by default must not change a byte: references uploaded beside the record loop (BQC_BG_REFS),
    k_short's XCD-contiguous chunk walk (BQC_SHORT_XCD), the inflate kernels in one phase (BQC_GI_TWO_PHASE=0), phase 1 by a lane
    per block (BQC_GI_WAVE=0, BQC_GI_LEAN)."""
    bam, fa = str(tmp_path / "m.bam"), str(tmp_path / "m.fa")
    hostio.synth_write(bam, fa, seed=17, n_reads=40_000, ref_names=["chr1", "chr2", "chrX"], ref_lens=[600_000, 300_000, 200_000], n_lanes=2)
    want = str(tmp_path / "oracle.bamqc")
    assert oracle_bamqualcheck(bam, fa, want) == 0
    for k, env in enumerate(({"BQC_BG_REFS": "1"}, {"BQC_SHORT_XCD": "1"}, {"BQC_GI_TWO_PHASE": "0"}, {"BQC_GI_TWO_PHASE": "0", "BQC_GI_LEAN": "32"}, {"BQC_GI_LEAN": "32"},
                            {"BQC_GI_WAVE": "0"})):
        got = str(tmp_path / ("got%d.bamqc" % k))
        r = subprocess.run([EXE, "-r", fa, "-o", got, bam], capture_output=True, text=True, env=dict(os.environ, **READER, **env))
        assert r.returncode == 0, (env, r.stderr)
        assert filecmp.cmp(got, want, shallow=False), env


def test_one_read_group_anchored_then_handed_over(tmp_path, reader):
    """One read group, the reader on the card, small batches: the first batches are anchored on the card (their columns never on the host);
    then a record the card does not decode (a second NM tag) makes ITS batch go through the host decoder — and from there on the host keeps the
    window state machine (the card's copy is behind by then); later, reads of a read group that is not in the header (lane 0 by the
    reference's getLane).  Same bytes as the host reader's run and as the oracle's."""
    import struct
    import numpy as np
    from bamqc_amd import synth
    from tests import pybam
    lens = [600_000, 300_000]
    refs = [synth.reference(31, i, ln) for i, ln in enumerate(lens)]
    cols = synth.batch(31, 24_000, lens, refs)
    n = len(cols["flag"])
    text = "@HD\tVN:1.6\n" + "".join("@SQ\tSN:chr%d\tLN:%d\n" % (i + 1, ln) for i, ln in enumerate(lens)) + "@RG\tID:L1\tSM:SYN\n"
    recs, so, qo, co = [], 0, 0, 0
    for i in range(n):
        L, nc = int(cols["l_seq"][i]), int(cols["n_cigar"][i])
        rg = b"other" if i >= 20_000 and i % 50 == 0 else b"L1"  # (a read group the header does not know: lane 0, bamqualcheck.cpp:86)
        tags = b"RGZ" + rg + b"\0" + b"ASi" + struct.pack("<i", int(cols["as_"][i]))
        nm = int(cols["nm"][i])
        if nm >= 0:
            tags += b"NMi" + struct.pack("<i", nm)
            if i == 13_000:
                tags += b"NMC" + struct.pack("<B", min(nm + 1, 255))  # a second NM tag: QualityCheck.hpp:201-218 looks at every one of them
        recs.append(dict(rid=int(cols["rid"][i]), pos=int(cols["pos"][i]), mapq=int(cols["mapq"][i]), flag=int(cols["flag"][i]) & 0xFFF,
                         rnext=0 if int(cols["flag"][i]) & 0x1000 else -1, tlen=int(cols["tlen"][i]), name="r%d" % i, cigar=cols["cigar"][co:co + nc],
                         seq=cols["seq"][so:so + (L + 1) // 2], qual=cols["qual"][qo:qo + L], l_seq=L, tags=tags))
        so += (L + 1) // 2; qo += L; co += nc
    bam, fa = str(tmp_path / "h.bam"), str(tmp_path / "h.fa")
    pybam.write_bam(bam, text, [("chr1", lens[0]), ("chr2", lens[1])], recs, rng=np.random.default_rng(2))
    hostio.write_fasta(fa, ["chr1", "chr2"], refs)
    got, want = str(tmp_path / "gpu.bamqc"), str(tmp_path / "oracle.bamqc")
    r = subprocess.run([EXE, "-r", fa, "-o", got, "-c", "chr1,chr2", "--batch-reads", "3000", bam], capture_output=True, text=True, env=dict(os.environ, BQC_TIMING="1", **READER))
    assert r.returncode == 0, r.stderr
    if reader == "gpu_reader":
        import re
        m = re.search(r"\[timing\] (\d+) batches anchored on the card", r.stderr)
        assert m and 1 <= int(m.group(1)) <= 5, r.stderr   # the batches in front of the second NM tag, none behind it
        assert "went through the host decoder" in r.stderr
    assert oracle_bamqualcheck(bam, fa, want, chroms="chr1,chr2", batch_reads=3000) == 0
    assert filecmp.cmp(got, want, shallow=False)
