"""Generates tests/golden/config1.bamqc: BASELINE.json configs[0] (10k-read 150 bp PE synthetic BAM vs
1 Mb FASTA, `-c chr1`, default -k 32 -q 17 -e 0.01 -s 1 -i 1000) pushed through the CPU oracle.
The reference binary itself cannot be built here (SeqAn 1.4.2 absent), so this fixture pins the
oracle/generator/host-reader chain against drift; the GPU program must reproduce it byte for byte."""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio  # noqa: E402
from tests.cli_oracle import oracle_bamqualcheck  # noqa: E402

d = tempfile.mkdtemp()
bam, fa = os.path.join(d, "c1.bam"), os.path.join(d, "c1.fa")
hostio.synth_write(bam, fa, seed=1001, n_reads=10_000, ref_names=["chr1"], ref_lens=[1_000_000])
out = os.path.join(HERE, "config1.bamqc")
assert oracle_bamqualcheck(bam, fa, out, chroms="chr1") == 0
print(out, os.path.getsize(out))
