"""Writes the benchmark's end-to-end input (config 2 as a BAM file + FASTA) into a directory. usage: python tools/make_e2e_input.py DIR [reads] [level]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import hostio  # noqa: E402

d = sys.argv[1]
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
os.makedirs(d, exist_ok=True)
hostio.synth_stream(os.path.join(d, "c2.bam"), os.path.join(d, "c2.fa"), 1002, reads, ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4, level=level)
print(os.path.getsize(os.path.join(d, "c2.bam")))
