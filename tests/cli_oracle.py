"""Test helper: the reference program's behaviour restated on top of the oracle — reads a BAM +
FASTA, pushes the records through the CPU oracle and writes the `.bamqc` with the oracle's own writer.  The records come
from the product's host reader (large inputs) or, with independent=True, from tests/pybam.py and a FASTA parser of its own:
then nothing of the product is on the checker's side.  (TEST INFRASTRUCTURE.)"""
import numpy as np

from bamqc_amd import hostio
from tests.oracle_lib import Oracle

DEFAULT_CHROMS = ",".join("chr%d" % i for i in range(1, 23))


def oracle_bamqualcheck(bam, fasta, out, chroms=DEFAULT_CHROMS, isize=1000, klist=(32,), qlist=(17,), max_read_len=65536,
                        hist_cap=65536, batch_reads=1 << 20, independent=False):
    if independent:
        return _independent(bam, fasta, out, chroms, isize, klist, qlist, max_read_len, hist_cap)
    f = hostio.BamFile(bam)
    main = np.array([1 if n in chroms.split(",") else 0 for n in f.ref_names], np.uint8)
    f.set_main_chrom(main)
    try:
        fa = hostio.load_fasta(fasta)
    except IOError:
        fa = []
    fidx = np.full(max(1, len(f.ref_names)), -1, np.int32)
    for r, name in enumerate(f.ref_names):
        for i, (n, _) in enumerate(fa):
            if n == name:
                fidx[r] = i
                break
    o = Oracle(n_lanes=f.lane_count, n_refs=len(f.ref_names), isize=isize, main_chrom=main, fasta_index=fidx,
               max_read_len=max_read_len, hist_cap=hist_cap, klist=klist, qlist=qlist)
    for r in range(len(f.ref_names)):
        if fidx[r] >= 0:
            o.reference(r, fa[fidx[r]][1])
    for cols in f.batches(max_reads=batch_reads):
        rc = o.process(cols)
        if rc:
            return rc
    lanes = f.lanes()
    o.finalize()
    o.write_bamqc(out, sample_id=f.sample_id, lane_names=[n for n, _ in lanes], lane_index=[i for _, i in lanes])
    return 0


def _independent(bam, fasta, out, chroms, isize, klist, qlist, max_read_len, hist_cap):
    from tests import pybam
    text, refs, _ = pybam.read_bam(bam)
    names = [n for n, _ in refs]
    main = np.array([1 if n in chroms.split(",") else 0 for n in names] or [0], np.uint8)
    cols, refs, lanes, sample = pybam.columns(bam, main)
    try:
        fa = pybam.read_fasta(fasta)
    except IOError:
        fa = []
    fidx = np.full(max(1, len(names)), -1, np.int32)
    for r, name in enumerate(names):
        for i, (n, _) in enumerate(fa):
            if n == name:
                fidx[r] = i
                break
    o = Oracle(n_lanes=len(lanes), n_refs=len(names), isize=isize, main_chrom=main, fasta_index=fidx, max_read_len=max_read_len,
               hist_cap=hist_cap, klist=klist, qlist=qlist)
    code = np.full(256, 4, np.uint8)
    for ch, v in (("A", 0), ("C", 1), ("G", 2), ("T", 3), ("U", 3)):
        code[ord(ch)] = code[ord(ch.lower())] = v
    for r in range(len(names)):
        if fidx[r] >= 0:
            o.reference(r, code[np.frombuffer(fa[fidx[r]][1].encode(), np.uint8)])
    rc = o.process(cols)
    if rc:
        return rc
    o.finalize()
    order = sorted(lanes.items())  # writeOutput iterates a std::map: lexicographic
    o.write_bamqc(out, sample_id=sample, lane_names=[n for n, _ in order], lane_index=[i for _, i in order])
    return 0
