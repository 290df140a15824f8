"""Test helper: the reference program's behaviour restated on top of the oracle — reads a BAM +
FASTA with the product's host reader, pushes the records through the CPU oracle and writes the
`.bamqc` with the oracle's own writer.  (TEST INFRASTRUCTURE.)"""
import numpy as np

from bamqc_amd import hostio
from tests.oracle_lib import Oracle

DEFAULT_CHROMS = ",".join("chr%d" % i for i in range(1, 23))


def oracle_bamqualcheck(bam, fasta, out, chroms=DEFAULT_CHROMS, isize=1000, klist=(32,), qlist=(17,), max_read_len=65536,
                        hist_cap=65536, batch_reads=1 << 20):
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
