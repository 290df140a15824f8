"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact.
Run on the GPU box with `pytest -m gpu`."""
import filecmp
import os

import numpy as np
import pytest

from tests import synth
from tests.parity import assert_parity, split

pytestmark = pytest.mark.gpu

P, PR, UN, MUN, REV, MREV, FIRST, LAST = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80
MM = 0x1000


def test_library_loads_and_device_present():
    from bamqc_amd import Aggregator
    a = Aggregator(n_refs=1)
    assert a.state_words > 65536
    a.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_synth_150bp(seed):
    cols, refs = synth.synth(seed=seed, n_reads=6000, n_refs=2, ref_len=120_000)
    assert_parity(cols, refs)


@pytest.mark.parametrize("L", [36, 100, 151, 250, 255])
def test_read_length_classes_two_read_groups(L):
    # lanes per read 3 / 7 / 10 / 16 / 16 in k_short (reads per wave 21 / 9 / 6 / 4 / 4), reads of both read groups interleaved
    from bamqc_amd import synth as csynth
    lens = [1_500_000] * 2
    refs = [csynth.reference(11, i, n) for i, n in enumerate(lens)]
    cols = csynth.batch(11 + L, 60_000, lens, refs, read_len=L, n_lanes=2)
    assert_parity(cols, refs, n_lanes=2)


def test_synth_deep_coverage_slides():
    # dense reads: many window slides, depth beyond the clamp
    cols, refs = synth.synth(seed=11, n_reads=8000, n_refs=1, ref_len=60_000, density=120)
    co, cg, _, _ = assert_parity(cols, refs)
    assert co[0]["poscov"][100] > 0


def test_synth_sparse_coverage_resets():
    cols, refs = synth.synth(seed=12, n_reads=1500, n_refs=3, ref_len=3_000_000)
    assert_parity(cols, refs)


def test_synth_multi_lane():
    cols, refs = synth.synth(seed=5, n_reads=5000, n_refs=2, ref_len=100_000, n_lanes=3)
    assert_parity(cols, refs, n_lanes=3)


def test_synth_variable_length_and_iupac_and_noqual():
    cols, refs = synth.synth(seed=6, n_reads=4000, n_refs=1, ref_len=100_000, var_len=True, p_iupac=0.01, p_noqual=0.01,
                             hardclip=True)
    assert_parity(cols, refs)


def test_synth_long_reads_beyond_lds_cycles():
    cols, refs = synth.synth(seed=7, n_reads=300, L=1500, n_refs=1, ref_len=400_000, long_cigar=True)
    assert_parity(cols, refs, max_read_len=2048, isize=3000)


def test_main_chrom_subset_and_fasta_index():
    cols, refs = synth.synth(seed=8, n_reads=4000, n_refs=3, ref_len=80_000)
    assert_parity(cols, refs, main_chrom=[1, 0, 1], fasta_index=[0, 1, 2])


@pytest.mark.parametrize("cuts", [[1000], [1, 2, 3], [500, 501, 2999], [2000, 2000]])
def test_batch_splits_carry_coverage(cuts):
    cols, refs = synth.synth(seed=21, n_reads=3000, n_refs=2, ref_len=40_000, density=40, n_lanes=2)
    assert_parity(split(cols, cuts), refs, n_lanes=2)


def test_many_small_batches():
    cols, refs = synth.synth(seed=22, n_reads=900, n_refs=1, ref_len=30_000, density=30)
    assert_parity(split(cols, list(range(37, 900, 37))), refs)


def test_empty_batch_and_no_reads():
    cols, refs = synth.synth(seed=23, n_reads=0, n_refs=1, ref_len=10_000)
    co, cg, _, _ = assert_parity(cols, refs)
    assert int(cg[0]["poscov"][0]) == 2000


def test_kat_cases_on_gpu():
    ref = np.array([0, 1, 2, 3] * 50, dtype=np.uint8)
    ok = P | PR | FIRST
    recs = [
        synth.single_read("ACGTACGTAC", [40] * 10, [(10, "M")], ok | MM, pos=0, mapq=60, as_=100),
        synth.single_read("ACGTTTACGT", [40] * 10, [(4, "M"), (2, "I"), (4, "M")], ok, pos=0, mapq=60, as_=100, nm=2),
        synth.single_read("ACGTACGTAC", [40] * 10, [(4, "M"), (4, "D"), (6, "M")], ok | REV, pos=0, mapq=60, as_=100, nm=4),
        synth.single_read("ACGTNCGTACGTACGT", [30] * 16, [(16, "M")], P | LAST | UN, rid=-1, pos=-1),
        synth.single_read("A" * 150, [30] * 150, [(5, "S"), (140, "M"), (5, "S")], P | FIRST | MM, pos=100, mapq=30),
        synth.single_read("A" * 150, [30] * 150, [(145, "M"), (5, "S")], P | LAST | REV | MM, pos=100, mapq=30),
        synth.single_read("A" * 150, [30] * 150, [(150, "M")], P | FIRST | MM, pos=1100, mapq=0, as_=0),
        synth.single_read("A" * 150, [30] * 150, [(150, "M")], P | FIRST | MM, pos=1101, mapq=0, as_=0),
        synth.single_read("A" * 150, [30] * 150, [(150, "M")], P | FIRST | MM, pos=3101, mapq=0, as_=0),  # exactly shift+2000
        synth.single_read("A" * 20, [30] * 20, [(20, "M")], 0x800 | 0x41, pos=10),
        synth.single_read("A" * 20, [30] * 20, [(20, "M")], 0x100 | 0x41, pos=10),
        synth.single_read("ACGTNRYC", [10] * 8, [], P | FIRST | UN, rid=-1, pos=-1),
        synth.single_read("AC", [10] * 2, [], P | FIRST | UN, rid=-1, pos=-1),
        synth.single_read("", [], [], P | LAST | UN, rid=-1, pos=-1),
    ]
    for r in recs:
        assert_parity(r, [ref])
    assert_parity(synth.concat(recs), [ref])


def test_triplet_weird_cigars():
    rng = np.random.default_rng(5)
    ref = rng.integers(0, 4, size=5000).astype(np.uint8)
    ok = P | PR | FIRST
    recs = []
    for trial in range(300):
        L = int(rng.integers(3, 200))
        pos = int(rng.integers(0, 4000))
        # random op sequence without clipping; lengths need not be consistent with L
        ops = []
        for _ in range(int(rng.integers(1, 8))):
            ops.append((int(rng.integers(0, 60)), "MIDNP=X"[int(rng.integers(0, 7))]))
        seq = "".join("ACGT"[c] for c in ref[pos:pos + L]) if rng.random() < 0.7 else "".join("ACGTN"[int(x)] for x in rng.integers(0, 5, L))
        q = rng.integers(10, 45, size=L).tolist()
        nm = sum(n for n, c in ops if c in "ID") + int(rng.integers(0, 3))
        recs.append(synth.single_read(seq, q, ops, ok | (REV if trial & 1 else 0) | (0x80 if trial & 2 else 0), pos=pos, mapq=60, as_=77, nm=nm))
    assert_parity(synth.concat(recs), [ref], hist_cap=1024)


def test_triplet_contig_edges_and_segments():
    """Reads hanging over both contig ends (with N in the overhang), segments whose virtual start is negative, the fast-path
    length limit (255 / 256 bases) and indel reads of several read groups in one batch."""
    rng = np.random.default_rng(9)
    reflen = 700
    ref = rng.integers(0, 4, size=reflen).astype(np.uint8)
    ref[5:9] = 4  # N run near the start
    ok = P | PR
    recs = []
    for trial in range(400):
        L = int(rng.choice([3, 17, 100, 150, 151, 254, 255, 256, 300]))
        pos = int(rng.choice([0, 1, 2, 7, reflen - L - 2, reflen - L, reflen - L + 3, reflen - 20, reflen - 2, int(rng.integers(0, reflen))]))
        pos = max(pos, 0)
        kind = trial % 4
        if kind == 0: ops = [(L, "M")]
        elif kind == 1: ops = [(3, "I"), (L - 3, "M")] if L > 6 else [(L, "M")]         # first op an insertion: segment starts before pos
        elif kind == 2: ops = [(L // 2, "M"), (2, "D"), (L - L // 2, "M")]
        else: ops = [(L // 3, "M"), (4, "N"), (L // 3, "M"), (1, "I"), (L - 2 * (L // 3) - 1, "M")]
        codes = np.pad(ref, (0, 400))[pos:pos + L].astype(np.int64)  # beyond the contig: 'A'
        codes[rng.random(L) < 0.03] = 4
        seq = "".join("ACGTN"[int(c)] for c in codes)
        q = rng.integers(15, 45, size=L).tolist()
        nm = sum(n for n, c in ops if c in "ID")
        flag = ok | (REV if trial & 4 else 0) | (FIRST if trial & 8 else LAST)
        recs.append(synth.single_read(seq, q, ops, flag, pos=pos, mapq=60, as_=90, nm=nm, lane=trial % 3))
    assert_parity(synth.concat(recs), [ref], n_lanes=3, hist_cap=1024, max_read_len=512)


def test_long_reads_many_short_segments_per_lane():
    """k_long hands the CIGAR's match-like segments to the lanes that hold the cycles (two per lane and pass): reads of several rows
    whose segments are shorter than a lane's 16 cycles (three and more per lane: the extra rounds), segments straddling lanes and rows,
    more than 64 operations (several blocks), both strands, reads that hang over the contig's ends."""
    rng = np.random.default_rng(44)
    reflen = 9000
    ref = rng.integers(0, 4, size=reflen).astype(np.uint8)
    ok = P | PR
    recs = []
    for trial in range(120):
        L = int(rng.choice([300, 993, 1500, 2100, 3000]))
        kind = trial % 4
        ops, left = [], L
        while left > 0:
            if kind == 0: m = int(rng.integers(1, 6))          # very short segments
            elif kind == 1: m = int(rng.integers(1, 40))
            elif kind == 2: m = int(rng.choice([15, 16, 17, 31, 32, 33]))
            else: m = int(rng.integers(100, 600))
            m = min(m, left)
            ops.append((m, "M=X"[int(rng.integers(0, 3))]))
            left -= m
            if left <= 0: break
            r = rng.random()
            if r < 0.4: ops.append((int(rng.integers(1, 4)), "D"))
            elif r < 0.8:
                k = min(int(rng.integers(1, 4)), left)
                ops.append((k, "I")); left -= k
            else: ops.append((int(rng.integers(1, 30)), "N"))
            if len(ops) > 600: ops.append((left, "M")); left = 0
        span = sum(n for n, c in ops if c in "M=XDN")
        pos = int(rng.choice([0, 1, max(reflen - span - 1, 0), max(reflen - span // 2, 0), int(rng.integers(0, max(reflen - span, 1)))]))
        # the read follows the reference along its CIGAR (so that triplets do get counted), with a few errors and N
        codes, rp = [], pos
        for n, c in ops:
            if c in "M=X": codes.extend(np.pad(ref, (0, 5000))[rp:rp + n].tolist()); rp += n
            elif c == "I": codes.extend(rng.integers(0, 4, n).tolist())
            else: rp += n
        codes = np.array(codes[:L] + [0] * (L - len(codes)), np.int64)
        err = rng.random(L) < 0.02
        codes[err] = rng.integers(0, 5, int(err.sum()))
        seq = "".join("ACGTN"[int(c)] for c in codes)
        q = rng.integers(15, 45, size=L).tolist()
        nm = sum(n for n, c in ops if c in "ID")
        flag = ok | (REV if trial & 4 else 0) | (FIRST if trial & 8 else LAST)
        recs.append(synth.single_read(seq, q, ops, flag, pos=pos, mapq=60, as_=90, nm=nm, lane=trial % 2))
    assert_parity(synth.concat(recs), [ref], n_lanes=2, hist_cap=4096, max_read_len=4096, isize=5000)


def test_error_codes_match():
    ref = np.zeros(1000, np.uint8)
    ok = P | PR | FIRST
    for cols in (
        synth.single_read("A" * 20, [30] * 20, [(20, "M")], 0x1, pos=10),                                  # no mate flag
        synth.single_read("A" * 20, [30] * 20, [(20, "M")], ok, pos=10, mapq=60, as_=synth.BQC_AS_ABSENT),  # AS missing
        synth.single_read("A" * 20, [30] * 20, [(20, "M")], ok, pos=10, mapq=60, as_=-3),                   # AS negative
        synth.single_read("A" * 20, [30] * 20, [(10, "M"), (2, "I"), (8, "M")], P | FIRST, pos=5, nm=1),    # NM < D+I
    ):
        assert_parity(cols, [ref])
    # contig missing from the FASTA / FASTA order violated
    cols = synth.single_read("A" * 20, [30] * 20, [(20, "M")], ok, pos=10, mapq=60, as_=60)
    assert_parity(cols, [None], n_refs=1)
    a = synth.single_read("A" * 20, [30] * 20, [(20, "M")], ok, pos=10, rid=1, mapq=60, as_=60)
    b = synth.single_read("A" * 20, [30] * 20, [(20, "M")], ok, pos=10, rid=0, mapq=60, as_=60)
    assert_parity(synth.concat([a, b]), [ref, ref], n_refs=2)


def test_multiple_nm_tags():
    cols = synth.concat([synth.single_read("A" * 20, [30] * 20, [(20, "M")], P | FIRST, pos=5, nm=1),
                         synth.single_read("A" * 20, [30] * 20, [(18, "M"), (2, "I")], P | LAST, pos=9, nm=3)])
    cols["nm_extra_read"] = np.array([1, 1], np.uint32)
    cols["nm_extra_val"] = np.array([4, 2], np.int32)
    co, cg, _, _ = assert_parity(cols, None)
    assert cg[0]["r2.mismatch"].tolist() == [1, 1, 1]


def test_bamqc_text_identical(tmp_path):
    cols, refs = synth.synth(seed=31, n_reads=5000, n_refs=2, ref_len=100_000, n_lanes=2)
    co, cg, o, a = assert_parity(cols, refs, n_lanes=2)
    p1, p2 = str(tmp_path / "oracle.bamqc"), str(tmp_path / "gpu.bamqc")
    o.write_bamqc(p1, sample_id="SYN", lane_names=["L1", "L2"])
    a.write_bamqc(p2, sample_id="SYN", lane_names=["L1", "L2"])
    assert filecmp.cmp(p1, p2, shallow=False)
    assert os.path.getsize(p1) > 100_000


def test_state_vector_is_additive_across_shards():
    # shard at chromosome boundaries, sum the flat state vectors, finalize once
    from bamqc_amd import Aggregator
    from bamqc_amd import _abi
    from tests.parity import run_oracle
    cols, refs = synth.synth(seed=41, n_reads=6000, n_refs=4, ref_len=50_000, density=20)
    rid = cols["rid"]
    cut = int(np.searchsorted(rid, 2))
    rc, co, _ = run_oracle([cols], refs, n_refs=4)
    shards = split(cols, [cut])
    total = None
    for sh in shards:
        a = Aggregator(n_refs=4)
        for i, r in enumerate(refs):
            a.set_reference(i, r)
        a.submit(sh)
        v = a.state_export_host()
        total = v if total is None else total + v
        a.close()
    m = Aggregator(n_refs=4)
    m.state_import_host(total)
    cg = m.finalize()
    d = _abi.diff_counts(co, cg)
    assert not d, "\n".join(d[:10])


def test_reset_and_device_resident_reprocess():
    from bamqc_amd import Aggregator
    from bamqc_amd import _abi
    from tests.parity import run_oracle
    cols, refs = synth.synth(seed=51, n_reads=3000, n_refs=1, ref_len=60_000)
    rc, co, _ = run_oracle([cols], refs, n_refs=1)
    a = Aggregator(n_refs=1)
    a.set_reference(0, refs[0])
    db = a.upload(cols)
    assert db.algorithmic_bytes == 48 * 3000 + len(cols["seq"]) + len(cols["qual"]) + 4 * len(cols["cigar"])
    a.set_timing(True)
    for _ in range(3):
        a.process(db)
    a.sync()
    t = a.last_timing()
    assert "k_short" in t and t["k_short"] > 0
    a.reset()
    a.process(db)
    cg = a.finalize()
    db.free()
    d = _abi.diff_counts(co, cg)
    assert not d, "\n".join(d[:10])
