"""GPU parity on records no aligner would write: every CIGAR operation in any order and with lengths unrelated to the
read, positions before / behind the contig and out of order, read lengths 0 .. 600 (both kernels' ranges and the
boundary between them), any IUPAC code, qualities up to 100 and missing qualities, extreme TLEN / MAPQ / NM values,
several read groups mixed.  The oracle defines the behaviour (the reference's where it has one, the DEFINED cases of
DESIGN.md elsewhere); the HIP path must produce the same counts or the same error code."""
import numpy as np
import pytest

from tests import synth
from tests.parity import assert_parity, split

pytestmark = pytest.mark.gpu

OPS = "MIDNSHP=X"


def wild_batch(seed, n_reads, n_refs=3, ref_len=30_000, n_lanes=3, max_len=600):
    rng = np.random.default_rng(seed)
    refs = [rng.integers(0, 5, size=ref_len + int(rng.integers(0, 17)), dtype=np.uint8) for _ in range(n_refs)]
    for r in refs:  # mostly A/C/G/T with stretches of N
        r[rng.random(len(r)) < 0.8] %= 4
    flag = np.zeros(n_reads, np.uint16); mapq = np.zeros(n_reads, np.uint8); lane = np.zeros(n_reads, np.uint8)
    rid = np.zeros(n_reads, np.int32); pos = np.zeros(n_reads, np.int32); tlen = np.zeros(n_reads, np.int32)
    nm = np.zeros(n_reads, np.int32); as_ = np.zeros(n_reads, np.int32); l_seq = np.zeros(n_reads, np.uint32)
    n_cigar = np.zeros(n_reads, np.uint16)
    seqs, quals, cigs = [], [], []
    cur_rid, cur_pos = 0, 0
    for i in range(n_reads):
        u = rng.random()
        L = int(rng.integers(0, 9)) if u < 0.08 else int(rng.integers(9, 256)) if u < 0.75 else int(rng.integers(min(256, max_len), max_len + 1))
        if u > 0.98:
            L = int(rng.choice([255, 256, 257, 16, 32, 48, 240]))
        f = int(rng.choice([0x40, 0x80]))  # (a primary record without either is an error, covered in test_error_codes_match)
        for bit, p in ((0x1, 0.95), (0x2, 0.85), (0x4, 0.05), (0x8, 0.05), (0x10, 0.5), (0x20, 0.5), (0x100, 0.03), (0x200, 0.03),
                       (0x400, 0.05), (0x800, 0.03), (0x1000, 0.8)):
            if rng.random() < p:
                f |= bit
        # position: mostly forward, sometimes far jumps, back steps, the contig's ends, negative
        v = rng.random()
        if v < 0.002:  # next contig (the FASTA scan is forward-only: going back is an error, tested elsewhere)
            cur_rid = min(cur_rid + 1, n_refs - 1)
            cur_pos = int(rng.integers(0, ref_len))
        elif v < 0.03:
            cur_pos = int(rng.integers(0, ref_len))
        elif v < 0.06:
            cur_pos = max(0, cur_pos - int(rng.integers(1, 5000)))
        elif v < 0.09:
            cur_pos = ref_len - int(rng.integers(-5, 700))
        elif v < 0.11:
            cur_pos = int(rng.integers(-3, 4))
        else:
            cur_pos += int(rng.integers(0, 60)) if rng.random() < 0.9 else int(rng.integers(900, 3500))
        rid[i] = -1 if (f & 0x4) and rng.random() < 0.3 else cur_rid
        pos[i] = -1 if rng.random() < 0.01 else cur_pos
        # CIGAR: nothing, one plausible M, or anything
        w = rng.random()
        if w < 0.06:
            ops = []
        elif w < 0.45:
            ops = [(max(1, L), "M")]
        elif w < 0.7:
            a = int(rng.integers(0, max(1, L // 2) + 1))
            ops = [(n, c) for n, c in ((a, "S"), (max(1, L - a), "M"), (int(rng.integers(0, 40)), "S")) if n]
        else:
            ops = []
            for _ in range(int(rng.integers(1, 9))):
                c = OPS[int(rng.integers(0, 9))]
                long_op = c not in "ID" and rng.random() < 0.1  # (long insertions / deletions would exceed hist_cap: an error, tested elsewhere)
                ops.append((int(rng.integers(200, 3000)) if long_op else int(rng.integers(1, 80)), c))
        codes = rng.integers(0, 4, size=L)
        if rng.random() < 0.6 and rid[i] >= 0:  # bases of the contig at the read's position (where there are any), 2 % substituted
            ref = refs[rid[i]]
            lo, hi = max(0, int(pos[i])), min(len(ref), int(pos[i]) + L)
            if hi > lo:
                seg = ref[lo:hi].astype(np.int64)
                sub = rng.random(hi - lo) < 0.02
                seg[sub] = rng.integers(0, 4, size=int(sub.sum()))
                codes[lo - int(pos[i]):hi - int(pos[i])] = seg
        nibs = synth.NIB[np.minimum(codes, 4)]
        x = rng.random(L)
        nibs[x < 0.02] = 15
        iu = x > 0.985
        nibs[iu] = rng.integers(0, 16, size=int(iu.sum())).astype(np.uint8)
        q = rng.integers(0, 46, size=L).astype(np.uint8)
        hq = rng.random(L) < 0.01
        q[hq] = rng.integers(90, 101, size=int(hq.sum())).astype(np.uint8)
        if L and rng.random() < 0.02:
            q[:] = 0xFF
        seqs.append(synth.pack_nibbles(nibs)); quals.append(q); cigs.append(synth.cigar_words(ops))
        flag[i] = f; lane[i] = int(rng.integers(0, n_lanes)); l_seq[i] = L; n_cigar[i] = len(ops)
        mapq[i] = int(rng.choice([0, 1, 29, 30, 59, 60, 60, 60, 60, 255]))
        t = rng.random()
        tlen[i] = int(rng.integers(-1500, 1500)) if t < 0.9 else int(rng.choice([-(2 ** 31), 2 ** 31 - 1, 0, 1_000_000, -1_000_000]))
        nm[i] = -1 if rng.random() < 0.05 else sum(n for n, c in ops if c in "ID") + int(rng.integers(0, 12))  # (NM < indels: unsigned wrap, an error)
        as_[i] = int(rng.integers(0, 2 * L + 2))
    cols = dict(flag=flag, mapq=mapq, lane=lane, rid=rid, pos=pos, tlen=tlen, nm=nm, as_=as_, l_seq=l_seq, n_cigar=n_cigar,
                seq=np.concatenate(seqs), qual=np.concatenate(quals), cigar=np.concatenate(cigs).astype(np.uint32))
    return cols, refs


@pytest.mark.parametrize("seed", range(16))
def test_wild_records(seed):
    cols, refs = wild_batch(100 + seed, 4000)
    assert_parity(cols, refs, n_lanes=3, max_read_len=1024, isize=2000)


def test_wild_records_in_small_batches_and_short_only():
    cols, refs = wild_batch(7, 3000, max_len=255)
    assert_parity(split(cols, [1, 2, 700, 701, 1500, 2999]), refs, n_lanes=3, max_read_len=1024, isize=2000)


@pytest.mark.parametrize("seed", range(2))
def test_wild_records_with_the_kmer_sketch(seed):
    # the sketch walks every read that passes the flag filter, whatever its CIGAR / position says; qualities around the cutoffs
    cols, refs = wild_batch(300 + seed, 2000)
    assert_parity(split(cols, [700]), refs, n_lanes=3, max_read_len=1024, isize=2000, klist=[5, 32], qlist=[17])


@pytest.mark.parametrize("n_lanes", [1, 2])
def test_hot_8mers_wrap_the_packed_counters(n_lanes):
    """Low-complexity reads: a few 8-mers receive millions of counts, so k_short's packed u8 LDS counters wrap thousands of
    times between two flushes (exact carry accounting), in the read group that uses the scratch rows and, with two read
    groups, in the one that flushes with global atomics; k_long (reads of 300) wraps its own table."""
    rng = np.random.default_rng(5)
    n = 240_000
    L = np.where(rng.random(n) < 0.97, 150, 300).astype(np.uint32)
    kind = rng.integers(0, 10, size=n)
    seqs, quals = [], []
    pat = {0: [1], 1: [1], 2: [1], 3: [1], 4: [1], 5: [8], 6: [1, 2], 7: [4, 4, 8], 8: None, 9: None}  # A.., T.., AC.., GGT.., random
    for i in range(n):
        p = pat[int(kind[i])]
        li = int(L[i])
        nib = np.resize(np.array(p, np.uint8), li) if p else synth.NIB[rng.integers(0, 4, size=li)]
        seqs.append(synth.pack_nibbles(nib))
        quals.append(np.full(li, 30, np.uint8))
    flag = (0x1 | 0x4 | 0x8 | np.where(np.arange(n) % 2 == 0, 0x40, 0x80) | np.where(rng.random(n) < 0.5, 0x10, 0)).astype(np.uint16)
    cols = dict(flag=flag, mapq=np.zeros(n, np.uint8), lane=rng.integers(0, n_lanes, size=n).astype(np.uint8) if n_lanes > 1 else np.zeros(n, np.uint8),
                rid=np.full(n, -1, np.int32), pos=np.full(n, -1, np.int32), tlen=np.zeros(n, np.int32), nm=np.full(n, -1, np.int32),
                as_=np.full(n, synth.BQC_AS_ABSENT, np.int32), l_seq=L, n_cigar=np.zeros(n, np.uint16),
                seq=np.concatenate(seqs), qual=np.concatenate(quals), cigar=np.zeros(0, np.uint32))
    co, cg, _, _ = assert_parity(cols, [np.zeros(1000, np.uint8)], n_lanes=n_lanes, max_read_len=1024)
    assert int(co[0]["eightmer"][0]) > 5_000_000 // n_lanes  # AAAAAAAA


def test_wild_records_many_read_groups():
    # 40 read groups interleaved at random: most chunks of a lane hold a handful of reads, every lane is flushed on its own
    cols, refs = wild_batch(77, 6000, n_lanes=40)
    assert_parity(split(cols, [2500]), refs, n_lanes=40, max_read_len=1024, isize=2000)
