"""Seeded synthetic BAM-record generator for tests (numpy, small/medium sizes).

Produces the column dict accepted by `bamqc_amd._abi.make_batch` plus the Dna5
reference arrays.  Mirrors the synthetic-input recipe of SURVEY.md §8d: 150 bp
paired reads, coordinate sorted, substitutions / indels / soft clips, flag mix,
NM and AS tags.  The large configurations are produced by the C++ generator
inside the product library (`bqc_synth_*`), not by this file.
"""
import numpy as np

NIB = np.array([1, 2, 4, 8, 15], dtype=np.uint8)  # A C G T N in BAM 4-bit code
OPS = {c: i for i, c in enumerate("MIDNSHP=X")}


def make_reference(rng, n_refs, ref_len, n_frac=0.001):
    refs = []
    for _ in range(n_refs):
        r = rng.integers(0, 4, size=ref_len, dtype=np.uint8)
        n_runs = max(1, int(ref_len * n_frac / 50))
        for _ in range(n_runs):
            s = int(rng.integers(0, max(1, ref_len - 50)))
            r[s:s + int(rng.integers(1, 100))] = 4
        refs.append(r)
    return refs


def pack_nibbles(nibs):
    """4-bit pack one read (high nibble first)."""
    if len(nibs) & 1:
        nibs = np.concatenate([nibs, np.zeros(1, np.uint8)])
    return (nibs[0::2] << 4) | nibs[1::2]


def cigar_words(ops):
    return np.array([(n << 4) | OPS[c] for n, c in ops], dtype=np.uint32)


def synth(seed=1, n_reads=2000, L=150, n_refs=1, ref_len=200_000, n_lanes=1, long_cigar=False,
          p_dup=0.01, p_qcfail=0.005, p_secondary=0.002, p_supp=0.002, p_unmapped=0.013,
          p_indel=0.02, p_clip=0.03, p_proper=0.96, var_len=False, refs=None, density=None,
          p_iupac=0.0, p_noqual=0.0, hardclip=False):
    rng = np.random.default_rng(seed)
    if refs is None:
        refs = make_reference(rng, n_refs, ref_len)
    n_refs = len(refs)
    # positions: coordinate sorted, reads confined to [0, len - 2L - 64)
    rid = np.sort(rng.integers(0, n_refs, size=n_reads)).astype(np.int32)
    pos = np.zeros(n_reads, np.int32)
    for r in range(n_refs):
        m = rid == r
        span = len(refs[r]) - 2 * L - 64
        if density is not None:  # cluster reads to get deep coverage / window slides
            span = min(span, max(1, int(m.sum() * L / density)))
        pos[m] = np.sort(rng.integers(1, max(2, span), size=int(m.sum())))
    cols = {k: [] for k in ("seq", "qual", "cigar")}
    flag = np.zeros(n_reads, np.uint16)
    mapq = np.zeros(n_reads, np.uint8)
    lane = rng.integers(0, n_lanes, size=n_reads).astype(np.uint8)
    tlen = np.zeros(n_reads, np.int32)
    nm = np.zeros(n_reads, np.int32)
    as_ = np.zeros(n_reads, np.int32)
    l_seq = np.zeros(n_reads, np.uint32)
    n_cigar = np.zeros(n_reads, np.uint16)
    qlevels = np.array([2, 12, 23, 37], dtype=np.uint8)
    for i in range(n_reads):
        Li = int(rng.integers(max(8, L // 3), L + 1)) if var_len else L
        f = 0x1
        f |= 0x40 if (i & 1) == 0 else 0x80
        if rng.random() < 0.5:
            f |= 0x10
        if rng.random() < 0.5:
            f |= 0x20
        u = rng.random()
        unmapped = u < p_unmapped
        if unmapped:
            f |= 0x4
            if rng.random() < 0.25:
                f |= 0x8
        elif rng.random() < 0.01:
            f |= 0x8
        if not unmapped and not (f & 0x8) and rng.random() < p_proper:
            f |= 0x2
        if rng.random() < p_dup:
            f |= 0x400
        if rng.random() < p_qcfail:
            f |= 0x200
        if rng.random() < p_secondary:
            f |= 0x100
        if rng.random() < p_supp:
            f |= 0x800
        if rng.random() < 0.95:
            f |= 0x1000  # BQC_FLAG_MATE_MAIN
        # alignment
        ref = refs[rid[i]]
        p = int(pos[i])
        ops = []
        mm = 0
        ind = 0
        if unmapped:
            codes = rng.integers(0, 4, size=Li, dtype=np.uint8)
            ops = []
        else:
            lead = trail = 0
            if rng.random() < p_clip:
                w = rng.integers(0, 3)
                if w in (0, 2):
                    lead = int(rng.integers(1, min(31, Li // 3)))
                if w in (1, 2):
                    trail = int(rng.integers(1, min(31, Li // 3)))
            body = Li - lead - trail
            parts = []
            rp = p
            if hardclip and rng.random() < 0.3:
                ops.append((int(rng.integers(1, 20)), "H"))
            if lead:
                ops.append((lead, "S"))
                parts.append(rng.integers(0, 4, size=lead, dtype=np.uint8))
            n_ev = 0
            if long_cigar:
                n_ev = int(rng.integers(8, 30))
            elif rng.random() < p_indel:
                n_ev = 1
            remaining = body
            for _ in range(n_ev):
                if remaining < 12:
                    break
                m = int(rng.integers(4, max(5, remaining // (2 if not long_cigar else 3))))
                ops.append((m, "M"))
                parts.append(ref[rp:rp + m].copy())
                rp += m
                remaining -= m
                k = int(rng.integers(1, 4))
                t = rng.random()
                if t < 0.45 and remaining > k + 4:
                    ops.append((k, "I"))
                    parts.append(rng.integers(0, 4, size=k, dtype=np.uint8))
                    remaining -= k
                    ind += k
                elif t < 0.9:
                    ops.append((k, "D"))
                    rp += k
                    ind += k
                elif long_cigar:
                    ops.append((k * 10, "N"))
                    rp += k * 10
            ops.append((remaining, "M"))
            parts.append(ref[rp:rp + remaining].copy())
            if trail:
                ops.append((trail, "S"))
                parts.append(rng.integers(0, 4, size=trail, dtype=np.uint8))
            codes = np.concatenate(parts)
            assert len(codes) == Li
            # substitutions on aligned bases (0.5 %)
            sub = rng.random(Li) < 0.005
            sub[:lead] = False
            if trail:
                sub[Li - trail:] = False
            mm = int(sub.sum())
            codes[sub] = (codes[sub] + rng.integers(1, 4, size=mm).astype(np.uint8)) % 4
            # random read N (0.1 %)
            codes[rng.random(Li) < 0.001] = 4
            # merge adjacent equal ops
            merged = []
            for n, c in ops:
                if n == 0:
                    continue
                if merged and merged[-1][1] == c:
                    merged[-1] = (merged[-1][0] + n, c)
                else:
                    merged.append((n, c))
            ops = merged
        nibs = NIB[np.minimum(codes, 4)]
        if p_iupac > 0:
            iu = rng.random(Li) < p_iupac
            nibs[iu] = rng.integers(0, 16, size=int(iu.sum())).astype(np.uint8)
        cyc = np.arange(Li) / max(1, Li)
        pr = np.stack([0.02 + 0.1 * cyc, 0.05 + 0.1 * cyc, 0.18 + 0 * cyc, 0.75 - 0.2 * cyc], axis=1)
        pr /= pr.sum(axis=1, keepdims=True)
        q = qlevels[(rng.random(Li)[:, None] > np.cumsum(pr, axis=1)).sum(axis=1).clip(0, 3)]
        if p_noqual > 0 and rng.random() < p_noqual:
            q = np.full(Li, 0xFF, np.uint8)
            f |= 0x8000
        cols["seq"].append(pack_nibbles(nibs))
        cols["qual"].append(q)
        cols["cigar"].append(cigar_words(ops))
        flag[i] = f
        r = rng.random()
        mapq[i] = 0 if unmapped else (60 if r < 0.9 else (0 if r < 0.93 else int(rng.integers(1, 60))))
        tlen[i] = int(np.clip(rng.normal(400, 80), -1200, 1200)) * (1 if rng.random() < 0.5 else -1)
        nm[i] = -1 if (unmapped or rng.random() < 0.01) else mm + ind
        as_[i] = BQC_AS_ABSENT if unmapped else max(0, Li - 5 * mm - 6 * ind)
        l_seq[i] = Li
        n_cigar[i] = len(ops)
    out = dict(flag=flag, mapq=mapq, lane=lane, rid=rid, pos=pos, tlen=tlen, nm=nm, as_=as_, l_seq=l_seq,
               n_cigar=n_cigar,
               seq=np.concatenate(cols["seq"]) if n_reads else np.zeros(0, np.uint8),
               qual=np.concatenate(cols["qual"]) if n_reads else np.zeros(0, np.uint8),
               cigar=np.concatenate(cols["cigar"]).astype(np.uint32) if n_reads else np.zeros(0, np.uint32))
    # unmapped reads without a mapped mate go to the end with rid -1 (as samtools sort does)
    return out, refs


BQC_AS_ABSENT = -(2 ** 31)


def single_read(seq, qual, cigar, flag, pos=0, rid=0, mapq=60, tlen=300, nm=0, as_=100, lane=0):
    """One hand-made record. seq: str over '=ACMGRSVTWYHKDBN', qual: list of Phred, cigar: [(n,'M'),...]"""
    table = "=ACMGRSVTWYHKDBN"
    nibs = np.array([table.index(c) for c in seq], dtype=np.uint8)
    return dict(flag=np.array([flag], np.uint16), mapq=np.array([mapq], np.uint8), lane=np.array([lane], np.uint8),
                rid=np.array([rid], np.int32), pos=np.array([pos], np.int32), tlen=np.array([tlen], np.int32),
                nm=np.array([nm], np.int32), as_=np.array([as_], np.int32), l_seq=np.array([len(seq)], np.uint32),
                n_cigar=np.array([len(cigar)], np.uint16), seq=pack_nibbles(nibs) if len(seq) else np.zeros(0, np.uint8),
                qual=np.array(qual, dtype=np.uint8), cigar=cigar_words(cigar))


def concat(batches):
    keys = batches[0].keys()
    return {k: np.concatenate([b[k] for b in batches]) for k in keys}


def slice_batch(cols, lo, hi):
    """Reads [lo, hi) of a column dict as a new dict (re-slicing the packed arrays)."""
    l = cols["l_seq"].astype(np.int64)
    so = np.concatenate([[0], np.cumsum((l + 1) // 2)])
    qo = np.concatenate([[0], np.cumsum(l)])
    co = np.concatenate([[0], np.cumsum(cols["n_cigar"].astype(np.int64))])
    out = {}
    for k, v in cols.items():
        if k == "seq":
            out[k] = v[so[lo]:so[hi]]
        elif k == "qual":
            out[k] = v[qo[lo]:qo[hi]]
        elif k == "cigar":
            out[k] = v[co[lo]:co[hi]]
        elif k in ("nm_extra_read", "nm_extra_val"):
            continue
        else:
            out[k] = v[lo:hi]
    return out
