"""Test helper: parse a `.bamqc` file (format of the reference's writeOutput, src/bamqualcheck.cpp:156-233) into
{lane: {key: int | [int] | [float] | (str, int)}} and check the size-independent properties that hold for ANY input:
they let a whole-genome-sized run be checked without an oracle of that size.  (TEST INFRASTRUCTURE.)"""
import numpy as np


def parse(path):
    lanes, cur, sample = {}, None, None
    for raw in open(path):
        t = raw.split()
        if not t:
            continue
        if t[0] == "sample_id":
            sample = t[1] if len(t) > 1 else ""
        elif t[0] == "lane":
            cur = lanes.setdefault(t[1] if len(t) > 1 else "", {"sample_id": sample})
        elif t[0].startswith("nr_"):
            cur[t[0]] = (t[1], int(t[2]))
        elif t[0].startswith("average_base_qual_by_position"):
            cur[t[0]] = np.array([float(x) for x in t[1:]])
        elif len(t) == 2 and "histogram" not in t[0] and "by_position" not in t[0] and not t[0].startswith(("8mer_count", "triplet_counts")):
            cur[t[0]] = int(t[1])
        else:
            cur[t[0]] = np.array([int(x) for x in t[1:]], dtype=np.int64)
    return lanes


def check_invariants(lane, n_records=None, read_len=None):
    """Properties of one lane's counters that hold whatever the reads are (counters are `unsigned` in the reference: the
    checks are written for totals below 2^32, i.e. up to ~4 G reads per lane)."""
    g = lane
    for sfx in ("first", "second"):
        rl = g["read_length_histogram_" + sfx]
        n_m = int(rl.sum())
        assert int(g["N_count_histogram_" + sfx].sum()) == n_m
        assert int(g["GC_content_histogram_" + sfx].sum()) == n_m
        assert int(g["average_base_qual_histogram_" + sfx].sum()) == n_m
        # every read contributes exactly one base code per cycle it reaches
        n_cyc = len(g["As_by_position_" + sfx])
        assert n_cyc == max(0, len(rl) - 1)
        reach = n_m - np.concatenate([[0], np.cumsum(rl)])[:n_cyc]
        tot = sum(g[c + "s_by_position_" + sfx] for c in "ACGTN")
        assert np.array_equal(tot, reach), sfx
        assert int(g["deletion_count_histogram_" + sfx].sum()) == int(g["insertion_count_histogram_" + sfx].sum()) == int(g["mapping_qual_histogram_" + sfx].sum())
        assert int(g["mismatch_count_histogram_" + sfx].sum()) <= int(g["mapping_qual_histogram_" + sfx].sum())
        if read_len is not None and n_m:
            assert len(rl) == read_len + 1 and int(rl[read_len]) == n_m
    n1, n2 = int(g["read_length_histogram_first"].sum()), int(g["read_length_histogram_second"].sum())
    prim = n1 + n2
    assert g["total_read_pairs"] == prim // 2
    bps = sum(int((np.arange(len(g["read_length_histogram_" + s])) * g["read_length_histogram_" + s]).sum()) for s in ("first", "second"))
    assert g["total_bps"] == bps
    if n_records is not None:
        assert prim + g["supplementary_alignments"] + g["not_primary_alignment"] == n_records
    assert int(g["genome_coverage_histogram"].sum()) % 1000 == 0 and len(g["genome_coverage_histogram"]) == 101
    assert len(g["8mer_count"]) == 65536
    if read_len is not None and read_len >= 8:
        assert int(g["8mer_count"].sum()) <= (read_len - 7) * prim
    assert g["first_read_unmapped"] >= g["both_reads_unmapped"]
    assert g["total_proper_pairs"] >= g["FF_RR_oriented_pairs"] and g["total_proper_pairs"] >= g["total_proper_pairs_autosome"]
    # a triplet is counted for one read base: at most (L - 2) per primary read
    trip = sum(int(v.sum()) for k, v in g.items() if k.startswith("triplet_counts_"))
    if read_len is not None:
        assert trip <= max(0, read_len - 2) * prim
    return dict(primary=prim, triplets=trip, eightmers=int(g["8mer_count"].sum()))
