"""GPU: BASELINE.json configs at (or scaled towards) full size, checked through size-independent
properties plus an oracle comparison on a prefix.  Inputs come from the C++ generator.

config 2: 10 M reads x 150 bp over 4 x 25 Mb contigs (here: the full 10 M reads)
config 5: long reads, 10 kb, indel / soft-clip heavy CIGARs (here: 20 k reads; 5 M x 10 kb = 76 GB does not fit a test)
"""
import numpy as np
import pytest

from bamqc_amd import Aggregator, _abi, synth as csynth
from tests import synth as tsynth
from tests.oracle_lib import Oracle

pytestmark = pytest.mark.gpu


def _check_invariants(c, n_reads_total):
    s = dict(zip(_abi.SCALAR_NAMES, c["scalars"].tolist()))
    prim = s["readcount"]
    assert prim + s["supplementary"] + s["not_primary_alignment"] == n_reads_total
    for m in ("r1.", "r2."):
        n_m = c[m + "qualcount_readnr"]
        rl = c[m + "readLength"]
        assert int(rl.sum()) == n_m and int(c[m + "Ncount"].sum()) == n_m and int(c[m + "GCcount"].sum()) == n_m
        assert int(c[m + "averageQual"].sum()) == n_m
        # every read contributes exactly one base code per cycle it reaches
        reach = n_m - np.concatenate([[0], np.cumsum(rl)])[:c[m + "n_cycles"]]
        tot = sum(c[m + "dnacount%d" % k] for k in range(5))
        assert np.array_equal(tot, reach.astype(np.uint64))
        assert int(c[m + "delhist"].sum()) == int(c[m + "inshist"].sum()) == int(c[m + "mapQ"].sum())
    assert c["r1.qualcount_readnr"] + c["r2.qualcount_readnr"] == prim
    assert int(c["poscov"].sum()) % 1000 == 0
    assert s["totalbps"] == int((np.arange(len(c["r1.readLength"])) * c["r1.readLength"]).sum() + (np.arange(len(c["r2.readLength"])) * c["r2.readLength"]).sum())


def test_config2_10M_reads_properties_and_prefix_parity():
    lens = [25_000_000] * 4
    refs = [csynth.reference(1002, i, n) for i, n in enumerate(lens)]
    n = 10_000_000
    cols = csynth.batch(1002, n, lens, refs)
    agg = Aggregator(n_refs=4)
    for i, r in enumerate(refs):
        agg.set_reference(i, r)
    agg.submit(cols)
    whole = agg.finalize()
    state_whole = agg.state_export_host()
    agg.close()
    _check_invariants(whole[0], n)
    # 8-mer windows: every primary 150-mer has 143 windows minus those blocked by an N
    assert 0.97 * 143 * whole[0]["scalars"][4] < int(whole[0]["eightmer"].sum()) <= 143 * int(whole[0]["scalars"][4])
    # additivity: the same reads submitted as five batches give the identical state vector
    agg2 = Aggregator(n_refs=4)
    for i, r in enumerate(refs):
        agg2.set_reference(i, r)
    for lo in range(0, n, 2_000_000):
        agg2.submit(tsynth.slice_batch(cols, lo, lo + 2_000_000))
    parts = agg2.finalize()
    assert np.array_equal(agg2.state_export_host(), state_whole)
    assert not _abi.diff_counts(whole, parts)
    agg2.close()
    # oracle on a 1 M-read prefix
    sub = tsynth.slice_batch(cols, 0, 1_000_000)
    o = Oracle(n_refs=4)
    for i, r in enumerate(refs):
        o.reference(i, r)
    assert o.process(sub) == 0
    g = Aggregator(n_refs=4)
    for i, r in enumerate(refs):
        g.set_reference(i, r)
    g.submit(sub)
    d = _abi.diff_counts(o.finalize(), g.finalize())
    assert not d, d[:5]
    g.close()


def test_config5_long_reads_parity_and_properties():
    lens = [50_000_000]
    refs = [csynth.reference(1005, 0, lens[0])]
    n = 20_000
    cols = csynth.batch(1005, n, lens, refs, read_len=10_000, isize=30_000, long_reads=True)
    opts = dict(n_refs=1, isize=30_000, max_read_len=16_384, hist_cap=16_384)
    agg = Aggregator(**opts)
    agg.set_reference(0, refs[0])
    agg.submit(cols)
    got = agg.finalize()
    agg.close()
    _check_invariants(got[0], n)
    o = Oracle(**opts)
    o.reference(0, refs[0])
    assert o.process(cols) == 0
    d = _abi.diff_counts(o.finalize(), got)
    assert not d, d[:5]


def test_twelve_read_groups_every_workgroup_meets_more_than_its_rows():
    """The packed 8-mer counters of a workgroup leave through at most BQC_T8_SPW = 8 scratch rows per launch, one per read group it meets
    (round 4: every read group's, with the read group noted in the slot's directory); a workgroup that meets more read groups than it has
    rows adds the rest by atomics (one bin per lane).  2.4 M reads of twelve read groups in ONE batch: every workgroup of k_short walks
    chunks of all twelve.  Bit-exact against the oracle, per read group."""
    lens = [6_000_000, 4_000_000]
    refs = [csynth.reference(77, i, n) for i, n in enumerate(lens)]
    n, n_lanes = 2_400_000, 12
    cols = csynth.batch(77, n, lens, refs, n_lanes=n_lanes)
    assert len(np.unique(cols["lane"])) == n_lanes
    agg = Aggregator(n_refs=2, n_lanes=n_lanes, klist=(), qlist=())
    for i, r in enumerate(refs):
        agg.set_reference(i, r)
    agg.submit(cols)
    got = agg.finalize()
    o = Oracle(n_refs=2, n_lanes=n_lanes, klist=(), qlist=())
    for i, r in enumerate(refs):
        o.reference(i, r)
    assert o.process(cols) == 0
    want = o.finalize()
    d = _abi.diff_counts(want, got)
    assert not d, "\n".join(d[:20])
    assert sum(int(c["eightmer"].sum()) for c in got) > 100 * n
