"""GPU: the k-mer sketch (N1) against the oracle (which is pinned to the reference's own kmerstream code)."""
import numpy as np
import pytest

from bamqc_amd import synth as csynth
from tests import synth
from tests.parity import assert_parity, split

pytestmark = pytest.mark.gpu


def _hq(cols, q=40):
    cols = dict(cols)
    cols["qual"] = np.full_like(cols["qual"], q)
    return cols


def test_sketch_default_k32_q17_high_quality_reads():
    lens = [400_000]
    refs = [csynth.reference(5, 0, lens[0])]
    cols = _hq(csynth.batch(5, 30_000, lens, refs))
    co, cg, _, _ = assert_parity(cols, refs, klist=[32], qlist=[17])
    (q, k, total, F0, f1, F2) = cg[0]["sketch"][0]
    assert (q, k) == (17, 32) and total > 2_000_000 and 0 < F0 < total


def test_sketch_multiple_k_and_q_with_clipping():
    lens = [300_000, 100_000]
    refs = [csynth.reference(6, i, n) for i, n in enumerate(lens)]
    cols = csynth.batch(6, 20_000, lens, refs, n_lanes=2)
    co, cg, _, _ = assert_parity(cols, refs, n_lanes=2, klist=[5, 32, 63], qlist=[2, 17, 30])
    assert len(cg[0]["sketch"]) == 9 and cg[0]["sketch"][0][2] > cg[0]["sketch"][8][2]


def test_sketch_iupac_noqual_varlen_and_batches():
    cols, refs = synth.synth(seed=16, n_reads=3000, n_refs=1, ref_len=100_000, var_len=True, p_iupac=0.01, p_noqual=0.02)
    cols["qual"] = np.where(cols["qual"] == 0xFF, 0xFF, np.maximum(cols["qual"], 25)).astype(np.uint8)
    assert_parity(split(cols, [700, 1500]), refs, klist=[8, 32], qlist=[17])


def test_sketch_restarts_inside_chunks_small_k_and_odd_thresholds():
    """The kernel works out a 16-base chunk's restarts as run masks: k-mers no longer than a chunk (k = 1, 2, 15, 16, 17: runs that start
    and end inside one chunk count), many restarts (a fifth of the bases below the threshold, IUPAC codes), reads shorter than k, and a
    threshold outside the byte-wise compare's range (q = 100: (signed char)(133) is negative, every quality passes)."""
    cols, refs = synth.synth(seed=23, n_reads=4000, n_refs=1, ref_len=80_000, var_len=True, p_iupac=0.03, p_noqual=0.01)
    rng = np.random.default_rng(5)
    q = cols["qual"].copy()
    low = (rng.random(q.size) < 0.2) & (q != 0xFF)
    q[low] = rng.integers(0, 17, int(low.sum())).astype(np.uint8)
    cols["qual"] = q
    assert_parity(split(cols, [1300]), refs, klist=[1, 2, 15, 16], qlist=[17])
    assert_parity(cols, refs, klist=[17, 33], qlist=[0, 100])


def test_sketch_empty_input_prints_the_reference_nan_value():
    cols, refs = synth.synth(seed=1, n_reads=0, n_refs=1, ref_len=10_000)
    co, cg, _, _ = assert_parity(cols, refs, klist=[32], qlist=[17])
    assert cg[0]["sketch"][0] == (17, 32, 0, 9223372036854775808, 9223372036854775808, 0)


def test_sketch_state_vector_additive():
    from bamqc_amd import Aggregator, _abi
    from tests.parity import run_oracle
    lens = [200_000, 200_000]
    refs = [csynth.reference(8, i, n) for i, n in enumerate(lens)]
    cols = _hq(csynth.batch(8, 20_000, lens, refs))
    cut = int(np.searchsorted(cols["rid"], 1))
    rc, co, _ = run_oracle([cols], refs, n_refs=2, klist=[31], qlist=[17])
    total = None
    for sh in split(cols, [cut]):
        a = Aggregator(n_refs=2, klist=[31], qlist=[17])
        for i, r in enumerate(refs):
            a.set_reference(i, r)
        a.submit(sh)
        v = a.state_export_host()
        total = v if total is None else total + v
        a.close()
    m = Aggregator(n_refs=2, klist=[31], qlist=[17])
    m.state_import_host(total)
    d = _abi.diff_counts(co, m.finalize())
    assert not d, d[:5]
