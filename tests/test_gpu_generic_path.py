"""GPU: the generic per-base kernel (any read length) stays covered now that short reads take the
fast path: BQC_NO_FAST=1 routes every read through k_reads + k_bases."""
import numpy as np
import pytest

from tests import synth
from tests.parity import assert_parity, split

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def no_fast(monkeypatch):
    monkeypatch.setenv("BQC_NO_FAST", "1")


@pytest.mark.parametrize("seed", [1, 2])
def test_generic_synth_150bp(seed):
    cols, refs = synth.synth(seed=seed, n_reads=6000, n_refs=2, ref_len=120_000)
    assert_parity(cols, refs)


def test_generic_multi_lane_varlen_iupac():
    cols, refs = synth.synth(seed=6, n_reads=4000, n_refs=1, ref_len=100_000, var_len=True, p_iupac=0.01, p_noqual=0.01,
                             hardclip=True, n_lanes=3)
    assert_parity(split(cols, [1500]), refs, n_lanes=3)


def test_mixed_fast_and_generic_reads(monkeypatch):
    # without the switch: reads longer than 256 bases go generic, the rest fast, in one batch
    monkeypatch.delenv("BQC_NO_FAST")
    a, refs = synth.synth(seed=7, n_reads=300, L=700, n_refs=1, ref_len=300_000, long_cigar=True)
    b, _ = synth.synth(seed=8, n_reads=3000, n_refs=1, ref_len=300_000, refs=refs)
    cols = synth.concat([synth.slice_batch(b, 0, 1500), a, synth.slice_batch(b, 1500, 3000)])
    assert_parity(cols, refs, max_read_len=1024, isize=2000)
