"""GPU: the window state machine of OverallNumbers::coverage (OverallNumbers.hpp:84-110) on the card (csrc/k_anchor.hip, include/bamqc.h:
bqc_anchor_*) against a restatement of the recurrence read by read, and — through bqc_submit_anchored — against the host's pass
(bqc_submit) on the same batches: sorted dense reads, sparse reads (gaps around the 1000 / 2000 thresholds), reads that sit at offset
2000 exactly, unsorted and wild records, several batches in a row (the state is carried on the card)."""
import ctypes as C

import numpy as np
import pytest

from bamqc_amd import Aggregator, BamQCError, _abi, _lib, synth
from tests.hipmem import Hip

pytestmark = pytest.mark.gpu


def reference_anchors(cols, state, n_refs, main):
    """the recurrence itself, one read at a time: returns (win relative to the batch's first window or 0xFFFFFFFF, offset) per read"""
    n = len(cols["flag"])
    win = np.full(n, 0xFFFFFFFF, np.uint64)
    off = np.zeros(n, np.uint32)
    first, sid, shift, w = state
    base = w
    M = 1 << 32
    for i in range(n):
        f, rid = int(cols["flag"][i]), int(cols["rid"][i])
        if (f & 0xD04) or not (f & 0xC0) or not (0 <= rid < n_refs) or not main[rid] or int(cols["lane"][i]) >= 1:
            continue
        b = int(cols["pos"][i]) % M
        if first:
            first, sid, shift = False, rid, b
        if sid != rid or (b - shift) % M > 2000:
            sid, shift, w = rid, b, w + 2
        p = (b - shift) % M
        if 1000 < p < 2000:
            w, shift, p = w + 1, (shift + 1000) % M, p - 1000
        win[i], off[i] = w - base, p
    return win, off, (first, sid, shift, w)


def device_batch(hip, cols):
    """the batch's columns in device memory: (Batch of device pointers, device buffer for the anchors)"""
    hb, keep = _abi.make_batch(cols)  # (the host arrays as bqc_submit would get them: the no-qualities annotation in the flag column included)
    b = _abi.Batch()
    n = len(cols["flag"])
    b.n_reads = n
    for name, dt, pt in _abi._BATCH_COLS:
        setattr(b, name, C.cast(hip.put(keep[name], extra=1024, front=1024), pt))  # (read in place: the kernels' vector loads run over both ends)
    return b, hip.put(np.zeros(2 * max(n, 1), np.uint32))


def anchor(lib, agg, hip, cols):
    b, d_cov = device_batch(hip, cols)
    h = C.c_void_p()
    rc = lib.bqc_anchor_enqueue(agg.h, C.byref(b), d_cov, None, C.byref(h))
    if rc:
        return rc, None, None, None
    assert hip.rt.hipDeviceSynchronize() == 0
    rc = lib.bqc_anchor_complete(agg.h, h, None)
    if rc:
        return rc, None, None, None
    cov = hip.get(d_cov, 8 * len(cols["flag"]), np.uint32).reshape(-1, 2)
    return 0, b, h, cov


def dense(seed, n, lens):
    refs = [synth.reference(seed, i, ln) for i, ln in enumerate(lens)]
    return synth.batch(seed, n, lens, refs), refs


def with_positions(cols, pos, rid=None):
    c = dict(cols)
    n = len(pos)
    for k in ("flag", "mapq", "lane", "rid", "pos", "tlen", "nm", "as_", "l_seq", "n_cigar"):
        c[k] = np.array(cols[k][:n], copy=True)
    c["pos"] = np.asarray(pos, np.int64).astype(np.int32)
    if rid is not None:
        c["rid"] = np.asarray(rid, np.int32)
    L = cols["l_seq"][:n].astype(np.int64)
    c["seq"] = cols["seq"][:int(((L + 1) // 2).sum())]
    c["qual"] = cols["qual"][:int(L.sum())]
    c["cigar"] = cols["cigar"][:int(cols["n_cigar"][:n].sum())]
    return c


CASES = ["dense", "sparse", "thresholds", "stuck", "unsorted", "wild"]


def make_case(kind, rng, base_cols, lens):
    n = 15_000  # (break-heavy cases: below the 16 384 breaks the card's chain takes per batch)
    if kind == "dense":
        return [base_cols]  # (60 000 reads)
    if kind == "sparse":  # gaps around the thresholds: < 1000 (run), 1000 .. 2000 (depends on the state), > 2000 (reset)
        gaps = rng.choice([3, 400, 999, 1000, 1001, 1500, 1999, 2000, 2001, 2600], size=2900)  # (~3.9 Mb: inside the first contig)
        return [with_positions(base_cols, np.cumsum(gaps))]
    if kind == "thresholds":  # every offset around 1000 and 2000 behind a reset
        pos, at = [], 0
        for d in list(range(990, 1012)) + list(range(1990, 2012)) + [0, 1, 999, 1000, 2000, 2000, 2000]:
            at += 5000
            pos += [at, at + d, at + d, at + d + 1, at + d + 999, at + d + 1000, at + d + 1001]
        return [with_positions(base_cols, pos)]
    if kind == "stuck":  # a read at offset 2000 exactly, then reads at the same position, then one further right (in the same and in the next batch)
        pos = [100, 2100, 2100, 2100, 2101, 2500, 3099, 3100, 3101, 5101, 7101, 7101]
        pos2 = [7101, 7101, 7102, 9102, 9102]
        pos3 = [9102, 9102, 9500]
        return [with_positions(base_cols, pos), with_positions(base_cols, pos2), with_positions(base_cols, pos3)]
    if kind == "unsorted":  # positions in any order (the chromosomes in FASTA order: TripletCounting.hpp:254-259 ends the run otherwise)
        pos = rng.integers(0, 3_000_000, size=n)
        rid = np.sort(rng.integers(0, len(lens), size=n))
        return [with_positions(base_cols, pos, rid)]
    from tests.test_gpu_fuzz import wild_batch
    cols, refs = wild_batch(91, 12_000)
    cols = {k: v for k, v in cols.items() if not k.startswith("nm_extra")}
    cols["lane"] = np.zeros(len(cols["flag"]), np.uint8)
    return [cols]


@pytest.mark.parametrize("kind", CASES)
def test_device_anchors_equal_the_recurrence_and_the_host_pass(kind):
    lib = _lib.load()
    rng = np.random.default_rng(7)
    lens = [4_000_000, 3_000_000]
    base_cols, refs = dense(5, 60_000, lens)
    batches = make_case(kind, rng, base_cols, lens)
    if kind == "wild":
        from tests.test_gpu_fuzz import wild_batch
        _, wrefs = wild_batch(91, 12_000)
        refs, lens = wrefs, [len(r) for r in wrefs]
    else:  # (a further batch on the last contig: the state carried on the card)
        tail = np.sort(rng.integers(2_000_000, 2_900_000, size=20_000))
        batches = batches + [with_positions(base_cols, tail, np.full(20_000, len(lens) - 1))]
    n_refs = len(lens)
    hip = Hip()
    try:
        dev = Aggregator(n_refs=n_refs, n_lanes=1, max_read_len=1024)
        host = Aggregator(n_refs=n_refs, n_lanes=1, max_read_len=1024)
        for i, r in enumerate(refs):
            dev.set_reference(i, r)
            host.set_reference(i, r)
        state = (True, 0, 0, 0)
        for cols in batches:
            rc, b, h, cov = anchor(lib, dev, hip, cols)
            assert rc == 0, (lib.bqc_anchor_error(dev.h) or b"").decode()
            win, off, state = reference_anchors(cols, state, n_refs, [1] * n_refs)
            assert np.array_equal(cov[:, 0].astype(np.uint64), win), np.flatnonzero(cov[:, 0].astype(np.uint64) != win)[:10]
            cand = win != 0xFFFFFFFF
            assert np.array_equal(cov[cand, 1], off[cand]), np.flatnonzero(cov[:, 1] != off)[:10]
            ticket = C.c_uint64()
            rc = lib.bqc_submit_anchored(dev.h, C.byref(b), h, C.byref(ticket))
            assert rc == 0 or kind == "wild", (lib.bqc_last_error(dev.h) or b"").decode()
            try:
                dev.sync()  # (the columns' device buffers are released after the test: the batch must be through)
                host.submit(cols)
            except BamQCError:
                assert kind == "wild"
        res = []
        for agg in (dev, host):  # (wild records may end the run: then both must end it the same way)
            try:
                res.append(agg.finalize())
            except BamQCError as e:
                res.append(e.code)
        if isinstance(res[1], int) or isinstance(res[0], int):
            assert res[0] == res[1], res
            assert kind == "wild"
        else:
            diffs = _abi.diff_counts(res[1], res[0])
            assert not diffs, diffs[:10]
        dev.close()
        host.close()
    finally:
        hip.free()


def test_too_many_breaks_leave_the_batch_and_the_stream_to_the_host():
    """More position breaks than the card's chain walks (AN_MAX_BREAKS = 16 384): bqc_anchor_complete says 1, the card's state is
    untouched, and from then on bqc_anchor_enqueue refuses — the batches go through bqc_submit, same result as a host-only stream."""
    lib = _lib.load()
    lens = [200_000_000]
    base_cols, refs = dense(9, 150_000, lens)
    base_cols = with_positions(base_cols, np.sort(np.random.default_rng(3).integers(0, 10_000_000, size=150_000)))  # dense: a break or two
    sparse = with_positions(base_cols, np.arange(150_000, dtype=np.int64) * 1200 + 7)
    hip = Hip()
    try:
        dev = Aggregator(n_refs=1, n_lanes=1, max_read_len=1024)
        host = Aggregator(n_refs=1, n_lanes=1, max_read_len=1024)
        dev.set_reference(0, refs[0])
        host.set_reference(0, refs[0])
        rc, b, h, cov = anchor(lib, dev, hip, base_cols)  # anchored
        assert rc == 0
        assert lib.bqc_submit_anchored(dev.h, C.byref(b), h, None) == 0
        rc, _, _, _ = anchor(lib, dev, hip, sparse)
        assert rc == 1
        dev.submit(sparse)
        rc, _, _, _ = anchor(lib, dev, hip, base_cols)  # the host keeps the state now
        assert rc == 1
        dev.submit(base_cols)
        for cols in (base_cols, sparse, base_cols):
            host.submit(cols)
        diffs = _abi.diff_counts(host.finalize(), dev.finalize())
        assert not diffs, diffs[:10]
        dev.close()
        host.close()
    finally:
        hip.free()
