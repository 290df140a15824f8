"""ctypes mirror of include/bamqc.h (the C ABI of the aggregation path).

Only structure layouts and small marshalling helpers live here; the library
itself is loaded in `bamqc_amd._lib`.  The test suite's CPU checker accepts the same
structures, so tests marshal one batch and hand it to both sides.
"""
import ctypes as C

import numpy as np

BQC_N_SCALARS = 13
BQC_COVSIZE = 100
BQC_N_8MER = 65536
BQC_N_TRIPLET = 64 * 4 * 4
BQC_FLAG_MATE_MAIN = 0x1000
BQC_FLAG_NO_QUAL = 0x8000
BQC_NM_ABSENT = -1
BQC_AS_ABSENT = -(2 ** 31)

SCALAR_NAMES = [
    "supplementary", "duplicates", "QCfailed", "not_primary_alignment", "readcount", "totalbps",
    "bothunmapped", "firstunmapped", "secondunmapped", "first_and_or_second_mapped",
    "FF_RR_orientation", "properpair_count", "auto_properpair_count",
]

ERR_NAMES = {0: "OK", 1: "ARG", 2: "DEVICE", 3: "NO_MATE_FLAG", 4: "AS_TAG", 5: "FASTA", 6: "RANGE", 7: "IO", 8: "STATE"}

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)


class SketchOptions(C.Structure):
    _fields_ = [("n_k", C.c_uint32), ("klist", i32p), ("n_q", C.c_uint32), ("qlist", u32p),
                ("e", C.c_double), ("seed", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_lanes", C.c_uint32), ("n_refs", C.c_uint32),
                ("isize", C.c_int32), ("max_read_len", C.c_uint32), ("hist_cap", C.c_uint32),
                ("main_chrom", u8p), ("fasta_index", i32p), ("device", C.c_int32),
                ("sketch", SketchOptions), ("shard_tail", C.c_uint32)]


class Batch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32),
                ("flag", u16p), ("mapq", u8p), ("lane", u8p), ("rid", i32p), ("pos", i32p),
                ("tlen", i32p), ("nm", i32p), ("as_", i32p), ("l_seq", u32p), ("n_cigar", u16p),
                ("seq", u8p), ("qual", u8p), ("cigar", u32p),
                ("n_nm_extra", C.c_uint32), ("nm_extra_read", u32p), ("nm_extra_val", i32p)]


class MateCounts(C.Structure):
    _fields_ = [("n_cycles", C.c_uint32), ("dnacount", u64p * 5), ("qualcount", u64p),
                ("qualcount_readnr", C.c_uint64), ("sc5", u64p), ("sc3", u64p),
                ("n_Ncount", C.c_uint32), ("Ncount", u64p),
                ("n_GCcount", C.c_uint32), ("GCcount", u64p),
                ("n_averageQual", C.c_uint32), ("averageQual", u64p),
                ("n_insertSize", C.c_uint32), ("insertSize", u64p),
                ("n_mapQ", C.c_uint32), ("mapQ", u64p),
                ("n_readLength", C.c_uint32), ("readLength", u64p),
                ("n_mismatch", C.c_uint32), ("mismatch", u64p),
                ("n_delhist", C.c_uint32), ("delhist", u64p),
                ("n_inshist", C.c_uint32), ("inshist", u64p)]


class SketchCounts(C.Structure):
    _fields_ = [("q", C.c_uint32), ("k", C.c_uint32), ("sumCount", C.c_uint64), ("F0", C.c_uint64),
                ("f1", C.c_uint64), ("F2", C.c_uint64)]


class LaneCounts(C.Structure):
    _fields_ = [("scalars", C.c_uint64 * BQC_N_SCALARS), ("poscov", C.c_uint64 * (BQC_COVSIZE + 1)),
                ("eightmer", u64p), ("mate", MateCounts * 2), ("triplet", u64p),
                ("n_sketch", C.c_uint32), ("sketch", C.POINTER(SketchCounts))]


class Counts(C.Structure):
    _fields_ = [("n_lanes", C.c_uint32), ("lanes", C.POINTER(LaneCounts))]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_read_index", C.c_uint64), ("n_reads", C.c_uint32),
                ("read_len", C.c_uint32), ("n_refs", C.c_uint32), ("ref_len", u32p), ("n_lanes", C.c_uint32),
                ("isize", C.c_int32), ("long_reads", C.c_int32)]


class ShardInfo(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("status", C.c_int32), ("begin_block", C.c_uint64), ("end_block", C.c_uint64),
                ("first", C.c_uint64), ("over", C.c_uint64), ("sample_id", C.c_char_p), ("n_lane_names", C.c_uint32),
                ("lane_names", C.POINTER(C.c_char_p)), ("lane_index", u32p)]


class ShardResult(C.Structure):
    _fields_ = [("n_lane_names", C.c_uint32), ("lane_names", C.POINTER(C.c_char_p)), ("lane_index", u32p)]


SHARD_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ShardInfo), C.POINTER(ShardResult))
SHARD_FAIL, SHARD_DONE, SHARD_FALLBACK, SHARD_WRITE = 0, 1, 2, 3  # (0 is what ctypes returns for a callback that raised)


class HeaderInfo(C.Structure):
    _fields_ = [("sample_id", C.c_char_p), ("n_names", C.c_uint32),
                ("lane_names", C.POINTER(C.c_char_p)), ("lane_index", u32p)]


_BATCH_COLS = [("flag", np.uint16, u16p), ("mapq", np.uint8, u8p), ("lane", np.uint8, u8p),
               ("rid", np.int32, i32p), ("pos", np.int32, i32p), ("tlen", np.int32, i32p),
               ("nm", np.int32, i32p), ("as_", np.int32, i32p), ("l_seq", np.uint32, u32p),
               ("n_cigar", np.uint16, u16p), ("seq", np.uint8, u8p), ("qual", np.uint8, u8p),
               ("cigar", np.uint32, u32p)]


def _ptr(a, ptype):
    return a.ctypes.data_as(ptype)


def make_batch(cols):
    """Build a `Batch` from a dict of numpy arrays (keys as in `_BATCH_COLS`, plus optional
    `nm_extra_read`/`nm_extra_val`).  Returns (batch, keepalive)."""
    n = len(cols["flag"])
    keep = {}
    b = Batch()
    b.n_reads = n
    if n:  # BQC_FLAG_NO_QUAL is the decoder's to set (include/bamqc.h): column dicts built by hand get it here
        l = np.asarray(cols["l_seq"], np.int64)
        qo = np.cumsum(l) - l
        q = np.asarray(cols["qual"], np.uint8)
        noq = (l > 0) & (q[np.minimum(qo, max(len(q) - 1, 0))] == 0xFF) if len(q) else np.zeros(n, bool)
        if noq.any():
            cols = dict(cols, flag=np.asarray(cols["flag"], np.uint16) | (noq.astype(np.uint16) << 15))
    for name, dt, pt in _BATCH_COLS:
        a = np.ascontiguousarray(cols[name], dtype=dt)
        if a.size == 0:
            a = np.zeros(1, dtype=dt)
        keep[name] = a
        setattr(b, name, _ptr(a, pt))
    xr = np.ascontiguousarray(cols.get("nm_extra_read", np.zeros(0)), dtype=np.uint32)
    xv = np.ascontiguousarray(cols.get("nm_extra_val", np.zeros(0)), dtype=np.int32)
    b.n_nm_extra = len(xr)
    if len(xr) == 0:
        xr = np.zeros(1, np.uint32)
        xv = np.zeros(1, np.int32)
    keep["xr"], keep["xv"] = xr, xv
    b.nm_extra_read = _ptr(xr, u32p)
    b.nm_extra_val = _ptr(xv, i32p)
    # consistency of the packed layout
    l = keep["l_seq"][:n].astype(np.int64)
    assert len(cols["seq"]) == int(((l + 1) // 2).sum()), "seq bytes != sum((l_seq+1)/2)"
    assert len(cols["qual"]) == int(l.sum()), "qual bytes != sum(l_seq)"
    assert len(cols["cigar"]) == int(keep["n_cigar"][:n].astype(np.int64).sum()), "cigar words != sum(n_cigar)"
    return b, keep


def make_options(n_lanes=1, n_refs=1, isize=1000, max_read_len=1024, hist_cap=4096, main_chrom=None,
                 fasta_index=None, device=0, klist=(), qlist=(), e=0.01, seed=1, shard_tail=0):
    keep = {}
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.n_lanes, o.n_refs, o.isize = n_lanes, n_refs, isize
    o.max_read_len, o.hist_cap, o.device = max_read_len, hist_cap, device
    mc = np.ones(max(n_refs, 1), np.uint8) if main_chrom is None else np.ascontiguousarray(main_chrom, np.uint8)
    keep["mc"] = mc
    o.main_chrom = _ptr(mc, u8p)
    if fasta_index is not None:
        fi = np.ascontiguousarray(fasta_index, np.int32)
        keep["fi"] = fi
        o.fasta_index = _ptr(fi, i32p)
    kl = np.ascontiguousarray(list(klist), np.int32)
    ql = np.ascontiguousarray(list(qlist), np.uint32)
    keep["kl"], keep["ql"] = kl, ql
    o.sketch.n_k, o.sketch.n_q = len(kl), len(ql)
    if len(kl):
        o.sketch.klist = _ptr(kl, i32p)
    if len(ql):
        o.sketch.qlist = _ptr(ql, u32p)
    o.sketch.e, o.sketch.seed = e, seed
    o.shard_tail = 1 if shard_tail else 0
    return o, keep


def _arr(p, n):
    if n == 0 or not p:
        return np.zeros(0, np.uint64)
    return np.ctypeslib.as_array(p, shape=(n,)).copy()


def counts_to_dict(cptr):
    """Deep-copy a `bqc_counts*` into {lane: {name: value/ndarray}} for comparison."""
    c = cptr.contents if hasattr(cptr, "contents") else cptr
    out = []
    for li in range(c.n_lanes):
        L = c.lanes[li]
        d = {"scalars": np.array(list(L.scalars), dtype=np.uint64),
             "poscov": np.array(list(L.poscov), dtype=np.uint64),
             "eightmer": _arr(L.eightmer, BQC_N_8MER),
             "triplet": _arr(L.triplet, BQC_N_TRIPLET)}
        for mi in range(2):
            m = L.mate[mi]
            p = "r%d." % (mi + 1)
            d[p + "n_cycles"] = int(m.n_cycles)
            for j in range(5):
                d[p + "dnacount%d" % j] = _arr(m.dnacount[j], m.n_cycles)
            d[p + "qualcount"] = _arr(m.qualcount, m.n_cycles)
            d[p + "qualcount_readnr"] = int(m.qualcount_readnr)
            d[p + "sc5"] = _arr(m.sc5, m.n_cycles)
            d[p + "sc3"] = _arr(m.sc3, m.n_cycles)
            for name in ("Ncount", "GCcount", "averageQual", "insertSize", "mapQ", "readLength",
                         "mismatch", "delhist", "inshist"):
                d[p + name] = _arr(getattr(m, name), getattr(m, "n_" + name))
        d["sketch"] = [(int(L.sketch[i].q), int(L.sketch[i].k), int(L.sketch[i].sumCount), int(L.sketch[i].F0),
                        int(L.sketch[i].f1), int(L.sketch[i].F2)) for i in range(L.n_sketch)]
        out.append(d)
    return out


def diff_counts(a, b):
    """Return a list of human-readable differences between two counts_to_dict results."""
    diffs = []
    if len(a) != len(b):
        return ["lane count %d != %d" % (len(a), len(b))]
    for li, (x, y) in enumerate(zip(a, b)):
        for k in x:
            u, v = x[k], y[k]
            if isinstance(u, np.ndarray):
                if u.shape != v.shape:
                    diffs.append("lane %d %s: length %d != %d" % (li, k, len(u), len(v)))
                elif not np.array_equal(u, v):
                    idx = np.nonzero(u != v)[0]
                    diffs.append("lane %d %s: %d bins differ, first at %d: %d != %d" % (
                        li, k, len(idx), idx[0], int(u[idx[0]]), int(v[idx[0]])))
            elif u != v:
                diffs.append("lane %d %s: %r != %r" % (li, k, u, v))
    return diffs
