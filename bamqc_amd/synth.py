"""Python face of the C++ synthetic-input generator (include/bamqc_host.h)."""
import ctypes as C

import numpy as np

from . import _abi, _lib

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
          58617616, 64444167, 46709983, 50818468, 156040895, 57227415]  # chr1..22, X, Y


def reference(seed, rid, length):
    lib = _lib.load()
    out = np.empty(length, np.uint8)
    rc = lib.bqc_synth_reference(seed, rid, length, out.ctypes.data_as(_abi.u8p))
    assert rc == 0
    return out


def batch(seed, n_reads, ref_lens, refs=None, read_len=150, n_lanes=1, isize=1000, long_reads=False, first_read_index=0):
    """Returns the column dict accepted by `Aggregator.submit` (numpy copies)."""
    lib = _lib.load()
    rl = np.ascontiguousarray(ref_lens, np.uint32)
    p = _abi.SynthParams(seed, first_read_index, n_reads, read_len, len(rl), rl.ctypes.data_as(_abi.u32p), n_lanes, isize,
                         1 if long_reads else 0)
    rp = None
    if refs is not None:
        keep = [np.ascontiguousarray(r, np.uint8) for r in refs]
        rp = (_abi.u8p * len(keep))(*[k.ctypes.data_as(_abi.u8p) for k in keep])
    out = C.POINTER(_abi.Batch)()
    rc = lib.bqc_synth_batch(C.byref(p), rp, C.byref(out))
    assert rc == 0, rc
    b = out.contents
    n = b.n_reads
    l = np.ctypeslib.as_array(b.l_seq, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
    ncg = np.ctypeslib.as_array(b.n_cigar, shape=(n,)).copy() if n else np.zeros(0, np.uint16)
    sb, qb, cw = int(((l.astype(np.int64) + 1) // 2).sum()), int(l.astype(np.int64).sum()), int(ncg.astype(np.int64).sum())

    def arr(ptr, count):
        return np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count else np.zeros(0, ptr._type_)

    cols = dict(flag=arr(b.flag, n), mapq=arr(b.mapq, n), lane=arr(b.lane, n), rid=arr(b.rid, n), pos=arr(b.pos, n),
                tlen=arr(b.tlen, n), nm=arr(b.nm, n), as_=arr(b.as_, n), l_seq=l, n_cigar=ncg,
                seq=arr(b.seq, sb), qual=arr(b.qual, qb), cigar=arr(b.cigar, cw))
    lib.bqc_synth_batch_free(out)
    return cols
