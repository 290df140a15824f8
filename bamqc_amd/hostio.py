"""Python face of the host-side I/O helpers of include/bamqc_host.h (BAM reader, FASTA loader,
synthetic BAM writer, and the `bamqualcheck` program entry point)."""
import ctypes as C

import numpy as np

from . import _abi, _lib


def batch_to_cols(b):
    """Copy a `bqc_batch` (ctypes) into the numpy column dict used by Aggregator.submit."""
    n = b.n_reads

    def arr(ptr, count, dtype):
        return np.ctypeslib.as_array(ptr, shape=(count,)).copy() if count else np.zeros(0, dtype)

    l = arr(b.l_seq, n, np.uint32)
    ncg = arr(b.n_cigar, n, np.uint16)
    sb = int(((l.astype(np.int64) + 1) // 2).sum())
    qb = int(l.astype(np.int64).sum())
    cw = int(ncg.astype(np.int64).sum())
    cols = dict(flag=arr(b.flag, n, np.uint16), mapq=arr(b.mapq, n, np.uint8), lane=arr(b.lane, n, np.uint8),
                rid=arr(b.rid, n, np.int32), pos=arr(b.pos, n, np.int32), tlen=arr(b.tlen, n, np.int32),
                nm=arr(b.nm, n, np.int32), as_=arr(b.as_, n, np.int32), l_seq=l, n_cigar=ncg,
                seq=arr(b.seq, sb, np.uint8), qual=arr(b.qual, qb, np.uint8), cigar=arr(b.cigar, cw, np.uint32))
    if b.n_nm_extra:
        cols["nm_extra_read"] = arr(b.nm_extra_read, b.n_nm_extra, np.uint32)
        cols["nm_extra_val"] = arr(b.nm_extra_val, b.n_nm_extra, np.int32)
    return cols


class BamFile:
    def __init__(self, path, begin_hint=None, end_hint=None, gpu=None):
        """The whole file, or (begin_hint / end_hint: compressed byte offsets) one shard of its record stream.
        gpu: a device index — the file is inflated and decoded on that GPU (batches are fetched back)."""
        self.lib = _lib.load()
        self.h = C.c_void_p()
        if gpu is not None and (begin_hint is not None or end_hint is not None):
            rc = self.lib.bqc_bam_open_gpu_range(path.encode(), gpu, begin_hint or 0, 2 ** 64 - 1 if end_hint is None else end_hint, C.byref(self.h))
        elif gpu is not None:
            rc = self.lib.bqc_bam_open_gpu(path.encode(), gpu, C.byref(self.h))
        elif begin_hint is None and end_hint is None:
            rc = self.lib.bqc_bam_open(path.encode(), C.byref(self.h))
        else:
            rc = self.lib.bqc_bam_open_range(path.encode(), begin_hint or 0, 2 ** 64 - 1 if end_hint is None else end_hint, C.byref(self.h))
        if rc:
            msg = (self.lib.bqc_bam_error(self.h) or b"").decode()
            self.lib.bqc_bam_close(self.h)
            self.h = None
            raise IOError("bqc_bam_open(%s): %s" % (path, msg))
        n = self.lib.bqc_bam_n_refs(self.h)
        self.ref_names = [self.lib.bqc_bam_ref_name(self.h, i).decode() for i in range(n)]
        self.ref_lens = [int(self.lib.bqc_bam_ref_len(self.h, i)) for i in range(n)]
        self.sample_id = (self.lib.bqc_bam_sample_id(self.h) or b"").decode()
        self.lane_count = int(self.lib.bqc_bam_lane_count(self.h))

    @property
    def range_info(self):
        """(begin block, end block, first, over) of a shard: see bqc_bam_open_range."""
        f = self.lib
        return (int(f.bqc_bam_range_begin_block(self.h)), int(f.bqc_bam_range_end_block(self.h)), int(f.bqc_bam_range_first(self.h)),
                int(f.bqc_bam_range_over(self.h)))

    @property
    def batches_handed_over(self):
        """GPU reader: batches that held a record the card does not decode and went through the host decoder."""
        return int(self.lib.bqc_bam_batches_handed_over(self.h))

    def lanes(self):
        """[(name, index)] in output order (lexicographic by @RG ID)."""
        n = self.lib.bqc_bam_n_lane_names(self.h)
        return [(self.lib.bqc_bam_lane_name(self.h, i).decode(), int(self.lib.bqc_bam_lane_index(self.h, i))) for i in range(n)]

    def set_main_chrom(self, mc):
        a = np.ascontiguousarray(mc, np.uint8)
        assert len(a) == len(self.ref_names)
        self.lib.bqc_bam_set_main_chrom(self.h, a.ctypes.data_as(_abi.u8p))

    def set_rid_filter(self, keep, keep_unplaced):
        a = np.ascontiguousarray(keep, np.uint8)
        assert len(a) == len(self.ref_names)
        self.lib.bqc_bam_set_rid_filter(self.h, a.ctypes.data_as(_abi.u8p), 1 if keep_unplaced else 0)

    def batches(self, max_reads=1 << 20, max_bases=1 << 28):
        while True:
            p = C.POINTER(_abi.Batch)()
            rc = self.lib.bqc_bam_next(self.h, max_reads, max_bases, C.byref(p))
            if rc == 0:
                return
            if rc < 0:
                raise IOError("bam read error %d: %s" % (-rc, (self.lib.bqc_bam_error(self.h) or b"").decode()))
            yield batch_to_cols(p.contents)

    def close(self):
        if self.h:
            self.lib.bqc_bam_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_fasta(path):
    lib = _lib.load()
    n = C.c_uint32()
    names = C.POINTER(C.c_char_p)()
    codes = C.POINTER(_abi.u8p)()
    lens = _abi.u64p()
    rc = lib.bqc_fasta_load(path.encode(), C.byref(n), C.byref(names), C.byref(codes), C.byref(lens))
    if rc:
        raise IOError("could not load fasta %s" % path)
    out = []
    for i in range(n.value):
        ln = int(lens[i])
        out.append((names[i].decode(), np.ctypeslib.as_array(codes[i], shape=(ln,)).copy() if ln else np.zeros(0, np.uint8)))
    lib.bqc_fasta_free(n, names, codes, lens)
    return out


def synth_write(bam_path, fasta_path, seed, n_reads, ref_names, ref_lens, read_len=150, n_lanes=1, isize=1000, long_reads=False):
    lib = _lib.load()
    rl = np.ascontiguousarray(ref_lens, np.uint32)
    p = _abi.SynthParams(seed, 0, n_reads, read_len, len(rl), rl.ctypes.data_as(_abi.u32p), n_lanes, isize, 1 if long_reads else 0)
    names = (C.c_char_p * len(ref_names))(*[s.encode() for s in ref_names])
    rc = lib.bqc_synth_write(C.byref(p), names, bam_path.encode(), fasta_path.encode() if fasta_path else None, 0)
    if rc:
        raise IOError("bqc_synth_write failed: %d" % rc)


def synth_slice(seed, n_total, lo, count, ref_lens, refs=None, read_len=150, n_lanes=1, isize=1000, long_reads=False):
    """Reads [lo, lo + count) of the plan of n_total reads as a column dict (refs: list of Dna5 arrays or None per contig)."""
    lib = _lib.load()
    rl = np.ascontiguousarray(ref_lens, np.uint32)
    p = _abi.SynthParams(seed, 0, n_total, read_len, len(rl), rl.ctypes.data_as(_abi.u32p), n_lanes, isize, 1 if long_reads else 0)
    rp = None
    if refs is not None:
        rp = (_abi.u8p * len(rl))()
        for i, r in enumerate(refs):
            if r is not None:
                rp[i] = r.ctypes.data_as(_abi.u8p)
    out = C.POINTER(_abi.Batch)()
    rc = lib.bqc_synth_slice(C.byref(p), lo, count, rp, C.byref(out))
    if rc:
        raise IOError("bqc_synth_slice failed: %d" % rc)
    cols = batch_to_cols(out.contents)
    lib.bqc_synth_batch_free(out)
    return cols


def write_bam(path, cols, ref_names, ref_lens, n_lanes=1, first_read_index=0, level=1):
    """A column dict as a BAM file (header and tags as the synthetic generator writes them)."""
    lib = _lib.load()
    b, keep = _abi.make_batch(cols)
    rl = np.ascontiguousarray(ref_lens, np.uint32)
    names = (C.c_char_p * len(ref_names))(*[s.encode() for s in ref_names])
    rc = lib.bqc_bam_write(path.encode(), C.byref(b), len(rl), names, rl.ctypes.data_as(_abi.u32p), n_lanes, first_read_index, level)
    if rc:
        raise IOError("bqc_bam_write failed: %d" % rc)


def write_fasta(path, names, codes):
    """Dna5 code arrays as a FASTA file (60 bases per line, like the synthetic generator's)."""
    with open(path, "wb") as f:
        for i, (name, c) in enumerate(zip(names, codes)):
            f.write((">%s synthetic contig %d\n" % (name, i)).encode())
            txt = np.frombuffer(b"ACGTN", np.uint8)[np.asarray(c, np.uint8)]
            full = len(txt) // 60 * 60
            body = np.concatenate([txt[:full].reshape(-1, 60), np.full((full // 60, 1), 10, np.uint8)], axis=1)
            f.write(body.tobytes())
            if full < len(txt):
                f.write(txt[full:].tobytes() + b"\n")


def synth_stream(bam_path, fasta_path, seed, n_reads, ref_names, ref_lens, read_len=150, n_lanes=1, isize=1000, long_reads=False,
                 slice_reads=1 << 21, level=1):
    """The same BAM as synth_write, generated and written slice by slice (any number of reads; bam_path may be a FIFO)."""
    lib = _lib.load()
    rl = np.ascontiguousarray(ref_lens, np.uint32)
    p = _abi.SynthParams(seed, 0, n_reads, read_len, len(rl), rl.ctypes.data_as(_abi.u32p), n_lanes, isize, 1 if long_reads else 0)
    names = (C.c_char_p * len(ref_names))(*[s.encode() for s in ref_names])
    rc = lib.bqc_synth_stream(C.byref(p), names, bam_path.encode(), fasta_path.encode() if fasta_path else None, slice_reads, level)
    if rc:
        raise IOError("bqc_synth_stream failed: %d" % rc)


def main(argv):
    """Run the bamqualcheck program in-process; returns its exit status."""
    lib = _lib.load()
    args = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    return int(lib.bqc_main(len(argv), args))
