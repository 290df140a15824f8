"""Host-side mirror of the aggregation seam of BamQC (the body of the record loop,
reference src/bamqualcheck.cpp:303-453) over the C ABI of include/bamqc.h.

    agg = Aggregator(n_lanes=1, n_refs=4, isize=1000, main_chrom=[1, 1, 1, 1])
    agg.set_reference(rid, dna5_codes)      # Genome (TripletCounting.hpp:60-104)
    agg.submit(cols)                        # one SoA batch of decoded BAM records
    counts = agg.finalize()                 # host mirror of `struct Counts`
    agg.write_bamqc(path, sample_id, lane_names)

All computation happens in hand-written HIP kernels inside libbamqc_gpu.so; this module only
marshals numpy arrays into the ABI structures.
"""
import ctypes as C

import numpy as np

from . import _abi, _lib


class BamQCError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bamqc error %d (%s): %s" % (code, _abi.ERR_NAMES.get(code, "?"), msg))
        self.code = code


class DeviceBatch:
    def __init__(self, agg, handle):
        self.agg, self.h = agg, handle

    @property
    def algorithmic_bytes(self):
        return int(self.agg.lib.bqc_dbatch_bytes(self.h))

    def free(self):
        if self.h:
            self.agg.lib.bqc_dbatch_free(self.agg.h, self.h)
            self.h = None


class Aggregator:
    def __init__(self, **options):
        self.lib = _lib.load()
        self.opt, self._keep = _abi.make_options(**options)
        self.h = C.c_void_p()
        rc = self.lib.bqc_create(C.byref(self.opt), C.byref(self.h))
        if rc:
            raise BamQCError(rc, (self.lib.bqc_last_error(None) or b"").decode())
        self._counts = None

    def _chk(self, rc):
        if rc:
            raise BamQCError(rc, (self.lib.bqc_last_error(self.h) or b"").decode())

    def set_reference(self, rid, dna5):
        a = np.ascontiguousarray(dna5, np.uint8)
        self._chk(self.lib.bqc_set_reference(self.h, rid, a.ctypes.data_as(_abi.u8p), len(a)))

    def submit(self, cols):
        b, keep = _abi.make_batch(cols)
        self._chk(self.lib.bqc_submit(self.h, C.byref(b)))

    def upload(self, cols):
        b, keep = _abi.make_batch(cols)
        h = C.c_void_p()
        self._chk(self.lib.bqc_upload(self.h, C.byref(b), C.byref(h)))
        return DeviceBatch(self, h)

    def process(self, dbatch):
        self._chk(self.lib.bqc_process(self.h, dbatch.h))

    def sync(self):
        self._chk(self.lib.bqc_sync(self.h))

    def reset(self):
        self._chk(self.lib.bqc_reset(self.h))

    def flush(self):
        self._chk(self.lib.bqc_flush(self.h))

    def set_timing(self, on=True):
        self._chk(self.lib.bqc_set_timing(self.h, 1 if on else 0))

    def last_timing(self):
        n = C.c_uint32()
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        self._chk(self.lib.bqc_last_timing(self.h, C.byref(n), C.byref(names), C.byref(ms)))
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}

    @property
    def state_words(self):
        return int(self.lib.bqc_state_words(self.h))

    def state_export_host(self):
        a = np.zeros(self.state_words, np.uint64)
        self._chk(self.lib.bqc_state_export_host(self.h, a.ctypes.data_as(_abi.u64p)))
        return a

    def state_import_host(self, a):
        a = np.ascontiguousarray(a, np.uint64)
        assert len(a) == self.state_words
        self._chk(self.lib.bqc_state_import_host(self.h, a.ctypes.data_as(_abi.u64p)))

    def state_export_device(self, data_ptr):
        self._chk(self.lib.bqc_state_export(self.h, C.c_void_p(data_ptr)))

    def state_import_device(self, data_ptr):
        self._chk(self.lib.bqc_state_import(self.h, C.c_void_p(data_ptr)))

    def finalize_raw(self):
        p = C.POINTER(_abi.Counts)()
        self._chk(self.lib.bqc_finalize(self.h, C.byref(p)))
        self._counts = p
        return p

    def finalize(self):
        return _abi.counts_to_dict(self.finalize_raw())

    def write_bamqc(self, path, sample_id="", lane_names=("",), lane_index=None):
        if self._counts is None:
            self.finalize_raw()
        hdr = _abi.HeaderInfo()
        names = (C.c_char_p * len(lane_names))(*[n.encode() for n in lane_names])
        idx = np.ascontiguousarray(lane_index if lane_index is not None else np.arange(len(lane_names)), np.uint32)
        hdr.sample_id = sample_id.encode()
        hdr.n_names = len(lane_names)
        hdr.lane_names = names
        hdr.lane_index = idx.ctypes.data_as(_abi.u32p)
        self._chk(self.lib.bqc_write_bamqc(self._counts, C.byref(hdr), path.encode()))

    def close(self):
        if self.h:
            self.lib.bqc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
