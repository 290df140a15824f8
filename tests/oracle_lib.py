"""ctypes loader for oracle/liboracle.so (TEST INFRASTRUCTURE: the CPU restatement).

Builds the library with `make -C oracle liboracle.so` when it is missing (gcc only,
no GPU needed).  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from bamqc_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # BQC_ORACLE_ASAN=1 (tools/asan_oracle.sh): the address / UB-sanitized build; needs LD_PRELOAD of libasan
    asan = os.environ.get("BQC_ORACLE_ASAN") == "1"
    path = os.path.join(ROOT, "oracle", "liboracle_asan.so" if asan else "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("bamqc_oracle.c", "sketch_oracle.c", "bamqc_oracle.h")]
    srcs.append(os.path.join(ROOT, "include", "bamqc.h"))
    if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), os.path.basename(path)], stdout=subprocess.DEVNULL)
    L = C.CDLL(path)
    L.orc_create.argtypes = [C.POINTER(_abi.Options), C.POINTER(C.c_void_p)]
    L.orc_set_reference.argtypes = [C.c_void_p, C.c_int32, _abi.u8p, C.c_uint64]
    L.orc_process_batch.argtypes = [C.c_void_p, C.POINTER(_abi.Batch)]
    L.orc_finalize.argtypes = [C.c_void_p, C.POINTER(C.POINTER(_abi.Counts))]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_destroy.restype = None
    L.orc_write_bamqc.argtypes = [C.POINTER(_abi.Counts), C.POINTER(_abi.HeaderInfo), C.c_char_p]
    L.orc_rephash_table.argtypes = [C.c_int, _abi.u64p]
    L.orc_rephash_table.restype = None
    L.orc_rephash_sequence.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_uint32, _abi.u64p]
    L.orc_rephash_sequence.restype = C.c_uint32
    L.orc_streamcounter_run.argtypes = [C.c_double, _abi.u64p, C.c_uint64, _abi.u64p]
    L.orc_streamcounter_run.restype = None
    _LIB = L
    return L


class Oracle:
    """Thin object wrapper: Oracle(**options).reference(...).process(cols) -> counts dict."""

    def __init__(self, **kw):
        self.L = lib()
        self.opt, self._keep = _abi.make_options(**kw)
        self.h = C.c_void_p()
        rc = self.L.orc_create(C.byref(self.opt), C.byref(self.h))
        assert rc == 0, "orc_create failed: %d" % rc
        self.counts_ptr = None

    def reference(self, rid, dna5):
        a = np.ascontiguousarray(dna5, np.uint8)
        rc = self.L.orc_set_reference(self.h, rid, a.ctypes.data_as(_abi.u8p), len(a))
        assert rc == 0
        return self

    def process(self, cols):
        """Returns the BQC_ERR_* code (0 = ok)."""
        b, keep = _abi.make_batch(cols)
        return self.L.orc_process_batch(self.h, C.byref(b))

    def finalize(self):
        p = C.POINTER(_abi.Counts)()
        rc = self.L.orc_finalize(self.h, C.byref(p))
        assert rc == 0
        self.counts_ptr = p
        return _abi.counts_to_dict(p)

    def write_bamqc(self, path, sample_id="SYN", lane_names=("L1",), lane_index=None):
        if self.counts_ptr is None:
            self.finalize()
        hdr, keep = make_header(sample_id, lane_names, lane_index)
        rc = self.L.orc_write_bamqc(self.counts_ptr, C.byref(hdr), path.encode())
        assert rc == 0

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_header(sample_id, lane_names, lane_index=None):
    hdr = _abi.HeaderInfo()
    names = (C.c_char_p * len(lane_names))(*[n.encode() for n in lane_names])
    idx = np.ascontiguousarray(lane_index if lane_index is not None else np.arange(len(lane_names)), np.uint32)
    sid = sample_id.encode()
    hdr.sample_id = sid
    hdr.n_names = len(lane_names)
    hdr.lane_names = names
    hdr.lane_index = idx.ctypes.data_as(_abi.u32p)
    return hdr, (names, idx, sid)
