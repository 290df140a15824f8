"""How many deflate blocks a BGZF block of a synthetic level-1 file holds (first block's BFINAL bit), and what k_inflate_wave counts for
the same blocks (BQC_GI_STATS=1 prints it at exit).  usage: BQC_GI_STATS=1 python tools/deflate_blocks.py"""
import ctypes as C
import os
import struct
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import _lib, hostio  # noqa: E402


class GiBlock(C.Structure):
    _fields_ = [("coff", C.c_uint64), ("uoff", C.c_uint64), ("csize", C.c_uint32), ("usize", C.c_uint32)]


def main():
    tmp = tempfile.mkdtemp(prefix="bqc_db_")
    bam = os.path.join(tmp, "x.bam")
    hostio.synth_stream(bam, None, 1002, 300_000, ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4, level=1)
    raw = open(bam, "rb").read()
    p, streams, sizes = 0, [], []
    while p < len(raw):
        bs = struct.unpack_from("<H", raw, p + 16)[0] + 1
        d = raw[p + 18:p + bs - 8]
        us = struct.unpack_from("<I", raw, p + bs - 4)[0]
        if us:
            streams.append(d)
            sizes.append(us)
        p += bs
    print("blocks:", len(streams), "first deflate block is not the last in", sum(1 for d in streams if not d[0] & 1), "of them")
    lib = _lib.load()
    lib.bqc_gpu_inflater_create.restype = C.c_void_p
    lib.bqc_gpu_inflater_create.argtypes = [C.c_int]
    lib.bqc_gpu_inflate.restype = C.c_int
    lib.bqc_gpu_inflate.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(GiBlock), C.c_size_t, C.c_void_p, C.c_size_t]
    g = lib.bqc_gpu_inflater_create(0)
    comp = b"".join(streams) + b"\0" * 64
    blocks = (GiBlock * len(streams))()
    co = uo = 0
    for i, s in enumerate(streams):
        blocks[i] = GiBlock(co, uo, len(s), sizes[i])
        co += len(s)
        uo += sizes[i]
    out = np.zeros(uo + 64, np.uint8)
    rc = lib.bqc_gpu_inflate(g, comp, len(comp), blocks, len(streams), out.ctypes.data, uo)
    want = b"".join(zlib.decompress(s, -15) for s in streams)
    print("rc", rc, "identical to zlib:", out[:uo].tobytes() == want)


if __name__ == "__main__":
    main()
