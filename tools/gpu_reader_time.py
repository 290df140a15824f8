"""The GPU BAM reader alone (no aggregation): batches per second and its BQC_GB_TIMING lines. usage: python tools/gpu_reader_time.py [reads] [level]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from bamqc_amd import _lib, hostio  # noqa: E402
import ctypes as C  # noqa: E402
from bamqc_amd import _abi  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
with tempfile.TemporaryDirectory(prefix="bqc_gr_") as tmp:
    bam = os.path.join(tmp, "x.bam")
    hostio.synth_stream(bam, None, 1002, reads, ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4, level=level)
    lib = _lib.load()
    for mode in ("gpu", "host"):
        h = C.c_void_p()
        t0 = time.perf_counter()
        rc = lib.bqc_bam_open_gpu(bam.encode(), 0, C.byref(h)) if mode == "gpu" else lib.bqc_bam_open(bam.encode(), C.byref(h))
        assert rc == 0
        n = 0
        while True:
            p = C.POINTER(_abi.Batch)()
            rc = lib.bqc_bam_next(h, 1 << 20, 256 << 20, C.byref(p))
            assert rc >= 0, rc
            if rc == 0:
                break
            n += p.contents.n_reads
        dt = time.perf_counter() - t0
        lib.bqc_bam_close(h)
        print("%s reader: %d reads in %.3f s = %.1f M reads/s (the gpu reader's batches are fetched back to the host here)" % (mode, n, dt, n / dt / 1e6), flush=True)
