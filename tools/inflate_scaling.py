"""How the inflate kernels scale with the number of BGZF blocks in ONE launch (a lane per block: the kernel's time is the time of
one block until the card is full): the blocks of a synthetic BAM, replicated into distinct output regions, launched 10 K .. 160 K
at a time through bqc_gpu_inflate_launch on device-resident operands; HIP events around the launch.
usage: python tools/inflate_scaling.py [reads] [level] [out.json]"""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bamqc_amd import _lib, hostio  # noqa: E402


def bgzf_blocks(raw):
    """(coff, csize, usize, crc) of every BGZF block with data"""
    out = []
    p, n = 0, len(raw)
    while p + 18 <= n:
        xlen = raw[p + 10] | (raw[p + 11] << 8)
        bsize = (raw[p + 16] | (raw[p + 17] << 8)) + 1
        isize = int.from_bytes(raw[p + bsize - 4:p + bsize], "little")
        crc = int.from_bytes(raw[p + bsize - 8:p + bsize - 4], "little")
        if isize:
            out.append((p + 12 + xlen, bsize - 12 - xlen - 8, isize, crc))
        p += bsize
    return out


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    _lib.load()
    fn = C.CDLL(_lib.LIB_PATH).bqc_gpu_inflate_launch
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    with tempfile.TemporaryDirectory(prefix="bqc_gis_") as tmp:
        bam = os.path.join(tmp, "x.bam")
        hostio.synth_stream(bam, None, 1002, reads, ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4, level=level)
        raw = open(bam, "rb").read()
    blocks = bgzf_blocks(raw)
    nb = len(blocks)
    print("file: %.0f MB, %d blocks, %.0f MB inflated" % (len(raw) / 1e6, nb, sum(b[2] for b in blocks) / 1e6), flush=True)
    dev = torch.device("cuda", 0)
    d_comp = torch.from_numpy(np.frombuffer(raw, np.uint8).copy()).to(dev)
    d_comp = torch.cat([d_comp, torch.zeros(256, dtype=torch.uint8, device=dev)])
    rows = []
    for mult in ((1, 2) if os.environ.get('BQC_GI_NO_RESOLVE') else (0.2, 0.4, 0.75, 1, 2, 3)):  # (BQC_GI_LEAN / BQC_GI_LANES in the environment choose the kernel and its workgroup width)
        n = int(nb * mult)
        tab = np.zeros(n, dtype=[("coff", "<u8"), ("uoff", "<u8"), ("csize", "<u4"), ("usize", "<u4")])
        crc = np.zeros(n, np.uint32)
        u = 0
        for i in range(n):
            b = blocks[i % nb]
            tab[i] = (b[0], u, b[1], b[2])
            crc[i] = b[3]
            u += b[2]
        d_tab = torch.from_numpy(tab.view(np.uint8)).to(dev)
        d_crc = torch.from_numpy(crc.view(np.int32)).to(dev)
        d_out = torch.empty(u + 4096, dtype=torch.uint8, device=dev)
        d_st = torch.zeros(16, dtype=torch.int32, device=dev)
        d_tok = torch.empty(u // 32 + n + 64, dtype=torch.int32, device=dev)
        d_ntok = torch.empty(n + 16, dtype=torch.int32, device=dev)
        ms = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(d_comp.data_ptr(), d_tab.data_ptr(), n, u, d_out.data_ptr(), d_crc.data_ptr(), d_st.data_ptr(), d_tok.data_ptr(), d_ntok.data_ptr(), None)
            e1.record()
            torch.cuda.synchronize()
            if it:
                ms.append(e0.elapsed_time(e1))
        assert int(d_st[0].item()) == 0 or os.environ.get("BQC_GI_NO_RESOLVE"), "inflate / crc status %d" % int(d_st[0].item())
        rows.append({"blocks": n, "inflated_MB": u / 1e6, "ms_inflate_plus_crc": float(np.median(ms)), "GB_per_s_out": u / np.median(ms) / 1e6,
                     "blocks_per_ms": n / float(np.median(ms))})
        print(rows[-1], flush=True)
        del d_out, d_tab, d_crc, d_tok, d_ntok
    if len(sys.argv) > 3:
        json.dump({"reads": reads, "level": level, "file_blocks": nb, "rows": rows}, open(sys.argv[3], "w"), indent=1)


main()
