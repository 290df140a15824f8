"""GPU: raw DEFLATE of BGZF blocks on the card (csrc/gpu_inflate.hip) against zlib — stored, fixed-code and dynamic-code blocks,
long codes, every match distance class, empty and maximal blocks, corrupt streams — and through the BAM reader."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

from bamqc_amd import _lib, hostio

pytestmark = pytest.mark.gpu


class GiBlock(C.Structure):
    _fields_ = [("coff", C.c_uint64), ("uoff", C.c_uint64), ("csize", C.c_uint32), ("usize", C.c_uint32)]


@pytest.fixture(autouse=True, params=["default", "lanes", "lean32", "lean64", "one_phase", "one_phase_lean32"])
def kernel_variant(request, monkeypatch):
    """Every test of this module runs through each inflate kernel: a wave per block (k_inflate_wave: the default) and a lane per
    block — with root tables in LDS and the lean one at two workgroup widths — all with the matches filled in by the second-phase
    kernel (k_inflate_resolve), and the lane-per-block kernels also with the lanes copying their matches themselves (one phase)."""
    env = {"default": {}, "lanes": {"BQC_GI_WAVE": "0"}, "lean32": {"BQC_GI_LEAN": "32"}, "lean64": {"BQC_GI_LEAN": "64"}, "one_phase": {"BQC_GI_TWO_PHASE": "0", "BQC_GI_LEAN": "0"},
           "one_phase_lean32": {"BQC_GI_TWO_PHASE": "0", "BQC_GI_LEAN": "32"}}[request.param]
    for k in ("BQC_GI_LEAN", "BQC_GI_TWO_PHASE", "BQC_GI_WAVE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    yield request.param


@pytest.fixture(scope="module")
def gi():
    lib = _lib.load()
    lib.bqc_gpu_inflater_create.restype = C.c_void_p
    lib.bqc_gpu_inflater_create.argtypes = [C.c_int]
    lib.bqc_gpu_inflater_destroy.argtypes = [C.c_void_p]
    lib.bqc_gpu_inflate.restype = C.c_int
    lib.bqc_gpu_inflate.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(GiBlock), C.c_size_t, C.c_void_p, C.c_size_t]
    g = lib.bqc_gpu_inflater_create(0)
    assert g
    yield lib, g
    lib.bqc_gpu_inflater_destroy(g)


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    return c.compress(data) + c.flush()


def run(gi, streams, sizes=None):
    lib, g = gi
    comp = b"".join(streams) + b"\0" * 16
    n = len(streams)
    blocks = (GiBlock * n)()
    co = uo = 0
    for i, s in enumerate(streams):
        us = sizes[i] if sizes else len(zlib.decompress(s, -15))
        blocks[i] = GiBlock(co, uo, len(s), us)
        co += len(s)
        uo += us
    out = np.zeros(uo + 64, np.uint8)
    rc = lib.bqc_gpu_inflate(g, comp, len(comp), blocks, n, out.ctypes.data, uo)
    return rc, out[:uo].tobytes()


def payloads():
    rng = np.random.default_rng(5)
    p = []
    p.append(b"")                                                       # empty block (the BGZF EOF marker's payload)
    p.append(b"a")
    p.append(b"abc" * 21000)                                            # distance 3, long matches
    p.append(b"\x07" * 65536)                                           # distance 1, maximal block
    p.append(bytes(rng.integers(0, 256, 65536, dtype=np.uint8)))        # incompressible
    p.append(bytes(rng.integers(0, 4, 60000, dtype=np.uint8)))          # short codes
    p.append(bytes((rng.geometric(0.02, 65000) % 256).astype(np.uint8)))  # skewed: long codes for the rare symbols
    txt = b"".join(b"read%07d\tACGT%s\t%d\n" % (i, b"ACGTTGCA"[i % 5:], i * 37) for i in range(2600))
    p.append(txt[:65000])                                               # text with matches at every distance
    a = bytes(rng.integers(0, 256, 30000, dtype=np.uint8))
    p.append(a + a)                                                     # distance 30000
    p.append(a[:17] * 1000)                                             # distance 17
    p.append(a[:5] * 3000)                                              # distance 5
    return p


def test_against_zlib_all_block_kinds(gi):
    streams, want = [], []
    for data in payloads():
        for level, strategy in [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (1, zlib.Z_FILTERED)]:
            s = raw_deflate(data, level, strategy)
            if len(s) > 65536 + 64:
                continue
            streams.append(s)
            want.append(data)
    # several deflate blocks inside one stream (Z_FULL_FLUSH between the parts: stored / dynamic / fixed mixed)
    rng = np.random.default_rng(9)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [bytes(rng.integers(0, 7, 9000, dtype=np.uint8)), b"xyz" * 2000, bytes(rng.integers(0, 256, 3000, dtype=np.uint8))]
    s = b"".join(c.compress(x) + c.flush(zlib.Z_FULL_FLUSH) for x in parts) + c.flush()
    streams.append(s)
    want.append(b"".join(parts))
    rc, got = run(gi, streams)
    assert rc == 0
    o = 0
    for i, w in enumerate(want):
        assert got[o:o + len(w)] == w, "stream %d differs" % i
        o += len(w)


def test_many_blocks_fill_the_card(gi):
    rng = np.random.default_rng(11)
    base = [bytes(rng.integers(0, 1 + 3 * (k % 80), 20000 + 500 * (k % 90), dtype=np.uint8)) for k in range(200)]
    streams = [raw_deflate(b, 1 + k % 9) for k, b in enumerate(base)] * 40       # 8000 blocks
    rc, got = run(gi, streams, [len(b) for b in base] * 40)
    assert rc == 0
    assert got == b"".join(base) * 40


def test_many_deflate_blocks_in_a_stream_and_long_codes(gi):
    """What the wave-per-block kernel cuts differently from a serial decoder: streams of twenty deflate blocks (the wave starts over with
    a header and new tables behind every end-of-block symbol, wherever in a piece it lies), blocks far shorter than a piece, codes
    longer than the root tables for literals (300 distinct symbols with a steep distribution) and for distances (matches at every
    distance class with few repeats), and a block that is one long run of literals."""
    rng = np.random.default_rng(23)
    streams, want = [], []
    for seed in range(6):
        r = np.random.default_rng(100 + seed)
        c = zlib.compressobj(1 + seed, zlib.DEFLATED, -15)
        parts, s = [], b""
        for k in range(20):
            kind = (k + seed) % 4
            if kind == 0:
                x = bytes(r.integers(0, 256, int(r.integers(1, 2500)), dtype=np.uint8))            # mostly stored or near-incompressible
            elif kind == 1:
                x = bytes((r.geometric(0.05, int(r.integers(10, 4000))) % 256).astype(np.uint8))    # long codes for the rare symbols
            elif kind == 2:
                x = (b"pattern%03d/" % k) * int(r.integers(1, 300))
            else:
                x = bytes(r.integers(0, 3, int(r.integers(1, 60)), dtype=np.uint8))                 # a block of a few symbols
            parts.append(x)
            s += c.compress(x) + c.flush(zlib.Z_FULL_FLUSH if k % 3 else zlib.Z_SYNC_FLUSH)
        s += c.flush()
        data = b"".join(parts)
        assert len(data) <= 65536 and len(s) <= 65536
        streams.append(s)
        want.append(data)
    # distances of every class, each used once or twice: many distance codes of 9 bits and more
    base = bytes(rng.integers(0, 256, 40000, dtype=np.uint8))
    far = bytearray(base)
    for k in range(400):
        d = int(rng.integers(1, 32000)); at = int(rng.integers(32100, 39000)); n = int(rng.integers(3, 40))
        far[at:at + n] = far[at - d:at - d + n]
    streams.append(raw_deflate(bytes(far), 9)); want.append(bytes(far))
    lit = bytes((rng.integers(0, 64, 60000, dtype=np.uint8) + 32).astype(np.uint8))                   # literals only, 6 bits each
    streams.append(raw_deflate(lit, 6, zlib.Z_HUFFMAN_ONLY)); want.append(lit)
    rc, got = run(gi, streams)
    assert rc == 0
    o = 0
    for i, w in enumerate(want):
        assert got[o:o + len(w)] == w, "stream %d differs" % i
        o += len(w)


def test_blocks_of_real_files(gi):
    """Not the synthetic generator's statistics: source text, bytecode, ELF sections and numpy arrays found on the machine, in BGZF-sized
    pieces at three compression levels (other code-length distributions, long codes, few and many deflate blocks per piece)."""
    import glob
    import sysconfig
    paths = sorted(glob.glob(os.path.join(sysconfig.get_paths()["stdlib"], "*.py")))[:40]
    paths += sorted(glob.glob(os.path.join(sysconfig.get_paths()["stdlib"], "__pycache__", "*.pyc")))[:20]
    import sys
    paths += [os.path.realpath(sys.executable)]  # (an ELF file)
    blob = b"".join(open(q, "rb").read()[:1_500_000] for q in paths if os.path.isfile(q))
    blob += np.arange(200_000, dtype=np.int32).tobytes() + np.linspace(0, 1, 100_000).tobytes()
    assert len(blob) > 2_000_000
    streams, want = [], []
    for k, at in enumerate(range(0, min(len(blob), 24_000_000) - 65280, 65280)):
        piece = blob[at:at + 65280]
        s = raw_deflate(piece, (1, 6, 9)[k % 3])
        if len(s) > 65536:
            continue
        streams.append(s)
        want.append(piece)
    assert len(streams) > 30
    rc, got = run(gi, streams, [len(w) for w in want])
    assert rc == 0
    assert got == b"".join(want)


def test_a_distance_before_the_block_in_the_middle_of_a_stream(gi):
    """A match that reaches in front of the block's first byte, met far inside the stream (a piece in the middle of the wave: found in
    the write pass): a stream compressed against a preset dictionary, inflated without it."""
    rng = np.random.default_rng(29)
    zdict = bytes(rng.integers(0, 256, 20000, dtype=np.uint8))
    data = bytes(rng.integers(0, 256, 6000, dtype=np.uint8)) + zdict[12000:16000] + bytes(rng.integers(0, 256, 30000, dtype=np.uint8))  # (the copy lies 14 000 bytes back, 6 000 are there)
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, zdict)
    bad = c.compress(data) + c.flush()
    with pytest.raises(zlib.error):
        zlib.decompress(bad, -15)
    good = raw_deflate(data)
    rc, _ = run(gi, [good] * 5 + [bad] + [good] * 5, [len(data)] * 11)
    assert rc > 0


@pytest.mark.parametrize("kind", ["truncated", "flipped", "wrong_size_short", "wrong_size_long", "bad_type", "bad_stored_len", "far_distance"])
def test_corrupt_streams_are_reported(gi, kind):
    rng = np.random.default_rng(3)
    data = bytes(rng.integers(0, 9, 30000, dtype=np.uint8))
    good = raw_deflate(data)
    ok = [good] * 70
    size = len(data)
    if kind == "truncated":
        bad = good[:len(good) // 2]
    elif kind == "flipped":
        bad = bytearray(good)
        for k in range(40, len(bad), 97):
            bad[k] ^= 0x5A
        bad = bytes(bad)
        if zlib_ok(bad, size):
            pytest.skip("the flipped stream is still a valid one")
    elif kind == "wrong_size_short":
        bad, size = good, len(data) - 1
    elif kind == "wrong_size_long":
        bad, size = good, len(data) + 1
    elif kind == "bad_type":
        bad = bytes([0x07]) + good[1:]           # BFINAL = 1, BTYPE = 3
    elif kind == "bad_stored_len":
        bad = bytes([0x01, 0x10, 0x00, 0x10, 0x00]) + b"x" * 16  # NLEN is not ~LEN
        size = 16
    else:  # a fixed-code block whose first symbol is a match: distance 1 before the start of the output
        # BFINAL=1 BTYPE=01, length code 257 (7 bits 0000001), distance code 0 (5 bits), end of block (7 bits 0)
        bits = "1" + "10" + "0000001" + "00000" + "0000000"
        v = int(bits[::-1], 2)
        bad, size = v.to_bytes(4, "little"), 3
    sizes = [len(data)] * 70
    streams = list(ok)
    streams[33] = bad
    sizes[33] = size
    rc, _ = run(gi, streams, sizes)
    assert rc > 0


def zlib_ok(s, size):
    try:
        return len(zlib.decompress(s, -15)) == size
    except zlib.error:
        return False


@pytest.mark.parametrize("level", [1, 6])
def test_reader_on_the_gpu_matches_the_cpu_decoder(tmp_path, level):
    lib = _lib.load()
    path = str(tmp_path / "x.bam")
    hostio.synth_stream(path, None, seed=21, n_reads=1_500_000, ref_names=["chr1", "chr2"], ref_lens=[3_000_000, 2_000_000], n_lanes=3, level=level)

    def columns():
        out = []
        b = hostio.BamFile(path)
        for batch in b.batches(400_000):
            out.append({k: np.array(v, copy=True) for k, v in batch.items() if isinstance(v, np.ndarray)})
        b.close()
        return out
    cpu = columns()
    before = lib.bqc_gpu_inflated_blocks()
    lib.bqc_gpu_inflate_device(0)
    try:
        gpu = columns()
    finally:
        lib.bqc_gpu_inflate_device(-1)
    assert lib.bqc_gpu_inflated_blocks() - before > 1000  # the card did work, not only the fallback (the share it gets beside the host's decoder threads varies with the box)
    assert len(cpu) == len(gpu) and sum(len(a["flag"]) for a in cpu) == 1_500_000
    for a, b in zip(cpu, gpu):
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]), k


from tests.hipmem import Hip as _Hip  # noqa: E402


def _launch_on_card(streams, want, crcs):
    """bqc_gpu_inflate_launch on device-resident operands (what csrc/gpu_bam.hip calls), with the blocks' expected CRC-32s: returns
    (status bits, output bytes)."""
    fn = C.CDLL(_lib.LIB_PATH).bqc_gpu_inflate_launch
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    n = len(streams)
    tab = np.zeros(n, dtype=[("coff", "<u8"), ("uoff", "<u8"), ("csize", "<u4"), ("usize", "<u4")])
    co = uo = 0
    for i, s in enumerate(streams):
        tab[i] = (co, uo, len(s), len(want[i]))
        co += len(s)
        uo += len(want[i])
    hip = _Hip()
    try:
        d_comp = hip.put(np.frombuffer(b"".join(streams), np.uint8), extra=256)
        d_tab = hip.put(tab.view(np.uint8))
        d_crc = hip.put(np.array(crcs, np.uint32))
        d_out = hip.put(np.zeros(uo, np.uint8), extra=4096)
        d_st = hip.put(np.zeros(16, np.uint32))
        d_tok = hip.put(np.zeros(uo // 32 + n + 64, np.uint32))
        d_ntok = hip.put(np.zeros(n + 16, np.uint32))
        fn(d_comp, d_tab, n, uo, d_out, d_crc, d_st, d_tok, d_ntok, None)
        assert hip.rt.hipDeviceSynchronize() == 0
        return int(hip.get(d_st, 4, np.uint32)[0]), hip.get(d_out, uo).tobytes()
    finally:
        hip.free()


def test_crc_checked_on_the_card():
    """The blocks' CRC-32s on the card — inside k_inflate_resolve (two phases: every final byte passes through that kernel; the CRC of
    1024 interleaved columns joined by polynomial arithmetic) or by k_gi_crc (one phase) — for every size class: empty, 1-9 bytes, around
    the 4096-byte rows of the resolve kernel's gather pass, maximal; blocks without any match (stored, incompressible) and blocks that
    are all matches; then ONE wrong CRC / ONE flipped output-relevant bit among many blocks must be noticed, wherever it sits."""
    rng = np.random.default_rng(17)
    datas = [b"", b"a", b"ab", b"abc", b"abcd", b"abcde", b"abcdefg", b"abcdefgh", b"abcdefghi"]
    for n in (63, 64, 65, 4093, 4094, 4095, 4096, 4097, 4099, 8191, 8192, 8193, 12288, 40001, 65533, 65534, 65535, 65536):
        datas.append(bytes(rng.integers(0, 5, n, dtype=np.uint8)))           # short codes, many matches
        datas.append(bytes(rng.integers(0, 256, n, dtype=np.uint8)))         # no matches at all
    datas += [p for p in payloads()]
    streams, want = [], []
    for d in datas:
        for level in (0, 1, 9):
            s = raw_deflate(d, level)
            if len(s) <= 65536 + 64:
                streams.append(s)
                want.append(d)
    crcs = [zlib.crc32(w) & 0xFFFFFFFF for w in want]
    st, got = _launch_on_card(streams, want, crcs)
    assert st == 0
    assert got == b"".join(want)
    for k in (0, 1, 5, len(crcs) // 2, len(crcs) - 1):      # a wrong expectation: the CRC bit, nothing else
        bad = list(crcs)
        bad[k] ^= 1 << int(rng.integers(0, 32))
        st, _ = _launch_on_card(streams, want, bad)
        assert st == 8, (k, st)
    for k in range(0, len(streams), 7):                       # a literal byte of a stored block changed: the block inflates, its CRC is wrong
        if len(want[k]) == 0 or raw_deflate(want[k], 0) != streams[k]:
            continue
        s = bytearray(streams[k])
        s[5 + len(want[k]) // 2] ^= 0x20
        st, _ = _launch_on_card(streams[:k] + [bytes(s)] + streams[k + 1:], want, crcs)
        assert st == 8, (k, st)
