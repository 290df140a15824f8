"""The BGZF reader's own DEFLATE decoder (bamqc_amd/host/inflate_fast.cpp) against zlib: every block type, code shapes
that need second-level tables, boundaries of the fast / careful paths, and rejection of damaged streams."""
import ctypes as C
import random
import zlib

import pytest

from bamqc_amd import _lib


def inflate_raw(stream: bytes, out_n: int, trailer: bytes = b"\xAA" * 8):
    lib = _lib.load()
    out = C.create_string_buffer(out_n + 16)
    C.memset(out, 0x5A, out_n + 16)
    ok = lib.bqc_inflate_raw(stream + trailer, len(stream), out, out_n)
    raw = out.raw
    assert raw[out_n:] == b"\x5A" * 16, "wrote past the end of the output"
    return ok, raw[:out_n]


def deflate_raw(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    return c.compress(data) + c.flush()


def samples(rng):
    yield b""
    yield b"a"
    yield bytes(rng.randrange(256) for _ in range(1000))                      # incompressible
    yield bytes(rng.randrange(256) for _ in range(65536))
    yield b"A" * 65536                                                          # distance 1 runs
    yield (b"ACGT" * 20000)[:65536]                                             # distance 4
    yield (b"ab" * 40000)[:65535]                                               # distance 2
    yield bytes(rng.choice(b"ACGTN") for _ in range(65536))                    # few symbols: short codes
    text = bytearray()
    while len(text) < 65536:                                                    # BAM-like: names, bases, qualities
        text += b"read%07d\x00" % rng.randrange(10 ** 7)
        text += bytes(rng.choice(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88") for _ in range(75))
        text += bytes(min(41, max(2, int(rng.gauss(34, 6)))) for _ in range(150))
    yield bytes(text[:65536])
    skew = bytearray()                                                          # very skewed literals: codes up to 15 bits
    for i in range(256):
        skew += bytes([i]) * max(1, 60000 >> i if i < 16 else 1)
    rng.shuffle(skew)
    yield bytes(skew[:65536])
    for n in (257, 258, 259, 272, 275, 280, 281, 282, 283, 284, 290, 300, 1023, 4097):  # around the careful-path margin
        yield bytes(rng.choice(b"ab") for _ in range(n))


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE),
                                            (9, zlib.Z_FILTERED)])
def test_round_trip_against_zlib(level, strategy):
    rng = random.Random(level * 100 + strategy)
    for data in samples(rng):
        stream = deflate_raw(data, level, strategy)
        ok, out = inflate_raw(stream, len(data))
        assert ok == 1 and out == data, (level, strategy, len(data))
        # the trailer bytes are never consumed, whatever they hold
        ok, out = inflate_raw(stream, len(data), trailer=b"\x00" * 8)
        assert ok == 1 and out == data
        ok, out = inflate_raw(stream, len(data), trailer=b"\xFF" * 8)
        assert ok == 1 and out == data


def test_multi_block_streams_and_small_windows():
    rng = random.Random(5)
    data = b"".join(bytes(rng.choice(b"ACGT") for _ in range(3000)) + bytes(rng.randrange(256) for _ in range(500)) for _ in range(18))[:65536]
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 1)  # memLevel 1: many small dynamic blocks
    stream = c.compress(data[:20000]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(data[20000:]) + c.flush()
    ok, out = inflate_raw(stream, len(data))
    assert ok == 1 and out == data


def test_wrong_output_size_is_rejected():
    data = bytes(random.Random(1).choice(b"ACGT") for _ in range(5000))
    stream = deflate_raw(data)
    assert inflate_raw(stream, len(data) - 1)[0] == 0
    assert inflate_raw(stream, len(data) + 1)[0] == 0
    assert inflate_raw(stream, 0)[0] == 0
    assert inflate_raw(b"", 0)[0] == 0          # no block at all
    assert inflate_raw(b"\x03\x00", 0) == (1, b"")  # an empty fixed block


def test_truncated_streams_are_rejected():
    rng = random.Random(2)
    for data in (bytes(rng.choice(b"ACGTN") for _ in range(20000)), bytes(rng.randrange(256) for _ in range(3000))):
        for level in (0, 6):
            stream = deflate_raw(data, level)
            for cut in sorted({1, 2, 5, len(stream) // 2, len(stream) - 2, len(stream) - 1}):
                if 0 < cut < len(stream):
                    ok, _ = inflate_raw(stream[:cut], len(data), trailer=b"\x00" * 8)
                    assert ok == 0, (level, cut, len(stream))


def test_corrupted_streams_agree_with_zlib():
    """Flip bits: whenever zlib rejects the stream or yields another length, so must the decoder; when zlib accepts, the
    bytes are the same (the CRC32 of the BGZF block is what catches that case in the reader)."""
    rng = random.Random(3)
    base = []
    for data in samples(random.Random(4)):
        if 0 < len(data) <= 5000 or len(data) == 65536:
            base.append((data, deflate_raw(data[:6000], 6)))
    n_accept = 0
    for data, stream in base:
        n = len(data[:6000])
        for _ in range(150):
            s = bytearray(stream)
            for _ in range(rng.choice((1, 1, 2, 5))):
                i = rng.randrange(len(s))
                s[i] ^= 1 << rng.randrange(8)
            s = bytes(s)
            d = zlib.decompressobj(-15)
            try:
                ref = d.decompress(s + b"\x00" * 8, n + 1)
                ref_ok = d.eof and len(ref) == n and len(s) + 8 - len(d.unused_data) <= len(s)
            except zlib.error:
                ref, ref_ok = b"", False
            ok, out = inflate_raw(s, n, trailer=b"\x00" * 8)
            assert bool(ok) == bool(ref_ok), (len(s), ok, ref_ok)
            if ok:
                n_accept += 1
                assert out == ref
    assert n_accept > 0


def test_invalid_headers():
    assert inflate_raw(b"\x07", 0)[0] == 0                       # block type 3
    assert inflate_raw(b"\x01\x01\x00\x00\x00", 1)[0] == 0       # stored: LEN / NLEN mismatch
    assert inflate_raw(b"\x01\x01\x00\xfe\xff", 1)[0] == 0       # stored: data missing
    assert inflate_raw(b"\x01\x01\x00\xfe\xffZ", 1) == (1, b"Z")
    # fixed block whose first symbol is a match: distance beyond the start of the output
    c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    stream = c.compress(b"abcabcabcabc") + c.flush()
    ok, out = inflate_raw(stream, 12)
    assert ok == 1 and out == b"abcabcabcabc"


def test_crc32_matches_zlib():
    lib = _lib.load()
    rng = random.Random(9)
    for n in list(range(0, 130)) + [255, 256, 257, 4095, 4096, 65535, 65536, 100003]:
        d = bytes(rng.randrange(256) for _ in range(n))
        assert lib.bqc_crc32(d, n) == zlib.crc32(d), n
