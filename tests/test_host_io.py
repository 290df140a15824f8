"""CPU tests of the host I/O layer (own BGZF/BAM/FASTA code replacing SeqAn): the C++ reader is
checked against an independent pure-Python decode of the same files."""
import os
import struct
import subprocess

import numpy as np
import pytest

from bamqc_amd import hostio, synth
from tests import pybam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def synth_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    bam, fa = str(d / "s.bam"), str(d / "s.fa")
    hostio.synth_write(bam, fa, seed=1001, n_reads=6000, ref_names=["chr1", "chrX"], ref_lens=[300_000, 120_000], n_lanes=2)
    return bam, fa


def test_bam_roundtrip_matches_generator_and_python_decode(synth_files):
    bam, fa = synth_files
    lens = [300_000, 120_000]
    refs = [synth.reference(1001, i, n) for i, n in enumerate(lens)]
    want = synth.batch(1001, 6000, lens, refs, n_lanes=2)
    f = hostio.BamFile(bam)
    assert f.ref_names == ["chr1", "chrX"] and f.ref_lens == lens
    assert f.sample_id == "SYN" and f.lane_count == 2 and f.lanes() == [("L1", 0), ("L2", 1)]
    f.set_main_chrom([1, 0])
    got = list(f.batches(max_reads=2500))
    assert [len(g["flag"]) for g in got] == [2500, 2500, 1000]
    from tests.synth import concat
    g = concat(got)
    for k in ("mapq", "lane", "rid", "pos", "tlen", "nm", "as_", "l_seq", "n_cigar", "seq", "qual", "cigar"):
        assert np.array_equal(g[k], want[k]), k
    # flag: BAM bits + MATE_MAIN recomputed from rnext (= rid in the synthetic files) and the main set
    exp_flag = (want["flag"] & 0x0FFF) | np.where(want["rid"] == 0, 0x1000, 0).astype(np.uint16)
    assert np.array_equal(g["flag"], exp_flag)
    # independent decode
    text, prefs, recs = pybam.read_bam(bam)
    assert prefs == [("chr1", 300_000), ("chrX", 120_000)] and len(recs) == 6000
    assert "@RG\tID:L1\tSM:SYN" in text
    for i in (0, 1, 17, 2999, 5999):
        r = recs[i]
        assert r["pos"] == int(want["pos"][i]) and r["flag"] == int(want["flag"][i] & 0x0FFF)
        assert r["tags"]["RG"][0][1] == "L%d" % (int(want["lane"][i]) + 1)
        if want["nm"][i] != -1:
            assert r["tags"]["NM"][0][1] == int(want["nm"][i])


def test_fasta_loader_matches_python(synth_files):
    bam, fa = synth_files
    got = hostio.load_fasta(fa)
    want = pybam.read_fasta(fa)
    assert [n for n, _ in got] == [n for n, _ in want] == ["chr1", "chrX"]
    for (n, codes), (_, s) in zip(got, want):
        assert len(codes) == len(s)
        assert "".join("ACGTN"[c] for c in codes[:500]) == s[:500]
    assert np.array_equal(got[0][1], synth.reference(1001, 0, 300_000))


def test_truncated_bam_is_an_error(synth_files, tmp_path):
    bam, _ = synth_files
    data = open(bam, "rb").read()
    bad = str(tmp_path / "trunc.bam")
    open(bad, "wb").write(data[:len(data) // 2])
    with pytest.raises(IOError):  # small files are inflated in one round: the error may surface at open
        list(hostio.BamFile(bad).batches())


def test_damaged_blocks_are_errors_not_crashes(synth_files, tmp_path):
    """Bit flips in the deflate data, the CRC32 or the ISIZE of a BGZF block: the reader reports an error (its own inflate
    rejects the stream, or the block checksum does), whatever the damage; an undamaged copy still reads."""
    import random
    bam, _ = synth_files
    data = bytearray(open(bam, "rb").read())
    rng = random.Random(11)
    bsize = struct.unpack_from("<H", data, 16)[0] + 1          # first block
    spots = [18 + rng.randrange(bsize - 26) for _ in range(12)] + [bsize - 8, bsize - 5, bsize - 4, bsize - 1]
    n_err = 0
    for k, pos in enumerate(spots):
        bad = bytearray(data)
        bad[pos] ^= 1 << rng.randrange(8)
        p = str(tmp_path / ("bad%d.bam" % k))
        open(p, "wb").write(bad)
        try:
            list(hostio.BamFile(p).batches())
        except IOError:
            n_err += 1
    assert n_err == len(spots)
    ok = str(tmp_path / "ok.bam")
    open(ok, "wb").write(data)
    assert sum(len(b["flag"]) for b in hostio.BamFile(ok).batches()) > 0


def test_not_a_bam(tmp_path):
    p = str(tmp_path / "x.bam")
    open(p, "wb").write(b"hello world, definitely not gzip")
    with pytest.raises(IOError):
        hostio.BamFile(p)


def _tiny_bam(path, records, rg_lines="@RG\tID:A\tSM:S1\n@RG\tID:B\tSM:S2\n"):
    """Hand-assembled BAM through python (independent of the C++ writer)."""
    import gzip
    text = "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\n" + rg_lines
    out = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1)
    out += struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1000)
    for tags in records:
        name = b"q\0"
        seq = bytes([0x12, 0x48])  # ACGT
        body = struct.pack("<iiBBHHHiiii", 0, 10, len(name), 30, 4680, 1, 0x41, 4, 0, 10, 100) + name + struct.pack("<I", 4 << 4) + seq + bytes([30] * 4) + tags
        out += struct.pack("<i", len(body)) + body
    with gzip.open(path, "wb") as f:  # plain gzip is not BGZF: add the BC field by hand
        pass
    import zlib
    def block(payload):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(payload) + c.flush()
        bsize = 18 + len(comp) + 8
        return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize - 1) + comp +
                struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))
    with open(path, "wb") as f:
        f.write(block(out[:100]) + block(out[100:]) + block(b""))


def test_rg_semantics(tmp_path):
    p = str(tmp_path / "t.bam")
    _tiny_bam(p, [b"RGZB\0" + b"NMC\x02" + b"NMi" + struct.pack("<i", 5) + b"ASs" + struct.pack("<h", -7),
                  b"XXZfoo\0RGZunknown\0"])
    f = hostio.BamFile(p)
    assert f.sample_id == "S2" and f.lane_count == 2
    (cols,) = list(f.batches())
    assert cols["lane"].tolist() == [1, 0]                 # unknown @RG ID maps to lane 0 (std::map::operator[])
    assert cols["nm"].tolist() == [2, -1] and cols["as_"].tolist() == [-7, -2 ** 31]
    assert cols["nm_extra_read"].tolist() == [0] and cols["nm_extra_val"].tolist() == [5]
    assert f.lanes() == [("A", 0), ("B", 1), ("unknown", 0)]  # ... and shows up as an extra output block
    # RG tag of a non-Z type is fatal (bamqualcheck.cpp:89-97); a missing RG tag is a defined error
    _tiny_bam(p, [b"RGi" + struct.pack("<i", 1)])
    with pytest.raises(IOError):
        list(hostio.BamFile(p).batches())
    _tiny_bam(p, [b"NMC\x02"])
    with pytest.raises(IOError):
        list(hostio.BamFile(p).batches())


def test_cli_argument_errors(tmp_path):
    exe = os.path.join(ROOT, "bin", "bamqualcheck")
    r = subprocess.run([exe, "--version"], capture_output=True, text=True)
    assert r.returncode == 0 and "dev" in r.stdout
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "--reference" in r.stdout
    r = subprocess.run([exe, "-o", "x", "in.bam"], capture_output=True, text=True)   # -r is required
    assert r.returncode == 1
    r = subprocess.run([exe, "-r", "g.fa", "-o", str(tmp_path / "o"), "in.txt"], capture_output=True, text=True)
    assert r.returncode == 1 and "extension" in r.stderr
    r = subprocess.run([exe, "-r", "g.fa", "-o", str(tmp_path / "o"), "--bogus", "in.bam"], capture_output=True, text=True)
    assert r.returncode == 1 and "illegal option" in r.stderr
    r = subprocess.run([exe, "-r", "g.fa", "-o", str(tmp_path / "o"), str(tmp_path / "missing.bam")], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not open" in r.stderr


def test_sam_text_decodes_to_the_same_batches(synth_files, tmp_path):
    """N4: the SAM text reader (stdin mode of the program) fills the same columns as the BAM reader."""
    bam, fa = synth_files
    sam = str(tmp_path / "s.sam")
    open(sam, "w").write(pybam.bam_to_sam_text(bam))
    fb, fs = hostio.BamFile(bam), hostio.BamFile(sam)
    assert fs.ref_names == fb.ref_names and fs.ref_lens == fb.ref_lens
    assert fs.sample_id == fb.sample_id and fs.lane_count == fb.lane_count and fs.lanes() == fb.lanes()
    fb.set_main_chrom([1, 0]); fs.set_main_chrom([1, 0])
    from tests.synth import concat
    gb, gs = concat(list(fb.batches(max_reads=2500))), concat(list(fs.batches(max_reads=1700)))
    assert set(gb) == set(gs)
    for k in gb:
        assert np.array_equal(gb[k], gs[k]), k
