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


def test_bsize_smaller_than_header_and_trailer_is_rejected(synth_files, tmp_path):
    """BSIZE from the BC subfield is checked against the block's own header + trailer (else the compressed size underflows and
    the trailer pointer lies before the block), and a subfield may not run past the extra field."""
    bam, _ = synth_files
    data = bytearray(open(bam, "rb").read())
    for k, patch in enumerate([(16, struct.pack("<H", 10)), (16, struct.pack("<H", 24)), (14, struct.pack("<H", 200))]):
        bad = bytearray(data)
        bad[patch[0]:patch[0] + 2] = patch[1]
        p = str(tmp_path / ("bs%d.bam" % k))
        open(p, "wb").write(bad)
        with pytest.raises(IOError):
            list(hostio.BamFile(p).batches())


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


def _wild_bam(path, seed, n_reads, extra_nm=True, adversarial=False, pad_header=0):
    """Wild records (tests/test_gpu_fuzz.py) as a BAM with every tag type around the three tags the decoder looks for."""
    from tests.test_gpu_fuzz import wild_batch
    from tests import pybam
    rng = np.random.default_rng(seed)
    cols, refs = wild_batch(seed, n_reads)
    n_refs = len(refs)
    rg = ["lane%d" % i for i in range(3)]
    text = "@HD\tVN:1.6\n" + "".join("@SQ\tSN:chr%d\tLN:%d\n" % (i + 1, len(r)) for i, r in enumerate(refs))
    text += "".join("@RG\tID:%s\tSM:S\n" % x for x in rg)

    def int_tag(key, v):
        fits = [t for t, lo, hi in (("c", -128, 127), ("C", 0, 255), ("s", -32768, 32767), ("S", 0, 65535), ("i", -2 ** 31, 2 ** 31 - 1),
                                    ("I", 0, 2 ** 32 - 1)) if lo <= v <= hi]
        t = fits[int(rng.integers(0, len(fits)))]
        return key + t.encode() + struct.pack({"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[t], v)

    def junk():
        k = int(rng.integers(0, 6))
        if k == 0: return b"XAZ" + bytes(rng.integers(33, 127, size=int(rng.integers(0, 40))).astype(np.uint8)) + b"\0"
        if k == 1: return b"XBBs" + struct.pack("<i", 3) + struct.pack("<hhh", -1, 2, 3)
        if k == 2: return b"XFf" + struct.pack("<f", 1.5)
        if k == 3: return b"XHH" + b"1AE301\0"
        if k == 4: return b"XCA" + b"Q"
        return b"NMZ" + b"7\0"  # an NM that is not an integer: ignored (QualityCheck.hpp:201-209)

    recs, so, qo, co = [], 0, 0, 0
    exp_nm, exp_extra = [], []
    for i in range(n_reads):
        L, nc = int(cols["l_seq"][i]), int(cols["n_cigar"][i])
        tags = [b"RGZ" + rg[int(cols["lane"][i])].encode() + b"\0", int_tag(b"AS", int(cols["as_"][i]))]
        nm = int(cols["nm"][i])
        if nm >= 0:
            tags.append(int_tag(b"NM", nm))
            if rng.random() < 0.05 and extra_nm:  # a second NM tag: reported as an extra value
                tags.append(int_tag(b"NM", nm + 1))
                exp_extra.append((i, nm + 1))
        for _ in range(int(rng.integers(0, 4))):
            tags.insert(int(rng.integers(0, len(tags) + 1)), junk())
        # the first integer NM in tag order is the read's value
        recs.append(dict(rid=int(cols["rid"][i]), pos=int(cols["pos"][i]), mapq=int(cols["mapq"][i]), flag=int(cols["flag"][i]) & 0xFFF,
                         rnext=0 if int(cols["flag"][i]) & 0x1000 else -1, tlen=int(cols["tlen"][i]), name="r%d" % i,
                         cigar=cols["cigar"][co:co + nc], seq=cols["seq"][so:so + (L + 1) // 2], qual=cols["qual"][qo:qo + L], l_seq=L,
                         tags=b"".join(tags)))
        so += (L + 1) // 2; qo += L; co += nc
    text += "".join("@CO\tpadding line %d of a long header\n" % i for i in range(pad_header))
    if adversarial:  # members of every DEFLATE kind, payloads of 1 .. 65 280 bytes, every record of the first half across member boundaries
        pybam.write_bam_adversarial(path, text, [("chr%d" % (i + 1), len(r)) for i, r in enumerate(refs)], recs, rng)
    else:
        pybam.write_bam(path, text, [("chr%d" % (i + 1), len(r)) for i, r in enumerate(refs)], recs, rng=rng)
    return cols, exp_extra


def test_wild_records_and_tags_decode(tmp_path):
    """Python-written BAM (records straddling BGZF blocks of random size and compression level, every tag type, repeated
    and non-integer NM tags) through the C++ reader: every column as written."""
    p = str(tmp_path / "wild.bam")
    cols, exp_extra = _wild_bam(p, 21, 3000)
    f = hostio.BamFile(p)
    f.set_main_chrom(np.ones(3, np.uint8))
    got = list(f.batches(max_reads=777))
    cat = {k: np.concatenate([b[k] for b in got]) for k in cols}
    noq = np.zeros(len(cols["flag"]), bool)
    qo = 0
    for i, L in enumerate(cols["l_seq"]):
        noq[i] = L > 0 and cols["qual"][qo] == 0xFF
        qo += int(L)
    want_flag = (cols["flag"] & 0x1FFF) | np.where(noq, 0x8000, 0).astype(np.uint16)
    assert np.array_equal(cat["flag"], want_flag)
    for k in ("mapq", "lane", "rid", "pos", "tlen", "nm", "as_", "l_seq", "n_cigar", "seq", "qual", "cigar"):
        assert np.array_equal(cat[k], cols[k]), k
    extra, base = [], 0
    for b in got:
        if "nm_extra_read" in b:
            extra += [(int(r) + base, int(v)) for r, v in zip(b["nm_extra_read"], b["nm_extra_val"])]
        base += len(b["flag"])
    assert extra == exp_extra


def test_adversarial_bgzf_members_through_the_host_reader(tmp_path):
    """A BGZF file of stored / fixed-Huffman / level-9 / flush-split / empty members with payloads of 1 .. 65 280 bytes, records across
    member boundaries everywhere in its first half, a 150 KB header across several members: the host reader's columns against the
    independent decoder's (tests/pybam.py: gzip + struct)."""
    from tests import pybam
    p = str(tmp_path / "adv.bam")
    _wild_bam(p, 31, 4000, extra_nm=False, adversarial=True, pad_header=4000)
    want, refs, _, _ = pybam.columns(p, [1, 1, 1])
    for threads in ("1", "4"):
        os.environ["BQC_IO_THREADS"] = threads
        try:
            f = hostio.BamFile(p)
            f.set_main_chrom(np.ones(3, np.uint8))
            got = list(f.batches(max_reads=999))
        finally:
            del os.environ["BQC_IO_THREADS"]
        for k in ("flag", "mapq", "lane", "rid", "pos", "tlen", "nm", "as_", "l_seq", "n_cigar", "seq", "qual", "cigar"):
            assert np.array_equal(np.concatenate([b[k] for b in got]), want[k]), (k, threads)


def test_parallel_record_walk_equals_the_serial_walk(tmp_path):
    """Batches of a 120 MB stream read with one thread (serial block_size walk), with several (segments walked in parallel from
    guessed record starts) and with every guess forced wrong (each segment walked again from the position the chain
    arrives at): same batch sizes, same bytes in every column, with and without a contig filter."""
    import hashlib, subprocess, sys, textwrap
    bam, fa = str(tmp_path / "p.bam"), str(tmp_path / "p.fa")
    hostio.synth_write(bam, fa, seed=11, n_reads=400_000, ref_names=["chr1", "chr2", "chr3"], ref_lens=[900_000, 600_000, 300_000], n_lanes=2)
    prog = textwrap.dedent("""
        import sys, hashlib, numpy as np
        sys.path.insert(0, %r)
        from bamqc_amd import hostio
        for filt in (False, True):
            f = hostio.BamFile(%r)
            f.set_main_chrom(np.ones(3, np.uint8))
            if filt: f.set_rid_filter(np.array([1, 0, 1], np.uint8), False)
            h, sizes = hashlib.sha256(), []
            for b in f.batches(max_reads=100_000):
                sizes.append(len(b["flag"]))
                for k in sorted(b): h.update(np.ascontiguousarray(b[k]).tobytes())
            print(sizes, h.hexdigest())
        """) % (ROOT, bam)
    outs = []
    for env in ({"BQC_IO_THREADS": "1"}, {"BQC_IO_THREADS": "4"}, {"BQC_IO_THREADS": "4", "BQC_TEST_WALK_SKEW": "1"}):
        r = subprocess.run([sys.executable, "-c", prog], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] and "[100000, 100000, 100000, 100000]" in outs[0]


def test_rid_filter_releases_skipped_records(tmp_path):
    """A rank that keeps only the LAST contig of a coordinate-sorted BAM must not hold the records before it in memory: peak RSS
    of the filtered read stays near the unfiltered one's (measured in child processes)."""
    import subprocess, sys
    bam = str(tmp_path / "f.bam")
    hostio.synth_stream(bam, None, 21, 1_500_000, ["chr1", "chr2", "chr3"], [3_000_000, 3_000_000, 60_000], level=1)
    code = ("import sys, resource; sys.path.insert(0, %r)\nfrom bamqc_amd import hostio\nimport numpy as np\n"
            "f = hostio.BamFile(%r)\n"
            "if sys.argv[1] == 'last': f.set_rid_filter(np.array([0, 0, 1], np.uint8), False)\n"
            "n = sum(len(b['flag']) for b in f.batches(max_reads=200000))\n"
            "print(n, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)\n") % (ROOT, bam)
    res = {}
    for mode in ("all", "last"):
        out = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, env=dict(os.environ, BQC_IO_THREADS="4"))
        assert out.returncode == 0, out.stderr
        n, rss = out.stdout.split()
        res[mode] = (int(n), int(rss))
    assert res["all"][0] == 1_500_000 and 0 < res["last"][0] < 40_000
    assert res["last"][1] < res["all"][1] + 100_000, res  # KiB: the inflated file is ~450 MB


def _fasta_rules(data):
    """The loader's rules, byte by byte: '>' outside a header line starts one (anywhere, not only at a line start), the name is
    the header up to the first space or tab without a trailing CR, LF and CR are dropped from sequence, what precedes the first
    header is ignored."""
    out, i, n = [], 0, len(data)
    cur = None
    while i < n:
        c = data[i:i + 1]
        if c == b">":
            j = data.find(b"\n", i)
            j = n if j < 0 else j
            hdr = data[i + 1:j].decode("latin-1")
            name = hdr
            for k, ch in enumerate(hdr):
                if ch in " \t":
                    name = hdr[:k]
                    break
            if name.endswith("\r"):
                name = name[:-1]
            cur = [name, bytearray()]
            out.append(cur)
            i = j + 1
            continue
        if cur is not None and c not in (b"\n", b"\r"):
            cur[1].append({65: 0, 97: 0, 67: 1, 99: 1, 71: 2, 103: 2, 84: 3, 116: 3, 85: 3, 117: 3}.get(data[i], 4))
        i += 1
    return [(nm, bytes(s)) for nm, s in out]


@pytest.mark.parametrize("shape", ["plain", "crlf", "no_final_newline", "gt_inside", "junk_first", "empty_records", "big", "header_only_tail"])
def test_fasta_loaders_agree_on_awkward_files(tmp_path, shape, monkeypatch):
    rng = np.random.default_rng(17)
    def seq(n, width=60, eol=b"\n"):
        s = bytes(rng.choice(np.frombuffer(b"ACGTNacgtnRYKMU", np.uint8), n))
        return eol.join(s[i:i + width] for i in range(0, n, width)) + (eol if n else b"")
    if shape == "plain":
        data = b">chr1 first contig\n" + seq(1000) + b">chr2\tother\n" + seq(77, 10)
    elif shape == "crlf":
        data = b">chr1 x\r\n" + seq(500, 50, b"\r\n") + b">chr2\r\n" + seq(123, 50, b"\r\n")
    elif shape == "no_final_newline":
        data = (b">a\n" + seq(130))[:-1]
    elif shape == "gt_inside":
        data = b">a des>cription > with more\nACGT>b mid-line header\nGGCC\n>c\n" + seq(200)
    elif shape == "junk_first":
        data = b"ACGTACGT\n\n>a\n" + seq(90) + b"\n\n>b\n\n" + seq(61)
    elif shape == "empty_records":
        data = b">a\n>b\n>c x\n" + seq(10) + b">d\n"
    elif shape == "header_only_tail":
        data = b">a\n" + seq(100) + b">tail without newline"
    else:  # several MB: many tasks and chunks, headers at chunk edges
        parts = []
        for k in range(40):
            parts.append(b">ctg%d some text\n" % k + seq(int(rng.integers(1, 700_000)), int(rng.integers(20, 200))))
        data = b"".join(parts)
    path = str(tmp_path / "x.fa")
    open(path, "wb").write(data)
    want = _fasta_rules(data)
    for sequential in (False, True):
        if sequential:
            monkeypatch.setenv("BQC_FASTA_SEQUENTIAL", "1")
        got = hostio.load_fasta(path)
        assert [n for n, _ in got] == [n for n, _ in want]
        for (n, codes), (_, w) in zip(got, want):
            assert codes.tobytes() == w, (shape, sequential, n)
    # a gzip-compressed copy goes through the sequential loader on its own
    monkeypatch.delenv("BQC_FASTA_SEQUENTIAL")
    import gzip
    gz = str(tmp_path / "x.fa.gz")
    with gzip.open(gz, "wb") as f:
        f.write(data)
    got = hostio.load_fasta(gz)
    assert [(n, c.tobytes()) for n, c in got] == want
