"""Independent pure-Python BAM/FASTA decoding for tests (gzip handles concatenated BGZF members)."""
import gzip
import struct

import numpy as np


def read_bam(path):
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", data, p)[0]
        name = data[p + 4:p + 4 + ln - 1].decode()
        rl = struct.unpack_from("<i", data, p + 4 + ln)[0]
        refs.append((name, rl))
        p += 8 + ln
    recs = []
    while p < len(data):
        bs = struct.unpack_from("<i", data, p)[0]
        r = data[p + 4:p + 4 + bs]
        rid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, rnext, pnext, tlen = struct.unpack_from("<iiBBHHHiiii", r, 0)
        q = 32 + l_name
        cigar = np.frombuffer(r, dtype="<u4", count=n_cig, offset=q).copy()
        q += 4 * n_cig
        seq = np.frombuffer(r, dtype=np.uint8, count=(l_seq + 1) // 2, offset=q).copy()
        q += (l_seq + 1) // 2
        qual = np.frombuffer(r, dtype=np.uint8, count=l_seq, offset=q).copy()
        q += l_seq
        tags = {}
        while q < bs:
            key = r[q:q + 2].decode()
            ty = chr(r[q + 2])
            q += 3
            if ty in "cCA":
                val = struct.unpack_from("<b" if ty == "c" else "<B", r, q)[0]
                q += 1
            elif ty in "sS":
                val = struct.unpack_from("<h" if ty == "s" else "<H", r, q)[0]
                q += 2
            elif ty in "iI":
                val = struct.unpack_from("<i" if ty == "i" else "<I", r, q)[0]
                q += 4
            elif ty == "f":
                val = struct.unpack_from("<f", r, q)[0]
                q += 4
            elif ty in "ZH":
                e = r.index(b"\0", q)
                val = r[q:e].decode()
                q = e + 1
            elif ty == "B":
                st = chr(r[q])
                cnt = struct.unpack_from("<i", r, q + 1)[0]
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
                val = (st, list(struct.unpack_from("<%d%s" % (cnt, fmt), r, q + 5)))
                q += 5 + cnt * struct.calcsize(fmt)
            else:
                raise ValueError("tag type " + ty)
            tags.setdefault(key, []).append((ty, val))
        recs.append(dict(rid=rid, pos=pos, mapq=mapq, flag=flag, l_seq=l_seq, rnext=rnext, tlen=tlen, cigar=cigar, seq=seq,
                         qual=qual, tags=tags, name=r[32:32 + l_name - 1].decode()))
        p += 4 + bs
    return text, refs, recs


def columns(path, main_chrom=None):
    """The file as the column dict the product's readers hand out (include/bamqc.h: bqc_batch) — decoded here from the records
    of read_bam, sharing nothing with the product: flag with the decoder's annotations (0x1000 mate on a main chromosome,
    0x8000 no qualities), lane through the header's @RG order, first integer NM, first AS.  Returns (cols, refs, lanes, sample)."""
    text, refs, recs = read_bam(path)
    lanes, sample = {}, ""
    for line in text.split("\n"):
        if line.startswith("@RG"):
            for f in line.split("\t")[1:]:
                if f.startswith("ID:"):
                    lanes[f[3:]] = len(lanes)
                if f.startswith("SM:"):
                    sample = f[3:]
    main = [1] * len(refs) if main_chrom is None else list(main_chrom)
    ints = "cCsSiI"
    want = dict(flag=[], mapq=[], lane=[], rid=[], pos=[], tlen=[], nm=[], as_=[], l_seq=[], n_cigar=[])
    for r in recs:
        f = r["flag"] & 0x0FFF
        if 0 <= r["rnext"] < len(refs) and main[r["rnext"]]:
            f |= 0x1000
        if r["l_seq"] > 0 and r["qual"][0] == 0xFF:
            f |= 0x8000
        nm = [v for key, vals in r["tags"].items() if key == "NM" for ty, v in vals if ty in ints]
        as_ = [v for key, vals in r["tags"].items() if key == "AS" for ty, v in vals]
        want["flag"].append(f); want["mapq"].append(r["mapq"]); want["lane"].append(lanes[r["tags"]["RG"][0][1]])
        want["rid"].append(r["rid"]); want["pos"].append(r["pos"]); want["tlen"].append(r["tlen"])
        want["nm"].append((nm[0] & 0xFFFFFFFF) if nm else 0xFFFFFFFF)
        want["as_"].append((int(as_[0]) & 0xFFFFFFFF) if as_ else 0x80000000)
        want["l_seq"].append(r["l_seq"]); want["n_cigar"].append(len(r["cigar"]))
    dt = dict(flag=np.uint16, mapq=np.uint8, lane=np.uint8, rid=np.int32, pos=np.int32, tlen=np.int32, l_seq=np.uint32, n_cigar=np.uint16)
    cols = {k: np.array(v, np.int64).astype(dt[k]) for k, v in want.items() if k in dt}
    cols["nm"] = np.array(want["nm"], np.uint32).view(np.int32)
    cols["as_"] = np.array(want["as_"], np.uint32).view(np.int32)
    cat = lambda key, d: np.concatenate([r[key] for r in recs]).astype(d) if recs else np.zeros(0, d)
    cols["seq"], cols["qual"], cols["cigar"] = cat("seq", np.uint8), cat("qual", np.uint8), cat("cigar", np.uint32)
    return cols, refs, lanes, sample


def bam_to_sam_text(path):
    """The BAM file as SAM text (header + one line per record; integer tags as type i)."""
    text, refs, recs = read_bam(path)
    table = "=ACMGRSVTWYHKDBN"
    lines = [text.rstrip("\n")]
    for r in recs:
        nib = []
        for b in r["seq"]:
            nib += [b >> 4, b & 15]
        seq = "".join(table[x] for x in nib[:r["l_seq"]]) or "*"
        qual = "*" if r["l_seq"] == 0 or r["qual"][0] == 0xFF else "".join(chr(int(q) + 33) for q in r["qual"])
        cig = "".join("%d%s" % (int(c) >> 4, "MIDNSHP=X"[int(c) & 15]) for c in r["cigar"]) or "*"
        rname = refs[r["rid"]][0] if r["rid"] >= 0 else "*"
        rnext = "*" if r["rnext"] < 0 else ("=" if r["rnext"] == r["rid"] else refs[r["rnext"]][0])
        tags = []
        for key, vals in r["tags"].items():
            for ty, val in vals:
                tags.append("%s:%s:%s" % (key, "i" if ty in "cCsSiI" else ty, val))
        lines.append("\t".join([r["name"] or "*", str(r["flag"]), rname, str(r["pos"] + 1), str(r["mapq"]), cig, rnext, "0", str(r["tlen"]), seq, qual] + tags))
    return "\n".join(lines) + "\n"


def read_fasta(path):
    out = []
    name, seq = None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                out.append((name, "".join(seq)))
            name = line[1:].split(" ")[0].split("\t")[0]
            seq = []
        else:
            seq.append(line)
    if name is not None:
        out.append((name, "".join(seq)))
    return out


def _bgzf_block(payload, level=6):
    import zlib
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(payload) + c.flush()
    bsize = 18 + len(comp) + 8
    assert bsize <= 65536
    return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize - 1) + comp +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def _bam_bytes(header_text, refs, records):
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(header_text)) + header_text.encode() + struct.pack("<i", len(refs)))
    for name, ln in refs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    for r in records:
        name = r["name"].encode() + b"\0"
        cig = np.asarray(r["cigar"], dtype="<u4").tobytes()
        body = struct.pack("<iiBBHHHiiii", r["rid"], r["pos"], len(name), r["mapq"], 4680, len(r["cigar"]), r["flag"], r["l_seq"],
                           r["rnext"], r.get("pnext", -1), r["tlen"])
        body += name + cig + bytes(r["seq"]) + bytes(r["qual"]) + r["tags"]
        out += struct.pack("<i", len(body)) + body
    return out


def write_bam(path, header_text, refs, records, block=40000, rng=None):
    """Independent BAM writer for tests.  records: dicts with rid pos mapq flag rnext pnext tlen name (str), cigar (uint32
    words), seq (packed nibbles), qual (bytes), l_seq, tags (bytes, already encoded).  BGZF blocks are cut at arbitrary
    byte offsets (records straddle them), with varying block sizes when `rng` is given."""
    out = _bam_bytes(header_text, refs, records)
    with open(path, "wb") as f:
        p = 0
        while p < len(out):
            n = block if rng is None else int(rng.integers(1, block + 1))
            f.write(_bgzf_block(bytes(out[p:p + n]), level=6 if rng is None else int(rng.integers(0, 10))))
            p += n
        f.write(_bgzf_block(b""))


MEMBER_KINDS = ("stored", "fixed", "level9", "flush_split", "level1", "empty")


def _bgzf_member(payload, kind, rng):
    """One BGZF member whose DEFLATE stream is of the given kind: `stored` (level 0: stored blocks only), `fixed` (Z_FIXED: fixed
    Huffman codes), `level9` / `level1` (dynamic codes), `flush_split` (the payload compressed in 2-5 pieces with Z_FULL_FLUSH
    between them: several deflate blocks, an empty stored block after each, no match reaches back over a flush point).
    Returns None when the member would not fit BGZF's 64 KiB."""
    import zlib
    if kind == "stored":
        c = zlib.compressobj(0, zlib.DEFLATED, -15)
        comp = c.compress(payload) + c.flush()
    elif kind == "fixed":
        c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
        comp = c.compress(payload) + c.flush()
    elif kind == "flush_split":
        c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, -15)
        cuts = sorted(int(x) for x in rng.integers(0, len(payload) + 1, size=int(rng.integers(1, 5))))
        comp, last = b"", 0
        for cut in cuts:
            comp += c.compress(payload[last:cut]) + c.flush(zlib.Z_FULL_FLUSH)
            last = cut
        comp += c.compress(payload[last:]) + c.flush()
    else:
        c = zlib.compressobj(9 if kind == "level9" else 1, zlib.DEFLATED, -15)
        comp = c.compress(payload) + c.flush()
    bsize = 18 + len(comp) + 8
    if bsize > 65536:
        return None
    return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", bsize - 1) + comp +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def write_bam_adversarial(path, header_text, refs, records, rng, tiny_until=0.5):
    """The same records as a BGZF file no writer in the field produces but every reader must take: members alternate between the kinds
    of MEMBER_KINDS (empty members in the middle of the file included); the first `tiny_until` of the stream is cut into payloads of
    1..200 bytes (every record of that part straddles at least one member boundary, most straddle several), the rest into payloads
    of 1..65 280 bytes with the extremes (1, 65 280) forced in.  Returns the list of (kind, payload bytes) written."""
    out = _bam_bytes(header_text, refs, records)
    members = []
    forced = [65280, 1, 65280, 2, 1]
    with open(path, "wb") as f:
        p, k = 0, 0
        while p < len(out):
            kind = MEMBER_KINDS[k % len(MEMBER_KINDS)]
            k += 1
            if kind == "empty":
                f.write(_bgzf_member(b"", "level1", rng))
                members.append((kind, 0))
                continue
            if p < tiny_until * len(out):
                n = int(rng.integers(1, 201))
            elif forced:
                n = forced.pop()
            else:
                n = int(rng.integers(1, 65281))
            while True:
                m = _bgzf_member(bytes(out[p:p + n]), kind, rng)
                if m is not None:
                    break
                n = n * 3 // 4  # (fixed codes on poorly compressible bytes: the member must stay within 64 KiB)
            f.write(m)
            members.append((kind, min(n, len(out) - p)))
            p += n
        f.write(_bgzf_block(b""))
    return members
