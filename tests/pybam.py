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
            else:
                raise ValueError("tag type " + ty)
            tags.setdefault(key, []).append((ty, val))
        recs.append(dict(rid=rid, pos=pos, mapq=mapq, flag=flag, l_seq=l_seq, rnext=rnext, tlen=tlen, cigar=cigar, seq=seq,
                         qual=qual, tags=tags, name=r[32:32 + l_name - 1].decode()))
        p += 4 + bs
    return text, refs, recs


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


def write_bam(path, header_text, refs, records, block=40000, rng=None):
    """Independent BAM writer for tests.  records: dicts with rid pos mapq flag rnext pnext tlen name (str), cigar (uint32
    words), seq (packed nibbles), qual (bytes), l_seq, tags (bytes, already encoded).  BGZF blocks are cut at arbitrary
    byte offsets (records straddle them), with varying block sizes when `rng` is given."""
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
    with open(path, "wb") as f:
        p = 0
        while p < len(out):
            n = block if rng is None else int(rng.integers(1, block + 1))
            f.write(_bgzf_block(bytes(out[p:p + n]), level=6 if rng is None else int(rng.integers(0, 10))))
            p += n
        f.write(_bgzf_block(b""))
