#!/bin/bash
# Host input code (BGZF reader, DEFLATE decoder, CRC, BAM record decoder) under ASan + UBSan on valid, bit-flipped and
# truncated files: damaged BGZF blocks and — re-blocked with correct checksums — damaged BAM records.  Usage: tools/asan_host_io.sh
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -o $T/check tools/host_io_check.cpp \
    bamqc_amd/host/bam_io.cpp bamqc_amd/host/bgzf.cpp bamqc_amd/host/inflate_fast.cpp bamqc_amd/host/crc32_fast.cpp tools/gpu_inflate_stub.cpp -lz -lpthread
python - "$T" <<'PY'
import sys, random, gzip
sys.path.insert(0, ".")
from bamqc_amd import hostio
from tests import pybam
from tests.test_host_io import _wild_bam
T = sys.argv[1]
hostio.synth_write(T + "/a.bam", T + "/a.fa", 3, 60000, ["chr1", "chr2"], [300000, 200000], n_lanes=2)
hostio.synth_write(T + "/big.bam", T + "/big.fa", 4, 400000, ["chr1", "chr2"], [900000, 600000], n_lanes=2)  # large enough for the parallel record walk
_wild_bam(T + "/w.bam", 5, 3000)
rng = random.Random(1)
data = open(T + "/a.bam", "rb").read()
for k in range(40):  # damaged BGZF framing / deflate data / checksums
    d = bytearray(data)
    for _ in range(rng.choice((1, 3, 10))):
        d[rng.randrange(len(d))] ^= 1 << rng.randrange(8)
    if k % 5 == 0:
        d = d[:rng.randrange(len(d))]
    open(T + "/c%02d.bam" % k, "wb").write(d)
raw = gzip.open(T + "/w.bam", "rb").read()
for k in range(80):  # damaged records behind valid blocks
    d = bytearray(raw)
    for _ in range(rng.choice((1, 2, 5, 20))):
        i = rng.randrange(0 if k % 10 == 0 else 2000, len(d))
        d[i] = rng.randrange(256) if rng.random() < 0.5 else d[i] ^ (1 << rng.randrange(8))
    if k % 7 == 0:
        d = d[:rng.randrange(2000, len(d))]
    with open(T + "/r%02d.bam" % k, "wb") as f:
        p = 0
        while p < len(d):
            n = rng.randrange(1, 40000)
            f.write(pybam._bgzf_block(bytes(d[p:p + n]), level=rng.randrange(0, 10)))
            p += n
        f.write(pybam._bgzf_block(b""))
PY
BQC_IO_THREADS=4 $T/check $T/big.bam > $T/big1.txt 2>&1 && BQC_IO_THREADS=4 BQC_TEST_WALK_SKEW=1 $T/check $T/big.bam > $T/big2.txt 2>&1 && grep -q "400000 records rc 0" $T/big1.txt && grep -q "400000 records rc 0" $T/big2.txt || { cat $T/big1.txt $T/big2.txt | tail -30; echo "FAILED (parallel record walk)"; exit 1; }
BQC_IO_THREADS=4 $T/check $T/a.bam $T/w.bam $T/c*.bam $T/r*.bam > $T/out.txt 2>&1 || { cat $T/out.txt | tail -40; echo "FAILED (sanitizer report or crash)"; exit 1; }
if grep -q "runtime error\|AddressSanitizer" $T/out.txt; then grep -n "runtime error\|AddressSanitizer" $T/out.txt | head; echo FAILED; exit 1; fi
echo "ok: $(grep -c 'records rc 0' $T/out.txt) files read completely, $(grep -c -v 'records rc 0' $T/out.txt) rejected with an error, no sanitizer report"
rm -rf $T
