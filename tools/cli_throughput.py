"""End-to-end throughput of bin/bamqualcheck on a synthetic coordinate-sorted BAM (N2): writes the BAM + FASTA to /tmp,
runs the program with and without the k-mer sketch and prints reads/s; BQC_TIMING shows where the wall time goes.
usage: python tools/cli_throughput.py [n_reads]"""
import sys, time, os, subprocess
sys.path.insert(0, ".")
from bamqc_amd import hostio
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
names = ["chr1", "chr2", "chr3", "chr4"]; lens = [10_000_000] * 4
t0 = time.time(); hostio.synth_write("/tmp/s.bam", "/tmp/s.fa", 7, n, names, lens); t1 = time.time()
print("write %d reads: %.1f s, bam %.1f MB" % (n, t1 - t0, os.path.getsize("/tmp/s.bam") / 1e6))
for extra in ([], ["--no-sketch"]):
    t0 = time.time()
    os.environ["BQC_TIMING"] = "2" if extra else "1"
    rc = subprocess.call(["bin/bamqualcheck", "-r", "/tmp/s.fa", "-o", "/tmp/s.bamqc"] + extra + ["/tmp/s.bam"])
    dt = time.time() - t0
    print(extra, "rc", rc, "%.2f s -> %.2f M reads/s" % (dt, n / dt / 1e6))
