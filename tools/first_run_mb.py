import os, subprocess, sys, tempfile, time, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from bamqc_amd import hostio
tmp = tempfile.mkdtemp(prefix="bqc_fm_")
names, lens = ["chr1", "chr2", "chr3", "chr4"], [25_000_000] * 4
bam, fa = os.path.join(tmp, "c2.bam"), os.path.join(tmp, "c2.fa")
hostio.synth_stream(bam, fa, 1002, 10_000_000, names, lens, read_len=150, level=1)
exe = os.path.join(ROOT, "bin", "bamqualcheck")
for rnd in range(2):
    for mb in (sys.argv[1:] or ["640", "320", "192", "128", "64"]):
        ts, loops = [], []
        for k in range(5):
            time.sleep(0.8)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "-r", fa, "-o", os.path.join(tmp, "o.bamqc"), "-c", ",".join(names), bam], capture_output=True, text=True,
                               env=dict(os.environ, BQC_GB_FIRST_MB=mb, BQC_TIMING="1"))
            ts.append(time.perf_counter() - t0)
            assert r.returncode == 0
            import re
            m = re.search(r"record loop ([0-9.]+) s", r.stderr); loops.append(float(m.group(1)))
        print(mb, "median wall %.3f min %.3f loops %s" % (statistics.median(ts), min(ts), loops), flush=True)
