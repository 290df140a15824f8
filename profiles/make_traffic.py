"""profiles/r1_traffic.json (read by bench.py for roofline.traffic) from a pmc_summary.json written by collect_r1.sh:
HBM bytes per k_short launch = FETCH_SIZE (KB, corrected by the factor the 1-GiB calibration kernel yields for this
counter on gfx950) + WRITE_SIZE (KB); cross-check: TCC_EA0_RDREQ_DRAM_32B x 32 B, calibrated the same way.
usage: python profiles/make_traffic.py profiles/r1_d_pmc_summary.json"""
import json, sys
src = sys.argv[1]
d = json.load(open(src))
def avg(key): return d[key]["avg"]
GiB_KB = float(1 << 20)
cal_fetch = avg("cal_fetch:k_calib_read4:FETCH_SIZE")
corr = GiB_KB / cal_fetch
fetch, write = avg("pmc_fetch:k_short:FETCH_SIZE"), avg("pmc_write:k_short:WRITE_SIZE")
cal_dram = avg("cal_dram:k_calib_read4:TCC_EA0_RDREQ_DRAM_32B_sum") * 32.0
dram = avg("pmc_tcc:k_short:TCC_EA0_RDREQ_DRAM_32B_sum") * 32.0 * ((1 << 30) / cal_dram)
hit, miss = avg("pmc_tcc:k_short:TCC_HIT_sum"), avg("pmc_tcc:k_short:TCC_MISS_sum")
out = {
    "workload": "bench.py default (10,000,000 reads x 150 bp per launch)",
    "kernel": "k_short",
    "fetch_size_KB": fetch, "write_size_KB": write, "fetch_correction": corr,
    "fetch_correction_note": "k_calib_read4 streams 1 GiB; FETCH_SIZE reports %.0f KB for 1,048,576 KB -> x%.3f (gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE taken as is" % (cal_fetch, corr),
    "read_bytes_from_TCC_EA0_RDREQ_DRAM_32B": dram,
    "calibration_TCC_EA0_RDREQ_DRAM_32B_for_1GiB": cal_dram,
    "l2_hit_rate": hit / (hit + miss),
    "traffic_bytes_per_launch": (fetch * corr + write) * 1024.0,
    "source": "%s (profiles/collect_r1.sh, profiles/make_traffic.py)" % src,
}
json.dump(out, open("profiles/r1_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
