"""profiles/r<N>_traffic.json (read by bench.py for roofline.traffic and roofline.limiter) from a pmc_summary.json written by
collect_r<N>.sh: HBM bytes per k_short launch = FETCH_SIZE (KB, corrected by the factor the 1-GiB calibration kernel yields for
this counter on gfx950) + WRITE_SIZE (KB); cross-check: TCC_EA0_RDREQ_DRAM_32B x 32 B, calibrated the same way; and what
limits the kernel: share of the SIMDs' VALU issue slots in use, share of the CU time the LDS is busy, bank-conflict share.
usage: python profiles/make_traffic.py profiles/r2_pmc_summary.json profiles/r2_traffic.json [kernel_ms]"""
import json, sys
src, dst = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "profiles/r1_traffic.json"
kernel_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
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
    "source": "%s (profiles/collect_r2.sh, profiles/make_traffic.py)" % src,
}
try:  # what limits k_short: issue slots, not bytes
    valu = avg("sq_a:k_short:SQ_INSTS_VALU")
    busy = avg("sq_a:k_short:SQ_BUSY_CYCLES") / 32.0       # (the counter sums the 32 shader engines)
    lds_act, lds_conf = avg("sq_c:k_short:SQ_LDS_IDX_ACTIVE"), avg("sq_b:k_short:SQ_LDS_BANK_CONFLICT")
    out["limiter"] = {
        "what": "valu+lds issue: integer SWAR and LDS atomics per base, not HBM",
        "valu_instructions_per_launch": valu, "valu_instructions_per_read": valu / 1e7,
        "kernel_cycles": busy,
        "valu_issue_frac": valu / (1024.0 * busy / 4.0),     # one wave64 VALU instruction per 4 cycles per SIMD, 1024 SIMDs
        "lds_busy_frac_of_cu_time": lds_act / (256.0 * busy),
        "lds_bank_conflict_share_of_lds_cycles": lds_conf / lds_act,
    }
except KeyError:
    pass
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
