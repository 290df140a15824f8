"""profiles/r<N>_traffic.json (read by bench.py for roofline.traffic and roofline.limiter) from a pmc_summary.json written by
collect_r<N>.sh: HBM bytes per k_short launch = FETCH_SIZE (KB, corrected by the factor the 1-GiB calibration kernel yields for
this counter on gfx950) + WRITE_SIZE (KB); cross-check: TCC_EA0_RDREQ_DRAM_32B x 32 B, calibrated the same way; and what
limits the kernel: share of the SIMDs' VALU issue slots in use, share of the CU time the LDS is busy, bank-conflict share.
The VALU ceiling is MEASURED (tools/micro/valu_issue.hip -> profiles/r3_valu_issue.json: wave64 instructions per microsecond and SIMD that
independent instruction streams reach at 1 / 2 / 4 / 8 waves per SIMD), not taken from a data sheet: round 2 divided by "one wave64
VALU instruction per 4 cycles", the round-2 review by "2 cycles" — the card does 0.37-0.42 per cycle with v_add_u32 and 0.23-0.26 with
the shift / permute / bit-field mix k_short is made of.
usage: python profiles/make_traffic.py profiles/r3_pmc_summary.json profiles/r3_traffic.json [kernel_ms] [profiles/r3_valu_issue.json]"""
import json, sys
src, dst = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "profiles/r1_traffic.json"
kernel_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
issue_json = sys.argv[4] if len(sys.argv) > 4 else "profiles/r3_valu_issue.json"
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
        "valu_per_cycle_per_simd": valu / (1024.0 * busy),
        "lds_busy_frac_of_cu_time": lds_act / (256.0 * busy),
        "lds_bank_conflict_share_of_lds_cycles": lds_conf / lds_act,
    }
    try:  # against what the card was measured to issue (independent streams, 4 waves per SIMD as in k_short: 16 waves per CU)
        rows = json.load(open(issue_json))["rows"]
        at = lambda op, w: [r for r in rows if r["op"] == op and r["waves_per_simd"] == w][0]["instr_per_us_per_simd"]
        L = out["limiter"]
        ms = kernel_ms if kernel_ms else busy / 2.4e6  # (SQ_BUSY_CYCLES at ~2.4 GHz when no event time is given)
        ach = valu / 1024.0 / (ms * 1e3)
        L["valu_instr_per_us_per_simd"] = ach
        L["measured_ceiling_instr_per_us_per_simd"] = {"v_add_u32_4_waves": at("add_u32", 4), "swar_mix_4_waves": at("swar_mix", 4), "swar_mix_8_waves": at("swar_mix", 8),
                                                        "source": issue_json + " (tools/micro/valu_issue.hip)"}
        L["valu_issue_frac_of_measured_mix_ceiling"] = ach / at("swar_mix", 4)
        L["valu_issue_frac_of_measured_add_ceiling"] = ach / at("add_u32", 4)
        L["kernel_ms_used"] = ms
    except Exception as e:  # noqa: BLE001
        out["limiter"]["measured_ceiling_error"] = str(e)
except KeyError:
    pass
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
