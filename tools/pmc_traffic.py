#!/usr/bin/env python3
"""profiles/*_traffic.json from the summary of the counter passes (tools/pmc_profile.sh -> summary.txt): HBM bytes per launch / per pair of
the fused DyGFormer kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts a 128-byte request as 64 bytes:
read bytes = 128 * TCC_EA0_RDREQ_128B + 64 * TCC_EA0_RDREQ_64B + 32 * ..._32B; WRITE_SIZE is exact, in KB), plus the matrix-pipe and wait
fractions.  usage: pmc_traffic.py SUMMARY.txt WORKLOAD WORKGROUPS PAIRS_PER_LAUNCH ALGORITHMIC_BYTES_PER_PAIR OUT.json [KERNEL_TRACE.csv]"""
import csv, json, re, sys
summary, workload, wgs, pairs, algo, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6]
m, n, on = {}, {}, False
for line in open(summary):
    h = re.match(r"-- (\S+): grid of (\d+) workgroups", line)
    if h:
        on = "k_dygformer_fused" in h.group(1) and int(h.group(2)) == wgs
        continue
    c = re.match(r"(\S+)\s+n=\s*(\d+)\s+mean=\s*([0-9.eE+-]+)", line)
    if on and c:
        m[c.group(1)], n[c.group(1)] = float(c.group(3)), int(c.group(2))
dur = []
if len(sys.argv) > 7:       # durations of the same launch shape from a kernel trace WITHOUT counters (the driver's command under --kernel-trace --stats)
    for r in csv.DictReader(open(sys.argv[7])):
        if "k_dygformer_fused" in r["Kernel_Name"] and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) == wgs:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    dur = sorted(dur)[:max(1, len(dur) // 2)]          # the faster half: launches after the clock ramp
rd = 128 * m["TCC_EA0_RDREQ_128B_sum"] + 64 * m["TCC_EA0_RDREQ_64B_sum"] + 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
wr = 1024 * m["WRITE_SIZE"]
d = {"source": "tools/pmc_profile.sh (rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_* / SQ_*) on bench.py "
               f"(CPU legs and secondary workloads off); the timed launch shape = {wgs} workgroups; summarised by tools/pmc_summary.py, tools/pmc_traffic.py",
     "workload": workload, "kernel": "k_dygformer_fused3<8>" if workload == "lastfm" else "k_dygformer_fused3<4>", "pairs_per_launch": pairs,
     "launches_averaged": n.get("WRITE_SIZE"),
     "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"], "TCC_EA0_RDREQ_128B": m["TCC_EA0_RDREQ_128B_sum"],
     "TCC_EA0_RDREQ_64B": m["TCC_EA0_RDREQ_64B_sum"], "TCC_EA0_RDREQ_32B": m.get("TCC_EA0_RDREQ_32B_sum", 0.0),
     "correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM): read bytes = 128*RDREQ_128B + 64*RDREQ_64B "
                   f"= {rd / 1e6:.1f} MB (2 x FETCH_SIZE = {2 * 1024 * m['FETCH_SIZE'] / 1e6:.1f} MB); WRITE_SIZE exact",
     "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_pair": (rd + wr) / pairs,
     "write_bytes_per_pair": wr / pairs, "algorithmic_bytes_per_pair": algo,
     "mfma_pipe_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128.0),        # as in profiles/r02_v4_traffic.json
     "mfma_f32_ops_per_launch": m.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
     "busy_cycles_per_launch": m["GRBM_GUI_ACTIVE"],
     "ms_per_launch_trace": (sum(dur) / len(dur) * 1e-6) if dur else None,
     "clock_GHz_under_load": (m["GRBM_GUI_ACTIVE"] / 8.0 / (sum(dur) / len(dur))) if dur else None,      # the counter sums the 8 XCDs' busy cycles
     "wave_cycles_waiting_frac": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], "wave_cycles_issue_stalled_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
     "lds_bank_conflict_frac": m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else None}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps({k: d[k] for k in ("workload", "kernel", "hbm_bytes_per_pair", "algorithmic_bytes_per_pair", "mfma_pipe_busy", "clock_GHz_under_load", "ms_per_launch_trace", "wave_cycles_waiting_frac")}))
