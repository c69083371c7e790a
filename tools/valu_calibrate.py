"""GPU box: summarise tools/valu_calibrate.sh — per one-instruction kernel: wave-instructions, launch duration in shader cycles
(GRBM_GUI_ACTIVE / 8 XCDs), hence TRUE issue cycles per wave-instruction per SIMD, the clock (cycles / kernel-trace duration) and
SQ_ACTIVE_INST_VALU per SQ_INSTS_VALU (the unit in which bench.py's roofline converts that counter into cycles)."""
import csv, glob, json, sys
root = sys.argv[1]
N_SIMD, N_XCD = 1024, 8


def counters(pattern):
    out = {}
    for f in glob.glob(root + "/" + pattern + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def durations():
    out = {}
    for f in glob.glob(root + "/trace_*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return out


sq, grbm, dur = counters("pmc_sq_*"), counters("pmc_grbm_*"), durations()
rows = []
for k in sorted(sq):
    if k not in grbm or k not in dur:
        continue
    # every kernel is launched twice (10 warm-up iterations, then 2000): the long launch is the last one of each list
    insts, active = sq[k]["SQ_INSTS_VALU"][-1], sq[k]["SQ_ACTIVE_INST_VALU"][-1]
    cycles = grbm[k]["GRBM_GUI_ACTIVE"][-1] / N_XCD
    rows.append({"kernel": k.split("(")[0], "SQ_INSTS_VALU": insts, "cycles": cycles, "seconds_kernel_trace": dur[k][-1],
                 "clock_GHz": round(cycles / dur[k][-1] / 1e9, 3),
                 "issue_cycles_per_inst_per_simd": round(cycles * N_SIMD / insts, 3),
                 "SQ_ACTIVE_INST_VALU_per_inst": round(active / insts, 3),
                 "SQ_ACTIVE_INST_VALU_x4_over_simd_cycles": round(active * 4 / (cycles * N_SIMD), 3),
                 "SQ_BUSY_CYCLES_over_cycles": round(sq[k]["SQ_BUSY_CYCLES"][-1] / cycles, 2) if "SQ_BUSY_CYCLES" in sq[k] else None})
print(json.dumps({"what": "one-instruction kernels, 5 waves per SIMD on every SIMD; each loop trip = 64 instructions of the named kind + s_add / s_cmp / s_cbranch",
                  "rows": rows}, indent=1))
