#!/usr/bin/env python3
"""Reduce the per-group csv files of tools/pmc_kernel.sh to one JSON: HBM traffic per launch (FETCH_SIZE doubled on
gfx950 for wide coalesced reads, MI355X_MICROARCH.md HBM section; WRITE_SIZE exact), MFMA-pipe utilisation, vector-ALU
instructions per MFMA, LDS bank-conflict share, kernel duration inside the counter passes.
usage: pmc_summary.py <dir> <flops per launch> <algorithmic bytes per launch> <peak TFLOP/s> [description]"""
import collections, csv, json, os, sys
d, flops, alg_bytes, peak = sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
desc = sys.argv[5] if len(sys.argv) > 5 else ""
ONLY = sys.argv[6] if len(sys.argv) > 6 else ""        # substring of the kernel name (when the csv files hold several)

def counters(name):
    p = os.path.join(d, f"pmc_{name}.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))     # counter -> dispatch -> sum
    if not os.path.exists(p):
        return {}
    for r in csv.DictReader(open(p)):
        if ONLY and ONLY not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: (sum(v.values()) / len(v), len(v)) for c, v in acc.items()}

def durations(name):
    p = os.path.join(d, f"trace_{name}.csv")
    if not os.path.exists(p):
        return None
    ds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(p))
          if not ONLY or ONLY in r["Kernel_Name"]]
    ds = ds[1:] if len(ds) > 1 else ds          # the first launch of a run is a warm-up
    return sum(ds) / len(ds) / 1e3 if ds else None

out = {"kernel": desc, "flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes}
f, w = counters("FETCH_SIZE"), counters("WRITE_SIZE")
if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
    fetch = f["FETCH_SIZE"][0] * 1024 * 2
    write = w["WRITE_SIZE"][0] * 1024
    out.update({"FETCH_SIZE_KB_raw": f["FETCH_SIZE"][0], "fetch_bytes_corrected_x2": fetch, "WRITE_SIZE_KB": w["WRITE_SIZE"][0],
                "write_bytes": write, "traffic_bytes_per_launch": fetch + write,
                "traffic_over_algorithmic": (fetch + write) / alg_bytes if alg_bytes else None,
                "launches_averaged": f["FETCH_SIZE"][1]})
m = counters("SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE")
if "GRBM_GUI_ACTIVE" in m:
    busy, act = m["SQ_VALU_MFMA_BUSY_CYCLES"][0], m["GRBM_GUI_ACTIVE"][0]
    dur = durations("SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE")
    out.update({"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE_sum_8_xcd": act,
                "mfma_pipe_utilisation": busy / 1024 / (act / 8),
                "mfma_util_note": "SQ_VALU_MFMA_BUSY_CYCLES summed over 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)",
                "duration_us_during_counter_pass": dur,
                "effective_clock_GHz_during_counter_pass": (act / 8) / (dur * 1e3) if dur else None})
i = counters("SQ_INSTS_VALU_SQ_INSTS_MFMA")
if "SQ_INSTS_MFMA" in i:
    valu, mfma = i["SQ_INSTS_VALU"][0], i["SQ_INSTS_MFMA"][0]
    out.update({"SQ_INSTS_MFMA": mfma, "SQ_INSTS_VALU_including_MFMA": valu,
                "valu_per_mfma": (valu - mfma) / mfma if mfma else None})
l = counters("SQ_LDS_BANK_CONFLICT_SQ_LDS_IDX_ACTIVE")
if "SQ_LDS_IDX_ACTIVE" in l and l["SQ_LDS_IDX_ACTIVE"][0]:
    out["lds_bank_conflict_share_of_lds_active_cycles"] = l["SQ_LDS_BANK_CONFLICT"][0] / l["SQ_LDS_IDX_ACTIVE"][0]
dur = durations("FETCH_SIZE")
out["duration_us_fetch_pass"] = dur
if dur:
    out["tflops_in_fetch_pass"] = flops / dur / 1e6
    out["frac_of_peak_in_fetch_pass"] = flops / dur / 1e6 / peak
    if "traffic_bytes_per_launch" in out:
        out["hbm_TBps_in_fetch_pass"] = out["traffic_bytes_per_launch"] / dur / 1e6
print(json.dumps(out, indent=1))
