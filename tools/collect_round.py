#!/usr/bin/env python3
"""A round's evidence: turn the raw outputs merged under gpurun_out/<dir>/ into the committed summaries under profiles/<round>/ and
profiles/traffic.json (what bench.py reports as roofline.traffic / valu_issue_frac_pmc / valu_flops_upper_bound_pmc, one entry per
BASELINE config).  One tool for every round (rounds 4 and 5 had a clone each: VERDICT r5 weak #11).

    python tools/collect_round.py r06 pmc   gpurun_out/<dir>     # PMC passes of tools/pmc_traffic.sh (modes cfg1 cfg2 cfg3 cfg4)
    python tools/collect_round.py r06 files gpurun_out/<dir> f1 f2 ...   # copy named result files as they are

FETCH_SIZE / WRITE_SIZE are corrected as MI355X_MICROARCH.md section HBM prescribes: calibrated on sqmod_kernel of the cfg1
pass (tools/profile_workload.py runs it at B = 64: 327.68 MB read, 163.84 MB written, the coalesced 4-byte-per-lane pattern of
the fused kernels)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1]
DST = os.path.join(ROOT, "profiles", ROUND)
os.makedirs(DST, exist_ok=True)
mode, src = sys.argv[2], sys.argv[3]

if mode == "files":
    for f in sys.argv[4:]:
        shutil.copy(os.path.join(src, f), os.path.join(DST, os.path.basename(f)))
        print("copied", f)
    sys.exit(0)

CONFIGS = {   # mode -> (dominant kernel, clips, samples, algorithmic bytes per launch = (hop + F) * io bytes * frames)
    "cfg1": ("leaf_fft_wg_kernel", 256, 16000, 800 * 256 * 100),
    "cfg2": ("leaf_fft_wg4k_kernel", 128, 160000, 1600 * 128 * 500),
    "cfg3": ("leaf_fft_wg_kernel", 512, 16000, 800 * 512 * 100),
    "cfg4": ("leaf_fft_wg_kernel", 256, 160000, 400 * 256 * 1000),
}


def counters(cfg, pass_name, key):
    files = glob.glob(os.path.join(src, f"pmc_{cfg}_{pass_name}", "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    if not files:
        return acc
    for r in csv.DictReader(open(files[0])):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return acc


def mean(v):
    return sum(x[0] for x in v) / len(v)


cal_r = counters("cfg1", "fetch", "sqmod")["FETCH_SIZE"]
cal_w = counters("cfg1", "write", "sqmod")["WRITE_SIZE"]
true_r, true_w = 64 * 80 * 16000 * 4, 64 * 40 * 16000 * 4
fr = true_r / (mean(cal_r) * 1024) if cal_r else 2.0
fw = true_w / (mean(cal_w) * 1024) if cal_w else 1.0
summary = {"source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ + GRBM set; each its own run, --kernel-trace only) on "
                     f"tools/profile_workload.py <mode>, one MI355X, {ROUND} (tools/pmc_traffic.sh)",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction factors calibrated on sqmod_kernel (known byte count)",
           "calibration": {"fetch_factor": round(fr, 4), "write_factor": round(fw, 4), "calibrated": bool(cal_r and cal_w)},
           "configs": {}}
tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
traffic["from"] = f"profiles/{ROUND}/pmc_summary.json"
traffic.setdefault("configs", {})
for cfg, (kernel, clips, samples, alg) in CONFIGS.items():
    f, w, sq = counters(cfg, "fetch", kernel), counters(cfg, "write", kernel), counters(cfg, "sq", kernel)
    if "FETCH_SIZE" not in f or "WRITE_SIZE" not in w:
        continue
    rd, wr = mean(f["FETCH_SIZE"]) * 1024 * fr, mean(w["WRITE_SIZE"]) * 1024 * fw
    e = {"kernel": kernel, "clips": clips, "samples": samples, "launches": len(f["FETCH_SIZE"]),
         "fetch_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr),
         "algorithmic_bytes_per_launch": alg, "traffic_ratio": round((rd + wr) / alg, 3)}
    if "GRBM_GUI_ACTIVE" in sq:
        cyc = mean(sq["GRBM_GUI_ACTIVE"]) / 8
        us = sum(x[1] for x in sq["GRBM_GUI_ACTIVE"]) / len(sq["GRBM_GUI_ACTIVE"]) / 1e3
        e.update({"avg_duration_us_under_pmc": round(us, 1), "effective_clock_GHz": round(cyc / us / 1e3, 3),
                  "valu_instructions": mean(sq["SQ_INSTS_VALU"]), "lds_instructions": mean(sq["SQ_INSTS_LDS"]),
                  "mfma_instructions": mean(sq["SQ_INSTS_MFMA"]),
                  "valu_issue_frac": round(mean(sq["SQ_INSTS_VALU"]) * 2 / 1024 / cyc, 4),       # wave64 VALU = 2 cycles, 1024 SIMDs
                  "wave_wait_fraction": round(mean(sq["SQ_WAIT_ANY"]) / mean(sq["SQ_WAVE_CYCLES"]), 4),
                  "lds_bank_conflict_cycles": mean(sq["SQ_LDS_BANK_CONFLICT"])})
    summary["configs"][cfg] = e
    traffic["configs"][cfg] = {"kernel": kernel, "clips": clips, "samples": samples, "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
                               "valu_issue_frac": e.get("valu_issue_frac"), "valu_instructions": e.get("valu_instructions")}
    if cfg == "cfg1":                      # the comparison kernels of the bench line's roofline_other_algo, from the same pass
        traffic["leaf_fft_wg_kernel_hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
        traffic["leaf_fft_wg_kernel_valu_issue_frac"] = e.get("valu_issue_frac")
        for other in ("leaf_fft_kernel", "leaf_fused_kernel"):
            fo, wo, so = counters(cfg, "fetch", other), counters(cfg, "write", other), counters(cfg, "sq", other)
            if "FETCH_SIZE" in fo and "WRITE_SIZE" in wo:
                traffic[other + "_hbm_bytes_per_launch"] = round(mean(fo["FETCH_SIZE"]) * 1024 * fr + mean(wo["WRITE_SIZE"]) * 1024 * fw)
            if "SQ_INSTS_VALU" in so and other == "leaf_fft_kernel":
                traffic[other + "_valu_issue_frac"] = round(mean(so["SQ_INSTS_VALU"]) * 2 / 1024 / (mean(so["GRBM_GUI_ACTIVE"]) / 8), 4)
json.dump(summary, open(os.path.join(DST, "pmc_summary.json"), "w"), indent=1)
json.dump(traffic, open(tpath, "w"))
print(json.dumps(summary["configs"], indent=1))
