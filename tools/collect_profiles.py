#!/usr/bin/env python3
"""Turn the raw rocprofv3 outputs merged under gpurun_out/<dir>/ into the committed summaries under profiles/<round>/.

    python tools/collect_profiles.py gpurun_out/prof6 profiles/r01

Expects <dir>/stats/bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py) and the three PMC passes
<dir>/pmc_{fetch,write,sq}/w_counter_collection.csv of tools/profile_workload.py.  FETCH_SIZE / WRITE_SIZE are
corrected as MI355X_MICROARCH.md section HBM prescribes: calibrated on sqmod_kernel, whose byte count is known and
whose access pattern (coalesced 4-byte-per-lane) matches the fused kernel's."""
import collections
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, "bench_kernel_stats.csv"))


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        for key in ("leaf_fused", "leaf_fft_kernel", "leaf_fft_wg_kernel", "finalize", "sqmod", "conv_staged", "pool_staged",
                    "fused_prep", "fft_prep"):
            if key in r["Kernel_Name"]:
                d[key][r["Counter_Name"]].append((float(r["Counter_Value"]),
                                                  int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return d


out, dur = {}, collections.defaultdict(list)
for f in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for k, v in load(os.path.join(src, f, "w_counter_collection.csv")).items():
        for c, vals in v.items():
            out.setdefault(k, {})[c] = {"n": len(vals), "mean": sum(x[0] for x in vals) / len(vals)}
            dur[k].extend(x[1] for x in vals)
true_r, true_w = 64 * 80 * 16000 * 4, 64 * 40 * 16000 * 4       # sqmod_kernel of profile_workload.py (B=64)
fr = true_r / (out["sqmod"]["FETCH_SIZE"]["mean"] * 1024)
fw = true_w / (out["sqmod"]["WRITE_SIZE"]["mean"] * 1024)
fu = out["leaf_fused"]
rd, wr = fu["FETCH_SIZE"]["mean"] * 1024 * fr, fu["WRITE_SIZE"]["mean"] * 1024 * fw
d_us = sum(dur["leaf_fused"]) / len(dur["leaf_fused"]) / 1e3
cycles = fu["GRBM_GUI_ACTIVE"]["mean"] / 8
summary = {
    "source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ+GRBM set; each its own run, --kernel-trace only) on "
              "tools/profile_workload.py, one MI355X",
    "units": "FETCH_SIZE/WRITE_SIZE in KiB; gfx950 correction factors calibrated on sqmod_kernel (known byte count, same "
             "coalesced 4-byte-per-lane pattern) per MI355X_MICROARCH.md section HBM",
    "calibration": {"fetch_factor": round(fr, 4), "write_factor": round(fw, 4)},
    "leaf_fused_kernel": {
        "fetch_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
        "avg_duration_us_under_pmc": round(d_us, 1),
        "effective_clock_GHz": round(cycles / d_us / 1e3, 3),
        "mfma_instructions": fu["SQ_INSTS_MFMA"]["mean"],
        "valu_instructions_incl_mfma": fu["SQ_INSTS_VALU"]["mean"],
        "lds_instructions": fu["SQ_INSTS_LDS"]["mean"],
        "matrix_pipe_busy_fraction": round(fu["SQ_INSTS_MFMA"]["mean"] * 32 / 1024 / cycles, 4),
        "lds_bank_conflict_cycles": fu["SQ_LDS_BANK_CONFLICT"]["mean"],
    },
    "leaf_fused_kernel_hbm_bytes_per_launch": round(rd + wr),
    "algorithmic_bytes_per_launch": 800 * 25600,
    "counters": out,
}
traffic = {"leaf_fused_kernel_hbm_bytes_per_launch": round(rd + wr), "from": os.path.join(dst, "pmc_summary.json")}
if "leaf_fft_kernel" in out and "FETCH_SIZE" in out["leaf_fft_kernel"]:
    ff = out["leaf_fft_kernel"]
    frd, fwr = ff["FETCH_SIZE"]["mean"] * 1024 * fr, ff["WRITE_SIZE"]["mean"] * 1024 * fw
    fd_us = sum(dur["leaf_fft_kernel"]) / len(dur["leaf_fft_kernel"]) / 1e3
    summary["leaf_fft_kernel"] = {"fetch_bytes_per_launch": round(frd), "write_bytes_per_launch": round(fwr),
                                  "avg_duration_us_under_pmc": round(fd_us, 1),
                                  "valu_instructions": ff.get("SQ_INSTS_VALU", {}).get("mean"),
                                  "lds_instructions": ff.get("SQ_INSTS_LDS", {}).get("mean"),
                                  "lds_bank_conflict_cycles": ff.get("SQ_LDS_BANK_CONFLICT", {}).get("mean"),
                                  "wave_wait_fraction": (ff["SQ_WAIT_ANY"]["mean"] / ff["SQ_WAVE_CYCLES"]["mean"])
                                  if "SQ_WAVE_CYCLES" in ff else None}
    summary["leaf_fft_kernel_hbm_bytes_per_launch"] = round(frd + fwr)
    traffic["leaf_fft_kernel_hbm_bytes_per_launch"] = round(frd + fwr)
if "leaf_fft_wg_kernel" in out and "FETCH_SIZE" in out["leaf_fft_wg_kernel"]:
    # the workgroup-per-block kernel (round 2 default): traffic + the VALU issue fraction the bench line quotes
    wg = out["leaf_fft_wg_kernel"]
    wrd, wwr = wg["FETCH_SIZE"]["mean"] * 1024 * fr, wg["WRITE_SIZE"]["mean"] * 1024 * fw
    w_us = sum(dur["leaf_fft_wg_kernel"]) / len(dur["leaf_fft_wg_kernel"]) / 1e3
    w_cycles = wg["GRBM_GUI_ACTIVE"]["mean"] / 8
    valu = wg.get("SQ_INSTS_VALU", {}).get("mean")
    issue = round(valu * 2 / 1024 / w_cycles, 4) if valu else None     # wave64 VALU = 2 cycles on a SIMD-32, 1024 SIMDs
    summary["leaf_fft_wg_kernel"] = {"fetch_bytes_per_launch": round(wrd), "write_bytes_per_launch": round(wwr),
                                     "avg_duration_us_under_pmc": round(w_us, 1),
                                     "effective_clock_GHz": round(w_cycles / w_us / 1e3, 3),
                                     "valu_instructions": valu, "valu_issue_fraction": issue,
                                     "mfma_instructions": wg.get("SQ_INSTS_MFMA", {}).get("mean"),
                                     "lds_instructions": wg.get("SQ_INSTS_LDS", {}).get("mean"),
                                     "lds_bank_conflict_cycles": wg.get("SQ_LDS_BANK_CONFLICT", {}).get("mean"),
                                     "wave_wait_fraction": (wg["SQ_WAIT_ANY"]["mean"] / wg["SQ_WAVE_CYCLES"]["mean"])
                                     if "SQ_WAVE_CYCLES" in wg else None}
    summary["leaf_fft_wg_kernel_hbm_bytes_per_launch"] = round(wrd + wwr)
    traffic["leaf_fft_wg_kernel_hbm_bytes_per_launch"] = round(wrd + wwr)
    traffic["leaf_fft_wg_kernel_valu_issue_frac"] = issue
if "leaf_fft_kernel" in out and "SQ_INSTS_VALU" in out["leaf_fft_kernel"] and "GRBM_GUI_ACTIVE" in out["leaf_fft_kernel"]:
    ff = out["leaf_fft_kernel"]
    traffic["leaf_fft_kernel_valu_issue_frac"] = round(ff["SQ_INSTS_VALU"]["mean"] * 2 / 1024 / (ff["GRBM_GUI_ACTIVE"]["mean"] / 8), 4)
# round 3: the opt-in streaming finalize and BASELINE configs[2]'s kernel, from their own passes (same calibration factors)
def extra(prefix, key, label, alg_bytes):
    acc, durs = {}, []
    for f in ("fetch", "write", "sq"):
        path = os.path.join(src, f"pmc_{prefix}_{f}", "w_counter_collection.csv")
        if not os.path.exists(path):
            return
        for r in csv.DictReader(open(path)):
            if key in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if "FETCH_SIZE" not in acc or "GRBM_GUI_ACTIVE" not in acc:
        return
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    rdb, wrb = mean["FETCH_SIZE"] * 1024 * fr, mean["WRITE_SIZE"] * 1024 * fw
    cyc = mean["GRBM_GUI_ACTIVE"] / 8
    summary[label] = {"kernel": key, "fetch_bytes_per_launch": round(rdb), "write_bytes_per_launch": round(wrb),
                      "hbm_bytes_per_launch": round(rdb + wrb), "algorithmic_bytes_per_launch": alg_bytes,
                      "traffic_ratio": round((rdb + wrb) / alg_bytes, 3),
                      "avg_duration_us_under_pmc": round(sum(durs) / len(durs) / 1e3, 1),
                      "effective_clock_GHz": round(cyc / (sum(durs) / len(durs) / 1e3) / 1e3, 3),
                      "valu_instructions": mean.get("SQ_INSTS_VALU"), "lds_instructions": mean.get("SQ_INSTS_LDS"),
                      "valu_issue_fraction": round(mean["SQ_INSTS_VALU"] * 2 / 1024 / cyc, 4) if "SQ_INSTS_VALU" in mean else None,
                      "wave_wait_fraction": round(mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"], 4) if "SQ_WAVE_CYCLES" in mean else None}


extra("stream", "leaf_fft_wg_kernel", "leaf_fft_wg_kernel_streaming_finalize", 800 * 25600)
extra("cfg2", "leaf_fft_wg4k_kernel", "leaf_fft_wg4k_kernel_cfg2", 1600 * 128 * 500)
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(os.path.dirname(dst.rstrip("/")) or ".", "traffic.json"), "w"))
print(json.dumps({k: summary[k] for k in summary if k not in ("counters", "source", "units")}, indent=1))
