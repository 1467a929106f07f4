#!/usr/bin/env python3
"""Copy the evidence of a tools/run_evidence.sh run into profiles/<round>/ and regenerate the two measured tables of DESIGN.md
(BASELINE configs, per-sample-rate) from the copied files.   usage: refresh_docs_tables.py gpurun_out/<run> profiles/r02"""
import json
import os
import shutil
import subprocess
import sys

src, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run([sys.executable, os.path.join(root, "tools", "collect_profiles.py"), src, dst], check=True, stdout=subprocess.DEVNULL)
for f in ("bench_n1.json", "bench_n2_dryrun_1gpu_gloo.json", "bench_n1_rccl_world1.json", "configs_1gpu.jsonl", "backward_timing.txt",
          "rates_1gpu.jsonl", "fft_vs_mfma.txt", "pmc_workgroup_kernels.json", "sweep_batch_wg.txt", "stage_times.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
for sr in (16000, 22050, 32000, 48000):
    with open(os.path.join(src, f"bwd_stats_{sr}", "b_kernel_stats.csv")) as fi, open(os.path.join(dst, f"training_step_kernel_stats_{sr}.csv"), "w") as fo:
        fo.writelines(fi.readlines()[:12])
with open(os.path.join(dst, "pytest_gpu.log"), "w") as fo:
    fo.writelines(open(os.path.join(src, "pytest_gpu.log")).readlines()[-2:])
    fo.write(open(os.path.join(src, "smoke.log")).read())

p = os.path.join(root, "DESIGN.md")
s = open(p).read()
rows = [json.loads(l) for l in open(os.path.join(dst, "configs_1gpu.jsonl")) if '"config"' in l]
keys = ["| cfg0 default, B = 4 × 1 s |", "| cfg1 default, B = 256 × 1 s, U(−1,1) |", "| cfg1, N(0,1) input |", "| cfg1, parameters perturbed ±10 % |",
        "| cfg2 80 filters / 32 kHz / 5 s (K = 801, hop 320), B = 128 |", "| cfg3 PCEN off, B = 512 × 1 s |", "| cfg4 10 s clips, B = 256, fp32 I/O |",
        "| cfg4 10 s clips, B = 256, bf16 I/O |", "| AudioSet cfg: 64 filters, B = 256 × 1 s |"]
for k, r in zip(keys, rows):
    a = s.index(k); b = s.index("\n", a); cols = s[a:b].split("|")
    cols[3] = f" {r['ms_median']:.3f} [{r['ms_p10']:.3f}, {r['ms_p90']:.3f}] "
    cols[4] = f" {r['frames_per_s'] / 1e6:.1f} M "
    if "frac_of_fp32_valu_peak" in r and len(cols) > 7:
        cols[6] = f" {r['frac_of_fp32_valu_peak']:.3f} " if r["algo"].startswith("fft") else " – "
    s = s[:a] + "|".join(cols) + s[b:]
rates = {json.loads(l)["sample_rate"]: json.loads(l) for l in open(os.path.join(dst, "rates_1gpu.jsonl"))}
lab = {8000: "| 8 kHz |", 11025: "| 11.025 kHz |", 16000: "| 16 kHz |", 22050: "| 22.05 kHz |", 24000: "| 24 kHz |", 32000: "| 32 kHz |",
       44100: "| 44.1 kHz |", 48000: "| 48 kHz |"}
for sr, k in lab.items():
    a = s.index(k); b = s.index("\n", a); cols = s[a:b].split("|")
    r = rates[sr]
    cols[4] = f" {r['forward_ms']:.3f} "; cols[5] = f" {r['forward_backward_ms']:.3f} "; cols[6] = f" {round(r['forward_frames_per_s'] / 1e6)} M "
    s = s[:a] + "|".join(cols) + s[b:]
open(p, "w").write(s)
d = json.loads(open(os.path.join(dst, "bench_n1.json")).read())
r = d["roofline"]
print("bench:", d["value"], d["ms_per_step"], "kernel_ms", r["kernel_ms"], "achieved", r["achieved"], "frac", r["frac"], "issue", r["valu_issue_frac_pmc"])
print(open(os.path.join(dst, "backward_timing.txt")).read())
