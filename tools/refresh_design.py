#!/usr/bin/env python3
"""Generate the measured block of DESIGN.md (between the `measured:begin` / `measured:end` markers of section 6) from the
committed evidence under profiles/<round>/ (the round named by the one-line file profiles/ROUND) -- so that no figure in it is typed by hand.

    python tools/refresh_design.py            # rewrite the block in place
    python tools/refresh_design.py --check    # exit 1 if DESIGN.md's block differs from what the files say (CPU test)
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = open(os.path.join(ROOT, "profiles", "ROUND")).read().strip()
R = os.path.join(ROOT, "profiles", ROUND)
BEGIN, END = "<!-- measured:begin -->", "<!-- measured:end -->"


def j(name):
    return json.load(open(os.path.join(R, name)))


def jl(name):
    return [json.loads(l) for l in open(os.path.join(R, name)) if l.startswith("{")]


def block():
    out = []
    pmc = j("pmc_summary.json")["configs"]
    out.append("**Bench lines** (`python bench.py --config cfgK --steps 50 --warmup 10`, N = 1; `bench_cfgK_n1.json`) and the PMC passes of "
               "the same workloads (`pmc_summary.json`):\n")
    out.append("| config | workload per GPU | ms / step | frames/s | dominant kernel | kernel ms | tasks per block (2048 / 8×256 / 4×512) | executed GFLOP | "
               "`frac` | of practical roof | VALU issue (PMC) | HBM traffic / algorithmic | CPU oracle frames/s (threads) |")
    out.append("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for c in ("cfg1", "cfg2", "cfg3", "cfg4"):
        b = j(f"bench_{c}_n1.json")
        r, cfg, cb = b["roofline"], b["config"], b["cpu_baseline"]
        pr = r.get("frac_of_practical_roof")
        bt = (r.get("band_tasks") or {}).get("tasks_per_block")
        if bt and "4096_point_filter" in bt:
            tasks = f"{bt['4096_point_filter']} on 4096 points / – / {bt['four_filters_on_512_points']}"
        else:
            tasks = f"{bt['2048_point_filter']} / {bt['eight_filters_on_256_points']} / {bt['four_filters_on_512_points']}" if bt else "all filters on full transforms"
        out.append(f"| {c} | {cfg['clips_per_gpu']} × {cfg['samples_per_clip']} samples, {cfg['io_dtype']} | {b['ms_per_step']:.4f} | "
                   f"{b['value'] / 1e6:.1f} M | `{r['kernel']}` | {r['kernel_ms']:.4f} | {tasks} | {r['executed_flops_per_launch'] / 1e9:.2f} | {r['frac']:.3f} | "
                   f"{('%.2f' % pr) if pr is not None else '–'} | {pmc[c]['valu_issue_frac']:.3f} | {pmc[c]['hbm_bytes_per_launch'] / 1e6:.1f} MB / "
                   f"{pmc[c]['algorithmic_bytes_per_launch'] / 1e6:.1f} MB = {pmc[c]['traffic_ratio']:.2f}× | {cb['value'] / 1e3:.1f} k ({cb['cores']}) |")
    b1 = j("bench_cfg1_n1.json")
    c0 = b1["cpu_baseline"].get("cfg0")
    if c0:
        out.append(f"\nBASELINE configs[0] (default Leaf, batch 4 × 1 s) on the host CPU: {c0['value'] / 1e3:.1f} k frames/s on {c0['cores']} threads; "
                   f"on the GPU: the one-launch kernel below.")
    rows = list(csv.DictReader(open(os.path.join(R, "bench_kernel_stats.csv"))))
    dom = next(r for r in rows if "leaf_fft_wg_kernel<401, 160, 12, false>" in r["Name"])
    out.append(f"\n`rocprofv3 --kernel-trace --stats` of the cfg1 bench command (`bench_kernel_stats.csv`): `leaf_fft_wg_kernel<401,160,12,false>` "
               f"{dom['Calls']} launches, average {float(dom['AverageNs']) / 1e3:.1f} µs (min {float(dom['MinNs']) / 1e3:.1f}) against "
               f"{b1['roofline']['kernel_ms'] * 1e3:.1f} µs from HIP events in the bench line.")
    roof = json.load(open(os.path.join(ROOT, "profiles", "valu_roof.json")))
    bw = roof["frac_of_peak_by_waves"]
    out.append(f"\nPractical VALU roof (`ubench_valu.txt`, `profiles/valu_roof.json`): the filter-task instruction mix issues at "
               f"{bw['1']:.3f} / {bw['2']:.3f} / {bw['3']:.3f} / {bw['4']:.3f} of the 157.3 TF peak at 1 / 2 / 3 / 4 waves per SIMD; the kernel runs three.")
    out.append("\n**Small batches** (`latency_breakdown.jsonl`: back-to-back eager calls of the dispatcher op, µs per call; device-bound):\n")
    out.append("| B × 1 s | one launch (`LEAF_ALGO_FFT_SMALL`, what AUTO runs) | three launches (per-wave kernel) |")
    out.append("|---|---|---|")
    lat = jl("latency_breakdown.jsonl")
    for B in sorted({r["B"] for r in lat}):
        one = next(r for r in lat if r["B"] == B and r["path"] == "one_launch")
        three = next(r for r in lat if r["B"] == B and r["path"] == "three_launches")
        out.append(f"| {B} | {one['dispatcher_op_loop_us']:.1f} | {three['dispatcher_op_loop_us']:.1f} |")
    out.append("\n**Training step** (`backward_timing.txt`, ms; `grad_out resident` = the output gradient already on the device, as in training):\n")
    out.append("```")
    out += [l.rstrip() for l in open(os.path.join(R, "backward_timing.txt")) if l.startswith("B=")]
    out.append("```")
    out.append("\n**All configs, median [p10, p90] of 50 event-timed module calls** (`configs_1gpu.jsonl`):\n")
    out.append("| workload | algo | ms | frames/s | frac of fp32 VALU peak (whole forward) |")
    out.append("|---|---|---|---|---|")
    for r in jl("configs_1gpu.jsonl"):
        if "config" not in r:
            continue
        out.append(f"| {r['config']} | {r['algo']} | {r['ms_median']:.4f} [{r['ms_p10']:.4f}, {r['ms_p90']:.4f}] | {r['frames_per_s'] / 1e6:.1f} M | "
                   f"{r['frac_of_fp32_valu_peak']:.3f} |")
    bc = [l.strip() for l in open(os.path.join(R, "band_check.txt")) if l.startswith("cfg") or l.startswith("worst")]
    out.append("\n**Band tasks against full transforms, same box** (`band_check.txt`: the module call with and without `LEAF_ALGO_FULL_TRANSFORMS`):\n")
    out.append("```")
    out += bc
    out.append("```")
    fz = [l[l.index("band fuzz"):].strip() for l in open(os.path.join(R, "band_fuzz.txt")) if "band fuzz" in l]   # (pytest -s: progress dots precede the line)
    if fz:
        wo = max(float(l.split("worst vs oracle ")[1].split(",")[0]) for l in fz)
        wf = max(float(l.split("full transforms ")[1]) for l in fz)
        out.append(f"\nSeeded (μ, σ, pooling width, signal) fuzz of the band choice (`band_fuzz.txt`, {len(fz)} seeds × 3–4 cases, both block lengths, pooling biases 0.02 … 3 and −50): worst error against the fp64 "
                   f"oracle {wo:.2e}, worst difference to the full-transform path {wf:.2e} (north star: 1e-4).")
    log = open(os.path.join(R, "pytest_gpu.log")).read().strip().splitlines()
    passed = next(l.strip() for l in log if " passed" in l)
    smoke = next(l.strip() for l in log if l.startswith("smoke ok:"))
    out.append(f"\nGPU test suite on the same box: `{passed}`; `{smoke[:200]}`.")
    return "\n".join(out)


def main():
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
    new = "\n" + block() + "\n"
    if "--check" in sys.argv:
        if s[a:b] != new:
            import difflib
            sys.stdout.writelines(list(difflib.unified_diff(s[a:b].splitlines(True), new.splitlines(True), "DESIGN.md", "profiles/" + ROUND))[:40])
            print(f"DESIGN.md section 6 does not match profiles/{ROUND}: run tools/refresh_design.py")
            return 1
        print(f"DESIGN.md section 6 matches profiles/{ROUND}")
        return 0
    open(p, "w").write(s[:a] + new + s[b:])
    print(f"DESIGN.md: {len(s[:a] + new + s[b:])} bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
