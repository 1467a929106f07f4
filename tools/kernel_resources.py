#!/usr/bin/env python3
"""Generated resource table of every kernel in libleaf_hip.so.

    python tools/kernel_resources.py [--out profiles/<round>/kernel_resources.csv] [--check]

Compiles each translation unit with the library's own flags plus -Rpass-analysis=kernel-resource-usage (objects are
discarded), parses the remarks into one CSV row per kernel (VGPRs, AGPRs, SGPRs, scratch bytes per lane, occupancy in
waves per SIMD, static LDS) and, with --check, fails when a kernel has scratch that is not explained in ALLOWED_SCRATCH
below.  DESIGN.md quotes the CSV instead of hand-typed figures.  Needs hipcc, not a GPU.
"""
import argparse
import concurrent.futures as cf
import csv
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from leaf_pytorch_amd import _native  # noqa: E402

# kernel-name regex -> why its scratch is accepted (everything else must be scratch-free)
ALLOWED_SCRATCH = {
    r"dtaps_mfma_kernel": "tap-gradient GEMM of the MFMA backward (short windows / K > 2049 only)",
    r"leaf_fft_kernelILi0ELi0E": "per-wave kernel, run-time geometry (small batches of non-LEAF windows): 12-32 B/lane in the frame switch",
    r"leaf_fft_wgg_bwd_kernelILi12ELi\d+ELb1ELb1E": "dL/dx on the workgroup structure (windows without a static instance, every batch size): 250-280 B/lane, all but "
                                                     "~20 spill / reload instructions per (block, filter) task of ~6 000 inside the branch the "
                                                     "wave that adds a block's LAST filter takes (wg_dx_finish: once per block)",
    r"leaf_fft_wg_bwd_kernelILi401ELi160ELi12ELb0E": "static 16 kHz backward with band tasks (leaf_band_bwd.hpp): 92 B/lane = 22 launch-invariant values (the "
                                                      "butterfly constants of the band network, hoisted out of the task loop) stored ONCE in the kernel's prologue; "
                                                      "17 reloads per band task of ~4 000 instructions, none in the full-transform task",
    r"leaf_fft_wg_bwd_kernelILi401ELi160ELi12ELb1E": "static 16 kHz backward with dL/dx and band tasks: 112 B/lane = the 22 launch-invariant butterfly constants of the band "
                                                      "network stored once in the prologue (as in the kernel without dL/dx) + the ~12 spill stores per full-transform task "
                                                      "of the other static dL/dx kernels",
    r"leaf_fft_wg_bwd_kernelILi\d+ELi\d+ELi12ELb1E": "dL/dx on the workgroup structure, static LEAF geometries: 44-96 B/lane, ~12 spill stores "
                                                       "per (block, filter) task of ~3 000 instructions, the rest in wg_dx_finish (once per block)",
    r"leaf_fft_blk_bwd_dx_kernel": "dL/dx at the static LEAF geometries (small batches; K = 801 at every batch): 32-64 B/lane outside the filter loop (the extra transform's "
                                   "temporaries); 0.15 ms of the 0.88 ms training step with dL/dx",
    r"leaf_fft_wgg4k_bwd_kernelILi12ELi7ELb1E": "static 32 kHz instance of the 4096-sample backward: 12 B/lane = two launch-invariant values "
                                                "stored once, reloaded three times per (block, filter) task of ~8 000 instructions",
    r"leaf_fft_wgg4k_bwd_kernel": "parameter gradients at 44.1 / 48 kHz (4096-sample plan): 76-84 B/lane around the half-transform "
                                  "hand-over = 11 spill stores + ~20 reloads per (block, filter) task of ~14 000 instructions (< 0.3 %)",
    r"leaf_fft_wg_kernel.*Lb1": "streaming finalize (what AUTO runs at BASELINE configs[3] / [4], and LEAF_ALGO_STREAM_FINALIZE elsewhere): 48 B/lane = the "
                                "FinCoef argument (8 floats, passed by reference) and the call frame of the OUT-OF-LINE point function that serves PCEN off "
                                "and filters with delta <= 0 -- kept out of line so that the kernel stays inside the instruction cache; "
                                "two 16-byte stores + three loads per finalized 64-filter group (once per block, one wave), none in the task loop",
}


def demangle(names):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else ["c++filt"],
                         input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [d.replace("(anonymous namespace)::", "") for d in out[:len(names)]]


def analyse(src):
    csrc = os.path.dirname(_native.SRC_PATH)
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize",
               "-fPIC", "-I", _native.INCLUDE_DIR, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o",
               os.path.join(tmp, "x.o")] + os.environ.get("LEAF_HIPCC_EXTRA", "").split()
        err = subprocess.run(cmd, capture_output=True, text=True, cwd=csrc).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {"tu": os.path.basename(src), "mangled": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", open(os.path.join(REPO, "profiles", "ROUND")).read().strip(), "kernel_resources.csv"))
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    units = _native._translation_units(os.path.dirname(_native.SRC_PATH))
    with cf.ThreadPoolExecutor(max_workers=os.cpu_count()) as ex:
        rows = [r for rs in ex.map(analyse, units) for r in rs]
    for r, name in zip(rows, demangle([r["mangled"] for r in rows])):
        r["kernel"] = re.sub(r"\(.*$", "", name).replace("void ", "")
    rows.sort(key=lambda r: (r["tu"], r["kernel"]))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    cols = ["tu", "kernel", "VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize", "VGPRs Spill", "Occupancy", "LDS Size"]
    with open(args.out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["translation_unit", "kernel", "vgprs", "agprs", "sgprs", "scratch_bytes_per_lane", "vgpr_spills", "occupancy_waves_per_simd",
                    "static_lds_bytes", "scratch_note"])
        bad = []
        for r in rows:
            scratch = int(r.get("ScratchSize", "0"))
            note = ""
            if scratch:
                note = next((why for pat, why in ALLOWED_SCRATCH.items() if re.search(pat, r["mangled"])), "")
                if not note:
                    bad.append((r["kernel"], scratch))
            w.writerow([r.get(c, "") for c in cols] + [note])
    print(f"{len(rows)} kernels -> {args.out}")
    for k, s in bad:
        print(f"UNEXPLAINED SCRATCH: {k}: {s} bytes/lane")
    if args.check and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
