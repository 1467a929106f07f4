#!/bin/bash
# rocprofv3 kernel stats of the cfg1 forward with band tasks (tools/debug_band.py runs band, then full)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
D=gpurun_out/prof_band
rm -rf "$D"; mkdir -p "$D"
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o w -- python tools/debug_band.py > "$D/run.log" 2>&1
find "$D" -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-8 {} | head -12'
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_band/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
# consecutive launches: band phase first (70 calls), then full (70 calls)
seq = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
wg = [d for n, d in seq if "leaf_fft_wg_kernel" in n]
bt = [d for n, d in seq if "fft_prep_band" in n]
pr = [d for n, d in seq if "fft_prep_kernel" in n]
h = len(wg) // 2
print("wg kernel, first half (band) avg us:", sum(wg[20:h]) / max(1, len(wg[20:h])) / 1e3, " second half (full):", sum(wg[h + 20:]) / max(1, len(wg[h + 20:])) / 1e3)
print("band_tables avg us:", sum(bt) / max(1, len(bt)) / 1e3, "n", len(bt), " fft_prep avg us:", sum(pr) / max(1, len(pr)) / 1e3)
PY
