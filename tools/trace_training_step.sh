#!/bin/bash
# per-dispatch kernel durations of one training step (forward + backward at BASELINE configs[1] or the arguments of
# tools/profile_backward.py) from rocprofv3 --kernel-trace: the last 14 dispatches of the run
#   bash tools/trace_training_step.sh [B F sr secs]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf "$R/gpurun_out/bwdtrace"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/bwdtrace" -o b -- python "$R/tools/profile_backward.py" ${@:-256 40 16000 1} > /dev/null 2>&1
f=$(ls "$R"/gpurun_out/bwdtrace/*/b_kernel_trace.csv "$R"/gpurun_out/bwdtrace/b_kernel_trace.csv 2>/dev/null | head -1)
python - "$f" <<PY
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-14:]:
    print(f'{r["Kernel_Name"][:64]:64s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us')
PY
