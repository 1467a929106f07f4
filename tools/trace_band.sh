#!/bin/bash
# phase stamps of the workgroup kernel with band tasks (tools/trace_wg.py, -DLEAF_TRACE=1 build on the GPU box)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
python tools/trace_wg.py > gpurun_out/trace_band.txt 2>&1
tail -150 gpurun_out/trace_band.txt
