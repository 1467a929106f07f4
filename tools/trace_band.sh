#!/bin/bash
# phase stamps of the workgroup kernel with band tasks (tools/trace_wg.py, -DLEAF_TRACE=<level> build on the GPU box)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
LEAF_TRACE_LEVEL=${1:-1} python tools/trace_wg.py > gpurun_out/trace_band_l${1:-1}${2:-}.txt 2>&1
tail -5 gpurun_out/trace_band_l${1:-1}${2:-}.txt
