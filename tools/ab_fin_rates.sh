#!/bin/bash
# A/B on one box: in-kernel tail finalize (LEAF_FIN_FUSED unset) against main kernel + row kernel (LEAF_FIN_FUSED=0) at the
# sample rates whose forward takes the run-time-geometry / 4096-sample kernels.  Needs the tools variant (-DLEAF_TOOLS=1).
for sr in 11025 22050 32000 44100 48000; do
  for i in 1 2; do
    echo "sr=$sr tail:  $(LEAF_CMP_SR=$sr LEAF_CMP_ALGO=4 timeout 120 python tools/compare_builds.py cur:-DLEAF_TOOLS=1 2>&1 | tail -1)"
    echo "sr=$sr rowk:  $(LEAF_CMP_SR=$sr LEAF_CMP_ALGO=4 LEAF_FIN_FUSED=0 timeout 120 python tools/compare_builds.py cur:-DLEAF_TOOLS=1 2>&1 | tail -1)"
  done
done
