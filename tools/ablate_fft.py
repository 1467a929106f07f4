#!/usr/bin/env python3
"""Ablation of the FFT kernel at BASELINE configs[1] (GPU box): builds with one phase removed each (results are wrong
by construction -- timing only) next to the full kernel.  The time that disappears with a phase is what that phase
costs at the SIMD level (what a per-wave phase trace cannot show, because the two waves of a SIMD overlap)."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
specs = ["full:-DLEAF_FFT_ABLATE=0", "no_spectrum_loads:-DLEAF_FFT_ABLATE=1", "spectrum_one_row_L1:-DLEAF_FFT_ABLATE=32",
         "no_inverse_fft:-DLEAF_FFT_ABLATE=2", "no_pool_fma_lds:-DLEAF_FFT_ABLATE=4", "no_row_dma:-DLEAF_FFT_ABLATE=16",
         "no_partial_store:-DLEAF_FFT_ABLATE=64", "no_loads_no_dma:-DLEAF_FFT_ABLATE=17"]
sys.exit(subprocess.call([sys.executable, os.path.join(REPO, "tools", "compare_builds.py")] + specs + sys.argv[1:]))
