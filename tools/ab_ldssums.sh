for i in 1 2 3; do
LEAF_CMP_ALGO=4 LEAF_LDS_SUMS=1 python tools/compare_builds.py cur:-DLEAF_TOOLS=1 2>&1 | tail -1
LEAF_CMP_ALGO=4 LEAF_LDS_SUMS=0 python tools/compare_builds.py cur:-DLEAF_TOOLS=1 2>&1 | tail -1
done
