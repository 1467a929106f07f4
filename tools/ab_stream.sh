for i in 1 2 3; do
LEAF_CMP_ALGO=4 LEAF_WG_STREAM=1 python tools/compare_builds.py stream:-DLEAF_TOOLS=1 2>&1 | tail -1
LEAF_CMP_ALGO=4 LEAF_WG_STREAM=0 python tools/compare_builds.py stream:-DLEAF_TOOLS=1 2>&1 | tail -1
done
