set -u
D=gpurun_out/r03_pmc1
export TMPDIR=/tmp
rm -rf "$D"; mkdir -p "$D"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
    set -- $pass; name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D/pmc_$name" -o w -- python tools/profile_workload.py > /dev/null 2>&1
done
ls -R $D | head -30
