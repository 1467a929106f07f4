#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant kernel of one workload mode of tools/profile_workload.py (cfg2 | cfg3 | cfg4 | "" = cfg1),
# each counter in its own rocprofv3 run (--kernel-trace only), into gpurun_out/$1/pmc_<mode>_{fetch,write}/.
#   bash tools/pmc_traffic.sh r04g cfg3 cfg4
set -u
export TMPDIR=/tmp
D=gpurun_out/$1; shift
mkdir -p "$D"
for mode in "$@"; do
    for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
        set -- $pass; name=$1; shift
        arg=$mode; [ "$mode" = "cfg1" ] && arg=""
        timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D/pmc_${mode}_$name" -o w -- python tools/profile_workload.py $arg > /dev/null 2>&1
    done
done
