#!/bin/bash
# Round evidence on one MI355X (run through gpurun): GPU tests, smoke, the bench line, the rocprofv3 kernel-trace stats of
# the same bench command, the three PMC passes (each its own run, --kernel-trace only), the per-config table and the
# FFT-kernel phase trace.  Everything lands under gpurun_out/$1/; tools/collect_profiles.py turns it into profiles/<round>/.
#   gpurun --timeout 2400 -- 'bash tools/run_evidence.sh ev1'
set -u
D=gpurun_out/${1:-evidence}
export TMPDIR=/tmp
rm -rf "$D"; mkdir -p "$D"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 > "$D/pytest_gpu.log"; cat "$D/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$D/smoke.log"
timeout 600 python bench.py --steps 50 --warmup 10 2>/dev/null | tail -1 > "$D/bench_n1.json"; cut -c1-600 "$D/bench_n1.json"
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$D/bench_n2_dryrun_1gpu_gloo.json"; cut -c1-200 "$D/bench_n2_dryrun_1gpu_gloo.json"
# the RCCL branches on the one GPU there is: a one-rank process group, all three gather modes (round 3)
LEAF_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > "$D/bench_n1_rccl_world1.json"; cut -c1-200 "$D/bench_n1_rccl_world1.json"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > "$D/bench_under_rocprof.log" 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
    set -- $pass; name=$1; shift
    timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D/pmc_$name" -o w -- python tools/profile_workload.py > /dev/null 2>&1
    # the same counters for the opt-in streaming finalize and for BASELINE configs[2]'s kernel (4096-sample plan)
    timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D/pmc_stream_$name" -o w -- python tools/profile_workload.py stream > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D/pmc_cfg2_$name" -o w -- python tools/profile_workload.py cfg2 > /dev/null 2>&1
done
timeout 120 python tools/stage_times.py 2>&1 | grep "us (prep" > "$D/stage_times.txt"; cat "$D/stage_times.txt"
timeout 600 python tools/bench_configs.py 2>&1 | grep '^{' > "$D/configs_1gpu.jsonl"; cat "$D/configs_1gpu.jsonl" | cut -c1-330
timeout 300 python tools/compare_algos.py > "$D/fft_vs_mfma.txt" 2>&1; tail -4 "$D/fft_vs_mfma.txt"
timeout 300 python tools/bench_backward.py > "$D/backward_timing.txt" 2>&1; python tools/bench_backward.py 128 80 32000 5 >> "$D/backward_timing.txt" 2>&1; tail -3 "$D/backward_timing.txt"
python tools/bench_backward.py 256 40 22050 1 >> "$D/backward_timing.txt" 2>&1; python tools/bench_backward.py 256 40 48000 1 >> "$D/backward_timing.txt" 2>&1; python tools/bench_backward.py 256 40 8000 1 >> "$D/backward_timing.txt" 2>&1; python tools/bench_backward.py 256 40 32000 1 >> "$D/backward_timing.txt" 2>&1
timeout 600 python tools/bench_rates.py 2>&1 | grep '^{' > "$D/rates_1gpu.jsonl"; cut -c1-200 "$D/rates_1gpu.jsonl"
for sr in 16000 22050 48000; do echo "sample rate $sr" >> "$D/sweep_batch_wg.txt"; timeout 300 python tools/sweep_batch_wg.py $sr 2>&1 | grep '^B=' >> "$D/sweep_batch_wg.txt"; done
# training step: per-kernel stats at 16 kHz (static kernels), 22.05 kHz (run-time geometry, even window), 32 kHz (static, 4096-sample
# plan) and 48 kHz (run-time geometry, 4096-sample plan)
for sr in 16000 22050 32000 48000; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/bwd_stats_$sr" -o b -- python tools/profile_backward.py 256 40 $sr 1 > /dev/null 2>&1
done
# instruction / wait counters of the workgroup kernels: forward at the 16 kHz geometry, training step at 16 / 22.05 / 48 kHz
# (LEAF_WG_GENERIC is a tools-build switch since round 3: the static-vs-generic forward comparison is in profiles/r02)
for pass in "a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH"; do
    set -- $pass; name=$1; shift
    timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D/pmcx/fwd_static_$name" -o w -- python tools/profile_workload.py > /dev/null 2>&1
    for sr in 16000 22050 48000; do
        timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$D/pmcx/train_${sr}_$name" -o w -- python tools/profile_backward.py 256 40 $sr 1 > /dev/null 2>&1
    done
done
python tools/pmc_report.py "$D/pmcx" leaf_fft_wg > "$D/pmc_workgroup_kernels.json" 2>/dev/null; head -c 600 "$D/pmc_workgroup_kernels.json"
head -6 "$D/stats/bench_kernel_stats.csv" | cut -c1-160
