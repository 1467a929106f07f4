#!/bin/bash
# A round's evidence on one MI355X (through gpurun), in two calls because the bench lines quote the PMC figures of the first:
#   gpurun --timeout 2400 -- 'bash tools/run_evidence_round.sh r06 pmc  r06pmc'   -> python tools/collect_round.py r06 pmc gpurun_out/r06pmc   (writes traffic.json)
#   gpurun --timeout 2400 -- 'bash tools/run_evidence_round.sh r06 main r06ev'    -> python tools/collect_round.py r06 files gpurun_out/r06ev ...
# (one script for every round: rounds 4 and 5 had a clone each)
set -u
ROUND=${1:?round, e.g. r06}
what=${2:-main}
D=gpurun_out/${3:-${ROUND}ev}
export TMPDIR=/tmp
mkdir -p "$D"
if [ "$what" = "pmc" ]; then
    bash tools/pmc_traffic.sh "${3:-${ROUND}ev}" cfg1 cfg2 cfg3 cfg4
    ls "$D"
    exit 0
fi
LEAF_GRAD_LOG="$D/grad_log.jsonl" timeout 1500 python -m pytest tests -m gpu -q --durations=25 > "$D/pytest_gpu_full.log" 2>&1
grep -E "s call|passed|failed" "$D/pytest_gpu_full.log" > "$D/pytest_gpu.log"; tail -3 "$D/pytest_gpu.log"
python tools/summarize_grad_log.py "$D/grad_log.jsonl" > "$D/backward_column_errors.txt"
LEAF_TEST_EXTENDED=1 timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k eight_rank 2>&1 | tail -3 > "$D/pytest_gpu_extended.log"; cat "$D/pytest_gpu_extended.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a "$D/pytest_gpu.log"
# the VALU microbenchmark first: its JSON line is profiles/valu_roof.json, which the bench lines quote as the practical roof
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -w -I "$GRAFT_REPO_ROOT/leaf_pytorch_amd/csrc" -I "$GRAFT_REPO_ROOT/include" \
    "$GRAFT_REPO_ROOT/tools/ubench_valu.hip" -o /tmp/ubench_valu && timeout 400 /tmp/ubench_valu) > "$D/ubench_valu.txt" 2>&1
tail -1 "$D/ubench_valu.txt" | python -c "
import json, sys
d = json.loads(sys.stdin.read()); d['from'] = 'profiles/$ROUND/ubench_valu.txt'
json.dump(d, open('profiles/valu_roof.json', 'w')); json.dump(d, open('$D/valu_roof.json', 'w')); print('valu roof', d['frac_of_peak'])"
for c in cfg1 cfg2 cfg3 cfg4; do
    timeout 600 python bench.py --config $c --steps 50 --warmup 10 2>/dev/null | tail -1 > "$D/bench_${c}_n1.json"; cut -c1-260 "$D/bench_${c}_n1.json"
done
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 2>/dev/null | tail -1 > "$D/bench_cfg1_n2_dryrun_1gpu_gloo.json"
timeout 600 python bench.py --config cfg2 --scaling strong --gpus 2 --steps 5 --warmup 2 2>/dev/null | tail -1 > "$D/bench_cfg2_strong_n2_dryrun_1gpu_gloo.json"
LEAF_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > "$D/bench_cfg1_n1_rccl_world1.json"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > "$D/bench_under_rocprof.log" 2>&1
cp "$D"/stats/*/bench_kernel_stats.csv "$D/bench_kernel_stats.csv" 2>/dev/null || cp "$D/stats/bench_kernel_stats.csv" "$D/bench_kernel_stats.csv"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats2" -o bench -- python bench.py --config cfg2 --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cp "$D"/stats2/*/bench_kernel_stats.csv "$D/bench_cfg2_kernel_stats.csv" 2>/dev/null || cp "$D/stats2/bench_kernel_stats.csv" "$D/bench_cfg2_kernel_stats.csv"
timeout 600 python tools/check_band.py 2>&1 | grep -v amdgpu.ids > "$D/band_check.txt"; tail -12 "$D/band_check.txt"
timeout 600 python tools/check_band4k.py 2>&1 | grep -v amdgpu.ids > "$D/band4k_check.txt"; tail -6 "$D/band4k_check.txt"
timeout 600 python tools/check_bias_bound.py 2>&1 | grep -v amdgpu.ids > "$D/bias_bound_check.txt"; grep -E "classes|worst|cfg" "$D/bias_bound_check.txt"
timeout 600 python tools/exp_bwd_bias.py 2>&1 | grep -v amdgpu.ids > "$D/exp_bwd_bias.txt"; cat "$D/exp_bwd_bias.txt"
timeout 600 python -m pytest tests/test_gpu_band.py -q -s -k fuzz 2>&1 | grep -E "band fuzz|passed|failed" > "$D/band_fuzz.txt"; tail -3 "$D/band_fuzz.txt"
timeout 600 python tools/bench_configs.py 2>&1 | grep '^{' > "$D/configs_1gpu.jsonl"; cut -c1-200 "$D/configs_1gpu.jsonl"
timeout 300 python tools/latency_breakdown.py 2>/dev/null | grep '^{' > "$D/latency_breakdown.jsonl"
(timeout 300 python tools/bench_backward.py; timeout 300 python tools/bench_backward.py 128 80 32000 5; timeout 300 python tools/bench_backward.py 256 40 22050 1; timeout 300 python tools/bench_backward.py 256 40 48000 1) 2>&1 | grep -v amdgpu.ids > "$D/backward_timing.txt"; cat "$D/backward_timing.txt"
timeout 600 python tools/bench_rates.py 2>&1 | grep '^{' > "$D/rates_1gpu.jsonl"
for sr in 16000 32000; do
    if [ $sr = 16000 ]; then a="256 40 16000 1"; else a="128 80 32000 5"; fi            # BASELINE configs[1] / [2]; 60 steps: steady-state averages
    LEAF_PROFILE_STEPS=60 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/bwd_stats_$sr" -o b -- python tools/profile_backward.py $a > /dev/null 2>&1
    f=$(ls "$D"/bwd_stats_$sr/*/b_kernel_stats.csv "$D"/bwd_stats_$sr/b_kernel_stats.csv 2>/dev/null | head -1); head -12 "$f" > "$D/training_step_kernel_stats_$sr.csv"
done
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w "$GRAFT_REPO_ROOT/tools/probe_wave_placement.hip" -o /tmp/probe_wp && /tmp/probe_wp) > "$D/wave_placement.txt" 2>&1
ls "$D"
