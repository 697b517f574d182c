#!/bin/bash
# Everything profiles/<round> holds, collected on the GPU box in one call: per configuration the kernel trace
# and the PMC passes (tools/profile_config.sh), the regression table, and the probes the design notes cite.
# usage: tools/profile_round.sh <dir under gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
D=${1:-prof_round}
OUT=$ROOT/gpurun_out/$D
mkdir -p "$OUT"
cd "$ROOT"
for cfg in c2 c3 c4; do
    bash tools/profile_config.sh $cfg $D/$cfg > "$OUT/profile_$cfg.log" 2>&1 || { echo "profile $cfg failed"; exit 1; }
done
timeout -k 10 200 python tools/regress.py "$OUT/regress.json" > "$OUT/regress.txt" 2>&1 || exit 1
timeout -k 10 200 python tools/feat_probe.py > "$OUT/feature_kernels.txt" 2>&1 || exit 1
timeout -k 10 120 python tools/depth_probe.py > "$OUT/depth_probe.txt" 2>&1 || exit 1
timeout -k 10 120 python tools/phase_shares.py > "$OUT/phase_shares_run.txt" 2>&1 || exit 1
timeout -k 10 120 python tools/share_sweep.py > "$OUT/share_sweep.txt" 2>&1 || exit 1
timeout -k 10 200 python tools/shard_probe.py > "$OUT/shard_probe.txt" 2>&1 || exit 1
timeout -k 10 200 python tools/pipeline_probe.py > "$OUT/pipeline_probe.txt" 2>&1 || exit 1
timeout -k 10 120 python tools/recon_probe.py > "$OUT/recon_probe.txt" 2>&1 || exit 1
timeout -k 10 120 python tools/upscale_probe.py > "$OUT/upscale_probe.txt" 2>&1 || exit 1
timeout -k 10 400 python tools/c4_sweep.py > "$OUT/c4_sweep.txt" 2>&1 || exit 1
timeout -k 10 60 python tools/ppm_probe.py > "$OUT/ppm_probe.txt" 2>&1 || exit 1
[ -x tools/micro/exec_mask_rate ] && timeout -k 10 60 ./tools/micro/exec_mask_rate > "$OUT/exec_mask_rate.txt" 2>&1
timeout -k 10 300 python bench.py --rehearse-collectives --no-cpu-baseline --no-other-configs --steps 20 > "$OUT/bench_rehearse_collectives.json" 2> "$OUT/bench_rehearse.err" || exit 1
timeout -k 10 300 python bench.py --config c5 --steps 12 > "$OUT/c5_bench.json" 2> "$OUT/c5_bench.err" || exit 1
timeout -k 10 400 python bench.py > "$OUT/c2_bench_as_the_driver_runs_it.json" 2> "$OUT/c2_bench.err" || exit 1
echo profile-round-done
