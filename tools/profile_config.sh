#!/bin/bash
# Collects the profiles of one bench configuration on the GPU box: one kernel-trace run and separate
# PMC passes (each pass is its own process; no trace domains are combined with --pmc).
# usage: tools/profile_config.sh <c2|c3|c4|c5> <out-dir under gpurun_out> [extra bench.py flags]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-c2}
OUT=$ROOT/gpurun_out/${2:-prof_$CFG}
shift; shift
EXTRA="$*"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$ROOT/bench.py" --config $CFG --steps 10 --warmup 2 --frames-in-flight 1 --no-other-configs $EXTRA > "$OUT/bench_traced.json" 2> "$OUT/bench_traced.err" || exit 1
pass() {   # name, counters...
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- \
        python3 "$ROOT/bench.py" --config $CFG --steps 2 --warmup 0 --frames-in-flight 1 --no-cpu-baseline --no-other-configs $EXTRA > "$OUT/pmc_$name.log" 2>&1 || exit 1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass tcc TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass grbm GRBM_GUI_ACTIVE
echo profile-done $CFG
