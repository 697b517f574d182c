#!/bin/bash
# PMC passes of the headline frame for one build of the library (LDS and wait counters, lanes per VALU
# instruction): one process per pass, --pmc alone.  usage: tools/pmc_variant.sh <out-dir under gpurun_out> [--lib path]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_lds" -- \
    python3 "$ROOT/tools/frame_loop.py" --frames 3 "$@" > "$OUT/pmc_lds.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d "$OUT/pmc_valu" -- \
    python3 "$ROOT/tools/frame_loop.py" --frames 3 "$@" > "$OUT/pmc_valu.log" 2>&1 || exit 1
echo pmc-done
