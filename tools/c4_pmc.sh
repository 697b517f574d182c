#!/bin/bash
# PMC passes over the 10^6-sphere frame (tools/c4_phase.py), one process per pass.
# usage: tools/c4_pmc.sh <out-dir under gpurun_out>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-c4pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- \
        python3 "$ROOT/tools/c4_phase.py" > "$OUT/pmc_$name.log" 2>&1 || exit 1
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM
pass tcc TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass fetch FETCH_SIZE
pass grbm GRBM_GUI_ACTIVE
echo c4-pmc-done
