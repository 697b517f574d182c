#!/bin/bash
# PMC passes over tools/wide_probe.py (binary against four-wide tree), one process per pass.
# usage: tools/wide_pmc.sh <out-dir under gpurun_out> <c2|c4>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-widepmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- \
        python3 "$ROOT/tools/wide_probe.py" ${WHICH} > "$OUT/pmc_$name.log" 2>&1 || exit 1
}
WHICH=${2:-c2}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
echo wide-pmc-done
