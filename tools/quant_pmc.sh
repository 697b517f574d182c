# PMC passes over tools/quant_case.py, quantised (q) and float (f) nodes; results under gpurun_out/q
set -u
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/q
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in q f; do
  F=""; [ $v = f ] && F="--float"
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/q/pmc1_$v -- python3 $R/tools/quant_case.py $F > $R/gpurun_out/q/pmc1_$v.log 2>&1 || exit 1
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/q/pmc2_$v -- python3 $R/tools/quant_case.py $F > $R/gpurun_out/q/pmc2_$v.log 2>&1 || exit 1
  timeout -s KILL 90 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/q/pmc3_$v -- python3 $R/tools/quant_case.py $F > $R/gpurun_out/q/pmc3_$v.log 2>&1 || exit 1
done
echo done
