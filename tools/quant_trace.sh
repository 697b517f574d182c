# kernel traces of tools/quant_case.py, quantised (q) and float (f) nodes; results under gpurun_out/q
set -u
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/q
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in q f; do
  F=""; [ $v = f ] && F="--float"
  timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q/tr_$v -- python3 $R/tools/quant_case.py $F > $R/gpurun_out/q/tr_$v.log 2>&1 || exit 1
done
echo done
