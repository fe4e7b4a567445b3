#!/bin/bash
# One rocprofv3 --pmc pass per named counter over tools/probe_unet.py (dev tool; counters in separate passes as on gfx950
# combined passes abort).  usage: tools/run_pmc_list.sh <out dir under gpurun_out> <precision> COUNTER...
set -e
OUT=$GRAFT_REPO_ROOT/$1
PREC=$2
shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for C in "$@"; do
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_unet.py $PREC > $OUT/$C.log 2>&1 || echo "$C FAILED"
  echo "$C done: $(find $OUT/$C -name '*counter_collection.csv' | wc -l) file(s)"
  find $OUT/$C -name '*kernel_trace.csv' -delete
done
