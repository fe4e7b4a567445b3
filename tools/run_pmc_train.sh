#!/bin/bash
# One rocprofv3 --pmc pass per named counter over `bench.py --mode train` (dev tool; separate passes: combined ones abort
# on gfx950).  usage: tools/run_pmc_train.sh <out dir under gpurun_out> COUNTER...
set -e
OUT=$GRAFT_REPO_ROOT/$1
shift 1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for C in "$@"; do
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o p -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 > $OUT/$C.log 2>&1 || echo "$C FAILED"
  echo "$C done: $(find $OUT/$C -name '*counter_collection.csv' | wc -l) file(s)"
  find $OUT/$C -name '*kernel_trace.csv' -delete
done
