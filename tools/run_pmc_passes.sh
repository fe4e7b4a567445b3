#!/bin/bash
# One rocprofv3 --pmc pass per counter over tools/probe_unet.py (a combined FETCH_SIZE + WRITE_SIZE pass aborts on gfx950).
# usage: tools/run_pmc_passes.sh <out dir under gpurun_out> [precision: bf16 | bf16x3 | f32]
set -e
OUT=$GRAFT_REPO_ROOT/$1
mkdir -p $OUT
PREC=${2:-bf16}
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout -k 10 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_unet.py $PREC > $OUT/$C.log 2>&1
  echo "$C done: $(find $OUT/$C -name '*counter_collection.csv' | wc -l) file(s)"
  find $OUT/$C -name '*kernel_trace.csv' -delete
done
