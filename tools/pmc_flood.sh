#!/bin/bash
# Dev tool: instruction mix of ws_flood_kernel (one rocprofv3 --pmc pass per counter over tools/probe_seg.py).
# usage: tools/pmc_flood.sh <out dir under gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for C in SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o p -- python3 $GRAFT_REPO_ROOT/tools/probe_seg.py > $OUT/$C.log 2>&1 || echo "$C FAILED"
  find $OUT/$C -name '*kernel_trace.csv' -delete
done
python3 - $OUT <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + '/SQ_*/')):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-40:]
            a = acc[k]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in acc.items():
        if 'ws_flood' in k or 'ws_seeds' in k:
            print(d.rstrip('/').split('/')[-1], k, 'dispatch rows', n, 'per launch', v / max(1, n))
P
