#!/bin/bash
# Everything profiles/ holds for a round, collected in one gpurun call.  usage: tools/collect_profiles.sh <dir under gpurun_out>
# Afterwards (here, from the merged gpurun_out/): tools/collect_profiles.sh --install <dir> <tag> copies the summaries into profiles/.
set -e
if [ "$1" = "--install" ]; then
  SRC=gpurun_out/$2; TAG=$3
  cp $SRC/bench_default_output.log profiles/${TAG}_bench_default_output.log
  cp $SRC/bench_driver_form_output.log profiles/${TAG}_bench_driver_form_output.log
  cp $SRC/bench_default_rocprof_output.log profiles/${TAG}_bench_default_rocprof_output.log
  cp $SRC/stats/b_kernel_stats.csv profiles/${TAG}_bench_default_kernel_stats.csv
  cp $SRC/probe_unet_bf16x3.log profiles/${TAG}_probe_unet_per_launch_bf16x3.log
  cp $SRC/train_default_output.log profiles/${TAG}_train_default_output.log
  python3 tools/pmc_traffic.py $SRC/pmc/FETCH_SIZE/p_counter_collection.csv $SRC/pmc/WRITE_SIZE/p_counter_collection.csv profiles/${TAG}_conv_traffic_pmc_bf16x3.json 39
  python3 tools/pmc_sq.py $SRC/pmc profiles/${TAG}_conv_sq_pmc_bf16x3.json > /dev/null
  exit 0
fi
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form_output.log 2>&1
python3 bench.py --no-drivers --no-train --no-whole-volume > $OUT/bench_default_output.log 2>&1
python3 tools/probe_unet.py bf16x3 > $OUT/probe_unet_bf16x3.log 2>&1
python3 bench.py --mode train > $OUT/train_default_output.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 $R/bench.py --no-drivers --no-train --no-whole-volume > $OUT/bench_default_rocprof_output.log 2>&1
find $OUT/stats -name '*kernel_trace.csv' -delete
cd $R
bash tools/run_pmc_passes.sh gpurun_out/$1/pmc bf16x3
echo "collected into $OUT"
