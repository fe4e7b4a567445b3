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
  cp $SRC/train_deterministic_output.log profiles/${TAG}_train_deterministic_output.log
  cp $SRC/train_step_trace.txt profiles/${TAG}_train_step_trace.txt
  cp $SRC/concurrent_forward_root_cause.log profiles/${TAG}_concurrent_forward_root_cause.log
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
python3 bench.py --mode train --train-deterministic --no-modes > $OUT/train_deterministic_output.log 2>&1
# the defect of round 4 and its cause: product library, then the dev build with the former head kernel
{
  echo "== product library, no forward chain: two / three engines of one process side by side"
  BSMI_FORWARD_CHAIN=0 python3 tools/debug_two_streams.py 2>&1 | grep differ
  ENGINES=3 BSMI_FORWARD_CHAIN=0 python3 tools/debug_two_streams.py 2>&1 | grep differ
  if [ -f bootstrapper_amd/libbsmi_headscratch.so ]; then
    echo "== dev build with the former head kernel (272-byte scratch segment): the same reproducer"
    BSMI_LIB=$R/bootstrapper_amd/libbsmi_headscratch.so BSMI_FORWARD_CHAIN=0 python3 tools/debug_two_streams.py 2>&1 | grep differ
    echo "== dev build: which launch of a corrupted forward pass differs (every launch's output tensor against a pass run alone)"
    BSMI_LIB=$R/bootstrapper_amd/libbsmi_headscratch.so CASES=1 python3 tools/debug_first_step.py 2>&1 | grep -v amdgpu.ids
  fi
  echo "== product library, guarded allocations (8 MiB of 0xFF on both sides of every device buffer of the engine)"
  python3 tools/debug_guards.py 2>&1 | grep -v amdgpu.ids && BSMI_GUARD_MB=8 python3 tools/debug_guards.py 2>&1 | grep -v amdgpu.ids
  echo "== product library: the whole pipeline as 1 rank, then 2 and 4 ranks SHARING the card, predict stages side by side"
  mkdir -p $OUT/ranks
  python3 tools/debug_ranks.py $OUT/ranks 2>&1 | grep "sha1"
  python3 -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 tools/debug_ranks.py $OUT/ranks 2>&1 | grep "equal\|DIFFER"
  python3 -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29656 tools/debug_ranks.py $OUT/ranks 2>&1 | grep "equal\|DIFFER"
  rm -rf $OUT/ranks
} > $OUT/concurrent_forward_root_cause.log 2>&1 || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_train -o t -- python3 $R/bench.py --mode train --steps 6 --warmup 2 --no-modes > /dev/null 2>&1
python3 $R/tools/trace_train.py $(find $OUT/trace_train -name '*kernel_trace.csv' | head -1) > $OUT/train_step_trace.txt 2>&1 || true
rm -rf $OUT/trace_train
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python3 $R/bench.py --no-drivers --no-train --no-whole-volume > $OUT/bench_default_rocprof_output.log 2>&1
find $OUT/stats -name '*kernel_trace.csv' -delete
cd $R
bash tools/run_pmc_passes.sh gpurun_out/$1/pmc bf16x3
echo "collected into $OUT"
